"""world_size-2 run of the multi-GPU path on CPU (gloo): contiguous read sharding + host-side
concatenation must reproduce the single-process result, in input order.  The per-rank searcher here
is the CPU oracle wrapped to look like binding.Result (there is no GPU in this container); on the GPU
box the same shard.search_sharded drives Engine.search_batch."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OracleResult:
    """Oracle output re-encoded as run-length CSR, shaped like binding.Result."""

    def __init__(self, orc, n):
        from pindel_amd.binding import RUN_DTYPE

        def encode(cnt, pts):
            off, runs = [0], []
            for i in range(n):
                p = pts[i][:cnt[i]]
                for q in p:   # one run per point is a valid encoding
                    flags = (1 if q["direction"] == b"-" else 0) | (2 if q["strand"] == b"-" else 0)
                    runs.append((q["abs_loc"], q["length"], q["length"], q["mismatches"], flags, q["chr_id"]))
                off.append(len(runs))
            return np.array(off, dtype=np.uint64), np.array(runs, dtype=RUN_DTYPE)
        self.close_off, self.close_runs = encode(orc["close_cnt"], orc["close_pts"])
        self.far_off, self.far_runs = encode(orc["far_cnt"], orc["far_pts"])
        self.rc_flag = orc["rc_flag"]


def _search(ref, batch):
    from oracle import pyoracle
    p = pyoracle.make_params()
    orc = pyoracle.search_batch(p, [ref], batch.seq, batch.seq_off, batch.anchor_strand, batch.anchor_pos,
                                batch.insert_size, batch.chr_id, n_threads=2)
    return _OracleResult(orc, batch.n)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pindel_amd import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = synth.make_reference(300_000, seed=41)
    batch = synth.make_reads(ref, 301, seed=42)           # odd size: uneven shards
    res = shard.search_sharded(lambda b: _search(ref, b), batch, rank, world, dist)
    dist.barrier()
    if rank == 0:
        np.savez(out_path, **res)
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    from pindel_amd import binding, shard, synth
    assert shard.shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    out = str(tmp_path / "sharded.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    ref = synth.make_reference(300_000, seed=41)
    batch = synth.make_reads(ref, 301, seed=42)
    whole = shard.result_arrays(_search(ref, batch))
    for k in ("close_off", "far_off", "rc_flag"):
        np.testing.assert_array_equal(got[k], whole[k], err_msg=k)
    for k in ("close_runs", "far_runs"):
        assert got[k].tobytes() == whole[k].tobytes(), k
    assert int(got["far_off"][-1]) > 1000
    # and the expanded points agree with pg_expand_runs semantics
    pts = binding.expand_runs(got["close_runs"].astype(binding.RUN_DTYPE))
    assert len(pts) == int(got["close_off"][-1])

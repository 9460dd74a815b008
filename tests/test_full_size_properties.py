"""-m gpu: the bench workload at BASELINE.json's full size (configs[2]: 10 M x 100 bp reads on a
chr20-shaped reference).  The oracle cannot run 10 M reads in a test, so the full-size run is checked
through size-independent properties of the path plus a sampled bit-exact comparison:

  * idempotence      -- searching the same device batch twice gives identical results
  * shard additivity -- reads are independent: the results of ragged contiguous shards, concatenated,
                        equal the result of the whole batch (what the multi-GPU path relies on)
  * structure        -- run-length-encoded lists are well formed for every read (lengths inside the
                        read, increasing along a list, far end only behind a close end, runs inside the
                        chromosome)
  * sampled parity   -- 20 000 reads drawn from all over the batch, bit-exact against the CPU oracle
"""
import numpy as np
import pytest

from pindel_amd import binding, hostio, shard, synth
from tests.parity import compare_result, run_oracle

pytestmark = pytest.mark.gpu

CHR20_LEN = 62_435_964
N_READS = 10_000_000
READ_LEN = 100


def _arrays(res):
    return shard.result_arrays(res)


def _assert_same(a, b, what):
    for k in ("close_off", "far_off", "rc_flag"):
        assert np.array_equal(a[k], b[k]), f"{what}: {k} differs"
    for k in ("close_runs", "far_runs"):
        assert a[k].tobytes() == b[k].tobytes(), f"{what}: {k} differs"


def test_bench_workload_properties(engine_factory):
    import torch
    dev = torch.device("cuda", 0)
    ref = synth.make_reference(CHR20_LEN, seed=20260927, device=dev)
    chroms = [("20", ref)]
    batch = synth.make_reads(ref, N_READS, read_len=READ_LEN, seed=20260928, device=dev)
    eng = engine_factory()
    eng.load_reference(chroms)

    db = eng.upload(batch)
    eng.search_device(db)
    whole_res = eng.download(db)
    whole = _arrays(whole_res)
    n_close = int((np.diff(whole["close_off"].astype(np.int64)) > 0).sum())
    n_far = int((np.diff(whole["far_off"].astype(np.int64)) > 0).sum())
    assert n_close > 0.7 * N_READS and n_far > 0.5 * N_READS

    # ---- idempotence
    eng.search_device(db)
    again = _arrays(eng.download(db))
    eng.free_device_batch(db)
    _assert_same(whole, again, "second search of the same device batch")

    # ---- shard additivity (ragged shards)
    parts = []
    for lo, hi in ((0, 3_333_333), (3_333_333, 7_000_001), (7_000_001, N_READS)):
        d = eng.upload(batch.slice(lo, hi))
        eng.search_device(d)
        parts.append(_arrays(eng.download(d)))
        eng.free_device_batch(d)
    _assert_same(whole, shard.concat_results(parts), "concatenated shards vs whole batch")

    # ---- structure of every list
    chr_size = len(ref)
    for key in ("close", "far"):
        off = whole[key + "_off"].astype(np.int64)
        runs = whole[key + "_runs"]
        assert off[0] == 0 and off[-1] == len(runs) and np.all(np.diff(off) >= 0)
        lf, ll = runs["len_first"].astype(np.int64), runs["len_last"].astype(np.int64)
        assert np.all(lf <= ll) and np.all(ll <= READ_LEN - 1) and np.all(lf >= (8 if key == "close" else 10))
        assert np.all(runs["abs_loc_first"] < chr_size) and np.all(runs["chr_id"] == 0)
        # lengths increase along a read's list: a run starts after the previous run of the same read ended
        same_read = np.ones(len(runs), dtype=bool)
        same_read[off[:-1][np.diff(off) > 0]] = False          # first run of each read
        assert np.all(lf[1:][same_read[1:]] > ll[:-1][same_read[1:]])
    has_close = np.diff(whole["close_off"].astype(np.int64)) > 0
    has_far = np.diff(whole["far_off"].astype(np.int64)) > 0
    assert not np.any(has_far & ~has_close)
    assert not np.any((whole["rc_flag"] != 0) & ~has_close)

    # ---- sampled parity against the oracle
    rng = np.random.default_rng(7)
    idx = np.sort(rng.choice(N_READS, size=20_000, replace=False))
    seq = np.asarray(batch.seq).reshape(N_READS, READ_LEN)[idx].reshape(-1)
    sub = hostio.ReadBatch(seq=seq, seq_off=(np.arange(len(idx) + 1, dtype=np.uint64) * READ_LEN),
                           anchor_strand=np.asarray(batch.anchor_strand)[idx], anchor_pos=np.asarray(batch.anchor_pos)[idx],
                           insert_size=np.asarray(batch.insert_size)[idx], chr_id=np.asarray(batch.chr_id)[idx])
    orc = run_oracle({}, chroms, sub)

    class Picked:                         # the sampled reads' slices of the full-size result
        n = len(idx)
        rc_flag = whole["rc_flag"][idx]

        @staticmethod
        def _gather(key):
            off = whole[key + "_off"].astype(np.int64)
            cnt = (off[1:] - off[:-1])[idx]
            new_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
            sel = np.concatenate([np.arange(off[i], off[i + 1]) for i in idx]) if len(idx) else np.zeros(0, np.int64)
            return new_off, whole[key + "_runs"][sel.astype(np.int64)]

        def close_points(self, i):
            return binding.expand_runs(self.close_runs[int(self.close_off[i]):int(self.close_off[i + 1])])

        def far_points(self, i):
            return binding.expand_runs(self.far_runs[int(self.far_off[i]):int(self.far_off[i + 1])])

    picked = Picked()
    picked.close_off, picked.close_runs = Picked._gather("close")
    picked.far_off, picked.far_runs = Picked._gather("far")
    compare_result(picked, orc, len(idx))

"""-m gpu: the BASELINE.json configurations other than configs[2] (which tests/test_full_size_properties.py covers).

  configs[0]  demo/simulated_reference.fa (the reference's own 4 x 200 kbp demo FASTA, committed as data under
              tests/golden/demo/; its three BAMs are not in the snapshot): reads generated on its 4 chromosomes
              with the demo's insert size (250), HIP path vs oracle bit-exact, FASTA through the C++ loader.
  configs[1]  COLO-829 chr20 + BreakDancer hints, deletions only: (a) the bench's `--workload colo-bd` generator
              at 1 M reads through the size-independent properties + 20 000 sampled reads against the oracle;
              (b) the reference's real demo/COLO-829.20.BreakDancer.sv parsed against hs_ref_chr20.fa.fai's
              chromosome by pg_bdhints (loadBDFile / loadRegion / getCorrespondingSearchWindowCluster,
              src/bddata.cpp:91-136, 814-979) -> per-read clusters -> pg_far_end_batch vs oracle with the same
              clusters (src/pindel.cpp:1006-1018).  The chr20 FASTA itself is not in the snapshot: the sequence
              is synthetic, the event coordinates are the real ones.
  configs[3]  GRCh38-shaped reference (24 chromosomes, 3.1 Gbp; pg_load_reference + packed cache) and one rank's
              shard of the 100 M x 150 bp reads (12.5 M) through the properties + sampled oracle parity.
  configs[4]  30x-WGS-shaped: coordinate-ordered 150-bp reads, 0.65 M per 5-Mbp bin, bin by bin on the same reference.
"""
import ctypes as C
import gzip
import os
import shutil
import types

import numpy as np
import pytest

from pindel_amd import binding, hostio, hostlib, synth
from tests.parity import compare_result, run_oracle
from tests.properties import check_workload

pytestmark = pytest.mark.gpu

DEMO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo")


def _gunzip(name, tmp_path):
    dst = os.path.join(str(tmp_path), name)
    with gzip.open(os.path.join(DEMO, name + ".gz"), "rb") as s, open(dst, "wb") as d:
        shutil.copyfileobj(s, d)
    return dst


# ------------------------------------------------------------------------------------------ configs[0]
def test_config0_demo_reference(engine_factory, tmp_path):
    fa = _gunzip("simulated_reference.fa", tmp_path)
    chroms = hostio.load_fasta(fa)
    fai = [l.split() for l in open(os.path.join(DEMO, "simulated_reference.fa.fai"))]
    assert [c[0] for c in chroms] == [f[0] for f in fai] == ["1", "2", "3", "4"]
    eng = engine_factory()
    eng.load_fasta(fa)                                   # Genome::loadChromosome semantics in C++
    info = eng.reference_info()
    for (name, seq), (iname, isize), f in zip(chroms, info, fai):
        # the reference's loader duplicates the final base of the LAST record (pindel.cpp:285-295)
        assert name == iname and isize == len(seq) and len(seq) - 2 * hostio.SPACER in (int(f[1]), int(f[1]) + 1)
        assert eng.reference_fetch(info.index((iname, isize)), 0, isize) == seq
    # demo/simulated_config.txt: insert size 250 for all three samples
    cfg = [l.split() for l in open(os.path.join(DEMO, "simulated_config.txt"))]
    assert {int(c[1]) for c in cfg} == {250}
    batch = synth.make_reads_genome(chroms, 12_000, seed=90, insert_size=250, max_del=4000)
    assert set(np.unique(batch.chr_id)) == {0, 1, 2, 3}
    gpu = eng.search_batch(batch)
    orc = run_oracle({}, chroms, batch)
    assert (orc["close_cnt"] > 0).sum() > 6000 and (orc["far_cnt"] > 0).sum() > 4000
    compare_result(gpu, orc, batch.n)
    # and through the two seams, with mixed read lengths
    batch2 = synth.make_reads_genome(chroms, 4000, seed=91, insert_size=250, max_del=4000, read_lens=(36, 76, 100, 150))
    close = eng.close_end_batch(batch2)
    both = eng.far_end_batch(batch2, close)
    compare_result(both, run_oracle({}, chroms, batch2), batch2.n)


# ------------------------------------------------------------------------------------------ configs[1]
def _bench_args(**kw):
    d = dict(workload="colo-bd", reads=10_000_000, read_len=100, chr_len=62_435_964, max_range_index=2,
             seed=20260927, scaling="weak", genome_scale=1.0)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_config1_colo_bd_workload(engine_factory):
    import torch
    import bench
    args = _bench_args()
    chroms, batch, bd, bd_off, desc, total = bench.build_workload(args, 0, 1, torch.device("cuda", 0))
    assert batch.n == 1_000_000 and "configs[1]" in desc and bd is not None
    eng = engine_factory()
    eng.load_reference(chroms)
    n_close, n_far = check_workload(eng, chroms, batch, bd=bd, bd_off=bd_off)
    assert n_close > 0.8 * batch.n and n_far > 0.6 * batch.n


def _last_close_absloc(res):
    off = res.close_off.astype(np.int64)
    has = off[1:] > off[:-1]
    last = res.close_runs[np.maximum(off[1:] - 1, 0)] if len(res.close_runs) else None
    out = np.zeros(res.n, dtype=np.uint32)
    if last is not None:
        span = last["len_last"].astype(np.int64) - last["len_first"].astype(np.int64)
        back = (last["flags"] & 1) != 0
        loc = last["abs_loc_first"].astype(np.int64) + np.where(back, -span, span)
        out[has] = loc[has].astype(np.uint32)
    return out, has


def test_config1_real_breakdancer_file(engine_factory, tmp_path):
    import torch
    bd_path = _gunzip("COLO-829.20.BreakDancer.sv", tmp_path)
    name, size = open(os.path.join(DEMO, "hs_ref_chr20.fa.fai")).read().split()[:2]
    assert name == "20" and int(size) == 62_435_964
    dev = torch.device("cuda", 0)
    ref = synth.make_reference(int(size), seed=20260927, device=dev)
    chroms = [(name, ref)]
    refa = np.frombuffer(ref, dtype=np.uint8)
    # the file's intra-chromosomal events with 600 bp .. 60 kb between the two coordinates
    ev = []
    for line in open(bd_path):
        if line.startswith("#"):
            continue
        f = line.split()
        if len(f) >= 6 and f[0] == f[3] == "20":
            p1, p2 = int(f[1]), int(f[4])
            if 600 <= p2 - p1 <= 60_000 and p1 > 2000 and p2 < int(size) - 2000:
                ev.append((p1, p2))
    assert len(ev) > 150
    rng = np.random.default_rng(11)
    ev = [ev[i] for i in rng.choice(len(ev), size=min(1500, len(ev)), replace=False)]
    S, L, ISZ = hostio.SPACER, 100, 450
    comp = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    seqs, strands, poss = [], [], []
    for p1, p2 in ev:
        for _ in range(16):
            # a deletion of [p1 + d, p2 + d): reads that cross it, anchored upstream ('+') or downstream ('-')
            d = int(rng.integers(-150, 150))
            sp = int(rng.integers(25, 75))
            a, b = p1 + d + S, p2 + d + S
            read = np.concatenate([refa[a - sp:a], refa[b:b + L - sp]]).copy()
            err = rng.random(L) < 0.01
            read[err] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(err.sum()))]
            if rng.random() < 0.5:                      # '+' anchor: mate upstream, read sequenced from the reverse strand
                seqs.append(comp[read[::-1]].tobytes())
                strands.append(b"+")
                poss.append(a - sp - int(rng.integers(0, ISZ - L - 20)) - S)
            else:
                seqs.append(read.tobytes())
                strands.append(b"-")
                poss.append(b + (L - sp) + int(rng.integers(0, ISZ - L - 20)) - S)
    n = len(seqs)
    order = np.argsort(np.array(poss), kind="stable")
    batch = hostio.batch_from_lists([seqs[i] for i in order], [strands[i] for i in order], [poss[i] for i in order],
                                    [ISZ] * n, [0] * n)
    eng = engine_factory()
    eng.load_reference(chroms)
    close = eng.close_end_batch(batch)
    last, has = _last_close_absloc(close)
    assert has.sum() > 0.8 * n
    # per 5-Mbp bin of the anchor (main()'s loop, src/pindel.cpp:1816-1853): loadRegion + one cluster per read
    Lh = hostlib.lib()
    Lh.pgh_bd_query.argtypes = [C.c_char_p, C.c_uint32, C.c_int32, C.POINTER(C.c_char_p), C.c_int32, C.c_uint32,
                                C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    names = (C.c_char_p * 1)(b"20")
    W = 5_000_000
    bins = batch.anchor_pos // W
    counts = np.zeros(n, dtype=np.int64)
    wins = [None] * n
    n_events = 0
    for b in np.unique(bins):
        sel = np.nonzero((bins == b) & has)[0]
        q = np.ascontiguousarray(last[sel], dtype=np.uint32)
        off = np.zeros(len(q) + 1, dtype=np.uint64)
        cap = 128 * max(len(q), 1)
        win = np.zeros(3 * cap, dtype=np.int32)
        nev = C.c_uint64()
        rc = Lh.pgh_bd_query(bd_path.encode(), S, 1, names, 0, int(b) * W + S, int(b + 1) * W + S, len(q), q.ctypes.data,
                             off.ctypes.data, win.ctypes.data, cap, C.byref(nev))
        assert rc == 0, (rc, Lh.pgh_last_error())
        n_events = nev.value
        for k, i in enumerate(sel):
            wins[i] = win[3 * int(off[k]):3 * int(off[k + 1])].reshape(-1, 3)
            counts[i] = len(wins[i])
    assert n_events > 250                                # the real file: 5 9xx lines, most are dropped by loadBDFile's
                                                         # |pos1 - pos2| < 500 rule (bddata.cpp:122), 310 stay
    assert (counts > 0).sum() > 0.5 * has.sum() and counts.max() <= 127
    bd_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    bd = np.zeros(int(bd_off[-1]), dtype=binding.WINDOW_DTYPE)
    flat = np.concatenate([w for w in wins if w is not None and len(w)]) if bd_off[-1] else np.zeros((0, 3), np.int32)
    bd["chr_id"], bd["start"], bd["end"] = flat[:, 0], flat[:, 1], flat[:, 2]
    both = eng.far_end_batch(batch, close, bd, bd_off)
    orc = run_oracle({}, chroms, batch, bd=bd, bd_off=bd_off)
    compare_result(both, orc, n)
    # the hints matter: deletions longer than the widest range (2 048 at -x 2) are only reachable through them
    plain = run_oracle({}, chroms, batch)
    assert (orc["far_cnt"] > 0).sum() > (plain["far_cnt"] > 0).sum() + 0.2 * n


# ------------------------------------------------------------------------------------------ configs[3]
def test_config3_grch38_shaped_shard(engine_factory, tmp_path):
    import torch
    import bench
    args = _bench_args(workload="grch38-150")
    chroms, batch, bd, bd_off, desc, total = bench.build_workload(args, 0, 1, torch.device("cuda", 0))
    assert len(chroms) == 24 and sum(len(s) for _, s in chroms) > 3.0e9 and batch.n == 12_500_000
    assert int(batch.lengths().max()) == 150 and len(np.unique(batch.chr_id)) == 24
    eng = engine_factory()
    eng.load_reference(chroms)
    # packed cache round trip at full size: 1.2 GB of planes
    packed = os.path.join(str(tmp_path), "grch38_shaped.pgref")
    eng.save_packed(packed)
    assert os.path.getsize(packed) > 1.1e9
    eng2 = engine_factory()
    eng2.load_packed(packed)
    os.remove(packed)
    assert eng2.reference_info() == eng.reference_info()
    probe = hostio.SPACER + 1234567
    for c in (0, 11, 23):
        assert eng2.reference_fetch(c, probe, 500) == chroms[c][1][probe:probe + 500]
    eng.close()
    n_close, n_far = check_workload(eng2, chroms, batch, n_sample=20_000)
    assert n_close > 0.7 * batch.n and n_far > 0.5 * batch.n


# ------------------------------------------------------------------------------------------ configs[4]
def test_config4_wgs_bins(engine_factory):
    """30x-WGS-shaped: 150-bp reads in coordinate order, 0.65 M per 5-Mbp bin, searched bin by bin on the full
    GRCh38-shaped reference (4 bins here; bench.py --workload wgs-bins runs 20).  Bin-by-bin results concatenated ==
    the whole batch (the shard cuts of the property check are the bin boundaries), sampled oracle parity."""
    import torch
    import bench
    args = _bench_args(workload="wgs-bins", reads=2_600_000)
    chroms, batch, bd, bd_off, desc, total = bench.build_workload(args, 0, 1, torch.device("cuda", 0))
    assert "configs[4]" in desc and len(chroms) == 24 and abs(batch.n - 2_600_000) < 10
    key = batch.chr_id.astype(np.int64) * 100_000 + batch.anchor_pos // 5_000_000
    assert np.all(np.diff(key) >= 0)
    cuts = (np.nonzero(np.diff(key))[0] + 1) / batch.n
    assert len(cuts) == 3
    eng = engine_factory()
    eng.load_reference(chroms)
    n_close, n_far = check_workload(eng, chroms, batch, n_sample=10_000, shard_cuts=tuple(float(c) for c in cuts))
    assert n_close > 0.7 * batch.n and n_far > 0.5 * batch.n


# ------------------------------------------------------------------------------------------ the retry path's workload
def test_wgs_real_read_mix(engine_factory):
    """bench.py --workload wgs-real (round-5 verdict, item 2): three reads in four find no close end and walk all four attempts;
    150 bp, coordinate order.  Size-independent properties on 1 M reads + 20 000 sampled reads against the oracle."""
    import torch
    import bench
    args = _bench_args()
    args.workload, args.reads, args.read_len = "wgs-real", 1_000_000, 100
    chroms, batch, bd, bd_off, desc, total = bench.build_workload(args, 0, 1, torch.device("cuda", 0))
    assert batch.n == 1_000_000 and args.read_len == 150 and bd is None and "WGS-like" in desc
    assert (np.diff(batch.anchor_pos) >= 0).all()
    eng = engine_factory()
    eng.load_reference(chroms)
    n_close, n_far = check_workload(eng, chroms, batch)
    assert 0.15 * batch.n < n_close < 0.25 * batch.n        # at least 75 % of the reads leave GetCloseEnd empty-handed

"""Pins the search against the reference's OWN golden files (devtools/gold_standard/simulated_test.*,
committed under tests/golden/sim1chrVs2): reads -> search -> C++ classifiers/reporters -> the four
reports must be byte-identical to the gold standard (coverage columns masked).

  * test_oracle_pinned_by_gold_reports (CPU): the search is the CPU oracle -> this is what pins the
    oracle that every other parity test trusts.
  * test_gpu_path_reproduces_gold_reports (-m gpu): the search is the HIP path through the C ABI.
"""
import numpy as np
import pytest

from oracle import pyoracle
from pindel_amd import hostio, hostlib
from tests import golden_util as gu


def _load(tmp_path):
    fa, reads_txt = gu.unpack(tmp_path)
    chroms = hostio.load_fasta(fa)
    batch = hostio.read_pindel_text(reads_txt, [n for n, _ in chroms], [len(s) - 200000 for _, s in chroms])
    assert batch.n == 14862 and len(chroms[0][1]) == 400001
    return fa, reads_txt, chroms, batch


def test_oracle_pinned_by_gold_reports(tmp_path):
    fa, reads_txt, chroms, batch = _load(tmp_path)
    p = pyoracle.make_params()
    r = pyoracle.search_batch(p, [s for _, s in chroms], batch.seq, batch.seq_off, batch.anchor_strand,
                              batch.anchor_pos, batch.insert_size, batch.chr_id)
    # figures the survey measured with the reference binary itself
    assert int((r["close_cnt"] > 0).sum()) == 14862
    assert int((r["far_cnt"] > 0).sum()) == 10968
    assert int(r["rc_flag"].sum()) == 30
    co, cp = gu.csr_from_strided(r["close_cnt"], r["close_pts"])
    fo, fp = gu.csr_from_strided(r["far_cnt"], r["far_pts"])
    st = hostlib.default_settings(pyoracle.max_mismatch_table())
    prefix = str(tmp_path / "oracle")
    hostlib.call_from_points(fa, reads_txt, prefix, st, co, cp, fo, fp, r["rc_flag"])
    gu.assert_reports_match_gold(prefix)


def test_reports_do_not_depend_on_host_threads(tmp_path, monkeypatch):
    """The classifiers run their per-read loops and the reporters format their boxes on several threads
    (PGH_THREADS; reads reach the boxes and boxes reach the files in the sequential order): same bytes, = gold."""
    fa, reads_txt, chroms, batch = _load(tmp_path)
    p = pyoracle.make_params()
    r = pyoracle.search_batch(p, [s for _, s in chroms], batch.seq, batch.seq_off, batch.anchor_strand,
                              batch.anchor_pos, batch.insert_size, batch.chr_id)
    co, cp = gu.csr_from_strided(r["close_cnt"], r["close_pts"])
    fo, fp = gu.csr_from_strided(r["far_cnt"], r["far_pts"])
    st = hostlib.default_settings(pyoracle.max_mismatch_table())
    outs = {}
    for threads in ("1", "2", "7", "16"):
        monkeypatch.setenv("PGH_THREADS", threads)
        prefix = str(tmp_path / f"t{threads}")
        hostlib.call_from_points(fa, reads_txt, prefix, st, co, cp, fo, fp, r["rc_flag"])
        gu.assert_reports_match_gold(prefix)
        outs[threads] = [open(f"{prefix}_{s}", "rb").read() for s in gu.SUFFIXES]
    assert outs["1"] == outs["2"] == outs["7"] == outs["16"]


def test_reports_thread_independent_on_a_larger_synthetic_set(tmp_path, monkeypatch):
    """40 000 synthetic reads (all SV types) on a 2-Mbp reference, points from the oracle: several thousand events in
    hundreds of boxes; the four reports are the same bytes on 1 and on 8 host threads."""
    from pindel_amd import synth
    n, length = 40_000, 2_000_000
    ref = synth.make_reference(length, seed=31)
    biol = ref[100000:-100000]
    fa = tmp_path / "ref.fa"
    with open(fa, "wb") as f:
        f.write(b">chrS\n")
        for i in range(0, len(biol), 60):
            f.write(biol[i:i + 60] + b"\n")
    b = synth.make_reads(ref, n, seed=32)
    order = np.argsort(b.anchor_pos, kind="stable")
    seq = np.asarray(b.seq).reshape(n, 100)
    reads_txt = tmp_path / "reads.txt"
    with open(reads_txt, "wb") as f:
        for k, i in enumerate(order):
            f.write(b"@r%d/1\n" % k + seq[i].tobytes() + b"\n" + bytes([b.anchor_strand[i]]) +
                    b"\tchrS\t%d\t60\t500\tS1\n" % int(b.anchor_pos[i]))
    p = pyoracle.make_params()
    r = pyoracle.search_batch(p, [ref], seq[order].reshape(-1), (np.arange(n + 1) * 100).astype(np.uint64), b.anchor_strand[order],
                              b.anchor_pos[order], b.insert_size[order], b.chr_id[order])
    co, cp = gu.csr_from_strided(r["close_cnt"], r["close_pts"])
    fo, fp = gu.csr_from_strided(r["far_cnt"], r["far_pts"])
    st = hostlib.default_settings(pyoracle.max_mismatch_table())
    outs = {}
    for threads in ("1", "8"):
        monkeypatch.setenv("PGH_THREADS", threads)
        prefix = str(tmp_path / f"t{threads}")
        hostlib.call_from_points(str(fa), str(reads_txt), prefix, st, co, cp, fo, fp, r["rc_flag"])
        outs[threads] = [open(f"{prefix}_{s}", "rb").read() for s in gu.SUFFIXES]
    assert outs["1"] == outs["8"]
    assert outs["1"][0].count(b"\tD ") > 300 and outs["1"][1].count(b"\tI ") > 300      # hundreds of events


@pytest.mark.gpu
def test_gpu_path_reproduces_gold_reports(tmp_path, engine_factory):
    from pindel_amd import binding
    fa, reads_txt, chroms, batch = _load(tmp_path)
    eng = engine_factory()
    eng.load_fasta(fa)                      # the library's own FASTA loader
    assert eng.reference_info() == [("1", 400001)]
    res = eng.search_batch(batch)
    assert int((res.close_off[1:] > res.close_off[:-1]).sum()) == 14862
    assert int((res.far_off[1:] > res.far_off[:-1]).sum()) == 10968
    assert int(res.rc_flag.sum()) == 30
    cp = binding.expand_runs(res.close_runs)
    fp = binding.expand_runs(res.far_runs)
    # CSR over points from CSR over runs
    def point_off(off, runs):
        per_run = (runs["len_last"].astype(np.int64) - runs["len_first"].astype(np.int64) + 1)
        cum = np.concatenate([[0], np.cumsum(per_run)])
        return cum[off.astype(np.int64)].astype(np.uint64)
    co = point_off(res.close_off, res.close_runs)
    fo = point_off(res.far_off, res.far_runs)
    st = hostlib.default_settings(eng.max_mismatch_table())
    prefix = str(tmp_path / "gpu")
    hostlib.call_from_points(fa, reads_txt, prefix, st, co, cp, fo, fp, res.rc_flag)
    gu.assert_reports_match_gold(prefix)
    # the same through pg_search_batch_multi (three contexts, contiguous shards, one host thread each)
    others = [engine_factory(), engine_factory()]
    for e in others:
        e.load_fasta(fa)
    multi = binding.Engine.search_batch_multi([eng] + others, batch)
    assert np.array_equal(multi.close_off, res.close_off) and np.array_equal(multi.far_off, res.far_off)
    assert np.array_equal(multi.rc_flag, res.rc_flag)
    assert multi.close_runs.tobytes() == res.close_runs.tobytes() and multi.far_runs.tobytes() == res.far_runs.tobytes()
    prefix = str(tmp_path / "gpu_multi")
    hostlib.call_from_points(fa, reads_txt, prefix, st, point_off(multi.close_off, multi.close_runs), binding.expand_runs(multi.close_runs),
                             point_off(multi.far_off, multi.far_runs), binding.expand_runs(multi.far_runs), multi.rc_flag)
    gu.assert_reports_match_gold(prefix)


@pytest.mark.gpu
def test_command_line_reproduces_gold_reports(tmp_path):
    """pindel_pg (C++ host + C ABI): -f/-p/-o like `pindel -f ... -p ... -o ...`."""
    import os
    import subprocess
    from pindel_amd import binding
    fa, reads_txt = gu.unpack(tmp_path)
    exe = os.path.join(os.path.dirname(binding.LIB_PATH), "pindel_pg")
    prefix = str(tmp_path / "cli")
    out = subprocess.run([exe, "-f", fa, "-p", reads_txt, "-o", prefix, "-T", "1"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "close end 14862, far end 10968" in out.stdout
    # the reference's own cross-check lines (ReportCloseAndFarEndCounts, "Checksum of far ends")
    assert "Total: 14862;\tClose_end_found 14862;\tFar_end_found 10968;" in out.stdout
    assert "Far ends already mapped 10968" in out.stdout and "Checksum of far ends: " in out.stdout
    gu.assert_reports_match_gold(prefix)
    # the close end in ReadBuffer-sized flushes (here 4 000 reads: four flushes), the far end on their filtered union
    prefix = str(tmp_path / "cli_flush")
    out = subprocess.run([exe, "-f", fa, "-p", reads_txt, "-o", prefix, "--flush-reads", "4000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "close end 14862, far end 10968" in out.stdout
    gu.assert_reports_match_gold(prefix)
    bad = subprocess.run([exe, "-f", fa, "-p", reads_txt, "-o", prefix, "-w", "0"], capture_output=True, text=True)
    assert bad.returncode == 2 and "-w must be" in bad.stderr


@pytest.mark.gpu
def test_command_line_sharded_over_devices(tmp_path):
    """`-G 0,0,0`: three contexts (here on the one visible GPU; on a node: `-G 0,1,...,7`), the reads of every
    bin sharded over them in contiguous ranges by one host thread per context -- same reports as one device,
    also with the reference's valueless switches on the command line and with BreakDancer hints."""
    import filecmp
    import os
    import subprocess
    from pindel_amd import binding
    fa, reads_txt = gu.unpack(tmp_path)
    exe = os.path.join(os.path.dirname(binding.LIB_PATH), "pindel_pg")
    prefix = str(tmp_path / "multi")
    out = subprocess.run([exe, "-f", fa, "-p", reads_txt, "-o", prefix, "-G", "0,0,0", "-k", "-s", "-l", "-T", "4"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "close end 14862, far end 10968" in out.stdout
    gu.assert_reports_match_gold(prefix)
    # flag errors are loud
    bad = subprocess.run([exe, "-f", fa, "-p", reads_txt, "-o", prefix, "-x", "foo"], capture_output=True, text=True)
    assert bad.returncode == 2 and "not a number" in bad.stderr
    bad = subprocess.run([exe, "-f", fa, "-p", reads_txt, "-o", prefix, "-x"], capture_output=True, text=True)
    assert bad.returncode == 2 and "lacking" in bad.stderr
    bad = subprocess.run([exe, "-f", fa, "-p", reads_txt, "-o", prefix, "--no-such-flag", "1"], capture_output=True, text=True)
    assert bad.returncode == 2 and "unknown argument" in bad.stderr


@pytest.mark.gpu
def test_two_contexts_over_two_shards_equal_the_whole_batch(engine_factory, tmp_path):
    """What bench.py --scaling strong and `pindel_pg -G` rely on, through the C ABI: two contexts, each searching a
    contiguous shard of ONE batch (ragged split), concatenated in order == the whole batch on one context."""
    from pindel_amd import shard, synth
    ref = [("chrS", synth.make_reference(1_200_000, seed=31))]
    batch = synth.make_reads(ref[0][1], 20_001, seed=32)
    a, b = engine_factory(), engine_factory()
    a.load_reference(ref)
    b.load_reference(ref)
    whole = a.search_batch(batch)
    cut = 7_123
    parts = [shard.result_arrays(a.search_batch(batch.slice(0, cut))), shard.result_arrays(b.search_batch(batch.slice(cut, batch.n)))]
    cat = shard.concat_results(parts)
    w = shard.result_arrays(whole)
    for k in ("close_off", "far_off", "rc_flag"):
        assert np.array_equal(w[k], cat[k]), k
    for k in ("close_runs", "far_runs"):
        assert w[k].tobytes() == cat[k].tobytes(), k

    class R:
        pass
    r = R()
    for k, v in cat.items():
        setattr(r, k, v)
    assert shard.digest_hex(shard.read_digests(whole)) == shard.digest_hex(shard.read_digests(r))
    # the same behind the C ABI: pg_search_batch_multi over three contexts (one host thread each)
    from pindel_amd import binding
    c = engine_factory()
    c.load_reference(ref)
    multi = shard.result_arrays(binding.Engine.search_batch_multi([a, b, c], batch))
    for k in ("close_off", "far_off", "rc_flag"):
        assert np.array_equal(w[k], multi[k]), k
    for k in ("close_runs", "far_runs"):
        assert w[k].tobytes() == multi[k].tobytes(), k


@pytest.mark.gpu
def test_command_line_breakdancer_hints(tmp_path):
    """`-b file` alone changes nothing (what 0.2.5b9 does for Pindel-text input); with `--bd-hints on` the
    command line searches the events' windows before the ranges: same reports as the CPU oracle given the
    same per-read window clusters (pgh_bd_query) and the C++ reporters."""
    import ctypes as C
    import filecmp
    import os
    import subprocess
    from pindel_amd import binding
    fa, reads_txt, chroms, batch = _load(tmp_path)
    exe = os.path.join(os.path.dirname(binding.LIB_PATH), "pindel_pg")
    # events between random places of the 200 kb chromosome and places 1-40 kb away
    rng = np.random.default_rng(9)
    lines = ["#Chr1\tPos1\tOri1\tChr2\tPos2\tOri2\tType\tSize\tScore\tReads"]
    for _ in range(300):
        p1 = int(rng.integers(1000, 190000))
        p2 = min(p1 + int(rng.integers(600, 40000)), 199000)
        lines.append(f"1\t{p1}\t5+0-\t1\t{p2}\t0+5-\tDEL\t{p2 - p1}\t99\t5")
    bd_path = tmp_path / "bd.txt"
    bd_path.write_text("\n".join(lines) + "\n")

    def run(prefix, *extra):
        out = subprocess.run([exe, "-f", fa, "-p", reads_txt, "-o", str(tmp_path / prefix), *extra],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        return out.stdout

    run("plain")
    so = run("ignored", "-b", str(bd_path))
    assert "BD events: 300" in so
    for sfx in ("_D", "_SI", "_TD", "_INV"):
        assert filecmp.cmp(tmp_path / ("plain" + sfx), tmp_path / ("ignored" + sfx), shallow=False)
    run("hinted", "-b", str(bd_path), "--bd-hints", "on")

    # the same with the oracle: close end, clusters of the last close-end point, far end with the windows
    p = pyoracle.make_params()
    seqs = [s for _, s in chroms]
    args = (batch.seq, batch.seq_off, batch.anchor_strand, batch.anchor_pos, batch.insert_size, batch.chr_id)
    close = pyoracle.search_batch(p, seqs, *args, do_far=False)
    last = np.array([int(close["close_pts"][i][close["close_cnt"][i] - 1]["abs_loc"]) if close["close_cnt"][i] else 0
                     for i in range(batch.n)], dtype=np.uint32)
    L = hostlib.lib()
    L.pgh_bd_query.argtypes = [C.c_char_p, C.c_uint32, C.c_int32, C.POINTER(C.c_char_p), C.c_int32, C.c_uint32,
                               C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    names = (C.c_char_p * 1)(b"1")
    off = np.zeros(batch.n + 1, dtype=np.uint64)
    cap = 64 * batch.n
    win = np.zeros(3 * cap, dtype=np.int32)
    nev = C.c_uint64()
    rc = L.pgh_bd_query(str(bd_path).encode(), 100000, 1, names, 0, 100000, 100000 + 5_000_000, batch.n,
                        last.ctypes.data, off.ctypes.data, win.ctypes.data, cap, C.byref(nev))
    assert rc == 0 and nev.value == 300
    has_close = close["close_cnt"] > 0
    cnt = np.diff(off.astype(np.int64))
    cnt[~has_close] = 0
    keep = np.repeat(has_close, np.diff(off.astype(np.int64)))
    bd = np.zeros(int(cnt.sum()), dtype=pyoracle.WINDOW_DTYPE)
    w3 = win[:3 * int(off[-1])].reshape(-1, 3)[keep]
    bd["chr_id"], bd["start"], bd["end"] = w3[:, 0], w3[:, 1], w3[:, 2]
    bd_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
    assert len(bd) > 1000
    r = pyoracle.search_batch(p, seqs, *args, bd=bd, bd_off=bd_off)
    co, cp = gu.csr_from_strided(r["close_cnt"], r["close_pts"])
    fo, fp = gu.csr_from_strided(r["far_cnt"], r["far_pts"])
    st = hostlib.default_settings(pyoracle.max_mismatch_table())
    hostlib.call_from_points(fa, reads_txt, str(tmp_path / "oracle_hinted"), st, co, cp, fo, fp, r["rc_flag"])
    differs_from_plain = False
    for sfx in ("_D", "_SI", "_TD", "_INV"):
        assert filecmp.cmp(tmp_path / ("hinted" + sfx), tmp_path / ("oracle_hinted" + sfx), shallow=False), sfx
        differs_from_plain |= not filecmp.cmp(tmp_path / ("hinted" + sfx), tmp_path / ("plain" + sfx), shallow=False)
    # (whether the hints change a report depends on the data; they must at least have been searched)
    assert differs_from_plain or int((r["far_cnt"] > 0).sum()) >= 10968

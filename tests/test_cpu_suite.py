"""-m "not gpu": oracle sanity, host logic, and that the C-ABI library loads and exports every
symbol include/pindel_pg.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import pyoracle
from pindel_amd import binding, hostio, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_max_mismatch_table_matches_survey():
    # value 2 for L=5-32, 3 for 33-74, 4 for 75-125, 5 for 126-180, 6 for 181-238; 0 for L<4
    t = pyoracle.max_mismatch_table()
    assert list(t[:4]) == [0, 0, 0, 0]
    assert set(t[5:33]) == {2} and set(t[33:75]) == {3} and set(t[75:126]) == {4}
    assert set(t[126:181]) == {5} and set(t[181:239]) == {6}
    assert np.all(np.diff(t[4:].astype(int)) >= 0), "g_maxMismatch must be monotone for the kernel's abort rule"


def test_header_symbols_exported():
    binding.build()
    lib = ctypes.CDLL(binding.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "pindel_pg.h")).read()
    declared = set(re.findall(r"\b(pg_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"pg_ctx", "pg_result"}
    assert declared == set(binding.EXPORTS), declared ^ set(binding.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_library_holds_both_kernel_families(tmp_path):
    """The shipped library carries the search kernels twice: generic (the five search parameters as kernel arguments) and compiled
    for Pindel's default parameter set (`DEF`).  Read from the code object, no GPU needed."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump in this image")
    binding.build()
    lib = str(tmp_path / "lib.so")
    shutil.copy(binding.LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", lib], cwd=tmp_path, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    cos = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert cos, "no gfx950 code object in the library"
    syms = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", str(tmp_path / cos[0])], stdout=subprocess.PIPE,
                          text=True, check=True).stdout             # (the kernels' names in the code object's metadata)
    for nb in (1, 2, 3, 4, 8):
        for mode in (1, 2, 3):                   # close end, far end, both
            assert f"pg_search_kernelILi{nb}ELi3EjLi{mode}ELb0E" in syms, ("generic kernel missing", nb, mode)
            assert f"pg_search_kernelILi{nb}ELi3EjLi{mode}ELb1E" in syms, ("default-parameter kernel missing", nb, mode)


def test_struct_layouts_match_header():
    assert ctypes.sizeof(binding.PgParams) == 56
    assert binding.RUN_DTYPE.itemsize == 12 and binding.POINT_DTYPE.itemsize == 12
    assert pyoracle.POINT_DTYPE.descr == binding.POINT_DTYPE.descr


def test_no_gpu_means_loud_failure():
    """Without a HIP device the product must fail, not fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(binding.PgError) as e:
        binding.Engine()
    assert e.value.code == binding.PG_E_DEVICE


def test_expand_runs_host_helper():
    runs = np.zeros(2, dtype=binding.RUN_DTYPE)
    runs[0] = (1000, 8, 10, 1, 0, 0)                       # FORWARD / SENSE
    runs[1] = (5000, 20, 21, 0, 3, 2)                      # BACKWARD / ANTISENSE
    pts = binding.expand_runs(runs)
    assert list(pts["abs_loc"]) == [1000, 1001, 1002, 5000, 4999]
    assert list(pts["length"]) == [8, 9, 10, 20, 21]
    assert list(pts["direction"]) == [b"+"] * 3 + [b"-"] * 2
    assert list(pts["strand"]) == [b"+"] * 3 + [b"-"] * 2
    assert list(pts["chr_id"]) == [0, 0, 0, 2, 2]


def test_fasta_loader_quirks(tmp_path):
    p = tmp_path / "t.fa"
    p.write_text(">c1 desc\nacgtnRYx\nAC\n>c2\nGGT\n")
    chroms = hostio.load_fasta(p, spacer=5)
    assert chroms[0] == ("c1", b"NNNNN" + b"ACGTNNNNAC" + b"NNNNN")
    # last record: the reference's extraction loop repeats the final base (pindel.cpp:288-299)
    assert chroms[1] == ("c2", b"NNNNN" + b"GGTT" + b"NNNNN")


def test_oracle_edge_cases():
    """First base N disables the far-end search; IUPAC codes never match; short reads yield nothing."""
    ref = [("c", synth.make_reference(300_000, seed=9))]
    b = synth.make_reads(ref[0][1], 400, seed=10)
    seq = b.seq.copy()
    off = b.seq_off.astype(np.int64)
    seq[off[:50]] = ord("N")                 # first base N
    seq[off[50:100] + 30] = ord("R")         # an IUPAC code inside the read
    p = pyoracle.make_params()
    r = pyoracle.search_batch(p, [ref[0][1]], seq, b.seq_off, b.anchor_strand, b.anchor_pos,
                              b.insert_size, b.chr_id)
    plain = pyoracle.search_batch(p, [ref[0][1]], b.seq, b.seq_off, b.anchor_strand, b.anchor_pos,
                                  b.insert_size, b.chr_id)
    assert (plain["far_cnt"] > 0).sum() > 100
    # reads whose (final-orientation) first base is N have no far end
    for i in range(50):
        first = r["seq"][off[i]]
        if first == ord("N"):
            assert r["far_cnt"][i] == 0
    tiny = hostio.batch_from_lists([b"ACGTACG"], [b"+"], [1000], [500], [0])
    rt = pyoracle.search_batch(p, [ref[0][1]], tiny.seq, tiny.seq_off, tiny.anchor_strand, tiny.anchor_pos,
                               tiny.insert_size, tiny.chr_id)
    assert rt["close_cnt"][0] == 0 and rt["far_cnt"][0] == 0


def test_fast_exchange_sort_equals_the_reference_loop():
    """The reporters' order of equal reads comes from the reference's O(n^2) exchange sort that also
    swaps equal elements (reporter.cpp:932-942); the host library replaces it by a stable sort of the
    reversed sequence -- identical permutation on random inputs with many ties."""
    import ctypes as C
    from pindel_amd import hostlib
    L = hostlib.lib()
    L.pgh_test_exchange_sort.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.pgh_test_exchange_sort.restype = None
    rng = np.random.default_rng(11)
    for trial in range(400):
        n = int(rng.integers(0, 200))
        keys = rng.integers(0, int(rng.integers(1, 12)), n).astype(np.int32)
        a = np.zeros(n, dtype=np.uint32)
        b = np.zeros(n, dtype=np.uint32)
        L.pgh_test_exchange_sort(keys.ctypes.data, n, a.ctypes.data, b.ctypes.data)
        assert np.array_equal(a, b), (trial, keys.tolist())
        assert np.all(np.diff(keys[a]) >= 0)


# ---- BreakDancer window hints (pindel_amd/csrc/host/pg_bdhints.hpp) against an independent restatement
def _bd_restatement(lines, spacer, chr_names, chr_id, start, end, queries):
    """bddata.cpp:91-136, 814-979 and control_state.cpp:71-131 restated in Python (PARITY UNPINNED: the
    reference never uses -b events on the text-input path, so there is no reference output to compare with)."""
    SPAN = 200
    ev = []
    for ln in lines:
        if ln.startswith("#") or not ln.strip():
            continue
        f = ln.split()
        c1, p1, c2, p2 = f[0], int(f[1]) + spacer, f[3], int(f[4]) + spacer
        if c1 == c2 and abs(p1 - p2) < 500:
            continue
        ev.append((c1, p1, c2, p2))
        ev.append((c2, p2, c1, p1))
    ev.sort()
    sow = lambda p: p - SPAN if p >= SPAN else 0
    eow = lambda p: p + SPAN
    ws = start - 3000 if start >= 3000 else 0
    we = end + 3000
    chr_ = chr_names[chr_id]
    import bisect
    first = bisect.bisect_left(ev, (chr_, ws, "", 0))
    # upper_bound with key (chr, we, "", 0): every event whose first coordinate is (chr, we) sorts after the key
    last = bisect.bisect_left(ev, (chr_, we, "", 0))
    while last < len(ev) and ev[last][:2] == (chr_, we) and ev[last][2:] <= ("", 0):
        last += 1
    mask = [0] * (we - ws + 1)
    clusters = [[]]
    b = e = first
    index = 0
    for pos in range(ws, we):
        changed = False
        k = b
        while k < e and pos > eow(ev[k][1]):
            b += 1
            k += 1
            changed = True
        for k in range(e, last):
            s_ = sow(ev[k][1])
            if pos < s_:
                break
            if pos == s_:
                e += 1
                changed = True
        if b != e:
            if changed:
                index += 1
                sub = sorted(ev[b:e], key=lambda t: (t[2], t[3], t[0], t[1]))
                cl, i = [], 0
                while i < len(sub):
                    cid = chr_names.index(sub[i][2])
                    w = [cid, sow(sub[i][3]), eow(sub[i][3])]
                    while i + 1 < len(sub) and sub[i + 1][2] == sub[i][2] and sow(sub[i + 1][3]) <= eow(sub[i][3]) + 1:
                        i += 1
                        w[2] = eow(sub[i][3])
                    cl.append(tuple(w))
                    i += 1
                clusters.append(cl)
            mask[pos - ws] = index
    out = []
    size = 1 + we - ws
    for q in queries:
        rel = (q - ws) & 0xFFFFFFFF
        if rel > size:
            out.append([])
        elif q > ws and rel < size - 1:
            out.append(clusters[mask[rel]])
        else:
            out.append([])
    return out, len(ev) // 2


def test_breakdancer_hints_match_the_restatement(tmp_path):
    import ctypes as C
    from pindel_amd import hostlib
    L = hostlib.lib()
    L.pgh_bd_query.argtypes = [C.c_char_p, C.c_uint32, C.c_int32, C.POINTER(C.c_char_p), C.c_int32, C.c_uint32,
                               C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    rng = np.random.default_rng(5)
    names = ["chrA", "chrB", "chr10"]
    spacer = 100000
    for trial in range(6):
        lines = ["#Chr1\tPos1\tOrientation1\tChr2\tPos2\tOrientation2\tType\tSize"]
        for _ in range(int(rng.integers(5, 400))):
            c1 = names[int(rng.integers(0, 3))]
            c2 = c1 if rng.random() < 0.8 else names[int(rng.integers(0, 3))]
            p1 = int(rng.integers(1, 60000))
            p2 = p1 + int(rng.integers(-300, 30000)) if rng.random() < 0.9 else int(rng.integers(1, 60000))
            p2 = max(p2, 1)
            lines.append(f"{c1}\t{p1}\t10+0-\t{c2}\t{p2}\t0+10-\tDEL\t{abs(p2 - p1)}\t99\t10")
            if rng.random() < 0.1:                       # clustered events: overlapping windows
                lines.append(f"{c1}\t{p1 + int(rng.integers(0, 150))}\t3+0-\t{c2}\t{p2 + int(rng.integers(0, 350))}\t0+3-\tDEL\t1\t50\t3")
        path = tmp_path / f"bd{trial}.txt"
        path.write_text("\n".join(lines) + "\n")
        chr_id = int(rng.integers(0, 3))
        start = spacer + int(rng.integers(0, 20000))
        end = start + int(rng.integers(5000, 50000))
        q = np.concatenate([rng.integers(start - 4000, end + 4000, 3000),
                            [start - 3000, start - 2999, end + 2999, end + 3000, end + 3001, 0]]).astype(np.uint32)
        want, n_ev = _bd_restatement(lines, spacer, names, chr_id, start, end, [int(x) for x in q])
        arr = (C.c_char_p * 3)(*[n.encode() for n in names])
        off = np.zeros(len(q) + 1, dtype=np.uint64)
        cap = 40 * len(q)
        win = np.zeros(3 * cap, dtype=np.int32)
        nev = C.c_uint64()
        rc = L.pgh_bd_query(str(path).encode(), spacer, 3, arr, chr_id, start, end, len(q), q.ctypes.data,
                            off.ctypes.data, win.ctypes.data, cap, C.byref(nev))
        assert rc == 0 and nev.value == n_ev
        got = [[tuple(int(v) for v in win[3 * k:3 * k + 3]) for k in range(int(off[i]), int(off[i + 1]))]
               for i in range(len(q))]
        assert got == want
        assert sum(len(g) for g in got) > 0
    bad = tmp_path / "bad.txt"
    bad.write_text("chrA\t10\t+\tchrA\tnotanumber\t-\n")
    rc = L.pgh_bd_query(str(bad).encode(), spacer, 3, arr, 0, spacer, spacer + 1000, 0, None, off.ctypes.data,
                        win.ctypes.data, cap, C.byref(nev))
    assert rc == 1 and nev.value == 0                    # ignored, like CheckBreakDancerFileFormat


def test_pindel_text_loader_framing_and_errors(tmp_path):
    """load_pindel_text: three lines per record from the top, the list ends at an empty name line or an incomplete record,
    the FIRST malformed record is the error (the parser runs on several threads)."""
    import ctypes as C
    import numpy as np
    from pindel_amd import hostlib
    from oracle import pyoracle
    fa = tmp_path / "r.fa"
    fa.write_text(">c1\n" + "ACGT" * 500 + "\n")
    st = hostlib.default_settings(pyoracle.max_mismatch_table())
    empty_pts = np.zeros(0, dtype=pyoracle.POINT_DTYPE)

    def run(text, n):
        rp = tmp_path / "reads.txt"
        rp.write_bytes(text)
        off = np.zeros(n + 1, dtype=np.uint64)
        hostlib.call_from_points(str(fa), str(rp), str(tmp_path / "o"), st, off, empty_pts, off, empty_pts, np.zeros(n, dtype=np.uint8))

    rec = lambda k, strand=b"+", name=None: (name or b"@r%d/1" % k) + b"\n" + b"ACGTACGTAC" * 3 + b"\n" + strand + b"\tc1\t%d\t60\t500\tS\n" % (100 + k)
    good = b"".join(rec(k) for k in range(20000))
    run(good, 20000)                                                        # 20000 records, all accepted
    run(good + b"@tail/1\nACGT\n", 20000)                                   # incomplete last record: dropped
    run(b"".join(rec(k) for k in range(100)) + b"\n" + good, 100)           # an empty name line ends the list
    for bad_at, bad, msg in ((15000, rec(15000, name=b"r15000"), "Something wrong with the read name: r15000"),
                             (7, rec(7, strand=b"x"), "+/- expected in read @r7/1")):
        text = b"".join(bad if k == bad_at else rec(k) for k in range(20000))
        with pytest.raises(RuntimeError, match=msg.replace("+", "\\+")):
            run(text, 20000)
    # two malformed records: the first one in the file is reported
    text = b"".join(rec(k, strand=b"x") if k in (300, 19000) else rec(k) for k in range(20000))
    with pytest.raises(RuntimeError, match="expected in read @r300/1"):
        run(text, 20000)


def test_adapter_compiles_against_reference_shapes(tmp_path):
    """The reference-side binding of INTEGRATION.md (pg_adapter.hpp: CloseEndBatch and both SearchFarEnds overloads)
    instantiated against the public interface of the reference's SPLIT_READ / SortedUniquePoints / UniquePoint
    (tests/ref_shapes.hpp restates src/pindel.h:137-197, 265-383: no reserve, no iterators, private storage)."""
    import subprocess
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "pindel_amd", "csrc", "host"), "-I" + os.path.join(ROOT, "tests"),
           os.path.join(ROOT, "tests", "adapter_ref_shapes.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the mock is not more generous than the reference: a vector-only member must fail to compile against it
    probe = tmp_path / "probe.cpp"
    probe.write_text('#include "ref_shapes.hpp"\nvoid f(SortedUniquePoints &p) { p.reserve(4); }\n')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "tests"), str(probe)],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "reserve" in r.stderr
    # ... and the same adapters still serve this repository's own read type (std::vector point lists)
    cmd[-1] = os.path.join(ROOT, "pindel_amd", "csrc", "host", "pindel_pg_main.cpp")
    r = subprocess.run([c for c in cmd if c != "-Werror"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr

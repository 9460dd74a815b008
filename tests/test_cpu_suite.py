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

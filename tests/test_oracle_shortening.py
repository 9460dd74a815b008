"""CPU: the oracle's restatement of setUnmatchedSeq's shortening (pindel.cpp:142-169, 966-970, 2037-2048, 2545) against
expectations written out from the source -- see tests/shortening_cases.py for the derivation."""
import numpy as np
import pytest

from pindel_amd import synth
from tests import shortening_cases as sc
from tests.parity import run_oracle


@pytest.fixture(scope="module")
def ref():
    return [("chrS", synth.make_reference(600_000, seed=31))]


@pytest.mark.parametrize("junk,read_len", [(b"R", 100), (b"RY", 100), (b"*", 100), (b"R", 125)])
def test_leading_junk_is_stripped_by_the_first_reverse_complement(ref, junk, read_len):
    clean = sc.clean_reads(ref[0][1], 400, read_len, seed=40 + len(junk) + read_len)
    oc = run_oracle({}, ref, clean)
    dirty = sc.lead_case(clean, junk)
    od = run_oracle({}, ref, dirty)
    hit = 0
    for i in range(clean.n):
        if oc["rc_flag"][i] != 0 or oc["close_cnt"][i] == 0:
            continue                     # (the clean read itself needed a retry: not this case)
        if od["rc_flag"][i] == 0 and od["close_cnt"][i] > 0:
            continue                     # attempt 0 placed the junk-led read by chance: nothing was stripped
        # attempt 0 found nothing: ONE reverse complement, the junk gone, the clean read's result
        assert od["rc_flag"][i] == 1, i
        assert od["len_out"][i] == read_len, (i, od["len_out"][i])
        assert sc.same_points(od, i, oc, i), i
        a, b = int(dirty.seq_off[i]), int(dirty.seq_off[i]) + read_len
        assert od["seq"][a:b].tobytes() == sc.seqs_of(clean)[i]           # UnmatchedSeq as GetCloseEnd left it
        hit += 1
    assert hit > 200
    if read_len == 125:                  # ReadLength 126 -> 125: one mismatch level fewer (g_maxMismatch 5 -> 4)
        from oracle import pyoracle
        t = pyoracle.max_mismatch_table()
        assert t[126] == 5 and t[125] == 4


def test_trailing_junk_goes_with_the_second_reverse_complement(ref):
    unmoved = sc.clean_reads(ref[0][1], 400, 100, seed=77)
    o0 = run_oracle({}, ref, unmoved)
    clean = sc.moved(unmoved)
    oc = run_oracle({}, ref, clean)
    dirty = sc.trail_case(clean, b"RK")
    od = run_oracle({}, ref, dirty)
    hit = 0
    for i in range(clean.n):
        # the clean read with the moved anchor must find the SAME close end as with the anchor in place -- which lies outside the
        # moved R = 0 window, so it was attempt 3 (R = 1, the read as it came) that found it ...
        nc = int(o0["close_cnt"][i])
        if not (oc["rc_flag"][i] == 0 and nc > 0 and int(oc["close_cnt"][i]) == nc and
                oc["close_pts"][i][:nc].tobytes() == o0["close_pts"][i][:nc].tobytes()):
            continue
        if od["rc_flag"][i] != 2 or od["close_cnt"][i] == 0:
            continue                     # ... and the dirty one at attempt 3 too
        assert od["len_out"][i] == 100
        assert sc.same_points(od, i, oc, i), i
        a = int(dirty.seq_off[i])
        assert od["seq"][a:a + 100].tobytes() == sc.seqs_of(clean)[i]
        hit += 1
    assert hit > 150
    # a read attempt 0 cannot seed (its first consumed character is junk) never ends with rc_flag 0 and points
    assert not np.any((od["rc_flag"] == 0) & (od["close_cnt"] > 0))


def test_inner_junk_keeps_the_length_and_flags_two_reverse_complements(ref):
    clean = sc.moved(sc.clean_reads(ref[0][1], 300, 100, seed=91))
    dirty = sc.inner_case(clean, 50)
    od = run_oracle({}, ref, dirty)
    assert (od["len_out"] == 100).all()
    two = od["rc_flag"] == 2
    assert two.sum() > 100                                              # attempt 3 or nothing: the R is a NUL now
    for i in np.nonzero(two)[0][:50]:
        a = int(dirty.seq_off[i])
        s = od["seq"][a:a + 100].tobytes()
        assert s[50] == 0 and s[:50] == sc.seqs_of(dirty)[i][:50] and s[51:] == sc.seqs_of(dirty)[i][51:]


def test_a_read_of_nothing_but_junk(ref):
    b = sc.batch_of([b"R" * 40, b"*" * 12, b"RRRRACGT"], [ord("+"), ord("-"), ord("+")], [200000, 200000, 200000], [500, 500, 500])
    o = run_oracle({}, ref, b)
    assert (o["close_cnt"] == 0).all() and (o["far_cnt"] == 0).all()
    assert list(o["len_out"]) == [0, 0, 4] and list(o["rc_flag"]) == [2, 2, 2]

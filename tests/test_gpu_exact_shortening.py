"""-m gpu: reads with characters outside ACGTN through the HIP path against the oracle -- the reference shortens a read that
BEGINS (first reverse complement) or ENDS (second) with such characters (setUnmatchedSeq, pindel.cpp:142-169, 2545; the derivation
and the hand-written expectations: tests/shortening_cases.py, tests/test_oracle_shortening.py).  The pack kernel lists these reads,
pg_search_exact_kernel searches them again with the shortening; every entry of the ABI goes through it."""
import numpy as np
import pytest

from pindel_amd import synth
from pindel_amd.synth import ReadBatch
from tests import shortening_cases as sc
from tests.parity import compare_result, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    return [("chrS", synth.make_reference(600_000, seed=31))]


def _concat(batches):
    off = [np.zeros(1, dtype=np.uint64)]
    base = 0
    for b in batches:
        off.append(b.seq_off[1:].astype(np.uint64) + np.uint64(base))
        base += len(b.seq)
    return ReadBatch(seq=np.concatenate([b.seq for b in batches]), seq_off=np.concatenate(off),
                     anchor_strand=np.concatenate([b.anchor_strand for b in batches]),
                     anchor_pos=np.concatenate([b.anchor_pos for b in batches]),
                     insert_size=np.concatenate([b.insert_size for b in batches]),
                     chr_id=np.concatenate([b.chr_id for b in batches]))


def _cases(ref):
    c100 = sc.clean_reads(ref[0][1], 300, 100, seed=41)
    c125 = sc.clean_reads(ref[0][1], 200, 125, seed=42)
    c150 = sc.clean_reads(ref[0][1], 200, 150, seed=43)
    mv = sc.moved(sc.clean_reads(ref[0][1], 300, 100, seed=77))
    odd = sc.batch_of([b"R" * 40, b"*" * 12, b"RRRRACGT", b"ACGTACGTACGTR", b"NACGTTGCAACGTTGCAACGTR*"],
                      [ord("+"), ord("-"), ord("+"), ord("-"), ord("+")], [200000] * 5, [500] * 5)
    return [sc.lead_case(c100, b"R"), sc.lead_case(c100, b"RY"), sc.lead_case(c100, b"*"), sc.lead_case(c125, b"R"),
            sc.lead_case(c150, b"KM"), sc.trail_case(mv, b"RK"), sc.trail_case(mv, b"r"), sc.inner_case(mv, 50),
            sc.inner_case(c100, 0), sc.inner_case(c100, 99), sc.lead_case(sc.trail_case(mv, b"S"), b"W"), odd]


def test_every_case_against_the_oracle(engine_factory, ref):
    eng = engine_factory()
    eng.load_reference(ref)
    for k, batch in enumerate(_cases(ref)):
        orc = run_oracle({}, ref, batch)
        gpu = eng.search_batch(batch)
        compare_result(gpu, orc, batch.n)
    # the flags the cases were built for do occur
    lead = sc.lead_case(sc.clean_reads(ref[0][1], 300, 100, seed=41), b"R")
    o = run_oracle({}, ref, lead)
    assert (o["rc_flag"] == 1).sum() > 150 and (o["len_out"][o["rc_flag"] == 1] == 100).all()
    trail = sc.trail_case(sc.moved(sc.clean_reads(ref[0][1], 300, 100, seed=77)), b"RK")
    o = run_oracle({}, ref, trail)
    assert ((o["rc_flag"] == 2) & (o["close_cnt"] > 0)).sum() > 100


def test_scattered_among_ordinary_reads_on_every_entry(engine_factory, ref, pg_env):
    """a few junk-ended reads inside a batch of ordinary ones: host path in several chunks, device-resident batch searched twice,
    the two seams with the close end's rc_flag (1 and 2) handed to the far end"""
    eng = engine_factory()
    eng.load_reference(ref)
    plain = synth.make_reads(ref[0][1], 20000, seed=5)
    cs = _cases(ref)
    parts = []
    for k in range(10):
        parts.append(plain.slice(2000 * k, 2000 * (k + 1)))
        parts.append(cs[k].slice(0, 25))
    parts.append(cs[11])
    batch = _concat(parts)
    orc = run_oracle({}, ref, batch)
    assert (orc["rc_flag"] == 2).sum() > 20 and (orc["rc_flag"] == 1).sum() > 1000
    compare_result(eng.search_batch(batch), orc, batch.n)
    pg_env.set("PG_HOST_CHUNK", "3000")
    compare_result(eng.search_batch(batch), orc, batch.n)
    pg_env.unset("PG_HOST_CHUNK")
    db = eng.upload(batch)
    for _ in range(2):
        eng.search_device(db)
        compare_result(eng.download(db), orc, batch.n)
    eng.repack(db)
    eng.search_device(db)
    compare_result(eng.download(db), orc, batch.n)
    eng.free_device_batch(db)
    close = eng.close_end_batch(batch)
    compare_result(close, orc, batch.n, check_far=False)
    compare_result(eng.far_end_batch(batch, close), orc, batch.n)


@pytest.mark.parametrize("kw", [dict(max_range_index=4), dict(additional_mismatch=2, min_close=6, max_mismatch_rate=0.05)])
def test_other_parameters(engine_factory, ref, kw):
    eng = engine_factory(**kw)
    eng.load_reference(ref)
    batch = _concat([c.slice(0, 60) for c in _cases(ref)[:11]])
    compare_result(eng.search_batch(batch), run_oracle(kw, ref, batch), batch.n)

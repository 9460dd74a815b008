"""-m gpu: the step as ONE launch -- pg_device_batch_pack_search, where pg_search_kernel builds the bit planes and records of each claim
itself before it searches them (PgDevBatch::soa).  Same results as pg_device_batch_repack + pg_device_batch_search: against the
oracle on small batches of every read-length class (PG_PACK_IN_PLACE_MIN lowered), and launch against launch at the size the
path switches on by itself.  The records are overwritten before every in-place step: what it searches is what it packed."""
import hashlib

import numpy as np
import pytest

from pindel_amd import synth
from tests import shortening_cases as sc
from tests.parity import compare_result, run_oracle
from tests.test_gpu_exact_shortening import _concat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    return [("chrP", synth.make_reference(700_000, seed=53))]


def _digest(res):
    h = hashlib.sha256()
    for a in (res.close_off, res.far_off, res.rc_flag, res.close_runs, res.far_runs):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("read_len", [36, 64, 100, 128, 150, 192, 250, 400])
def test_small_batches_against_the_oracle(engine_factory, ref, pg_env, read_len):
    pg_env.set("PG_PACK_IN_PLACE_MIN", "1")
    eng = engine_factory()
    eng.load_reference(ref)
    n = 3000 if read_len <= 150 else 1200
    plain = synth.make_reads(ref[0][1], n, read_len=read_len, seed=600 + read_len)
    parts = [plain]
    if read_len >= 100:          # reads the exact kernel searches again: the pack inside the search kernel lists them
        clean = sc.clean_reads(ref[0][1], 60, read_len, seed=61)
        parts += [sc.lead_case(clean, b"RY"), sc.trail_case(sc.moved(clean), b"K"), sc.inner_case(clean, read_len // 2)]
    batch = _concat(parts)
    orc = run_oracle({}, ref, batch)
    db = eng.upload(batch)
    for claim in ("16", "5", "64"):                  # claims that do and do not divide the parts of the launch
        pg_env.set("PG_PACK_CLAIM", claim)
        eng.scribble_records(db)
        eng.pack_search_device(db)
        assert eng.last_step_in_place()
        compare_result(eng.download(db), orc, batch.n)
    # ... and the batch stays searchable the ordinary way afterwards (records in place, exact list counted)
    eng.search_device(db)
    compare_result(eng.download(db), orc, batch.n)
    eng.free_device_batch(db)


@pytest.mark.parametrize("kw", [dict(max_range_index=5), dict(additional_mismatch=2, min_close=6, max_mismatch_rate=0.05)])
def test_other_parameters_generic_kernels(engine_factory, ref, pg_env, kw):
    pg_env.set("PG_PACK_IN_PLACE_MIN", "1")
    eng = engine_factory(**kw)
    eng.load_reference(ref)
    batch = synth.make_reads(ref[0][1], 2500, read_len=100, seed=77)
    orc = run_oracle(kw, ref, batch)
    db = eng.upload(batch)
    eng.scribble_records(db)
    eng.pack_search_device(db)
    assert eng.last_step_in_place()
    compare_result(eng.download(db), orc, batch.n)
    eng.free_device_batch(db)


def test_window_clusters_and_wide_ids_fall_back_or_pack(engine_factory, ref, pg_env):
    """a batch whose kernels' class is not its plane layout (64-bit candidate ids, reads of up to 64 bases: two blocks per read in
    the kernel, one in the planes) takes the two-launch route inside the same call"""
    pg_env.set("PG_PACK_IN_PLACE_MIN", "1")
    pg_env.set("PG_FORCE_WIDE_CELLS", "1")
    eng = engine_factory()
    eng.load_reference(ref)
    batch = synth.make_reads(ref[0][1], 2000, read_len=50, seed=78)
    orc = run_oracle({}, ref, batch)
    db = eng.upload(batch)
    eng.scribble_records(db)
    eng.pack_search_device(db)
    assert not eng.last_step_in_place()
    compare_result(eng.download(db), orc, batch.n)
    eng.free_device_batch(db)


def test_two_million_reads_one_launch_equals_two(engine_factory):
    """the size at which the path switches on by itself; characters outside ACGTN in one read in 997"""
    big = synth.make_reference(8_000_000, seed=91)
    eng = engine_factory()
    eng.load_reference([("chrB", big)])
    batch = synth.make_reads(big, 2_100_000, seed=92)
    at = batch.seq_off[:-1][::997].astype(np.int64)
    batch.seq[at + (np.arange(len(at)) % 3) * 45] = ord("K")
    db = eng.upload(batch)
    eng.repack(db)
    eng.search_device(db)
    two = eng.download(db)
    assert (two.rc_flag == 2).sum() > 50                   # the exact kernel had work
    eng.scribble_records(db)
    eng.pack_search_device(db)
    assert eng.last_step_in_place()
    one = eng.download(db)
    assert _digest(one) == _digest(two)
    # idempotent, and the ordinary search of the records it left agrees too
    eng.pack_search_device(db)
    assert _digest(eng.download(db)) == _digest(two)
    eng.search_device(db)
    assert _digest(eng.download(db)) == _digest(two)
    eng.free_device_batch(db)

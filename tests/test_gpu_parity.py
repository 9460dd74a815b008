"""-m gpu: the HIP path (C ABI) against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from pindel_amd import synth
from tests.parity import compare_result, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_ref():
    return [("chrS", synth.make_reference(1_500_000, seed=11))]


def test_default_params_100bp(engine_factory, small_ref):
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 6000, seed=5)
    gpu = eng.search_batch(batch)
    orc = run_oracle({}, small_ref, batch)
    assert (orc["close_cnt"] > 0).sum() > 3000 and (orc["far_cnt"] > 0).sum() > 2000
    compare_result(gpu, orc, batch.n)


def test_mixed_lengths(engine_factory, small_ref):
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 4000, seed=6, read_lens=[36, 76, 100, 150, 250])
    gpu = eng.search_batch(batch)
    orc = run_oracle({}, small_ref, batch)
    compare_result(gpu, orc, batch.n)


def test_close_then_far_seams(engine_factory, small_ref):
    """pg_close_end_batch + pg_far_end_batch (the two reference seams) == fused search."""
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 3000, seed=7)
    close = eng.close_end_batch(batch)
    orc = run_oracle({}, small_ref, batch)
    compare_result(close, orc, batch.n, check_far=False)
    assert close.far_off[-1] == 0
    both = eng.far_end_batch(batch, close)
    compare_result(both, orc, batch.n)

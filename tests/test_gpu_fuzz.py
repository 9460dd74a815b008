"""-m gpu: a few fixed seeds of the randomised parity hunt (scripts/fuzz_parity.py): random parameters,
read lengths and repeat-ridden references, HIP path vs CPU oracle, bit-exact."""
import pytest

from scripts.fuzz_parity import one_iteration

pytestmark = pytest.mark.gpu


# 61026: a close-end window (R = 1) that starts exactly where the innermost far-end chunk starts -- the chunk
# must be re-staged to its full extent before the filter masks of all nested ranges are computed
# 5002 / 5015 / 5027: -x 3 / 4, the state-dependent bound of the seed filter (a DPP shift under a diverged EXEC mask
# once made it too tight; 5011 / 5026 failed with them in round 2); 1011 / 8018 / 200203: more than 64 survivors in the first of two paired chunks of a wide
# far-end window (the rest of the first half must be queued before the second half is filtered)
@pytest.mark.parametrize("seed", [1000, 1006, 1011, 1015, 1020, 1029, 1038, 2024, 61026, 5002, 5011, 5015, 5026, 5027, 8018, 200203])
def test_fuzz_seed(seed):
    assert one_iteration(seed, verbose=False)


# a wider driver-observed slice of the hunt: 60 consecutive seeds (a third of them with a second chromosome and
# per-read BreakDancer window clusters, scripts/fuzz_parity.py)
@pytest.mark.parametrize("block", range(6))
def test_fuzz_block(block):
    for seed in range(3000 + 10 * block, 3010 + 10 * block):
        assert one_iteration(seed, verbose=False), seed


# ... and 20 seeds of the wide stream (-x up to 6, more insert sizes: scripts/fuzz_parity.py, seeds >= 200000)
@pytest.mark.parametrize("block", range(2))
def test_fuzz_block_wide(block):
    for seed in range(200400 + 10 * block, 200410 + 10 * block):
        assert one_iteration(seed, verbose=False), seed


# ... and 30 seeds with Pindel's default search parameters (seeds 300000-399999): the kernels that hold the five parameters as
# compile-time constants, on the same nasty references, read-length mixes, window clusters and error-rate tables
@pytest.mark.parametrize("block", range(3))
def test_fuzz_block_default_parameters(block):
    for seed in range(300000 + 10 * block, 300010 + 10 * block):
        assert one_iteration(seed, verbose=False), seed


# ... and 30 more default-parameter seeds through BOTH kernel families on the same inputs and the same oracle: the kernels compiled for
# Pindel's default parameter set (what the launch picks) and, forced with PG_GENERIC_KERNELS=1, the generic ones that every other
# parameter set runs
@pytest.mark.parametrize("block", range(3))
def test_fuzz_block_both_kernel_families(block):
    for seed in range(300100 + 10 * block, 300110 + 10 * block):
        assert one_iteration(seed, verbose=False), seed
        assert one_iteration(seed, verbose=False, generic=True), (seed, "generic kernels")


# ... and 30 seeds of the stream with characters outside ACGTN in one read in twelve (seeds >= 600000): the reads the reference shortens
# when it reverse-complements them, searched by pg_search_exact_kernel (random parameters, lengths, window clusters as above)
@pytest.mark.parametrize("block", range(3))
def test_fuzz_block_characters_outside_acgtn(block):
    for seed in range(600000 + 10 * block, 600010 + 10 * block):
        assert one_iteration(seed, verbose=False), seed

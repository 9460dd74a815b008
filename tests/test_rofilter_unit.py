"""The read-order seed filter (seed_filter_ro, pg_kernels.hip: VGPR index mode, fixed registers) against a brute-force count per
window position: 64 random windows / symbol programs / thresholds x 2 orientations x plain and wide depth x kinds F, B and both x
three and four counter slices (up to 8 / 16 mismatch levels),
with ONE wave, one wave per CU and SEVEN waves per SIMD -- the first version of the asm passed every single-wave test and faulted
at full occupancy (an indexed v_bitop3).  The binary is built by `make -C pindel_amd/csrc` from tests/rofilter_unit.hip, which
includes the shipped pg_kernels.hip."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "pindel_amd", "test_rofilter")


@pytest.mark.gpu
def test_read_order_filter_against_brute_force_at_every_occupancy():
    assert os.path.exists(EXE), "pindel_amd/test_rofilter is missing: __graft_entry__.build() / make -C pindel_amd/csrc"
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    grids = re.findall(r"slices (\d) grid\s+(\d+): no error; lanes that differ from the brute force: F (\d+) B (\d+) DUAL (\d+)", p.stdout)
    assert [(int(g[0]), int(g[1])) for g in grids] == [(3, 1), (3, 256), (3, 7168), (4, 1), (4, 256), (4, 7168)], p.stdout
    assert all(int(x) == 0 for g in grids for x in g[2:]), p.stdout
    kinds = re.findall(r"slices \d kind (\w+)\s*: (\d+) lane results differ of (\d+) \(survivors in the expectation: (\d+)\)", p.stdout)
    assert len(kinds) == 6 and all(int(k[1]) == 0 and int(k[3]) > 10000 for k in kinds), p.stdout

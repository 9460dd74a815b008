#!/bin/bash
# The round's randomised parity campaign on the GPU box: scripts/fuzz_campaign.sh <tag>  -> gpurun_out/<tag>/fuzz_log.txt
# (one summary line per seed range; a mismatch stops the range and is logged with its parameters)
tag=${1:-fuzz}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag; mkdir -p "$out"
cd "$root" || exit 1
{
    echo "library: $(sha256sum pindel_amd/libpindel_pg.so | cut -c1-16)  kernels source: $(sha256sum pindel_amd/csrc/pg_kernels.hip | cut -c1-16)  $(date -u +%FT%TZ)"
    echo "== random parameters, lengths, references, window clusters (seeds < 200000: -x <= 4)"
    FUZZ_QUIET=1 timeout 900 python scripts/fuzz_parity.py ${N1:-400} 70000 2>&1 | grep -v amdgpu.ids | tail -3
    echo "== the wide stream (seeds >= 200000: -x <= 6, more insert sizes)"
    FUZZ_QUIET=1 timeout 900 python scripts/fuzz_parity.py ${N2:-600} 500000 2>&1 | grep -v amdgpu.ids | tail -3
    echo "== Pindel's default search parameters (seeds 300000-399999), each seed through the default-parameter AND the generic kernels"
    FUZZ_QUIET=1 FUZZ_BOTH_FAMILIES=1 timeout 1200 python scripts/fuzz_parity.py ${N3:-400} 310000 2>&1 | grep -v amdgpu.ids | tail -3
    echo "== characters outside ACGTN in one read in twelve (seeds >= 600000): the exact kernel"
    FUZZ_QUIET=1 timeout 900 python scripts/fuzz_parity.py ${N4:-300} 600000 2>&1 | grep -v amdgpu.ids | tail -3
    echo "== every launch packing in place (PG_PACK_IN_PLACE_MIN=1: the chunks of the host path, any size), seeds of the first and the fourth stream"
    PG_PACK_IN_PLACE_MIN=1 FUZZ_QUIET=1 timeout 900 python scripts/fuzz_parity.py ${N5:-200} 90000 2>&1 | grep -v amdgpu.ids | tail -3
    PG_PACK_IN_PLACE_MIN=1 FUZZ_QUIET=1 timeout 900 python scripts/fuzz_parity.py ${N5:-200} 620000 2>&1 | grep -v amdgpu.ids | tail -3
} | tee "$out/fuzz_log.txt"

#!/bin/bash
# Diagnostics (GPU box): instruction counts per read of the step's launch (pack in place) beside the search of packed records.
#   scripts/step_pmc.sh <tag> [lib.so] [reads]
tag=${1:-step_pmc}; root=$(cd "$(dirname "$0")/.." && pwd)
lib=${2:-$root/pindel_amd/libpindel_pg.so}; reads=${3:-2000000}
out=$root/gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp || exit 1
for mode in PG_STEP=1 PG_NONE=1; do
    rm -rf /tmp/rp_sp
    env $mode rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
        --output-format csv -d /tmp/rp_sp -- python "$root/scripts/run_variant.py" "$lib" "$reads" > /tmp/rp_sp.log 2>&1
    echo "== $mode"; python "$root/scripts/pmc_brief.py" /tmp/rp_sp "$reads"
done | tee "$out/step_pmc.txt"

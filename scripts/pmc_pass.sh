#!/bin/bash
# Diagnostics: one rocprofv3 counter pass (per-read SQ instruction counts) of the bench workload with a given
# build of the library:  scripts/pmc_pass.sh <lib.so> [reads]   (PG_X=<n> selects -x n)
lib=$(readlink -f "${1:-pindel_amd/libpindel_pg.so}"); reads=${2:-2000000}
root=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp || exit 1
rm -rf /tmp/rp_pass
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM \
    --output-format csv -d /tmp/rp_pass -- python "$root/scripts/run_variant.py" "$lib" "$reads" 2>/dev/null | grep "kernel ms"
python "$root/scripts/pmc_brief.py" /tmp/rp_pass "$reads"

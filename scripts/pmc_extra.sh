#!/bin/bash
# Diagnostics: the lane-utilisation / wait counters of the search kernel (bench workload):
#   scripts/pmc_extra.sh <lib.so> [reads]     (PG_X / PG_LEN as in run_variant.py)
lib=$(readlink -f "${1:-pindel_amd/libpindel_pg.so}"); reads=${2:-2000000}
root=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp || exit 1
rm -rf /tmp/rp_x1 /tmp/rp_x2
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES \
    --output-format csv -d /tmp/rp_x1 -- python "$root/scripts/run_variant.py" "$lib" "$reads" > /tmp/rp_x1.log 2>&1
python "$root/scripts/pmc_brief.py" /tmp/rp_x1 "$reads"
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_ANY \
    --output-format csv -d /tmp/rp_x2 -- python "$root/scripts/run_variant.py" "$lib" "$reads" > /tmp/rp_x2.log 2>&1
python "$root/scripts/pmc_brief.py" /tmp/rp_x2 "$reads"

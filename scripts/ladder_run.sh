#!/bin/bash
# Diagnostics (GPU box): PMC instruction counts per read of the stop-ladder builds, one process / one rocprofv3 pass.
#   scripts/ladder_run.sh <tag> [reads]     -> gpurun_out/<tag>/ladder.txt
tag=${1:-ladder}; reads=${2:-1000000}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag; mkdir -p "$out"
names="plain"; libs="$root/pindel_amd/libpindel_pg_plain.so"
for k in ${PTS:-1 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24 25 26 27 28 29 30 31 32}; do
    [ -f "$root/pindel_amd/libpindel_pg_stop$k.so" ] && { names="$names stop$k"; libs="$libs $root/pindel_amd/libpindel_pg_stop$k.so"; }
done
export TMPDIR=/tmp PG_LAUNCHES=2
cd /tmp || exit 1
rm -rf /tmp/rp_ladder
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU \
    --output-format csv -d /tmp/rp_ladder -- python "$root/scripts/run_variants_multi.py" "$reads" $libs > "$out/ladder_run.log" 2>&1
grep "kernel ms" "$out/ladder_run.log"
python "$root/scripts/pmc_multi.py" /tmp/rp_ladder "$reads" 2 $names | tee "$out/ladder.txt"

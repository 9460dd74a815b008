#!/bin/bash
# Diagnostics (GPU box): A/B of variant builds: kernel ms (2 M reads, no profiler) + PMC instruction counts (1 M reads), digests.
#   scripts/ab_run.sh <tag> name1 name2 ...      (pindel_amd/libpindel_pg_<name>.so)
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag; mkdir -p "$out"
libs=""; for v in "$@"; do libs="$libs $root/pindel_amd/libpindel_pg_$v.so"; done
export TMPDIR=/tmp
cd /tmp || exit 1
PG_LAUNCHES=4 python "$root/scripts/run_variants_multi.py" ${READS:-2000000} $libs 2>/dev/null | tee "$out/ab_time.txt"
rm -rf /tmp/rp_ab
PG_LAUNCHES=2 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU \
    --output-format csv -d /tmp/rp_ab -- python "$root/scripts/run_variants_multi.py" 1000000 $libs > "$out/ab_pmc.log" 2>&1
python "$root/scripts/pmc_multi.py" /tmp/rp_ab 1000000 2 "$@" | tee "$out/ab_pmc.txt"
if [ -n "$AB_EXTRA" ]; then
    for e in "PG_X=5" "PG_LEN=150"; do
        echo "== $e"; env $e PG_LAUNCHES=3 python "$root/scripts/run_variants_multi.py" 1000000 $libs 2>/dev/null | tee -a "$out/ab_time.txt"
    done
fi

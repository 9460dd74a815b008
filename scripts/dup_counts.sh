#!/bin/bash
# Diagnostics: dynamic VALU / SALU instructions per read of each phase of pg_search_kernel.  Variant k runs
# phase k twice (-DPG_DUP=k, results unchanged), so counter(variant k) - counter(base) = cost of phase k per
# read.  Run on the GPU box; libraries are built on the build host first with:
#   for k in 0 1 2 3 4 5; do scripts/build_variant.sh dup$k -DPG_DUP=$k; done
root=$(cd "$(dirname "$0")/.." && pwd)
reads=${1:-1000000}
export TMPDIR=/tmp
cd /tmp || exit 1
for k in 0 1 2 3 4 5; do
    rm -rf /tmp/rp_dup$k
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES \
        --output-format csv -d /tmp/rp_dup$k -- python "$root/scripts/run_variant.py" "$root/pindel_amd/libpindel_pg_dup$k.so" "$reads" > /tmp/rp_dup$k.log 2>&1
    echo "== PG_DUP=$k (0 = base, 1 read planes, 2 stage window, 3 seed filter, 4 candidate pass + fold, 5 evaluate)"
    tail -1 /tmp/rp_dup$k.log
    python "$root/scripts/pmc_brief.py" /tmp/rp_dup$k "$reads"
done

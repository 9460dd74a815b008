#!/bin/bash
# Diagnostics: rocprofv3 PC sampling of the bench workload's search kernel (which instructions the waves sit on).
#   scripts/pc_sample.sh <lib.so> <tag> [reads] [method: stochastic|host_trap] [interval]
# writes gpurun_out/<tag>/pcs_summary.txt (+ the raw csv, gzipped, if small enough)
lib=$(readlink -f "${1:-pindel_amd/libpindel_pg.so}"); tag=${2:-pcs}; reads=${3:-2000000}
method=${4:-stochastic}; interval=${5:-65536}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp || exit 1
rm -rf /tmp/pcs
unit=cycles; [ "$method" = host_trap ] && unit=time
ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method "$method" \
    --pc-sampling-interval "$interval" --kernel-trace --output-format csv -d /tmp/pcs -- \
    python "$root/scripts/run_variant.py" "$lib" "$reads" > "$out/pcs.log" 2>&1
echo "rocprofv3 rc=$?" >> "$out/pcs.log"
tail -5 "$out/pcs.log"
find /tmp/pcs -type f | head -20
python "$root/scripts/pc_sample_summary.py" /tmp/pcs > "$out/pcs_summary.txt" 2>&1
head -60 "$out/pcs_summary.txt"

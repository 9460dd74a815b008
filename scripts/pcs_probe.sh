#!/bin/bash
# Diagnostics: what PC-sampling configurations does this box offer, and does host_trap sampling of the bench kernel work?
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pcs_probe
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp || exit 1
ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 120 rocprofv3 -L > "$out/list_avail.txt" 2>&1
grep -n -i -B2 -A12 "pc.sampl" "$out/list_avail.txt" | head -80
for cfg in "host_trap time 1" "host_trap time 10" "stochastic cycles 1048576" "stochastic cycles 65536"; do
    set -- $cfg
    rm -rf /tmp/pcs
    ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $2 --pc-sampling-method $1 \
        --pc-sampling-interval $3 --kernel-trace --output-format csv -d /tmp/pcs -- \
        python "$root/scripts/run_variant.py" "$root/pindel_amd/libpindel_pg.so" 2000000 > "$out/pcs_$1_$3.log" 2>&1
    echo "== $cfg rc=$?"; tail -3 "$out/pcs_$1_$3.log"
    n=$(find /tmp/pcs -name '*pc_sampling*' | head -1)
    if [ -n "$n" ]; then
        ls -la /tmp/pcs/*/* | head
        python "$root/scripts/pc_sample_summary.py" /tmp/pcs > "$out/pcs_summary_$1_$3.txt" 2>&1
        head -30 "$out/pcs_summary_$1_$3.txt" | cut -c1-250
        f=$(find /tmp/pcs -name '*pc_sampling*.csv' | head -1)
        [ -n "$f" ] && gzip -c "$f" | head -c 40000000 > "$out/pcs_$1_$3.csv.gz"
        break
    fi
done

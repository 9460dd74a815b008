#!/bin/bash
# Regenerates the judged profile set from the CURRENT tree on the GPU box (run through gpurun):
#   scripts/profile_round.sh <tag> [reads_for_profiles]
# writes gpurun_out/<tag>/{bench_default.json, kernel_stats.csv, bench_under_rocprof.json, pmc_sq.txt,
# hbm_traffic.json}.  Copy what should be judged into profiles/rNN/ afterwards.
# PMC passes use --kernel-trace only (no sys/hip/hsa trace domains), one counter group per pass.
tag=${1:-prof}
reads=${2:-2000000}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root" || exit 1

python bench.py --steps 5 --warmup 1 > "$out/bench_default.json" 2> "$out/bench_default.err"
tail -c 600 "$out/bench_default.json"

cd /tmp || exit 1
rm -rf /tmp/rp_stats /tmp/rp_sq /tmp/rp_fetch /tmp/rp_write /tmp/rp_calib
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -- \
    python "$root/bench.py" --reads "$reads" --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > "$out/bench_under_rocprof.json" 2> /tmp/rp_stats.err
f=$(find /tmp/rp_stats -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -12 "$f" > "$out/kernel_stats.csv"

LIBSO=$root/pindel_amd/libpindel_pg.so
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d /tmp/rp_sq -- python "$root/scripts/run_variant.py" "$LIBSO" "$reads" > /tmp/rp_sq.log 2>&1
python "$root/scripts/pmc_brief.py" /tmp/rp_sq "$reads" > "$out/pmc_sq.txt"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT \
    --output-format csv -d /tmp/rp_sq2 -- python "$root/scripts/run_variant.py" "$LIBSO" "$reads" > /tmp/rp_sq2.log 2>&1
python "$root/scripts/pmc_brief.py" /tmp/rp_sq2 "$reads" >> "$out/pmc_sq.txt"
cat "$out/pmc_sq.txt"

rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/rp_fetch -- python "$root/scripts/run_variant.py" "$LIBSO" "$reads" > /tmp/rp_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/rp_write -- python "$root/scripts/run_variant.py" "$LIBSO" "$reads" > /tmp/rp_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/rp_calib -- python "$root/scripts/calib_fetch.py" > /tmp/rp_calib.log 2>&1
python "$root/scripts/traffic_summary.py" /tmp/rp_fetch /tmp/rp_write /tmp/rp_calib "$reads" "$out/bench_under_rocprof.json" > "$out/hbm_traffic.json"
cat "$out/hbm_traffic.json"

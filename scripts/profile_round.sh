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

# per-phase instruction counts: cumulative SQ_INSTS_* per read of builds that return early (-DPG_STOP=k, built
# here by scripts/build_variant.sh stop$k -DPG_STOP=$k before the gpurun call)
cd "$root" || exit 1
{
    echo "# cumulative instructions per read of builds that stop early (bench workload, $reads reads)"
    echo "# stop1 = record + read planes loaded, window of the first close-end attempt filled; stop2 = + close end, first attempt up to the end of its scan; stop6 = + its evaluation;"
    echo "# stop7 = + emission of its points; stop3 = whole close end (retries included); stop4 = + far end up to the end of"
    echo "# the first range's scan; stop5 = + its evaluation; full = the shipped kernel"
    for k in 1 2 6 7 3 4 5; do
        [ -f pindel_amd/libpindel_pg_stop$k.so ] || continue
        echo "stop$k"
        bash scripts/pmc_pass.sh pindel_amd/libpindel_pg_stop$k.so "$reads" | grep -E "VALU|SALU|INSTS_LDS|VMEM"
    done
    echo "full"
    bash scripts/pmc_pass.sh pindel_amd/libpindel_pg.so "$reads" | grep -E "VALU|SALU|INSTS_LDS|VMEM"
} > "$out/phase_instruction_counts.txt"

# the other workloads and parameter points quoted in DESIGN.md section 8 (one bench line each)
{
    for w in colo-bd repeat-rich wgs-bins grch38-150; do
        python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1
    done
    python bench.py --steps 3 --warmup 1 --max-range-index 5 --reads 2000000 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1
    python bench.py --steps 3 --warmup 1 --read-len 150 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1
    python bench.py --steps 3 --warmup 1 --read-len 250 --reads 4000000 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1
} > "$out/bench_workloads.jsonl"
python - "$out/bench_workloads.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["config"]["workload"][:70], "|", round(d["value"] / 1e6, 1), "M reads/s | cand/read", round(d["config"].get("candidates_per_read", 0), 1))
PY

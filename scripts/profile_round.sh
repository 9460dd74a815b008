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
pmc_set() {   # pmc_set <outfile> <env...>: the SQ counter passes of the search kernel (bench workload unless PG_X / PG_LEN say otherwise)
    local out_file=$1; shift
    : > "$out_file"
    for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
               "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" \
               "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH SQC_ICACHE_MISSES SQ_INST_LEVEL_VMEM"; do
        rm -rf /tmp/rp_sq
        env "$@" rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/rp_sq -- python "$root/scripts/run_variant.py" "$LIBSO" "$reads" > /tmp/rp_sq.log 2>&1
        python "$root/scripts/pmc_brief.py" /tmp/rp_sq "$reads" >> "$out_file"
    done
}
pmc_set "$out/pmc_sq.txt" PG_NONE=1
cat "$out/pmc_sq.txt"
pmc_set "$out/pmc_sq_x5.txt" PG_X=5
pmc_set "$out/pmc_sq_150bp.txt" PG_LEN=150

rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/rp_fetch -- python "$root/scripts/run_variant.py" "$LIBSO" "$reads" > /tmp/rp_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/rp_write -- python "$root/scripts/run_variant.py" "$LIBSO" "$reads" > /tmp/rp_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/rp_calib -- python "$root/scripts/calib_fetch.py" > /tmp/rp_calib.log 2>&1
python "$root/scripts/traffic_summary.py" /tmp/rp_fetch /tmp/rp_write /tmp/rp_calib "$reads" "$out/bench_under_rocprof.json" > "$out/hbm_traffic.json"
cat "$out/hbm_traffic.json"

# kernel_stats for -x 5 and 150 bp as well
for v in "x5 --max-range-index 5" "150bp --read-len 150"; do
    set -- $v
    tag2=$1; shift
    rm -rf /tmp/rp_stats2
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats2 -- \
        python "$root/bench.py" --reads "$reads" --steps 3 --warmup 1 --no-cpu-baseline --no-host-path "$@" > "$out/bench_under_rocprof_$tag2.json" 2> /tmp/rp_stats2.err
    f=$(find /tmp/rp_stats2 -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && head -8 "$f" > "$out/kernel_stats_$tag2.csv"
done

# the host path with a download: the delivery kernels (HBM-bound) next to the search kernel
rm -rf /tmp/rp_host
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_host -- python "$root/scripts/host_path_rate.py" 4000000 > "$out/host_path_rate.txt" 2> /tmp/rp_host.err
f=$(find /tmp/rp_host -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -14 "$f" > "$out/kernel_stats_host_path.csv"
python "$root/scripts/host_path_rate.py" 50000 | tail -1 >> "$out/host_path_rate.txt"

# wave-cycles per phase of a read (s_memtime at the phase boundaries; -DPG_TIMING build made by scripts/build_variant.sh tim)
cd "$root" || exit 1
if [ -f pindel_amd/libpindel_pg_tim.so ]; then
    {
        python scripts/phase_timing.py pindel_amd/libpindel_pg_tim.so 1000000
        echo "== -x 5"
        PG_X=5 python scripts/phase_timing.py pindel_amd/libpindel_pg_tim.so 500000
        echo "== 150 bp"
        PG_LEN=150 python scripts/phase_timing.py pindel_amd/libpindel_pg_tim.so 1000000
    } 2>/dev/null > "$out/phase_wave_cycles.txt"
fi
if [ -f pindel_amd/libpindel_pg_diag.so ]; then
    {
        python scripts/diag_counts.py pindel_amd/libpindel_pg_diag.so 1000000
        echo "== -x 5"
        PG_X=5 python scripts/diag_counts.py pindel_amd/libpindel_pg_diag.so 500000
    } 2>/dev/null > "$out/per_read_event_counts.txt"
fi

# the pack stage on the 10 M-read batch: HIP events (scripts/pack_rate.py) and the tracer's view of the same kernel
python scripts/pack_rate.py pindel_amd/libpindel_pg.so 10000000 2>/dev/null | tail -1 > "$out/pack_rate.txt"
PG_LEN=150 python scripts/pack_rate.py pindel_amd/libpindel_pg.so 10000000 2>/dev/null | tail -1 >> "$out/pack_rate.txt"
( cd /tmp && rm -rf /tmp/rp_pack && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_pack -- python "$root/scripts/pack_rate.py" "$LIBSO" 10000000 > /tmp/rp_pack.log 2>&1
  f=$(find /tmp/rp_pack -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep -E '"Name"|pg_pack_kernel|pg_search_kernel' "$f" > "$out/kernel_stats_pack.csv" )
cat "$out/pack_rate.txt"

# what one more instruction costs inside the shipped kernel (-DPG_PAD_* builds: 128 extra dependent instructions per filter run)
if [ -f pindel_amd/libpindel_pg_pad_s.so ]; then
    bash scripts/variants.sh plain pad_s pad_vf pad_vs > "$out/issue_calibration_raw.txt" 2>&1
    cat "$out/issue_calibration_raw.txt"
fi

# where the instructions of a read go: builds that run one component twice (-DPG_DUP=n) or stop early (-DPG_STOP=n), PMC per read
if [ -f pindel_amd/libpindel_pg_dup3.so ]; then
    PMC=1 bash scripts/variants.sh plain dup3 dup4 dup5 dup6 dup7 dup8 dup9 stop1 stop2 > "$out/component_instruction_counts_raw.txt" 2>&1
    grep -E "^==|SALU|VALU" "$out/component_instruction_counts_raw.txt" | paste - - - | head -12
fi

# Pindel's own flush size and a 4 M-read batch through the host-buffer entry, steady state (six calls each)
python scripts/host_path_calls.py 50000 4000000 > "$out/host_path_calls.txt" 2>&1
tail -4 "$out/host_path_calls.txt"

# the other workloads and parameter points quoted in DESIGN.md section 8 (one bench line each)
{
    for w in colo-bd repeat-rich wgs-bins grch38-150; do
        python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1
    done
    python bench.py --steps 3 --warmup 1 --max-range-index 5 --reads 2000000 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1
    python bench.py --steps 3 --warmup 1 --read-len 150 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1
    python bench.py --steps 3 --warmup 1 --read-len 250 --reads 4000000 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1
    python bench.py --steps 3 --warmup 1 --read-len 400 --reads 2000000 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1
} > "$out/bench_workloads.jsonl"
python - "$out/bench_workloads.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["config"]["workload"][:70], "|", round(d["value"] / 1e6, 1), "M reads/s | cand/read", round(d["config"].get("candidates_per_read", 0), 1))
PY


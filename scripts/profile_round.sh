#!/bin/bash
# Regenerates the judged profile set from the CURRENT tree on the GPU box (run through gpurun):
#   scripts/profile_round.sh <tag> [reads_for_the_counter_passes]
# writes gpurun_out/<tag>/{bench_traced_10m.json, kernel_stats_10m.csv, pmc_sq*.txt, hbm_traffic.json, kernel_stats_*.csv,
# bench_workloads.jsonl, pack_rate.txt, bench_default.json}.  Copy what should be judged into profiles/rNN/ afterwards
# (bench.py reads profiles/rNN/{hbm_traffic.json, pmc_sq.txt} for roofline.traffic / roofline.issue).
# PMC passes use --kernel-trace only (no sys/hip/hsa trace domains), one counter group per pass.
tag=${1:-prof}
reads=${2:-2000000}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
LIBSO=$root/pindel_amd/libpindel_pg.so

# 1. ONE-PROCESS RECONCILIATION: the judged command (10 M reads, 20 steps, 5 warm-up) under the tracer; the tracer's average
#    kernel duration, the HIP-event kernel_ms and ms_per_step of this same run go side by side (profiles/rNN/README.md)
cd /tmp || exit 1
rm -rf /tmp/rp_10m
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_10m -- \
    python "$root/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-standalone > "$out/bench_traced_10m.json" 2> /tmp/rp_10m.err
f=$(find /tmp/rp_10m -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -8 "$f" > "$out/kernel_stats_10m.csv"
python - "$out/bench_traced_10m.json" "$out/kernel_stats_10m.csv" <<'PY' | tee "$out/reconciliation.txt"
import csv, json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "pg_search_kernel" in r["Name"]]
avg = float(rows[0]["AverageNs"]) / 1e6 if rows else float("nan")
calls = rows[0]["Calls"] if rows else "?"
prow = [r for r in csv.DictReader(open(sys.argv[2])) if "pg_pack_kernel" in r["Name"]]
pcalls = prow[0]["Calls"] if prow else "0"
c = d["config"]
ev = c["device_ms_per_step"]
print(f"same run, 10 M reads, a step = ONE launch (pg_search_kernel packs its claims): tracer AverageNs {avg:.3f} ms over {calls} launches "
      f"(pg_pack_kernel launches in the run: {pcalls} = the upload's) | HIP-event device_ms_per_step {ev:.3f} | wall ms_per_step {d['ms_per_step']:.3f} | "
      f"value {d['value'] / 1e6:.1f} M reads/s | spread of the three step times {100 * (max(ev, avg, d['ms_per_step']) / min(ev, avg, d['ms_per_step']) - 1):.2f} %")
PY
# ... and the judged command as the driver runs it (no tracer; with the stages as launches of their own measured after the timed region)
python "$root/bench.py" --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > "$out/bench_default.json"
python - "$out/bench_default.json" <<'PY' | tee -a "$out/reconciliation.txt"
import json, sys
d = json.loads(open(sys.argv[1]).read())
c = d["config"]
print(f"untraced run of the same command: value {d['value'] / 1e6:.1f} M reads/s, ms_per_step {d['ms_per_step']:.3f}, device {c['device_ms_per_step']:.3f} | "
      f"stages as launches of their own (outside the timed region): pack {c['pack_ms_standalone']:.3f} ms + search {c['search_ms_standalone']:.3f} ms = "
      f"{c['pack_ms_standalone'] + c['search_ms_standalone']:.3f} ms; search only {c['value_search_only'] / 1e6:.1f} M reads/s (rounds 1-5's `value`)")
PY

# 2. the SQ counter passes of the search kernel (bench workload unless PG_X / PG_LEN say otherwise)
pmc_set() {
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
pmc_set "$out/pmc_sq.txt" PG_STEP=1                     # the step's launch (pack in place)
cat "$out/pmc_sq.txt"
pmc_set "$out/pmc_sq_search_only.txt" PG_NONE=1         # the search of packed records (what rounds 1-5 counted)
pmc_set "$out/pmc_sq_x5.txt" PG_X=5
pmc_set "$out/pmc_sq_150bp.txt" PG_LEN=150
pmc_set "$out/pmc_sq_wgsreal.txt" PG_LEN=150 PG_SORT=1 PG_MIX=0.02,0.01,0.01,0.01,0.95

# 3. HBM traffic of the search kernel: FETCH_SIZE / WRITE_SIZE in passes of their own + the calibration stream
rm -rf /tmp/rp_fetch /tmp/rp_write /tmp/rp_calib /tmp/rp_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -- \
    python "$root/bench.py" --reads "$reads" --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > "$out/bench_under_rocprof.json" 2> /tmp/rp_stats.err
f=$(find /tmp/rp_stats -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -8 "$f" > "$out/kernel_stats.csv"
PG_STEP=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/rp_fetch -- python "$root/scripts/run_variant.py" "$LIBSO" "$reads" > /tmp/rp_fetch.log 2>&1
PG_STEP=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/rp_write -- python "$root/scripts/run_variant.py" "$LIBSO" "$reads" > /tmp/rp_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/rp_calib -- python "$root/scripts/calib_fetch.py" > /tmp/rp_calib.log 2>&1
python "$root/scripts/traffic_summary.py" /tmp/rp_fetch /tmp/rp_write /tmp/rp_calib "$reads" "$out/bench_under_rocprof.json" > "$out/hbm_traffic.json"
cat "$out/hbm_traffic.json"

# 4. kernel_stats for -x 5 and 150 bp
for v in "x5 --max-range-index 5" "150bp --read-len 150"; do
    set -- $v
    tag2=$1; shift
    rm -rf /tmp/rp_stats2
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats2 -- \
        python "$root/bench.py" --reads "$reads" --steps 3 --warmup 1 --no-cpu-baseline --no-host-path "$@" > "$out/bench_under_rocprof_$tag2.json" 2> /tmp/rp_stats2.err
    f=$(find /tmp/rp_stats2 -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && head -8 "$f" > "$out/kernel_stats_$tag2.csv"
done

# 5. the pack stage on the 10 M-read batch (HIP events) and the other workloads / parameter points quoted in DESIGN.md
cd "$root" || exit 1
python scripts/pack_rate.py pindel_amd/libpindel_pg.so 10000000 2>/dev/null | tail -1 > "$out/pack_rate.txt"
PG_LEN=150 python scripts/pack_rate.py pindel_amd/libpindel_pg.so 10000000 2>/dev/null | tail -1 >> "$out/pack_rate.txt"
cat "$out/pack_rate.txt"
{
    for w in colo-bd repeat-rich wgs-bins grch38-150 wgs-real; do
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
python scripts/host_path_calls.py 50000 4000000 > "$out/host_path_calls.txt" 2>&1
tail -3 "$out/host_path_calls.txt"

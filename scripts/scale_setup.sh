#!/bin/bash
# Round-5 verdict, item 6: what does a rank's set-up cost before the timed region, alone and with eight ranks side by side on one
# host?  (GPU box, 1 GPU: the eight ranks share device 0 -- PG_BENCH_SHARE_GPU=1 -- so only the HOST side of the contention is real.)
#   scripts/scale_setup.sh <tag>  -> gpurun_out/<tag>/scale_setup.txt
tag=${1:-scale_setup}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag; mkdir -p "$out"
cd "$root" || exit 1
TIMEFORMAT="  wall time of the process: %R s"
show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        s = d["config"]["setup_seconds"]
        print(f"  n_gpus {d['n_gpus']}  reads/GPU {d['config']['reads_per_gpu']}  set-up: inputs {s['synthetic_inputs']:.1f} s, reference -> HBM {s['reference_to_hbm']:.1f} s, "
              f"reads -> HBM {s['reads_to_hbm']:.1f} s | value {d['value'] / 1e6:.1f} M reads/s, {d['ms_per_step']:.2f} ms/step")
PY
}
{
    echo "host: $(nproc) hardware threads visible, cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
    echo "== the driver's command at N = 1 (BASELINE configs[2], 10 M reads): wall time of the whole process"
    time python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > "$out/n1.json" 2> "$out/n1.err"; show "$out/n1.json"
    echo "== the same command with EIGHT ranks (self-spawned, sharing device 0; rank 0's record): host-side contention of eight set-ups"
    time env PG_BENCH_SHARE_GPU=1 python bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > "$out/n8.json" 2> "$out/n8.err"; show "$out/n8.json"
    echo "== BASELINE configs[3] shape at full genome scale (3.1 Gbp, 24 chromosomes, 12.5 M x 150 bp), one rank"
    time python bench.py --workload grch38-150 --genome-scale 1.0 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > "$out/grch38_n1.json" 2> "$out/grch38_n1.err"; show "$out/grch38_n1.json"
    echo "== ... and four ranks of it side by side at a quarter of the genome scale (host memory: four full-scale set-ups would hold 4 x 10 GB)"
    time env PG_BENCH_SHARE_GPU=1 python bench.py --gpus 4 --workload grch38-150 --genome-scale 0.25 --reads 2000000 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > "$out/grch38_n4.json" 2> "$out/grch38_n4.err"; show "$out/grch38_n4.json"
} 2>&1 | tee "$out/scale_setup.txt"

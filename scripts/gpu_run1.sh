#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04a/pytest.txt
cat gpurun_out/r04a/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err
tail -c 1500 gpurun_out/r04a/bench.json
cd /tmp && rm -rf /tmp/rp_stats && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -- python $GRAFT_REPO_ROOT/bench.py --reads 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $GRAFT_REPO_ROOT/gpurun_out/r04a/bench_under_rocprof.json 2> /tmp/rp.err
f=$(find /tmp/rp_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" > $GRAFT_REPO_ROOT/gpurun_out/r04a/kernel_stats.csv; cat $GRAFT_REPO_ROOT/gpurun_out/r04a/kernel_stats.csv

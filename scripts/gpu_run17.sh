#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04d
{
for c in 1 2 3 4; do
  echo "== PG_COPY_THREADS=$c"
  PG_COPY_THREADS=$c python scripts/host_path_calls.py 4000000 2>/dev/null | tail -2
done
PG_COPY_THREADS=3 python scripts/host_path_calls.py 1000000 10000000 2>/dev/null | grep best
} > gpurun_out/r04d/copy_threads.txt 2>&1
cat gpurun_out/r04d/copy_threads.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3

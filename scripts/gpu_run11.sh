#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04c
{
bash scripts/variants.sh cur c4 c16 j2 j6
PG_X=5 bash scripts/variants.sh cur jw0 jw4 j2 j6
PG_LEN=150 bash scripts/variants.sh cur j2 j6
python scripts/host_path_rate.py 50000 2>/dev/null | tail -2
python scripts/host_path_rate.py 4000000 2>/dev/null | tail -2
PG_HOST_TIMING=1 python scripts/host_path_rate.py 4000000 2>&1 | grep "pg_search_batch:" | tail -1
python scripts/host_path_rate.py 1000000 2>/dev/null | tail -1
} > gpurun_out/r04c/retune.txt 2>&1
cat gpurun_out/r04c/retune.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3

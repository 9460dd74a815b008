#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04c
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04c/pytest.txt
cat gpurun_out/r04c/pytest.txt
{
python scripts/host_path_rate.py 50000 2>/dev/null | tail -2
python scripts/host_path_rate.py 4000000 2>/dev/null | tail -2
PG_HOST_TIMING=1 python scripts/host_path_rate.py 4000000 2>&1 | grep "pg_search_batch:" | tail -2
PG_HOST_TIMING=1 python scripts/host_path_rate.py 50000 2>&1 | grep "pg_search_batch:" | tail -2
} > gpurun_out/r04c/host_path.txt 2>&1
cat gpurun_out/r04c/host_path.txt

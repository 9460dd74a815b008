#!/bin/bash
export TMPDIR=/tmp
PG_HOST_TIMING=1 python scripts/host_path_calls.py 10000000 2>&1 | grep -v amdgpu | tail -4
python scripts/host_path_calls.py 50000 1000000 4000000 2>&1 | grep best

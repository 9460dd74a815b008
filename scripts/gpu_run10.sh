#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04c
{
bash scripts/variants.sh ka ka5
PG_X=5 bash scripts/variants.sh ka ka5
PG_LEN=150 bash scripts/variants.sh ka ka5
} > gpurun_out/r04c/ka5.txt 2>&1
cat gpurun_out/r04c/ka5.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04d
{
bash scripts/variants.sh w5 new new2
PG_LEN=150 bash scripts/variants.sh w5 new2
PG_LEN=150 PG_X=5 bash scripts/variants.sh w5 new2
PG_X=5 bash scripts/variants.sh new new2
} > gpurun_out/r04d/w6c.txt 2>&1
cat gpurun_out/r04d/w6c.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4

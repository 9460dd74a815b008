#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04d
{
bash scripts/variants.sh cur w6 w7
PG_X=5 bash scripts/variants.sh cur w6 w7
PG_LEN=150 bash scripts/variants.sh cur w6
READS=50000 bash scripts/variants.sh cur w6
} > gpurun_out/r04d/w6.txt 2>&1
cat gpurun_out/r04d/w6.txt

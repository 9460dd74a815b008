#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04b
{
PG_X=5 bash scripts/variants.sh nopair pair
bash scripts/variants.sh nopair pair
PG_X=4 bash scripts/variants.sh nopair pair
PG_LEN=150 PG_X=5 bash scripts/variants.sh nopair pair
} > gpurun_out/r04b/pair.txt 2>&1
cat gpurun_out/r04b/pair.txt

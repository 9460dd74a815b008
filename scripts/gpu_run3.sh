#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04b
{
for v in pu2 pe1 pe2 pe4 pe7; do
  PG_PACK_ONLY=1 timeout 120 python scripts/pack_rate.py pindel_amd/libpindel_pg_$v.so 10000000 2>/dev/null | tail -1
done
} > gpurun_out/r04b/pack_exp.txt 2>&1
cat gpurun_out/r04b/pack_exp.txt

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04b
{
for v in pv1 pv2 pv4 pv8; do
  timeout 120 python scripts/pack_rate.py pindel_amd/libpindel_pg_$v.so 10000000 2>/dev/null | tail -1
done
PG_LEN=150 timeout 120 python scripts/pack_rate.py pindel_amd/libpindel_pg_pv2.so 10000000 2>/dev/null | tail -1
} > gpurun_out/r04b/pack4.txt 2>&1
cat gpurun_out/r04b/pack4.txt

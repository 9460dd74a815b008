#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04b
{
for v in main pu1 pu2 pu8; do
  lib=pindel_amd/libpindel_pg_$v.so; [ $v = main ] && lib=pindel_amd/libpindel_pg.so
  python scripts/pack_rate.py $lib 10000000 2>/dev/null | tail -1
done
PG_LEN=150 python scripts/pack_rate.py pindel_amd/libpindel_pg.so 10000000 2>/dev/null | tail -1
bash scripts/variants.sh f_base f_ilp f_mem f_o2 f_trk
} > gpurun_out/r04b/variants.txt 2>&1
cat gpurun_out/r04b/variants.txt

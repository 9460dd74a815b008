#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04b
{
for v in main pu1 pu4; do
  lib=pindel_amd/libpindel_pg_$v.so; [ $v = main ] && lib=pindel_amd/libpindel_pg.so
  timeout 120 python scripts/pack_rate.py $lib 10000000 2>/dev/null | tail -1
done
PG_LEN=150 timeout 120 python scripts/pack_rate.py pindel_amd/libpindel_pg.so 10000000 2>/dev/null | tail -1
PG_LEN=250 timeout 120 python scripts/pack_rate.py pindel_amd/libpindel_pg.so 4000000 2>/dev/null | tail -1
} > gpurun_out/r04b/pack2.txt 2>&1
cat gpurun_out/r04b/pack2.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_li_pin.py -m gpu -x -q 2>&1 | tail -5

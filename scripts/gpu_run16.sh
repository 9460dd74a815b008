#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04d
{
for c in 0 25000 16700 12500; do
  echo "== PG_HOST_CHUNK=$c"
  if [ $c = 0 ]; then python scripts/host_path_calls.py 50000 2>/dev/null | tail -1; else PG_HOST_CHUNK=$c python scripts/host_path_calls.py 50000 2>/dev/null | tail -1; fi
done
} > gpurun_out/r04d/flush_chunks.txt 2>&1
cat gpurun_out/r04d/flush_chunks.txt

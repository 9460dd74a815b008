#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04c
{
for n in 50000 262144 650000 2000000; do
  READS=$n bash scripts/variants.sh notail tail
done
python scripts/host_path_rate.py 50000 2>/dev/null | tail -2
python scripts/host_path_rate.py 4000000 2>/dev/null | tail -2
python bench.py --workload wgs-bins --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | cut -c1-200
} > gpurun_out/r04c/tail.txt 2>&1
cat gpurun_out/r04c/tail.txt

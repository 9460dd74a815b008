#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04d
bash scripts/variants.sh plain nofill > gpurun_out/r04d/nofill.txt 2>&1
cat gpurun_out/r04d/nofill.txt

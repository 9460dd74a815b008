#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04b
TR_N=150 bash scripts/trace_flush.sh > gpurun_out/r04b/trace_flush.txt 2>&1
grep -n "us  +" gpurun_out/r04b/trace_flush.txt | grep -v "CallConfiguration\|hipGetLastError" | head -75

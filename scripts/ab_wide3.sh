#!/bin/bash
# Diagnostics (GPU box): the three-words-per-lane fill of wide window pairs (w3) against the build without it (w0: -DPG_NO_WIDE3),
# default-parameter and generic kernels, 100 / 150 bp, -x 2 / -x 5.  scripts/build_variant.sh w3 / w0 first.
cd "$(dirname "$0")/.." || exit 1
L="pindel_amd/libpindel_pg_w0.so pindel_amd/libpindel_pg_w3.so"
run() { echo "== $1"; shift; env "$@" PG_LAUNCHES=3 timeout 300 python scripts/run_variants_multi.py $N $L 2>/dev/null | tail -2; }
N=1000000 run "150 bp, default kernels" PG_LEN=150
N=1000000 run "150 bp, generic kernels" PG_LEN=150 PG_GENERIC_KERNELS=1
N=2000000 run "100 bp, default kernels" PG_NONE=1
N=2000000 run "100 bp, generic kernels" PG_GENERIC_KERNELS=1
N=2000000 run "100 bp, -x 5" PG_X=5
N=1000000 run "150 bp, -x 5" PG_X=5 PG_LEN=150

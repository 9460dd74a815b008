#!/bin/bash
# Diagnostics: build pindel_amd/libpindel_pg_<name>.so with extra -D flags for the kernels (the host side of the ABI is
# taken from the regular build: run `make -C pindel_amd/csrc` first).  usage: build_variant.sh name flags...   (KFLAGS= in the
# environment drops the Makefile's kernel flags)
cd "$(dirname "$0")/../pindel_amd/csrc" || exit 1
n=$1; shift
mkdir -p build
[ -f build/pg_api.o ] || make build/pg_api.o > /dev/null || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -pthread --offload-arch=gfx950 -I../../include -I. ${KFLAGS--mllvm -disable-machine-licm -mllvm -phi-elim-split-all-critical-edges} "$@" -c -x hip pg_kernels.hip -o build/pg_kernels_$n.o 2>build/pg_kernels_$n.log || { grep -A6 -E "error" build/pg_kernels_$n.log | head -40; exit 1; }
/opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 build/pg_kernels_$n.o build/pg_api.o -o ../libpindel_pg_$n.so

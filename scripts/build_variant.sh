#!/bin/bash
# Diagnostics: build pindel_amd/libpindel_pg_<name>.so with extra -D flags.  usage: build_variant.sh name flags...
cd "$(dirname "$0")/../pindel_amd/csrc" || exit 1
n=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -pthread --offload-arch=gfx950 -I../../include -I. "$@" -shared -x hip pg_api.cpp pg_kernels.hip -o ../libpindel_pg_$n.so 2>/dev/null

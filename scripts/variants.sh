#!/bin/bash
# Diagnostics: kernel time (and, with PMC=1, VALU / SALU instructions per read) of several builds of the library on the
# bench workload:  scripts/variants.sh name1 name2 ...   (pindel_amd/libpindel_pg_<name>.so; "main" = the shipped one)
root=$(cd "$(dirname "$0")/.." && pwd)
reads=${READS:-2000000}
for v in "$@"; do
    lib=$root/pindel_amd/libpindel_pg_$v.so
    [ "$v" = main ] && lib=$root/pindel_amd/libpindel_pg.so
    [ -f "$lib" ] || { echo "$v: no such build"; continue; }
    if [ -n "$PMC" ]; then
        echo "== $v"
        bash "$root/scripts/pmc_pass.sh" "$lib" "$reads" | grep -E "kernel ms|VALU|SALU|WAVE_CYCLES"
    else
        python "$root/scripts/run_variant.py" "$lib" "$reads" 2>/dev/null | tail -1
    fi
done

#!/bin/bash
# Diagnostics: disassemble the headline kernel (or $2 = symbol substring) of a variant library into /tmp/pgk/<name>.s
lib=$(readlink -f "$1"); sym=${2:-pg_search_kernelILi2ELi3EjLi3ELb1EE}
name=$(basename "$lib" .so)
mkdir -p /tmp/pgk/$name && cd /tmp/pgk/$name || exit 1
cp "$lib" lib.so
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1
co=$(ls lib.so.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$co" | awk -v s="$sym" '/^[0-9a-f]+ <.*>:$/ {p = index($0, s) > 0} p' > ../$name.s
wc -l ../$name.s

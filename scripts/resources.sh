#!/bin/bash
# Register / spill / LDS figures of every kernel of the shipped library, from the code object's metadata
# (llvm-objdump --offloading + llvm-readelf --notes):  scripts/resources.sh [lib.so] > profiles/rNN/resources.txt
lib=$(readlink -f "${1:-$(dirname "$0")/../pindel_amd/libpindel_pg.so}")
tmp=$(mktemp -d); cp "$lib" "$tmp/lib.so"; cd "$tmp" || exit 1
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1
co=$(ls lib.so.*gfx950* 2>/dev/null | head -1)
[ -n "$co" ] || { echo "no gfx950 code object in $lib"; exit 1; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" > notes.txt
python3 - <<'PY'
import re
t = open("notes.txt").read()
rows = []
for blk in t.split("\n  - ")[1:]:
    m = re.search(r"\.name:\s+(\S+)", blk)
    if not m:
        continue
    g = lambda k: (re.search(k + r":\s+(\d+)", blk) or [None, "?"])[1]
    name = m.group(1)
    d = re.match(r"_Z16pg_search_kernelILi(\d+)ELi(\d+)E([jy])Li(\d)ELb([01])E", name)
    if d:
        name = (f"pg_search_kernel<NB={d.group(1)}, NS={d.group(2)}, {'u32' if d.group(3) == 'j' else 'u64'}, "
                f"{['', 'CLOSE', 'FAR', 'BOTH'][int(d.group(4))]}, {'defaults' if d.group(5) == '1' else 'generic'}>")
    else:
        k = re.match(r"_Z(\d+)", name)
        if k:
            n = int(k.group(1))
            name = name[2 + len(k.group(1)):][:n] + name[2 + len(k.group(1)) + n:][:12]
    rows.append((name, g(r"\.vgpr_count"), g(r"\.sgpr_count"), g(r"\.sgpr_spill_count"), g(r"\.vgpr_spill_count"),
                 g(r"\.private_segment_fixed_size"), g(r"\.group_segment_fixed_size")))
print(f"{'kernel':72s} {'vgpr':>5s} {'sgpr':>5s} {'sgpr_spill':>10s} {'vgpr_spill':>10s} {'scratch B':>9s} {'LDS B':>6s}")
for r in sorted(rows):
    if "hipcub" in r[0] or "rocprim" in r[0]:
        continue
    print(f"{r[0]:72s} {r[1]:>5s} {r[2]:>5s} {r[3]:>10s} {r[4]:>10s} {r[5]:>9s} {r[6]:>6s}")
PY
rm -rf "$tmp"

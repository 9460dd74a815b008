"""Diagnostics (no GPU needed): static instruction counts of one kernel per source line and instruction class, from
`llvm-objdump -d -l` of a code object built with -gline-tables-only.

  scripts/isa_lines.py <dev.s> <kernel-symbol-substring> [--dump LO HI] [--top N]

--dump LO HI prints the instructions attributed to source lines LO..HI of pg_kernels.hip in address order (with their line)."""
import collections
import re
import sys


def klass(m):
    if m.startswith("s_nop"):
        return "nop"
    if m.startswith(("s_waitcnt", "s_barrier", "s_sleep", "s_setprio", "s_endpgm", "s_sethalt")):
        return "wait"
    if m.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if m.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_dcache")):
        return "smem"
    if m.startswith("s_"):
        return "salu"
    if m.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "vlane"
    if m.startswith(("ds_",)):
        return "lds"
    if m.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if m.startswith("v_"):
        return "valu"
    return "other"


def parse(path, sym):
    rows = []           # (addr, mnemonic, text, line)
    cur = False
    line = 0
    rx_sym = re.compile(r"^[0-9a-f]+ <(.+)>:")
    rx_line = re.compile(r"^; .*pg_kernels\.hip:(\d+)")
    rx_other = re.compile(r"^; .*:(\d+)")
    rx_ins = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):")
    for l in open(path, errors="replace"):
        m = rx_sym.match(l)
        if m:
            cur = sym in m.group(1)
            continue
        if not cur:
            continue
        m = rx_line.match(l)
        if m:
            line = int(m.group(1))
            continue
        if rx_other.match(l):
            continue           # a header's line: keep the last pg_kernels.hip line
        m = rx_ins.match(l)
        if m:
            rows.append((int(m.group(3), 16), m.group(1), m.group(1) + " " + m.group(2), line))
    return rows


def main():
    path, sym = sys.argv[1], sys.argv[2]
    rows = parse(path, sym)
    if "--dump" in sys.argv:
        i = sys.argv.index("--dump")
        lo, hi = int(sys.argv[i + 1]), int(sys.argv[i + 2])
        for a, m, t, ln in rows:
            if lo <= ln <= hi:
                print(f"{a:08x} L{ln:<5d} {klass(m):6s} {t}")
        return
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 80
    tot = collections.Counter()
    by_line = collections.defaultdict(collections.Counter)
    for a, m, t, ln in rows:
        k = klass(m)
        tot[k] += 1
        by_line[ln][k] += 1
    print("instructions", len(rows), dict(tot))
    keys = ["salu", "valu", "vlane", "branch", "nop", "wait", "lds", "vmem", "smem"]
    print(f"{'line':>6s} " + " ".join(f"{k:>6s}" for k in keys) + "  total")
    for ln, c in sorted(by_line.items(), key=lambda kv: -sum(kv[1].values()))[:top]:
        print(f"{ln:6d} " + " ".join(f"{c[k]:6d}" for k in keys) + f"  {sum(c.values()):5d}")


if __name__ == "__main__":
    main()

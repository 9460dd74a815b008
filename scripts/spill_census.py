"""Diagnostics: how many of a read's executed instructions are register-allocation artefacts -- SGPR spills and reloads (v_writelane /
v_readlane on the compiler's spill VGPR), scratch traffic, register copies (v_mov / s_mov) and s_nop padding?

Method (round-5 verdict, item 1a): the stop ladder gives the EXECUTED vector / scalar instructions of every segment of a read's
timeline (profiles/r06/ladder_raw.txt: PMC differences between -DPG_STOP=k builds); the -DPG_STOP=32 build, whose read runs to the
end, carries every point's marker in its ISA (s_nop 14, s_nop k & 7, s_nop k >> 3), so the static instruction mix of the code
between two markers can be counted; executed artefacts of a segment ~ executed instructions of the segment x static share of the
artefact class in the segment's code.  (Within a segment every instruction is taken to run equally often; loops inside a segment --
the carry-save groups, the tier A fold -- hold few artefacts, so the estimate leans high rather than low.)
  python scripts/spill_census.py <stop32.s> <ladder.txt>"""
import collections
import re
import sys

isa, ladder = sys.argv[1], sys.argv[2]
rows = []
rx = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):")
for l in open(isa, errors="replace"):
    m = rx.match(l)
    if m:
        rows.append((m.group(1), m.group(2)))
# the compiler's SGPR-spill VGPR: the destination of the v_writelane instructions
wl = collections.Counter(a.split(",")[0].strip() for o, a in rows if o.startswith("v_writelane"))
spill_vgpr = wl.most_common(1)[0][0] if wl else None


def klass(o, a):
    if o.startswith("s_nop"):
        return "nop"
    if o.startswith(("s_waitcnt", "s_barrier", "s_endpgm", "s_sleep")):
        return "wait"
    if o.startswith(("s_cbranch", "s_branch", "s_setpc", "s_getpc", "s_swappc")):
        return "branch"
    if o.startswith("s_load"):
        return "smem"
    if o.startswith(("s_mov_b32", "s_mov_b64", "s_cmov")):
        return "s_copy"
    if o.startswith("s_"):
        return "salu"
    if o.startswith("v_writelane") and a.split(",")[0].strip() == spill_vgpr:
        return "spill"
    if o.startswith("v_readlane") and len(a.split(",")) > 1 and a.split(",")[1].strip() == spill_vgpr:
        return "spill"
    if o.startswith("scratch_"):
        return "scratch"
    if o.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr")):
        return "v_copy"
    if o.startswith(("ds_", "global_", "flat_", "buffer_")):
        return "mem"
    if o.startswith("v_"):
        return "valu"
    return "other"


# segments of the ISA in address order, keyed by the marker that opens them
seg = collections.defaultdict(collections.Counter)
cur = 0
i = 0
while i < len(rows):
    o, a = rows[i]
    if o == "s_nop" and a.strip() == "14" and i + 2 < len(rows) and rows[i + 1][0] == "s_nop" and rows[i + 2][0] == "s_nop":
        cur = int(rows[i + 1][1]) + 8 * int(rows[i + 2][1])
        i += 3
        continue
    seg[cur][klass(o, a)] += 1
    i += 1

# executed instructions per build (cumulative along a read); segment k = what runs between reaching point k and the next point
cum = {}
for l in open(ladder):
    f = l.split()
    if len(f) >= 9 and (f[0].startswith("stop") or f[0] == "plain"):
        cum[f[0]] = dict(valu=float(f[6]), salu=float(f[4]))
pts = sorted((int(k[4:]) for k in cum if k.startswith("stop")), key=lambda k: cum[f"stop{k}"]["valu"])
V_CLASSES = ("valu", "spill", "v_copy")            # what SQ_INSTS_VALU counts (lane moves and copies are vector instructions)
S_CLASSES = ("salu", "s_copy")                     # SQ_INSTS_SALU (s_nop / s_waitcnt / branches are not scalar-ALU instructions)
tot = collections.Counter()
print(f"spill VGPR {spill_vgpr}; static instructions of the whole kernel: " + ", ".join(f"{k} {sum(s[k] for s in seg.values())}" for k in
      ("valu", "spill", "v_copy", "scratch", "salu", "s_copy", "nop", "branch", "smem", "mem", "wait")))
print(f"{'segment':>8} {'exec V':>8} {'exec S':>8} | static: {'valu':>5} {'spill':>5} {'v_copy':>6} {'salu':>5} {'s_copy':>6} {'nop':>4} | est. executed: {'spill':>6} {'v_copy':>6} {'s_copy':>6} {'nop':>6}")
order = [0] + pts
for idx, k in enumerate(order):
    lo = cum[f"stop{k}"] if k else dict(valu=0.0, salu=0.0)
    hi = cum[f"stop{order[idx + 1]}"] if idx + 1 < len(order) else cum["plain"]
    dv, ds = max(hi["valu"] - lo["valu"], 0.0), max(hi["salu"] - lo["salu"], 0.0)
    s = seg.get(k, collections.Counter())
    nv, ns = sum(s[c] for c in V_CLASSES), sum(s[c] for c in S_CLASSES)
    e = {c: (dv * s[c] / nv if nv else 0.0) for c in ("spill", "v_copy")}
    e["s_copy"] = ds * s["s_copy"] / ns if ns else 0.0
    e["nop"] = (dv + ds) * s["nop"] / max(nv + ns, 1)      # (padding rides with the instructions around it)
    for c, v in e.items():
        tot[c] += v
    tot["V"] += dv
    tot["S"] += ds
    print(f"{k:>8} {dv:8.1f} {ds:8.1f} |         {s['valu']:5d} {s['spill']:5d} {s['v_copy']:6d} {s['salu']:5d} {s['s_copy']:6d} {s['nop']:4d} |"
          f"                {e['spill']:6.1f} {e['v_copy']:6.1f} {e['s_copy']:6.1f} {e['nop']:6.1f}")
allv, alls = cum["plain"]["valu"], cum["plain"]["salu"]
print(f"\nper read: {allv:.0f} vector + {alls:.0f} scalar instructions executed (plain build).  Estimated artefacts: "
      f"SGPR spill moves {tot['spill']:.0f} ({100 * tot['spill'] / allv:.1f} % of the vector instructions), vector copies {tot['v_copy']:.0f} "
      f"({100 * tot['v_copy'] / allv:.1f} %), scalar copies {tot['s_copy']:.0f} ({100 * tot['s_copy'] / alls:.1f} % of the scalar instructions), "
      f"s_nop {tot['nop']:.0f} (not in either count).")
print(f"spill + copies: {tot['spill'] + tot['v_copy'] + tot['s_copy']:.0f} of {allv + alls:.0f} = "
      f"{100 * (tot['spill'] + tot['v_copy'] + tot['s_copy']) / (allv + alls):.1f} % of the executed vector + scalar instructions")

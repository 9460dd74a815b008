"""Diagnostics: per-read counter averages per build from ONE rocprofv3 --pmc run of scripts/run_variants_multi.py:
   python scripts/pmc_multi.py <rocprof dir> <reads> <launches per build> <name> [<name> ...]"""
import collections
import csv
import glob
import sys

d, reads, per = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
names = sys.argv[4:]
rows = collections.defaultdict(dict)          # dispatch id -> counter -> value
for path in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(path)):
        if "pg_search_kernel" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(rows)
ctrs = sorted({c for v in rows.values() for c in v})
print(f"{'build':28s} " + " ".join(f"{c:>20s}" for c in ctrs))
for k, name in enumerate(names):
    mine = ids[k * per:(k + 1) * per]
    if not mine:
        break
    print(f"{name:28s} " + " ".join(f"{sum(rows[i].get(c, 0.0) for i in mine) / len(mine) / reads:20.1f}" for c in ctrs))

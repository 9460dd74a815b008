"""Diagnostics: per-read averages of the rocprofv3 --pmc counter CSVs under a directory."""
import collections
import csv
import glob
import sys

reads = float(sys.argv[2]) if len(sys.argv) > 2 else 2e6
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "pg_search_kernel" in r["Kernel_Name"]:      # (not pg_search_exact_kernel: a step launches it on an empty list)
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:24s} {sum(v) / len(v) / reads:12.1f} per read  ({len(v)} launches)")

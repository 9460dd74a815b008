"""Diagnostics: sums of the counters of a rocprofv3 --pmc run per kernel name.  usage: pmc_by_kernel.py <output dir>"""
import collections
import csv
import glob
import sys

t = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        t[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in t.items():
    print(k, {a: round(b) for a, b in v.items()})

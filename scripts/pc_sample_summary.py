"""Diagnostics: aggregates rocprofv3 PC-sampling CSVs: samples per instruction / per source line of pg_search_kernel."""
import collections
import csv
import glob
import sys

csv.field_size_limit(1 << 30)
for path in sorted(glob.glob(sys.argv[1] + "/**/*pc_sampling*.csv", recursive=True)):
    print("==", path)
    rd = csv.DictReader(open(path))
    print("columns:", rd.fieldnames)
    rows = 0
    by_inst = collections.Counter()
    by_line = collections.Counter()
    by_other = collections.defaultdict(collections.Counter)
    first = []
    for r in rd:
        rows += 1
        if rows <= 3:
            first.append(dict(r))
        inst = r.get("Instruction", "")
        com = r.get("Instruction_Comment", "")
        by_inst[(inst, com)] += 1
        by_line[com] += 1
        for k, v in r.items():
            if k in ("Instruction", "Instruction_Comment", "Sample_Timestamp", "Exec_Mask", "Dispatch_Id", "Correlation_Id",
                     "Timestamp"):
                continue
            if v is not None and len(v) < 40:
                by_other[k][v] += 1
    print("rows", rows)
    for f in first:
        print(f)
    for k, c in by_other.items():
        if len(c) <= 40:
            print("--", k, dict(c.most_common(40)))
    print("-- top source lines")
    for (k, v) in by_line.most_common(150):
        print(f"{v:9d} {100.0 * v / max(rows, 1):6.2f}%  {k}")
    print("-- top instructions")
    for (k, v) in by_inst.most_common(400):
        print(f"{v:9d} {100.0 * v / max(rows, 1):6.2f}%  {k[0]}   ; {k[1]}")

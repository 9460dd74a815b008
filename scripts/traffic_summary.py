"""Diagnostics: HBM traffic per read from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 64 B per 128-B request on gfx950 -> x2, checked
against the calibration kernel with the search kernel's own 4-B-per-lane pattern).

    traffic_summary.py <fetch_dir> <write_dir> <calib_dir> <reads per launch> [bench line json]
"""
import csv
import glob
import json
import sys


def counter(directory, kernel, name):
    vals = []
    for path in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


fetch_dir, write_dir, calib_dir, reads = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
fetch_kb, nf = counter(fetch_dir, "pg_search_kernel", "FETCH_SIZE")
write_kb, nw = counter(write_dir, "pg_search_kernel", "WRITE_SIZE")
calib_kb, _ = counter(calib_dir, "pg_calib_stream", "FETCH_SIZE")
true_bytes = 2147483648
factor = true_bytes / (calib_kb * 1024.0) if calib_kb else 2.0
alg = None
if len(sys.argv) > 5:
    try:
        line = [l for l in open(sys.argv[5]) if l.startswith("{")][-1]
        d = json.loads(line)
        alg = d["roofline"]["algorithmic_bytes_per_launch"] / d["config"]["reads_per_gpu"]
    except Exception:
        pass
out = {
    "workload": f"scripts/run_variant.py PG_STEP=1 (the step of bench.py: pg_search_kernel packing its claims in place), {int(reads)} reads "
                f"(100 bp, -x 2), one launch; averages over {nf} / {nw} launches",
    "FETCH_SIZE_raw_KB_per_launch": fetch_kb,
    "WRITE_SIZE_raw_KB_per_launch": write_kb,
    "calibration": {"kernel": "pg_calib_stream (4 B per lane, coalesced, 2 GiB > Infinity Cache)",
                    "true_bytes": true_bytes, "FETCH_SIZE_raw_KB": calib_kb, "factor": factor},
    "fetch_bytes_per_read": fetch_kb * 1024.0 * factor / reads if fetch_kb else None,
    "write_bytes_per_read_uncalibrated": write_kb * 1024.0 / reads if write_kb else None,
    "algorithmic_bytes_per_read": alg,
    "pack_bytes_per_read": "of the fetched / written bytes the pack inside the launch accounts for len + 19 read and 64 x blocks + 128 written per read "
                           "(100 bp: 119 + 256), the latter read back by the same wave through L2",
    "note": "FETCH_SIZE on gfx950 counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM): scaled by the "
            "factor of the calibration kernel. WRITE_SIZE is uncalibrated. Separate --pmc passes with --kernel-trace only.",
}
print(json.dumps(out, indent=1))

"""Diagnostics: per-read event counts of a -DPG_DIAG build (LDS fills, seed-filter runs, candidate passes, evaluations)
on the bench workload:  python scripts/diag_counts.py <lib.so> [reads] (PG_X=<n> selects -x n, PG_LEN the read length)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pindel_amd import binding, synth

binding.use_library(os.path.abspath(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
dev = torch.device("cuda", 0)
ref = synth.make_reference(62_435_964, seed=20260927, device=dev)
kw, rkw = {}, {}
if os.environ.get("PG_X"):
    kw["max_range_index"] = int(os.environ["PG_X"])
if os.environ.get("PG_LEN"):
    rkw["read_len"] = int(os.environ["PG_LEN"])
batch = synth.make_reads(ref, n, seed=20260928, device=dev, **rkw)
eng = binding.Engine(**kw)
eng.load_reference([("20", ref)])
db = eng.upload(batch)
eng.search_device(db)
L = binding.lib()
L.pg_debug_read_reserved.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
out = np.zeros(n, dtype=np.uint32)
assert L.pg_debug_read_reserved(eng._h, db, out.ctypes.data, n) == 0
names = ["LDS fills", "seed-filter runs", "candidate passes", "evaluations"]
for k, nm in enumerate(names):
    f = (out >> (8 * k)) & 0xff
    hist = np.bincount(f, minlength=8)
    print(f"{nm:18s} mean {f.mean():6.3f}  histogram 0..9: {(hist[:10] / n).round(3).tolist()}")
print("kernel ms", round(eng.last_stats()[0], 2))

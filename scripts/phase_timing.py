"""Diagnostics: wave-cycles per phase of the search kernel (a -DPG_TIMING build: s_memtime at the phase boundaries,
summed over all waves) on the bench workload:  python scripts/phase_timing.py <lib.so> [reads]   (PG_X / PG_LEN as usual)"""
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
eng.search_device(db)
L = binding.lib()
L.pg_debug_read_phase_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
out = np.zeros(12, dtype=np.uint64)
assert L.pg_debug_read_phase_cycles(eng._h, db, out.ctypes.data, 12) == 0
names = ["record, bases, planes, first window", "close 1st attempt: fill + filter + queue", "close 1st attempt: candidate pass",
         "close 1st attempt: evaluate + emit", "close retries: fill + filter + queue", "close retries: candidate passes",
         "close retries: evaluate + emit", "far: fill + filter + queue", "far: candidate pass(es)",
         "far: tier B folds + evaluations + emits", "output record", "claim + record load"]
tot = float(out.sum())
for nm, v in zip(names, out):
    print(f"{nm:46s} {float(v) / n:9.0f} cycles/read  {100.0 * float(v) / tot:5.1f} %")
print("total", round(tot / n), "cycles per read and wave; kernel ms", round(eng.last_stats()[0], 2))

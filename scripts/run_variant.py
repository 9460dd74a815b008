"""Diagnostics: run the device-resident search of the bench workload with another build of the library
(e.g. one compiled with -DPG_STOP_AFTER=n or -DPG_ABL_*), for rocprofv3 counter passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pindel_amd import binding, synth

binding.use_library(os.path.abspath(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
dev = torch.device("cuda", 0)
ref = synth.make_reference(62_435_964, seed=20260927, device=dev)
batch = synth.make_reads(ref, n, seed=20260928, device=dev)
kw = {}
if os.environ.get("PG_X"):
    kw["max_range_index"] = int(os.environ["PG_X"])
eng = binding.Engine(**kw)
eng.load_reference([("20", ref)])
db = eng.upload(batch)
for _ in range(3):
    eng.search_device(db)
print(os.path.basename(sys.argv[1]), "kernel ms", round(eng.last_stats()[0], 2), "candidates per read", round(eng.candidates(db) / n, 1))

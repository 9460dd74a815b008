"""Diagnostics: several builds of the library on the SAME bench reads in ONE process (one start-up, one synthetic batch):
   python scripts/run_variants_multi.py <reads> <lib.so> [<lib.so> ...]
Every build runs PG_LAUNCHES (default 3) search launches; prints name, best kernel ms, candidates per read, result digest.
Under `rocprofv3 --kernel-trace --pmc ...` the pg_search_kernel dispatches appear in this order, PG_LAUNCHES per build
(scripts/pmc_multi.py groups them).  PG_X=<n> selects -x n, PG_LEN=<bases> the read length."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pindel_amd import binding, synth

n = int(sys.argv[1])
libs = sys.argv[2:]
launches = int(os.environ.get("PG_LAUNCHES", "3"))
dev = torch.device("cuda", 0)
ref = synth.make_reference(62_435_964, seed=20260927, device=dev)
kw, rkw = {}, {}
if os.environ.get("PG_X"):
    kw["max_range_index"] = int(os.environ["PG_X"])
if os.environ.get("PG_LEN"):
    rkw["read_len"] = int(os.environ["PG_LEN"])
batch = synth.make_reads(ref, n, seed=20260928, device=dev, **rkw)
for path in libs:
    binding.use_library(os.path.abspath(path))
    eng = binding.Engine(**kw)
    eng.load_reference([("20", ref)])
    db = eng.upload(batch)
    ms = []
    for _ in range(launches):
        eng.search_device(db)
        ms.append(eng.last_stats()[0])
    res = eng.download(db)
    h = hashlib.sha256()
    for a in (res.close_off, res.far_off, res.rc_flag, res.close_runs, res.far_runs):
        h.update(a.tobytes())
    print(os.path.basename(path), "kernel ms", round(min(ms), 3), "candidates per read", round(eng.candidates(db) / n, 1),
          "digest", h.hexdigest()[:16], flush=True)
    eng.free_device_batch(db)
    eng.close()

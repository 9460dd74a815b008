"""Diagnostics: run the device-resident search of the bench workload with another build of the library
(e.g. one compiled with -DPG_STOP=n or -DPG_DUP=n), for rocprofv3 counter passes and A/B timing.  Prints the kernel
time, the candidates per read and a digest of the downloaded result (equal digests = bit-identical results).
  PG_X=<n> selects -x n, PG_LEN=<bases> the read length, PG_SORT=1 the reads in coordinate order, PG_STEP=1 the step of bench.py
  (pg_device_batch_pack_search: a million reads and more = one launch that packs in place) instead of the search of packed records."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pindel_amd import binding, synth

binding.use_library(os.path.abspath(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
dev = torch.device("cuda", 0)
ref = synth.make_reference(62_435_964, seed=20260927, device=dev)
kw, rkw = {}, {}
if os.environ.get("PG_X"):
    kw["max_range_index"] = int(os.environ["PG_X"])
if os.environ.get("PG_LEN"):
    rkw["read_len"] = int(os.environ["PG_LEN"])
if os.environ.get("PG_MIX"):          # fractions of (D, SI, TD, INV, none), e.g. the wgs-real workload's 0.04,0.02,0.02,0.02,0.9
    rkw["mix"] = tuple(float(x) for x in os.environ["PG_MIX"].split(","))
batch = synth.make_reads(ref, n, seed=20260928, device=dev, **rkw)
if os.environ.get("PG_SORT"):
    # coordinate order (every window a neighbour's window: the L2-resident upper bound of any prefetching)
    import numpy as np
    order = np.argsort(batch.anchor_pos, kind="stable")
    L = rkw.get("read_len", 100)
    batch = type(batch)(seq=batch.seq.reshape(batch.n, L)[order].reshape(-1), seq_off=batch.seq_off,
                        anchor_strand=batch.anchor_strand[order], anchor_pos=batch.anchor_pos[order],
                        insert_size=batch.insert_size[order], chr_id=batch.chr_id[order])
eng = binding.Engine(**kw)
eng.load_reference([("20", ref)])
db = eng.upload(batch)
ms = []
for _ in range(3):
    if os.environ.get("PG_STEP"):
        eng.pack_search_device(db)
    else:
        eng.search_device(db)
    ms.append(eng.last_stats()[0])
res = eng.download(db)
h = hashlib.sha256()
for a in (res.close_off, res.far_off, res.rc_flag, res.close_runs, res.far_runs):
    h.update(a.tobytes())
print(os.path.basename(sys.argv[1]), "kernel ms", round(min(ms), 3), "candidates per read", round(eng.candidates(db) / n, 1),
      "digest", h.hexdigest()[:16])

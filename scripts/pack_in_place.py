"""Diagnostics (GPU box): the step of bench.py both ways on the SAME reads in ONE process --
   A  pg_device_batch_repack + pg_device_batch_search   (pack launch, then search launch)
   B  pg_device_batch_pack_search                        (one launch: the search kernel packs its own claims)
   python scripts/pack_in_place.py <reads> [lib.so]      PG_X / PG_LEN as run_variants_multi.py; PG_PACK_CLAIM=<reads per claim>
Prints wall ms per step (best of PG_LAUNCHES), the search kernel's HIP-event ms, the result digests (must agree).
The records are overwritten with garbage before every B step: what B searches is what B packed."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pindel_amd import binding, synth

n = int(sys.argv[1])
if len(sys.argv) > 2:
    binding.use_library(os.path.abspath(sys.argv[2]))
launches = int(os.environ.get("PG_LAUNCHES", "5"))
dev = torch.device("cuda", 0)
ref = synth.make_reference(62_435_964, seed=20260927, device=dev)
kw, rkw = {}, {}
if os.environ.get("PG_X"):
    kw["max_range_index"] = int(os.environ["PG_X"])
if os.environ.get("PG_LEN"):
    rkw["read_len"] = int(os.environ["PG_LEN"])
batch = synth.make_reads(ref, n, seed=20260928, device=dev, **rkw)
if os.environ.get("PG_JUNK"):            # a character outside ACGTN in every PG_JUNK-th read (the exact kernel's list)
    import numpy as np
    at = batch.seq_off[:-1][::int(os.environ["PG_JUNK"])].astype(np.int64)
    batch.seq[at + (np.arange(len(at)) % 3) * 40] = ord("K")
eng = binding.Engine(**kw)
eng.load_reference([("20", ref)])
db = eng.upload(batch)


def digest():
    res = eng.download(db)
    h = hashlib.sha256()
    for a in (res.close_off, res.far_off, res.rc_flag, res.close_runs, res.far_runs):
        h.update(a.tobytes())
    return h.hexdigest()[:16]


def timed(step):
    wall, kern = [], []
    for _ in range(launches):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
        kern.append(eng.last_stats()[0])
    return min(wall), min(kern)


def step_a():
    eng.repack(db)
    eng.search_device(db)


wa, ka = timed(step_a)
da = digest()
eng.scribble_records(db)
wb, kb = timed(lambda: eng.pack_search_device(db))
dbg = digest()
print(f"reads {n}  A pack + search: wall {wa:.3f} ms (search kernel {ka:.3f})   B pack in place: wall {wb:.3f} ms (kernel {kb:.3f})   "
      f"B/A {wb / wa:.4f}   digests {da} {dbg} {'equal' if da == dbg else 'DIFFERENT'}", flush=True)
sys.exit(0 if da == dbg else 1)

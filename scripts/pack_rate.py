"""Diagnostics: HIP-event duration of the pack stage (pg_pack_kernel) on the bench workload's batch with a given build of
the library, and the digest of a search on what it packed (equal digests = the same planes and records).
  scripts/pack_rate.py <lib.so> [reads]      PG_LEN=<bases> selects the read length"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pindel_amd import binding, synth

binding.use_library(os.path.abspath(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
dev = torch.device("cuda", 0)
ref = synth.make_reference(62_435_964, seed=20260927, device=dev)
rkw = {"read_len": int(os.environ["PG_LEN"])} if os.environ.get("PG_LEN") else {}
batch = synth.make_reads(ref, n, seed=20260928, device=dev, **rkw)
eng = binding.Engine()
eng.load_reference([("20", ref)])
db = eng.upload(batch)
ms = [eng.repack(db) for _ in range(6)]
h = hashlib.sha256()
if not os.environ.get("PG_PACK_ONLY"):        # (experiment builds whose pack output is deliberately wrong: timing only)
    eng.search_device(db)
    res = eng.download(db)
    for a in (res.close_off, res.far_off, res.rc_flag, res.close_runs, res.far_runs):
        h.update(a.tobytes())
lens = batch.lengths()
ml = int(lens.max())
blocks = 1 if ml <= 64 else 2 if ml <= 128 else 3 if ml <= 192 else 4 if ml <= 256 else 8
nbytes = float(lens.sum()) + n * (8 + 11 + 64 * blocks + 128)
print(os.path.basename(sys.argv[1]), "pack ms", round(min(ms), 4), "GB/s", round(nbytes / min(ms) / 1e6, 1), "search ms",
      round(eng.last_stats()[0], 3), "digest", h.hexdigest()[:16])

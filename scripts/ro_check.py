"""Diagnostics: a -DPG_RO_CHECK build runs both seed filters on every run and counts the runs in which the read-order filter lost a
seed the symbol-by-symbol one keeps:  python scripts/ro_check.py <lib.so> [reads]"""
import ctypes as C
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pindel_amd import binding, synth

binding.use_library(os.path.abspath(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
dev = torch.device("cuda", 0)
ref = synth.make_reference(62_435_964, seed=20260927, device=dev)
batch = synth.make_reads(ref, n, seed=20260928, device=dev)
eng = binding.Engine()
eng.load_reference([("20", ref)])
db = eng.upload(batch)
eng.search_device(db)
L = binding.lib()
L.pg_debug_read_phase_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
out = np.zeros(12, dtype=np.uint64)
assert L.pg_debug_read_phase_cycles(eng._h, db, out.ctypes.data, 12) == 0
res = eng.download(db)
h = hashlib.sha256()
for a in (res.close_off, res.far_off, res.rc_flag, res.close_runs, res.far_runs):
    h.update(a.tobytes())
print(os.path.basename(sys.argv[1]), "runs", int(out[0]), "runs with a lost seed: F", int(out[1]), "B", int(out[2]), "DUAL", int(out[3]),
      "lanes", int(out[4]), "bad programs", int(out[5]), "digest", h.hexdigest()[:16])

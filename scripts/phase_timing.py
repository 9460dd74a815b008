"""Diagnostics: per-phase wave cycles of the search kernels.  Needs pindel_amd/libpindel_pg_timing.so
(built with -DPG_PHASE_TIMING: the kernels store s_memtime deltas of the first 65536 reads at the end
of the alg-bytes array)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pindel_amd import binding, synth

binding.LIB_PATH = os.path.join(os.path.dirname(binding.LIB_PATH), "libpindel_pg_timing.so")
dev = torch.device("cuda", 0)
ref = synth.make_reference(62_435_964, seed=20260927, device=dev)
n = 4_000_000
batch = synth.make_reads(ref, n, seed=20260928, device=dev)
eng = binding.Engine()
eng.load_reference([("20", ref)])
db = eng.upload(batch)
eng.search_device(db)
eng.search_device(db)
L = binding.lib()
raw = np.zeros(n, dtype=np.uint32)
L.pg_debug_read_alg.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
assert L.pg_debug_read_alg(eng._h, db, raw.ctypes.data, n) == 0
d = raw[n - 65536 * 16:].reshape(65536, 16).astype(np.float64)
names = ["load planes", "scan (stage+filter+dense)", "evaluate", "publish/clean", "zero hist", "configure", "tail", "-"]
for k, kern in ((0, "close kernel"), (8, "far kernel")):
    tot = d[:, k:k + 8].sum(axis=1)
    print(kern, "mean cycles per read", round(tot.mean()))
    for j in range(8):
        print(f"   {names[j]:32s} {d[:, k + j].mean():10.0f}  {100 * d[:, k + j].mean() / tot.mean():5.1f} %")

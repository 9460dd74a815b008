"""PCIe-inclusive rate of the host-buffer entry point pg_search_batch (upload + kernel + D2H of the runs
+ host CSR rebuild), for DESIGN.md.  Never used as bench `value`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pindel_amd import binding, synth

dev = torch.device("cuda", 0)
ref = synth.make_reference(62_435_964, seed=20260927, device=dev)
n = 4_000_000
batch = synth.make_reads(ref, n, seed=20260928, device=dev)
eng = binding.Engine()
eng.load_reference([("20", ref)])
s, keep = binding._batch_struct(batch)
import ctypes as C
for rep in range(3):
    h = C.c_void_p()
    t0 = time.perf_counter()
    rc = eng._L.pg_search_batch(eng._h, C.byref(s), C.byref(h))
    dt = time.perf_counter() - t0
    assert rc == 0
    eng._L.pg_result_free(h)
    print(f"pg_search_batch (host in / host out): {n / dt / 1e6:.2f} M reads/s  ({dt * 1e3:.1f} ms for {n} reads)")

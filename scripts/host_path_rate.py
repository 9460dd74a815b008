"""Diagnostics: PCIe-inclusive rate of pg_search_batch (host buffers in, host CSR out) on the bench workload."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pindel_amd import binding, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = torch.device("cuda", 0)
ref = synth.make_reference(62_435_964, seed=20260927, device=dev)
batch = synth.make_reads(ref, n, seed=20260928, device=dev)
eng = binding.Engine()
eng.load_reference([("20", ref)])
s, keep = binding._batch_struct(batch)
import ctypes as C
L = binding.lib()
for it in range(3):
    h = C.c_void_p()
    t0 = time.perf_counter()
    rc = L.pg_search_batch(eng._h, C.byref(s), C.byref(h))
    dt = time.perf_counter() - t0
    assert rc == 0
    L.pg_result_free(h)
    print(f"pg_search_batch {n} reads: {dt * 1e3:.1f} ms -> {n / dt / 1e6:.1f} M reads/s (kernel {eng.last_stats()[0]:.1f} ms)")

"""Diagnostics: pg_search_batch (host buffers in, host CSR out, PCIe both ways) called six times on the same batch, per size:
the first calls pin the result buffers and grow the device arena, the later ones are the steady state.
  scripts/host_path_calls.py <reads> [<reads> ...]        PG_HOST_TIMING=1 adds the library's own stage times"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pindel_amd import binding, synth

dev = torch.device("cuda", 0)
ref = synth.make_reference(62_435_964, seed=20260927, device=dev)
eng = binding.Engine()
eng.load_reference([("20", ref)])
L = binding.lib()
for n in [int(a) for a in sys.argv[1:]] or [50000, 4000000]:
    batch = synth.make_reads(ref, n, seed=20260928, device=dev)
    s, keep = binding._batch_struct(batch)
    best = None
    for it in range(6):
        h = C.c_void_p()
        t0 = time.perf_counter()
        rc = L.pg_search_batch(eng._h, C.byref(s), C.byref(h))
        dt = time.perf_counter() - t0
        assert rc == 0
        L.pg_result_free(h)
        best = dt if best is None else min(best, dt)
        print(f"pg_search_batch {n} reads, call {it}: {dt * 1e3:.3f} ms = {n / dt / 1e6:.1f} M reads/s (kernels {eng.last_stats()[0]:.2f} ms)", flush=True)
    print(f"pg_search_batch {n} reads: best {best * 1e3:.3f} ms = {n / best / 1e6:.1f} M reads/s")

# the two seams (pg_close_end_batch at ReadBuffer::flush, pg_far_end_batch_from_close before the far-end searches), 50 000 reads
batch = synth.make_reads(ref, 50000, seed=20260928, device=dev)
tc, tf = [], []
for it in range(6):
    t0 = time.perf_counter()
    close = eng.close_end_batch(batch)
    t1 = time.perf_counter()
    eng.far_end_batch(batch, close)
    t2 = time.perf_counter()
    tc.append(t1 - t0)
    tf.append(t2 - t1)
print(f"seams, 50000 reads (python binding): close end best {min(tc) * 1e3:.3f} ms, far end from close best {min(tf) * 1e3:.3f} ms")

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04c
{
for n in 1000000 4000000; do
PG_HOST_TIMING=1 python - $n <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, ctypes as C
from pindel_amd import binding, synth
n = int(sys.argv[1])
dev = torch.device("cuda", 0)
ref = synth.make_reference(62_435_964, seed=20260927, device=dev)
batch = synth.make_reads(ref, n, seed=20260928, device=dev)
eng = binding.Engine(); eng.load_reference([("20", ref)])
s, keep = binding._batch_struct(batch)
L = binding.lib()
for it in range(6):
    h = C.c_void_p(); t0 = time.perf_counter()
    rc = L.pg_search_batch(eng._h, C.byref(s), C.byref(h)); dt = time.perf_counter() - t0
    assert rc == 0
    t1 = time.perf_counter(); L.pg_result_free(h); t2 = time.perf_counter()
    print(f"call {it}: {dt*1e3:.2f} ms ({n/dt/1e6:.1f} M/s), free {1e3*(t2-t1):.2f} ms", flush=True)
PY
done
} > gpurun_out/r04c/host2.txt 2>&1
cat gpurun_out/r04c/host2.txt

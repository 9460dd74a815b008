"""Calibrates rocprofv3's FETCH_SIZE for this kernel's staging pattern (4 bytes per lane, coalesced):
reads a 2 GiB buffer (> the 256 MiB Infinity Cache) once; run under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- python scripts/calib_fetch.py
and compare the counter of pg_calib_stream with the known byte count printed here."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pindel_amd import binding

L = binding.lib()
n = 512 * 1024 * 1024          # dwords = 2 GiB
x = torch.randint(0, 2**31 - 1, (n,), dtype=torch.int32, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
L.pg_debug_calib_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
for _ in range(2):
    assert L.pg_debug_calib_stream(x.data_ptr(), n, sink.data_ptr()) == 0
print("bytes per launch:", n * 4)

"""ctypes binding of the host-side library (pindel_amd/libpindel_host.so): loaders, SV
classifiers and text reporters written in C++ (pindel_amd/csrc/host/).  No search code."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpindel_host.so")


class HostSettings(C.Structure):
    _fields_ = [
        ("spacer", C.c_uint32), ("min_support", C.c_uint32), ("balance_cutoff", C.c_uint32),
        ("seq_error_rate", C.c_double), ("min_num_matched_bases", C.c_int32),
        ("min_inversion_size", C.c_int32), ("analyze_td", C.c_int32), ("analyze_inv", C.c_int32),
        ("window_mbp", C.c_double), ("max_mismatch", C.c_uint32 * 500)]


def build(force=False):
    src_dir = os.path.join(_HERE, "csrc")
    host_dir = os.path.join(src_dir, "host")
    srcs = [os.path.join(host_dir, f) for f in os.listdir(host_dir)]
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", src_dir, "host"] + (["-B"] if force else []),
                              stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.pgh_last_error.restype = C.c_char_p
        L.pgh_call_from_points.argtypes = [
            C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(HostSettings), C.c_uint32,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def default_settings(max_mismatch) -> HostSettings:
    """Pindel 0.2.5b9 defaults of the flags the classifiers/reporters read."""
    s = HostSettings()
    s.spacer = 100000
    s.min_support = 1          # -M
    s.balance_cutoff = 100     # -B
    s.seq_error_rate = 0.01    # -e
    s.min_num_matched_bases = 30   # -d
    s.min_inversion_size = 50      # -v
    s.analyze_td = 1
    s.analyze_inv = 1
    s.window_mbp = 5.0
    for i in range(500):
        s.max_mismatch[i] = int(max_mismatch[i])
    return s


def call_from_points(fasta, reads_txt, out_prefix, settings, close_off, close_pts, far_off, far_pts,
                     rc_flag):
    """Classify + report (_D, _SI, _TD, _INV) from per-read UP_Close / UP_Far points (CSR over
    all reads of the file, 12-byte pg_point records)."""
    L = lib()
    close_off = np.ascontiguousarray(close_off, dtype=np.uint64)
    far_off = np.ascontiguousarray(far_off, dtype=np.uint64)
    close_pts = np.ascontiguousarray(close_pts)
    far_pts = np.ascontiguousarray(far_pts)
    rc_flag = np.ascontiguousarray(rc_flag, dtype=np.uint8)
    assert close_pts.dtype.itemsize == 12 and far_pts.dtype.itemsize == 12
    rc = L.pgh_call_from_points(str(fasta).encode(), str(reads_txt).encode(), str(out_prefix).encode(),
                                C.byref(settings), len(close_off) - 1, close_off.ctypes.data,
                                close_pts.ctypes.data, far_off.ctypes.data, far_pts.ctypes.data,
                                rc_flag.ctypes.data)
    if rc:
        raise RuntimeError("pgh_call_from_points: " + (L.pgh_last_error() or b"").decode())

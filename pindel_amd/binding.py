"""ctypes binding of the C ABI in include/pindel_pg.h (pindel_amd/libpindel_pg.so).

This is glue only: every computation happens in the HIP library.  If the library
is missing or no MI355X is visible the calls raise -- there is no Python or CPU
implementation of the search behind this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PG_LIBRARY") or os.path.join(_HERE, "libpindel_pg.so")      # (PG_LIBRARY: an experiment build, scripts/build_variant.sh)

PG_OK = 0
PG_E_DEVICE = -3


class PgError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"pindel_pg error {code}: {msg}")
        self.code = code


class PgParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("max_range_index", C.c_int32),
        ("additional_mismatch", C.c_int32), ("min_perfect_match_around_bp", C.c_int32),
        ("min_close", C.c_int32), ("max_allowed_mismatch_rate", C.c_double),
        ("seq_error_rate", C.c_double), ("sensitivity", C.c_double),
        ("spacer", C.c_uint32), ("reserved", C.c_uint32)]


class PgReadBatch(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint32), ("seq", C.c_void_p), ("seq_off", C.c_void_p),
        ("anchor_strand", C.c_void_p), ("anchor_pos", C.c_void_p),
        ("insert_size", C.c_void_p), ("chr_id", C.c_void_p)]


class PgWindows(C.Structure):
    _fields_ = [("offset", C.c_void_p), ("windows", C.c_void_p)]


class PgResultView(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint32), ("close_off", C.POINTER(C.c_uint64)), ("close_runs", C.c_void_p),
        ("far_off", C.POINTER(C.c_uint64)), ("far_runs", C.c_void_p),
        ("rc_flag", C.POINTER(C.c_uint8))]


RUN_DTYPE = np.dtype([("abs_loc_first", "<u4"), ("len_first", "<u2"), ("len_last", "<u2"),
                      ("mismatches", "u1"), ("flags", "u1"), ("chr_id", "<i2")], align=True)
POINT_DTYPE = np.dtype([("abs_loc", "<u4"), ("length", "<i2"), ("mismatches", "<i2"),
                        ("chr_id", "<i2"), ("direction", "S1"), ("strand", "S1")], align=True)
WINDOW_DTYPE = np.dtype([("chr_id", "<i4"), ("start", "<i4"), ("end", "<i4")], align=True)
assert RUN_DTYPE.itemsize == 12 and POINT_DTYPE.itemsize == 12 and WINDOW_DTYPE.itemsize == 12

# every symbol include/pindel_pg.h declares
EXPORTS = [
    "pg_default_params", "pg_create", "pg_destroy", "pg_last_error", "pg_get_max_mismatch",
    "pg_load_reference", "pg_load_fasta", "pg_reference_save_packed", "pg_reference_load_packed", "pg_reference_n_chr", "pg_reference_name",
    "pg_reference_comp_size", "pg_reference_fetch", "pg_close_end_batch", "pg_far_end_batch",
    "pg_far_end_batch_from_close",
    "pg_search_batch", "pg_search_batch_multi", "pg_result_view_get", "pg_result_free", "pg_expand_runs",
    "pg_device_batch_upload", "pg_device_batch_set_windows", "pg_device_batch_search", "pg_device_batch_download",
    "pg_device_batch_free", "pg_last_search_stats", "pg_device_batch_algorithmic_bytes",
    "pg_device_batch_candidates", "pg_device_batch_repack", "pg_device_batch_pack_search"]


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the library (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir)
            if f.endswith((".hip", ".cpp", ".h"))] + [
        os.path.join(_HERE, "..", "include", "pindel_pg.h")]
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        r = subprocess.run(["make", "-C", src_dir, "-B", "all"], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"building {LIB_PATH} failed (make exit {r.returncode}):\n{r.stdout[-8000:]}")
    return LIB_PATH


_lib = None


def use_library(path):
    """Tests: bind a differently configured build of the library (call before the first use)."""
    global _lib, LIB_PATH
    _lib = None
    LIB_PATH = path


def reload_env():
    """The library reads its PG_* environment switches once per process; tests that change one call this afterwards."""
    L = lib()
    L.pg_debug_reload_env.restype = None
    L.pg_debug_reload_env()


def lib():
    """Load the shared library (building it first if the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, u64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64
    L.pg_default_params.argtypes = [C.POINTER(PgParams)]
    L.pg_default_params.restype = None
    L.pg_create.argtypes = [C.POINTER(PgParams), C.POINTER(vp)]
    L.pg_destroy.argtypes = [vp]
    L.pg_destroy.restype = None
    L.pg_last_error.argtypes = [vp]
    L.pg_last_error.restype = C.c_char_p
    L.pg_get_max_mismatch.argtypes = [vp, vp]
    L.pg_load_reference.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(vp), C.POINTER(u64)]
    L.pg_load_fasta.argtypes = [vp, C.c_char_p]
    L.pg_reference_save_packed.argtypes = [vp, C.c_char_p]
    L.pg_reference_load_packed.argtypes = [vp, C.c_char_p]
    L.pg_reference_n_chr.argtypes = [vp]
    L.pg_reference_name.argtypes = [vp, i32]
    L.pg_reference_name.restype = C.c_char_p
    L.pg_reference_comp_size.argtypes = [vp, i32]
    L.pg_reference_comp_size.restype = u64
    L.pg_reference_fetch.argtypes = [vp, i32, u64, u64, vp]
    L.pg_close_end_batch.argtypes = [vp, C.POINTER(PgReadBatch), C.POINTER(vp)]
    L.pg_far_end_batch.argtypes = [vp, C.POINTER(PgReadBatch), vp, C.POINTER(PgWindows)]
    L.pg_far_end_batch_from_close.argtypes = [vp, C.POINTER(PgReadBatch), vp, vp, C.POINTER(PgWindows), C.POINTER(vp)]
    L.pg_search_batch.argtypes = [vp, C.POINTER(PgReadBatch), C.POINTER(vp)]
    L.pg_search_batch_multi.argtypes = [C.POINTER(vp), i32, C.POINTER(PgReadBatch), C.POINTER(vp)]
    L.pg_result_view_get.argtypes = [vp, C.POINTER(PgResultView)]
    L.pg_result_free.argtypes = [vp]
    L.pg_result_free.restype = None
    L.pg_expand_runs.argtypes = [vp, u64, vp]
    L.pg_expand_runs.restype = u64
    L.pg_device_batch_upload.argtypes = [vp, C.POINTER(PgReadBatch), C.POINTER(vp)]
    L.pg_device_batch_set_windows.argtypes = [vp, vp, C.POINTER(PgWindows)]
    L.pg_device_batch_search.argtypes = [vp, vp]
    L.pg_device_batch_download.argtypes = [vp, vp, C.POINTER(vp)]
    L.pg_device_batch_free.argtypes = [vp, vp]
    L.pg_device_batch_free.restype = None
    L.pg_last_search_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64)]
    L.pg_device_batch_algorithmic_bytes.argtypes = [vp, vp, C.POINTER(C.c_double)]
    L.pg_device_batch_candidates.argtypes = [vp, vp, C.POINTER(C.c_double)]
    L.pg_device_batch_repack.argtypes = [vp, vp, C.POINTER(C.c_double)]
    L.pg_device_batch_pack_search.argtypes = [vp, vp]
    _lib = L
    return L


def expand_runs(runs: np.ndarray) -> np.ndarray:
    """pg_expand_runs: run-length-encoded runs -> UniquePoint records."""
    L = lib()
    runs = np.ascontiguousarray(runs, dtype=RUN_DTYPE)
    n = L.pg_expand_runs(runs.ctypes.data, len(runs), None)
    out = np.zeros(n, dtype=POINT_DTYPE)
    if n:
        L.pg_expand_runs(runs.ctypes.data, len(runs), out.ctypes.data)
    return out


class Result:
    """Host copy of a pg_result: CSR runs for UP_Close / UP_Far + rc flags."""

    def __init__(self, handle, owner):
        self._h = handle
        self._owner = owner
        v = PgResultView()
        rc = lib().pg_result_view_get(handle, C.byref(v))
        if rc:
            raise PgError(rc)
        n = v.n_reads
        self.n = n
        self.close_off = np.ctypeslib.as_array(v.close_off, shape=(n + 1,)).copy()
        self.far_off = np.ctypeslib.as_array(v.far_off, shape=(n + 1,)).copy()
        self.rc_flag = np.ctypeslib.as_array(v.rc_flag, shape=(n,)).copy() if n else np.zeros(0, np.uint8)

        def runs(ptr, cnt):
            if not cnt:
                return np.zeros(0, dtype=RUN_DTYPE)
            buf = (C.c_uint8 * (cnt * 12)).from_address(ptr)
            return np.frombuffer(buf, dtype=RUN_DTYPE).copy()
        self.close_runs = runs(v.close_runs, int(self.close_off[-1]))
        self.far_runs = runs(v.far_runs, int(self.far_off[-1]))

    def refresh(self):
        self.__init__(self._h, self._owner)

    def close_points(self, i):
        return expand_runs(self.close_runs[int(self.close_off[i]):int(self.close_off[i + 1])])

    def far_points(self, i):
        return expand_runs(self.far_runs[int(self.far_off[i]):int(self.far_off[i + 1])])

    def free(self):
        if self._h:
            lib().pg_result_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _batch_struct(batch):
    """batch: pindel_amd.hostio.ReadBatch -> (PgReadBatch, keepalive)."""
    seq = np.ascontiguousarray(batch.seq, dtype=np.uint8)
    off = np.ascontiguousarray(batch.seq_off, dtype=np.uint64)
    st = np.ascontiguousarray(batch.anchor_strand, dtype=np.uint8)
    pos = np.ascontiguousarray(batch.anchor_pos, dtype=np.int32)
    isz = np.ascontiguousarray(batch.insert_size, dtype=np.int16)
    cid = np.ascontiguousarray(batch.chr_id, dtype=np.int32)
    s = PgReadBatch(len(off) - 1, seq.ctypes.data, off.ctypes.data, st.ctypes.data,
                    pos.ctypes.data, isz.ctypes.data, cid.ctypes.data)
    return s, (seq, off, st, pos, isz, cid)


class Engine:
    """One pg_ctx = one GPU."""

    def __init__(self, device=0, max_range_index=2, additional_mismatch=1, min_perfect_match=3,
                 min_close=8, max_mismatch_rate=0.02, seq_error_rate=0.01, sensitivity=0.95,
                 spacer=100000):
        L = lib()
        p = PgParams()
        L.pg_default_params(C.byref(p))
        p.device = device
        p.max_range_index = max_range_index
        p.additional_mismatch = additional_mismatch
        p.min_perfect_match_around_bp = min_perfect_match
        p.min_close = min_close
        p.max_allowed_mismatch_rate = max_mismatch_rate
        p.seq_error_rate = seq_error_rate
        p.sensitivity = sensitivity
        p.spacer = spacer
        h = C.c_void_p()
        rc = L.pg_create(C.byref(p), C.byref(h))
        if rc:
            raise PgError(rc, "pg_create failed (no usable HIP device?)" if rc == PG_E_DEVICE else "pg_create")
        self._h = h
        self._L = L

    def _check(self, rc):
        if rc:
            raise PgError(rc, (self._L.pg_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            self._L.pg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def max_mismatch_table(self):
        t = np.zeros(500, dtype=np.uint32)
        self._check(self._L.pg_get_max_mismatch(self._h, t.ctypes.data))
        return t

    def load_reference(self, chroms):
        """chroms: [(name, padded_bytes)] as hostio.load_fasta returns."""
        n = len(chroms)
        names = (C.c_char_p * n)(*[nm.encode() for nm, _ in chroms])
        bufs = [np.frombuffer(s, dtype=np.uint8) for _, s in chroms]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        lens = (C.c_uint64 * n)(*[len(b) for b in bufs])
        self._check(self._L.pg_load_reference(self._h, n, names, ptrs, lens))

    def load_fasta(self, path):
        self._check(self._L.pg_load_fasta(self._h, str(path).encode()))

    def save_packed(self, path):
        self._check(self._L.pg_reference_save_packed(self._h, str(path).encode()))

    def load_packed(self, path):
        self._check(self._L.pg_reference_load_packed(self._h, str(path).encode()))

    def reference_info(self):
        n = self._L.pg_reference_n_chr(self._h)
        return [(self._L.pg_reference_name(self._h, c).decode(),
                 int(self._L.pg_reference_comp_size(self._h, c))) for c in range(n)]

    def reference_fetch(self, chr_id, start, n):
        out = np.zeros(n, dtype=np.uint8)
        self._check(self._L.pg_reference_fetch(self._h, chr_id, start, n, out.ctypes.data))
        return out.tobytes()

    # ---- host in / host out
    def close_end_batch(self, batch) -> Result:
        s, keep = _batch_struct(batch)
        h = C.c_void_p()
        self._check(self._L.pg_close_end_batch(self._h, C.byref(s), C.byref(h)))
        return Result(h, self)

    def far_end_batch(self, batch, close: Result, bd=None, bd_off=None) -> Result:
        s, keep = _batch_struct(batch)
        w = None
        if bd is not None:
            bd = np.ascontiguousarray(bd, dtype=WINDOW_DTYPE)
            bd_off = np.ascontiguousarray(bd_off, dtype=np.uint64)
            w = PgWindows(bd_off.ctypes.data, bd.ctypes.data)
        self._check(self._L.pg_far_end_batch(self._h, C.byref(s), close._h,
                                             C.byref(w) if w is not None else None))
        close.refresh()
        return close

    def far_end_batch_from_close(self, batch, close_last, close_max, bd=None, bd_off=None) -> Result:
        """pg_far_end_batch_from_close: the reads as the close end left them + UP_Close.back()'s AbsLoc / LengthStr."""
        s, keep = _batch_struct(batch)
        cl = np.ascontiguousarray(close_last, dtype=np.uint32)
        cm = np.ascontiguousarray(close_max, dtype=np.int16)
        w = None
        if bd is not None:
            bd = np.ascontiguousarray(bd, dtype=WINDOW_DTYPE)
            bd_off = np.ascontiguousarray(bd_off, dtype=np.uint64)
            w = PgWindows(bd_off.ctypes.data, bd.ctypes.data)
        h = C.c_void_p()
        self._check(self._L.pg_far_end_batch_from_close(self._h, C.byref(s), cl.ctypes.data, cm.ctypes.data,
                                                        C.byref(w) if w is not None else None, C.byref(h)))
        return Result(h, self)

    def search_batch(self, batch) -> Result:
        s, keep = _batch_struct(batch)
        h = C.c_void_p()
        self._check(self._L.pg_search_batch(self._h, C.byref(s), C.byref(h)))
        return Result(h, self)

    @staticmethod
    def search_batch_multi(engines, batch) -> "Result":
        """pg_search_batch_multi: contiguous shards of one batch on several contexts (one GPU each), concatenated."""
        s, keep = _batch_struct(batch)
        hs = (C.c_void_p * len(engines))(*[e._h for e in engines])
        h = C.c_void_p()
        rc = lib().pg_search_batch_multi(hs, len(engines), C.byref(s), C.byref(h))
        if rc:
            raise PgError(rc, (lib().pg_last_error(engines[0]._h) or b"").decode())
        return Result(h, engines[0])

    # ---- device resident
    def upload(self, batch):
        s, keep = _batch_struct(batch)
        h = C.c_void_p()
        self._check(self._L.pg_device_batch_upload(self._h, C.byref(s), C.byref(h)))
        return h

    def set_windows(self, dbatch, bd=None, bd_off=None):
        """Per-read BreakDancer window clusters for a device-resident batch (None detaches them)."""
        w = None
        if bd is not None:
            bd = np.ascontiguousarray(bd, dtype=WINDOW_DTYPE)
            bd_off = np.ascontiguousarray(bd_off, dtype=np.uint64)
            w = PgWindows(bd_off.ctypes.data, bd.ctypes.data)
        self._check(self._L.pg_device_batch_set_windows(self._h, dbatch, C.byref(w) if w is not None else None))

    def search_device(self, dbatch):
        self._check(self._L.pg_device_batch_search(self._h, dbatch))

    def pack_search_device(self, dbatch):
        """repack + search_device as one step (large batches: one launch, the search kernel packs its own reads)"""
        self._check(self._L.pg_device_batch_pack_search(self._h, dbatch))

    def last_step_in_place(self) -> bool:
        """tests: did the last pack_search_device build its records inside the search kernel (one launch)?"""
        self._L.pg_debug_last_pack_in_place.argtypes = [C.c_void_p]
        return bool(self._L.pg_debug_last_pack_in_place(self._h))

    def scribble_records(self, dbatch):
        """tests: overwrite the packed records and planes of the batch"""
        self._L.pg_debug_scribble_records.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self._L.pg_debug_scribble_records(self._h, dbatch))

    def repack(self, dbatch) -> float:
        """The pack stage (ASCII bases -> bit planes + packed records) again on the resident batch; HIP-event ms."""
        ms = C.c_double()
        self._check(self._L.pg_device_batch_repack(self._h, dbatch, C.byref(ms)))
        return ms.value

    def download(self, dbatch) -> Result:
        h = C.c_void_p()
        self._check(self._L.pg_device_batch_download(self._h, dbatch, C.byref(h)))
        return Result(h, self)

    def free_device_batch(self, dbatch):
        self._L.pg_device_batch_free(self._h, dbatch)

    def last_stats(self):
        ms = C.c_double()
        runs = C.c_uint64()
        self._check(self._L.pg_last_search_stats(self._h, C.byref(ms), C.byref(runs)))
        return ms.value, runs.value

    def candidates(self, dbatch):
        """Diagnostics: seed-filter survivors folded by the last search of the batch."""
        n = C.c_double()
        self._check(self._L.pg_device_batch_candidates(self._h, dbatch, C.byref(n)))
        return n.value

    def algorithmic_bytes(self, dbatch):
        b = C.c_double()
        self._check(self._L.pg_device_batch_algorithmic_bytes(self._h, dbatch, C.byref(b)))
        return b.value

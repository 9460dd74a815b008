"""Shared helpers for the parity tests: compare the HIP path (through the C ABI) with the
CPU oracle point by point."""
import numpy as np

from oracle import pyoracle
from pindel_amd import binding


def oracle_points(res, i, which):
    cnt = res[which + "_cnt"][i]
    return res[which + "_pts"][i][:cnt]


def assert_same_points(gpu_pts, orc_pts, what):
    assert len(gpu_pts) == len(orc_pts), f"{what}: {len(gpu_pts)} points vs oracle {len(orc_pts)}"
    for f in ("abs_loc", "length", "mismatches", "chr_id", "direction", "strand"):
        if not np.array_equal(gpu_pts[f], orc_pts[f]):
            k = int(np.nonzero(gpu_pts[f] != orc_pts[f])[0][0])
            raise AssertionError(f"{what}: field {f} differs at point {k}: gpu {gpu_pts[k]} oracle {orc_pts[k]}")


def points_per_read(off, runs):
    """UniquePoints each read owns, from its CSR offsets (in runs) and the run lengths."""
    pts = runs["len_last"].astype(np.int64) - runs["len_first"].astype(np.int64) + 1
    assert (pts > 0).all(), "a run with len_last < len_first"
    cum = np.concatenate([[0], np.cumsum(pts)])
    off = np.asarray(off, dtype=np.int64)
    assert off[0] == 0 and (np.diff(off) >= 0).all() and off[-1] == len(runs), "CSR offsets are not a partition of the runs"
    return cum[off[1:]] - cum[off[:-1]]


def compare_result(gpu: "binding.Result", orc: dict, n: int, check_far=True):
    """Bit-exact comparison of UP_Close, UP_Far and the rc flag for every read."""
    assert gpu.n == n
    np.testing.assert_array_equal(gpu.rc_flag, orc["rc_flag"], err_msg="rc_flag")
    # read boundaries first: a list attributed to the neighbouring read must not pass the concatenated comparison below
    np.testing.assert_array_equal(points_per_read(gpu.close_off, gpu.close_runs), orc["close_cnt"][:n],
                                  err_msg="UP_Close points per read")
    if check_far:
        np.testing.assert_array_equal(points_per_read(gpu.far_off, gpu.far_runs), orc["far_cnt"][:n],
                                      err_msg="UP_Far points per read")
    # whole-batch comparison on the expanded point arrays (fast path), then per read on failure
    g_close = binding.expand_runs(gpu.close_runs)
    o_close = np.concatenate([oracle_points(orc, i, "close") for i in range(n)]) if n else g_close
    g_far = binding.expand_runs(gpu.far_runs)
    o_far = np.concatenate([oracle_points(orc, i, "far") for i in range(n)]) if n else g_far
    ok = len(g_close) == len(o_close) and g_close.tobytes() == o_close.tobytes()
    if check_far:
        ok = ok and len(g_far) == len(o_far) and g_far.tobytes() == o_far.tobytes()
    if ok:
        return
    for i in range(n):
        assert_same_points(gpu.close_points(i), oracle_points(orc, i, "close"), f"read {i} UP_Close")
        if check_far:
            assert_same_points(gpu.far_points(i), oracle_points(orc, i, "far"), f"read {i} UP_Far")
    raise AssertionError("byte comparison failed but per-read comparison passed (padding?)")


def run_oracle(params_kw, chroms, batch, bd=None, bd_off=None, do_far=True):
    p = pyoracle.make_params(**params_kw)
    return pyoracle.search_batch(p, [s for _, s in chroms], batch.seq, batch.seq_off,
                                 batch.anchor_strand, batch.anchor_pos, batch.insert_size,
                                 batch.chr_id, bd=bd, bd_off=bd_off, do_far=do_far, n_threads=0)

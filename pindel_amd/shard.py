"""Multi-GPU sharding of a read batch: reads are independent, so rank r
searches a CONTIGUOUS range of read indices on its own GPU (reference replicated per GPU) and the
per-rank run lists are concatenated on the host in rank order == input order.  No collective is
on the data path; torch.distributed is only used to hand the (small) result arrays to rank 0.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, world: int):
    """Balanced contiguous split of n reads over `world` ranks: list of (lo, hi)."""
    base, rem = divmod(n, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def result_arrays(res):
    """The arrays of a binding.Result (or anything shaped like it) as a plain dict."""
    return dict(close_off=np.asarray(res.close_off), close_runs=np.asarray(res.close_runs),
                far_off=np.asarray(res.far_off), far_runs=np.asarray(res.far_runs),
                rc_flag=np.asarray(res.rc_flag))


def concat_results(parts):
    """Concatenate per-rank CSR results (in rank order) into one CSR over all reads."""
    def cat_off(key):
        offs, base = [np.zeros(1, dtype=np.uint64)], 0
        for p in parts:
            o = p[key].astype(np.uint64)
            offs.append(o[1:] + np.uint64(base))
            base += int(o[-1])
        return np.concatenate(offs)
    return dict(close_off=cat_off("close_off"), far_off=cat_off("far_off"),
                close_runs=np.concatenate([p["close_runs"] for p in parts]),
                far_runs=np.concatenate([p["far_runs"] for p in parts]),
                rc_flag=np.concatenate([p["rc_flag"] for p in parts]))


def search_sharded(search_fn, batch, rank: int, world: int, dist=None):
    """Run `search_fn(sub_batch)` on this rank's shard; rank 0 returns the concatenated result
    (other ranks return None).  `dist` is torch.distributed (initialised) or None for world == 1."""
    lo, hi = shard_bounds(batch.n, world)[rank]
    local = result_arrays(search_fn(batch.slice(lo, hi)))
    if world == 1 or dist is None:
        return local
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0)
    return concat_results(gathered) if rank == 0 else None

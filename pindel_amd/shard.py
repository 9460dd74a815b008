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


def read_digests(res) -> np.ndarray:
    """One 64-bit digest per read over everything the search returned for it (rc flag, UP_Close runs,
    UP_Far runs, in order).  sha256 over the digests of a batch is the "checksum of checksums" bench.py
    prints: shards searched on different GPUs and concatenated must reproduce the single-GPU value."""
    a = result_arrays(res)
    n = len(a["rc_flag"])
    M = np.uint64(0x9E3779B97F4A7C15)

    def mix(x):
        x = x.astype(np.uint64)
        with np.errstate(over="ignore"):
            x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))

    def per_read(off, runs, salt):
        off = off.astype(np.int64)
        out = np.zeros(n, dtype=np.uint64)
        if len(runs) == 0:
            return out
        r = runs
        with np.errstate(over="ignore"):
            w0 = r["abs_loc_first"].astype(np.uint64) | (r["len_first"].astype(np.uint64) << np.uint64(32)) | \
                (r["len_last"].astype(np.uint64) << np.uint64(48))
            w1 = r["mismatches"].astype(np.uint64) | (r["flags"].astype(np.uint64) << np.uint64(8)) | \
                (r["chr_id"].astype(np.int64).astype(np.uint64) << np.uint64(16))
            h = mix(w0 + np.uint64(salt)) ^ mix(w1 * M + np.uint64(salt + 1))
            k = np.arange(len(r), dtype=np.int64) - np.repeat(off[:-1], np.diff(off))     # index inside the read
            h = h * (np.uint64(2) * k.astype(np.uint64) + np.uint64(3))
            cs = np.concatenate([[np.uint64(0)], np.cumsum(h, dtype=np.uint64)])
            out = cs[off[1:]] - cs[off[:-1]]
        return out

    with np.errstate(over="ignore"):
        d = mix(a["rc_flag"].astype(np.uint64) + np.uint64(17))
        d = d * M + per_read(a["close_off"], a["close_runs"], 101)
        d = d * M + per_read(a["far_off"], a["far_runs"], 202)
    return d


def digest_hex(digests: np.ndarray) -> str:
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(digests, dtype=np.uint64).tobytes()).hexdigest()

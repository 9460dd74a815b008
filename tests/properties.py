"""Size-independent checks of a search over a large batch (shared by the full-size tests):
idempotence, shard additivity, structure of every list, and a sampled bit-exact comparison with the oracle."""
import numpy as np

from pindel_amd import binding, hostio, shard
from tests.parity import compare_result, run_oracle


def arrays(res):
    return shard.result_arrays(res)


def assert_same(a, b, what):
    for k in ("close_off", "far_off", "rc_flag"):
        assert np.array_equal(a[k], b[k]), f"{what}: {k} differs"
    for k in ("close_runs", "far_runs"):
        assert a[k].tobytes() == b[k].tobytes(), f"{what}: {k} differs"


def slice_windows(bd, bd_off, lo, hi):
    if bd is None:
        return None, None
    b0, b1 = int(bd_off[lo]), int(bd_off[hi])
    return bd[b0:b1], (bd_off[lo:hi + 1] - bd_off[lo]).astype(np.uint64)


def search_device(eng, batch, bd=None, bd_off=None, twice=False):
    db = eng.upload(batch)
    if bd is not None:
        eng.set_windows(db, bd, bd_off)
    eng.search_device(db)
    out = arrays(eng.download(db))
    again = None
    if twice:
        eng.search_device(db)
        again = arrays(eng.download(db))
    eng.free_device_batch(db)
    return out, again


def gather_reads(batch, idx):
    o = batch.seq_off.astype(np.int64)
    lens = (o[1:] - o[:-1])[idx]
    if len(idx) and (lens == lens[0]).all() and (np.diff(o) == lens[0]).all():
        L = int(lens[0])
        seq = np.asarray(batch.seq).reshape(batch.n, L)[idx].reshape(-1)
    else:
        seq = np.concatenate([batch.seq[o[i]:o[i + 1]] for i in idx]) if len(idx) else np.zeros(0, np.uint8)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    return hostio.ReadBatch(seq=seq, seq_off=off, anchor_strand=np.asarray(batch.anchor_strand)[idx],
                            anchor_pos=np.asarray(batch.anchor_pos)[idx], insert_size=np.asarray(batch.insert_size)[idx],
                            chr_id=np.asarray(batch.chr_id)[idx])


class Picked:
    """The sampled reads' slices of a full-size result, shaped like binding.Result."""

    def __init__(self, whole, idx):
        self.n = len(idx)
        self.rc_flag = whole["rc_flag"][idx]
        self.close_off, self.close_runs = self._gather(whole, "close", idx)
        self.far_off, self.far_runs = self._gather(whole, "far", idx)

    @staticmethod
    def _gather(whole, key, idx):
        off = whole[key + "_off"].astype(np.int64)
        cnt = (off[1:] - off[:-1])[idx]
        new_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
        sel = np.concatenate([np.arange(off[i], off[i + 1]) for i in idx]) if len(idx) else np.zeros(0, np.int64)
        return new_off, whole[key + "_runs"][sel.astype(np.int64)]

    def close_points(self, i):
        return binding.expand_runs(self.close_runs[int(self.close_off[i]):int(self.close_off[i + 1])])

    def far_points(self, i):
        return binding.expand_runs(self.far_runs[int(self.far_off[i]):int(self.far_off[i + 1])])


def check_structure(whole, chroms, batch):
    n = batch.n
    sizes = np.array([len(s) for _, s in chroms], dtype=np.int64)
    lens = batch.lengths()
    for key in ("close", "far"):
        off = whole[key + "_off"].astype(np.int64)
        runs = whole[key + "_runs"]
        assert off[0] == 0 and off[-1] == len(runs) and np.all(np.diff(off) >= 0)
        owner = np.repeat(np.arange(n), np.diff(off))
        lf, ll = runs["len_first"].astype(np.int64), runs["len_last"].astype(np.int64)
        assert np.all(lf <= ll) and np.all(ll <= lens[owner] - 1) and np.all(lf >= (8 if key == "close" else 10))
        cid = runs["chr_id"].astype(np.int64)
        assert np.all((cid >= 0) & (cid < len(chroms))) and np.all(runs["abs_loc_first"] < sizes[cid])
        # lengths increase along a read's list: a run starts after the previous run of the same read ended
        same_read = np.ones(len(runs), dtype=bool)
        same_read[off[:-1][np.diff(off) > 0]] = False          # first run of each read
        assert np.all(lf[1:][same_read[1:]] > ll[:-1][same_read[1:]])
    has_close = np.diff(whole["close_off"].astype(np.int64)) > 0
    has_far = np.diff(whole["far_off"].astype(np.int64)) > 0
    assert not np.any(has_far & ~has_close)
    assert not np.any((whole["rc_flag"] != 0) & ~has_close)
    return int(has_close.sum()), int(has_far.sum())


def check_workload(eng, chroms, batch, params_kw=None, bd=None, bd_off=None, n_sample=20_000,
                   shard_cuts=(0.3333333, 0.7000001), seed=7):
    """idempotence + ragged-shard additivity + structure + sampled oracle parity; returns (n_close, n_far)."""
    n = batch.n
    whole, again = search_device(eng, batch, bd, bd_off, twice=True)
    assert_same(whole, again, "second search of the same device batch")
    cuts = [0] + [int(round(n * c)) for c in shard_cuts] + [n]
    parts = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        w, wo = slice_windows(bd, bd_off, lo, hi)
        parts.append(search_device(eng, batch.slice(lo, hi), w, wo)[0])
    assert_same(whole, shard.concat_results(parts), "concatenated shards vs whole batch")
    n_close, n_far = check_structure(whole, chroms, batch)
    rng = np.random.default_rng(seed)
    idx = np.sort(rng.choice(n, size=min(n_sample, n), replace=False))
    sub = gather_reads(batch, idx)
    sbd = sbd_off = None
    if bd is not None:
        o = bd_off.astype(np.int64)
        cnt = (o[1:] - o[:-1])[idx]
        sbd_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
        sel = np.concatenate([np.arange(o[i], o[i + 1]) for i in idx]) if len(idx) else np.zeros(0, np.int64)
        sbd = bd[sel.astype(np.int64)]
    orc = run_oracle(params_kw or {}, chroms, sub, bd=sbd, bd_off=sbd_off)
    compare_result(Picked(whole, idx), orc, len(idx))
    return n_close, n_far

"""Randomised parity hunt: HIP path (C ABI) vs the CPU oracle, bit-exact, over random parameters, read
lengths and deliberately nasty references (segmental duplications, tandem repeats, homopolymers, N runs).

    python scripts/fuzz_parity.py [iterations] [first_seed]

Prints one line per iteration and stops at the first mismatch (exit code 1) after saving the failing
configuration under gpurun_out/fuzz_fail_<seed>.npz.  tests/test_gpu_fuzz.py runs a few fixed seeds of it.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pindel_amd import binding, synth
from tests.parity import compare_result, run_oracle

SPACER = 100000


def nasty_reference(rng, length):
    """Spacer-padded chromosome: random ACGT with copied segments (exact and mutated), tandem repeats with
    periods 1-6, and a few N runs."""
    seq = rng.integers(0, 4, length).astype(np.uint8)
    for _ in range(int(rng.integers(20, 80))):           # segmental duplications
        ln = int(rng.integers(30, 400))
        src = int(rng.integers(0, length - ln))
        seg = seq[src:src + ln].copy()
        for _ in range(int(rng.integers(1, 5))):
            dst = int(rng.integers(0, length - ln))
            cp = seg.copy()
            nmut = int(rng.integers(0, 4))
            if nmut:
                cp[rng.integers(0, ln, nmut)] = rng.integers(0, 4, nmut)
            if rng.random() < 0.3:                        # inverted copy
                cp = (3 - cp)[::-1]
            seq[dst:dst + ln] = cp
    for _ in range(int(rng.integers(10, 40))):           # tandem repeats / homopolymers
        period = int(rng.integers(1, 7))
        ln = int(rng.integers(20, 300))
        dst = int(rng.integers(0, length - ln))
        unit = rng.integers(0, 4, period).astype(np.uint8)
        seq[dst:dst + ln] = np.resize(unit, ln)
    asc = np.frombuffer(b"ACGT", dtype=np.uint8)[seq]
    for _ in range(int(rng.integers(0, 4))):
        ln = int(rng.integers(1, 2000))
        dst = int(rng.integers(0, length - ln))
        asc[dst:dst + ln] = ord("N")
    pad = np.full(SPACER, ord("N"), dtype=np.uint8)
    return np.concatenate([pad, asc, pad]).tobytes()


def random_params(rng, wide=False):
    kw = {}
    if rng.random() < 0.7:
        kw["max_range_index"] = int(rng.integers(1, 7 if wide else 5))
    if rng.random() < 0.5:
        kw["additional_mismatch"] = int(rng.integers(1, 4))
    if rng.random() < 0.5:
        kw["min_perfect_match"] = int(rng.integers(1, 8))
    if rng.random() < 0.4:
        kw["max_mismatch_rate"] = float(rng.choice([0.0, 0.01, 0.02, 0.05, 0.1]))
    if rng.random() < 0.4:
        kw["seq_error_rate"] = float(rng.choice([0.001, 0.01, 0.03, 0.05]))
    if rng.random() < 0.3:
        kw["sensitivity"] = float(rng.choice([0.8, 0.95, 0.99]))
    if rng.random() < 0.3:
        kw["min_close"] = int(rng.integers(6, 15))
    return kw


def one_iteration(seed, n_reads=1500, verbose=True, generic=False):
    """generic=True forces the generic kernel family (PG_GENERIC_KERNELS=1) for this iteration; otherwise the launch picks the
    default-parameter kernels whenever the parameters equal Pindel's defaults."""
    if generic:
        old = os.environ.get("PG_GENERIC_KERNELS")
        os.environ["PG_GENERIC_KERNELS"] = "1"
        binding.reload_env()
        try:
            return one_iteration(seed, n_reads, verbose)
        finally:
            if old is None:
                del os.environ["PG_GENERIC_KERNELS"]
            else:
                os.environ["PG_GENERIC_KERNELS"] = old
            binding.reload_env()
    rng = np.random.default_rng(seed)
    length = int(rng.integers(150_000, 400_000))
    ref = nasty_reference(rng, length)
    chroms = [("f", ref)]
    wide = seed >= 200000            # seeds from 200000 on draw from wider ranges (-x up to 6, more insert sizes);
    kw = random_params(rng, wide)    # lower seeds keep their original streams (61026 is a regression seed)
    if 300000 <= seed < 400000:      # seeds 300000-399999: Pindel's default search parameters (the kernels compiled with them as
        kw = {k: v for k, v in kw.items() if k in ("max_mismatch_rate", "seq_error_rate", "sensitivity")}   # constants), the rest random
    lens = sorted(set(int(x) for x in rng.choice([24, 36, 50, 64, 76, 100, 101, 128, 129, 150, 192, 200, 250, 300],
                                                 size=int(rng.integers(1, 4)))))
    isz = int(rng.choice([120, 200, 350, 480, 500, 700, 800, 1200] if wide else [200, 350, 500, 800]))
    batch = synth.make_reads(ref, n_reads, seed=seed + 1, read_lens=lens, insert_size=isz,
                             error_rate=float(rng.choice([0.0, 0.01, 0.03])),
                             n_rate=float(rng.choice([0.0, 0.001, 0.02])),
                             max_del=int(rng.choice([50, 2000, 20000])))
    # a third of the reads are re-anchored at random places of the nasty reference: their windows are
    # full of repeats and most of them have no close end at all
    k = n_reads // 3
    batch.anchor_pos[:k] = rng.integers(2 * isz + 10, length - 2 * isz - 10, k).astype(np.int32)
    # seeds from 600000 on: characters outside ACGTN (IUPAC codes, lower case, '*', '=') at the start, the end or inside of one
    # read in twelve -- the reads the reference shortens when it reverse-complements them (setUnmatchedSeq, pindel.cpp:142-169);
    # they go through pg_search_exact_kernel
    if seed >= 600000:
        junk = np.frombuffer(b"RYKMSWBDHVrykmacgtn*=.", dtype=np.uint8)
        off = batch.seq_off.astype(np.int64)
        for i in np.nonzero(rng.random(n_reads) < 1.0 / 12)[0]:
            a, b = int(off[i]), int(off[i + 1])
            how = int(rng.integers(0, 6))
            k1, k2 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
            if how in (0, 2, 5):
                batch.seq[a:a + k1] = junk[rng.integers(0, len(junk), k1)]
            if how in (1, 2, 5):
                batch.seq[b - k2:b] = junk[rng.integers(0, 10, k2)]          # (alphanumeric at the very end: setUnmatchedSeq at creation)
            if how in (3, 4, 5):
                m = int(rng.integers(1, 4))
                batch.seq[rng.integers(a, b, m)] = junk[rng.integers(0, len(junk), m)]
            if how == 4 and b - a > 8:                                        # reverse-complemented behind the junk: attempt 1 finds it
                body = batch.seq[a + 1:b].copy()
                comp = np.zeros(256, dtype=np.uint8)
                for x, y in zip(b"ACGTN", b"TGCAN"):
                    comp[x] = y
                batch.seq[a + 1:b] = comp[body][::-1]
                batch.seq[a] = junk[rng.integers(0, 10)]
            if not ((48 <= batch.seq[b - 1] <= 57) or (65 <= batch.seq[b - 1] <= 90) or (97 <= batch.seq[b - 1] <= 122)):
                batch.seq[b - 1] = ord("R")
    # a third of the iterations: a second chromosome and per-read BreakDancer window clusters (0-4 windows,
    # some on the other chromosome, some with start < 0, some overlapping), searched before the ranges
    bd = bd_off = None
    if rng.random() < 0.34:
        chroms.append(("g", nasty_reference(rng, int(rng.integers(60_000, 150_000)))))
        cnt = rng.integers(0, 5, n_reads)
        cnt[rng.random(n_reads) < 0.3] = 0
        bd_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
        owner = np.repeat(np.arange(n_reads), cnt)
        bd = np.zeros(len(owner), dtype=binding.WINDOW_DTYPE)
        other = rng.random(len(owner)) < 0.2
        bd["chr_id"] = np.where(other, 1, 0)
        size = np.where(other, len(chroms[1][1]), len(ref))
        centre = np.where(other, rng.integers(SPACER + 300, 1 << 30, len(owner)) % (size - 2 * SPACER - 600) + SPACER + 300,
                          batch.anchor_pos[owner].astype(np.int64) + SPACER + rng.integers(-30000, 30000, len(owner)))
        centre = np.clip(centre, SPACER + 300, size - SPACER - 300)
        half = rng.integers(20, 1500, len(owner))
        bd["start"] = np.maximum(centre - half, 1)
        bd["end"] = np.minimum(centre + half, size - 1)
        neg = rng.random(len(owner)) < 0.03
        bd["start"][neg] = -1
    try:
        eng = binding.Engine(**kw)
    except binding.PgError as e:          # e.g. a parameter set the library rejects (> 16 levels)
        if verbose:
            print(f"seed {seed}: parameters rejected ({e}); skipped")
        return True
    try:
        eng.load_reference(chroms)
        if bd is None:
            gpu = eng.search_batch(batch)
            orc = run_oracle(kw, chroms, batch)
        else:
            close = eng.close_end_batch(batch)
            gpu = eng.far_end_batch(batch, close, bd=bd, bd_off=bd_off)
            orc = run_oracle(kw, chroms, batch, bd=bd, bd_off=bd_off)
            db = eng.upload(batch)                    # and the fused device-resident launch with windows
            eng.set_windows(db, bd, bd_off)
            if os.environ.get("PG_PACK_IN_PLACE_MIN"):     # the step as one launch: the search kernel packs its claims (records overwritten first)
                eng.scribble_records(db)
                eng.pack_search_device(db)
            else:
                eng.search_device(db)
            compare_result(eng.download(db), orc, batch.n)
            eng.free_device_batch(db)
        compare_result(gpu, orc, batch.n)
    except binding.PgError as e:
        if verbose:
            print(f"seed {seed}: rejected by the library ({e}); skipped")
        return True
    except AssertionError as e:
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez(f"gpurun_out/fuzz_fail_{seed}.npz", seed=seed, kw=str(kw), lens=lens, isz=isz)
        print(f"seed {seed}: MISMATCH params={kw} lens={lens} isz={isz}: {e}")
        return False
    finally:
        eng.close()
    if verbose:
        nc = int((orc["close_cnt"] > 0).sum())
        nf = int((orc["far_cnt"] > 0).sum())
        print(f"seed {seed}: ok  params={kw} lens={lens} isz={isz} close {nc} far {nf}" +
              (f" bd windows {len(bd)}" if bd is not None else ""))
    return True


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    quiet = os.environ.get("FUZZ_QUIET") is not None          # one summary line per call (the committed fuzz log)
    both = os.environ.get("FUZZ_BOTH_FAMILIES") is not None   # default-parameter seeds also through the generic kernels
    for s in range(first, first + iters):
        if not one_iteration(s, verbose=not quiet):
            sys.exit(1)
        if both and not one_iteration(s, verbose=False, generic=True):
            sys.exit(1)
    print("seeds", first, "..", first + iters - 1, ":", iters, "iterations bit-exact" + (" (both kernel families)" if both else ""))

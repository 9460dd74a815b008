"""-m gpu: the HIP path (C ABI) against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from pindel_amd import synth
from tests.parity import compare_result, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_ref():
    return [("chrS", synth.make_reference(1_500_000, seed=11))]


def test_default_params_100bp(engine_factory, small_ref):
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 6000, seed=5)
    gpu = eng.search_batch(batch)
    orc = run_oracle({}, small_ref, batch)
    assert (orc["close_cnt"] > 0).sum() > 3000 and (orc["far_cnt"] > 0).sum() > 2000
    compare_result(gpu, orc, batch.n)


def test_mixed_lengths(engine_factory, small_ref):
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 4000, seed=6, read_lens=[36, 76, 100, 150, 250])
    gpu = eng.search_batch(batch)
    orc = run_oracle({}, small_ref, batch)
    compare_result(gpu, orc, batch.n)


@pytest.mark.parametrize("read_len", [40, 64, 150, 192])
def test_kernel_block_count_variants(engine_factory, small_ref, read_len):
    """One launch per 64-base block count the kernel is instantiated for (1 and 3 besides 2/4/8)."""
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 3000, seed=60 + read_len, read_len=read_len)
    gpu = eng.search_batch(batch)
    orc = run_oracle({}, small_ref, batch)
    assert (orc["close_cnt"] > 0).sum() > 1000
    compare_result(gpu, orc, batch.n)


def test_close_then_far_seams(engine_factory, small_ref):
    """pg_close_end_batch + pg_far_end_batch (the two reference seams) == fused search."""
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 3000, seed=7)
    close = eng.close_end_batch(batch)
    orc = run_oracle({}, small_ref, batch)
    compare_result(close, orc, batch.n, check_far=False)
    assert close.far_off[-1] == 0
    both = eng.far_end_batch(batch, close)
    compare_result(both, orc, batch.n)


ALT = dict(max_range_index=5, additional_mismatch=2, max_mismatch_rate=0.05, min_perfect_match=5,
           min_close=10, seq_error_rate=0.02, sensitivity=0.9)


def test_non_default_parameters(engine_factory, small_ref):
    """-x 5 -a 2 -u 0.05 -m 5 -H 10 -e 0.02 -E 0.9 (the survey's second parameter set)."""
    eng = engine_factory(**ALT)
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 1500, seed=8, read_lens=[50, 100, 150], error_rate=0.02)
    gpu = eng.search_batch(batch)
    orc = run_oracle(ALT, small_ref, batch)
    assert (orc["far_cnt"] > 0).sum() > 300
    compare_result(gpu, orc, batch.n)
    from oracle import pyoracle
    np.testing.assert_array_equal(eng.max_mismatch_table(),
                                  pyoracle.max_mismatch_table(ALT["seq_error_rate"], ALT["sensitivity"]))


def test_noisy_reads_and_ns(engine_factory, small_ref):
    """5 % N's, 3 % errors, first-base N's, IUPAC codes and lower-case bases in reads."""
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 3000, seed=9, error_rate=0.03, n_rate=0.05)
    off = batch.seq_off.astype(np.int64)
    batch.seq[off[:200]] = ord("N")
    batch.seq[off[200:400] + 17] = ord("R")
    batch.seq[off[400:600] + 40] = ord("a")
    batch.seq[off[600:700] + 99] = ord("N")
    gpu = eng.search_batch(batch)
    orc = run_oracle({}, small_ref, batch)
    compare_result(gpu, orc, batch.n)


def test_repeats_gaps_and_chromosome_ends(engine_factory):
    """Reads anchored inside the repeat family, the AC microsatellite, next to N gaps and at both
    chromosome ends (far-end windows get clipped at the spacers, pindel.cpp:1034-1043)."""
    ref = [("chrR", synth.make_reference(1_400_000, seed=21))]
    L = 1_400_000
    eng = engine_factory(max_range_index=4)
    eng.load_reference(ref)
    parts = []
    hot = [int(L * 0.07), int(L * 0.18), int(L * 0.61), int(L * 1 / 4.5), 300, L - 400]
    base = synth.make_reads(ref[0][1], 2400, seed=22)
    rng = np.random.default_rng(5)
    pos = np.array([hot[i % len(hot)] for i in range(base.n)]) + rng.integers(-600, 600, base.n)
    base.anchor_pos[:] = np.clip(pos, 0, L).astype(np.int32)
    # give these reads sequence from where they are anchored so that close ends exist
    refb = np.frombuffer(ref[0][1], dtype=np.uint8)
    off = base.seq_off.astype(np.int64)
    for i in range(0, base.n, 2):
        p = int(base.anchor_pos[i]) + 100000 + int(rng.integers(0, 300))
        p = min(max(p, 100000), 100000 + L - 200)
        frag = refb[p:p + 60].copy()
        far = refb[p + 5000:p + 5040] if p + 5040 < len(refb) else refb[p - 5000:p - 4960]
        s = np.concatenate([frag, far])
        if base.anchor_strand[i] == ord("+"):
            comp = {65: 84, 67: 71, 71: 67, 84: 65, 78: 78}
            s = np.array([comp[int(c)] for c in s[::-1]], dtype=np.uint8)
        base.seq[off[i]:off[i] + 100] = s
    gpu = eng.search_batch(base)
    orc = run_oracle(dict(max_range_index=4), ref, base)
    assert (orc["close_cnt"] > 0).sum() > 200
    compare_result(gpu, orc, base.n)


def test_multi_chromosome_and_breakdancer_hints(engine_factory):
    """Two chromosomes; per-read BreakDancer clusters (some on the other chromosome, some with
    start < 0, farend_searcher.cpp:69-71) searched before the ranges (pindel.cpp:1006-1018)."""
    from pindel_amd.binding import WINDOW_DTYPE
    chroms = [("a", synth.make_reference(700_000, seed=31)), ("b", synth.make_reference(500_000, seed=32))]
    eng = engine_factory()
    eng.load_reference(chroms)
    b0 = synth.make_reads(chroms[0][1], 1200, seed=33, chr_id=0, max_del=200000)
    b1 = synth.make_reads(chroms[1][1], 800, seed=34, chr_id=1, max_del=100000)
    from pindel_amd.hostio import ReadBatch
    batch = ReadBatch(
        seq=np.concatenate([b0.seq, b1.seq]),
        seq_off=np.concatenate([b0.seq_off, b1.seq_off[1:] + b0.seq_off[-1]]),
        anchor_strand=np.concatenate([b0.anchor_strand, b1.anchor_strand]),
        anchor_pos=np.concatenate([b0.anchor_pos, b1.anchor_pos]),
        insert_size=np.concatenate([b0.insert_size, b1.insert_size]),
        chr_id=np.concatenate([b0.chr_id, b1.chr_id]))
    close = eng.close_end_batch(batch)
    orc_close = run_oracle({}, chroms, batch, do_far=False)
    compare_result(close, orc_close, batch.n, check_far=False)
    # hints: 0-4 windows per read around plausible far ends
    rng = np.random.default_rng(7)
    wins, offs = [], [0]
    for i in range(batch.n):
        k = int(rng.integers(0, 5)) if orc_close["close_cnt"][i] else 0
        for _ in range(k):
            c = int(batch.chr_id[i]) if rng.random() < 0.8 else 1 - int(batch.chr_id[i])
            size = len(chroms[c][1])
            centre = int(orc_close["close_pts"][i][0]["abs_loc"]) + int(rng.integers(-150000, 150000))
            centre = min(max(centre, 100300), size - 100300)
            start, end = centre - 200, centre + 200
            if rng.random() < 0.05:
                start = -1
            wins.append((c, start, end))
        offs.append(len(wins))
    bd = np.array(wins, dtype=WINDOW_DTYPE)
    bd_off = np.array(offs, dtype=np.uint64)
    both = eng.far_end_batch(batch, close, bd=bd, bd_off=bd_off)
    orc = run_oracle({}, chroms, batch, bd=bd, bd_off=bd_off)
    assert (orc["far_cnt"] > 0).sum() > 500
    compare_result(both, orc, batch.n)
    # the same through the device-resident path: windows attached to the uploaded batch, one fused launch
    db = eng.upload(batch)
    eng.set_windows(db, bd, bd_off)
    eng.search_device(db)
    compare_result(eng.download(db), orc, batch.n)
    eng.set_windows(db)                      # detached again: plain range search
    eng.search_device(db)
    compare_result(eng.download(db), run_oracle({}, chroms, batch), batch.n)
    eng.free_device_batch(db)


def test_empty_and_tiny_inputs(engine_factory, small_ref):
    from pindel_amd import hostio
    eng = engine_factory()
    eng.load_reference(small_ref)
    empty = hostio.batch_from_lists([], [], [], [], [])
    res = eng.search_batch(empty)
    assert res.n == 0 and len(res.close_runs) == 0 and len(res.far_runs) == 0
    tiny = hostio.batch_from_lists([b"ACGTACG", b"A", b"ACGTACGTAC" * 3], [b"+", b"-", b"+"],
                                   [1000, 2000, 3000], [500, 500, 500], [0, 0, 0])
    gpu = eng.search_batch(tiny)
    orc = run_oracle({}, small_ref, tiny)
    compare_result(gpu, orc, tiny.n)


def test_wide_ranges_streaming_windows(engine_factory, small_ref):
    """-x 7: far-end windows up to 524 288 bases, i.e. many LDS chunks per read."""
    eng = engine_factory(max_range_index=7)
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 120, seed=12, max_del=400000, mix=(0.7, 0.0, 0.1, 0.1, 0.1))
    gpu = eng.search_batch(batch)
    orc = run_oracle(dict(max_range_index=7), small_ref, batch)
    compare_result(gpu, orc, batch.n)


@pytest.mark.parametrize("read_len", [150, 192])
def test_chunk_pairs_in_dynamic_lds_for_129_to_192_base_reads(engine_factory, small_ref, read_len):
    """129..192-base reads keep ONE window chunk in static LDS (6.3 KB: 24 workgroups per CU); the second chunk of a chunk
    pair is dynamic LDS that only -x >= 3 launches request.  -x 4 (chunk pairs through the dynamic tail) and -x 2 with window
    clusters wider than a chunk (no dynamic LDS: chunk by chunk) against the oracle."""
    from pindel_amd.binding import WINDOW_DTYPE
    batch = synth.make_reads(small_ref[0][1], 700, seed=70 + read_len, read_len=read_len, max_del=60000, mix=(0.6, 0.1, 0.1, 0.1, 0.1))
    eng = engine_factory(max_range_index=4)
    eng.load_reference(small_ref)
    orc = run_oracle(dict(max_range_index=4), small_ref, batch)
    assert (orc["far_cnt"] > 0).sum() > 300
    compare_result(eng.search_batch(batch), orc, batch.n)
    # default -x 2 with 9 000-base window clusters (four and a half chunks each)
    eng2 = engine_factory()
    eng2.load_reference(small_ref)
    close = eng2.close_end_batch(batch)
    rng = np.random.default_rng(read_len)
    size = len(small_ref[0][1])
    wins, offs = [], [0]
    for i in range(batch.n):
        if close.close_off[i + 1] > close.close_off[i] and rng.random() < 0.5:
            c = int(np.clip(int(batch.anchor_pos[i]) + 100000 + int(rng.integers(-40000, 40000)), 110000, size - 110000))
            wins.append((0, c - 4500, c + 4500))
        offs.append(len(wins))
    bd, bd_off = np.array(wins, dtype=WINDOW_DTYPE), np.array(offs, dtype=np.uint64)
    orc2 = run_oracle({}, small_ref, batch, bd=bd, bd_off=bd_off)
    compare_result(eng2.far_end_batch(batch, close, bd=bd, bd_off=bd_off), orc2, batch.n)


def test_long_reads_and_wide_close_windows(engine_factory, small_ref):
    """300-450 bp reads (8 blocks of 64 bases per read) and insert size 1200: the R=1 close-end
    window (3 x InsertSize = 3600 bases) spans two LDS fills."""
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 600, seed=13, read_lens=[300, 450], insert_size=1200)
    gpu = eng.search_batch(batch)
    orc = run_oracle({}, small_ref, batch)
    assert (orc["far_cnt"] > 0).sum() > 100
    compare_result(gpu, orc, batch.n)


def test_pool_overflow_is_retried_on_the_gpu(engine_factory, small_ref, pg_env):
    """A pool that is far too small makes the launch repeat with a regrown pool; results unchanged."""
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 5000, seed=14)
    orc = run_oracle({}, small_ref, batch)
    pg_env.set("PG_TEST_TINY_POOL", "1")
    gpu = eng.search_batch(batch)
    compare_result(gpu, orc, batch.n)


def test_delivery_overflow_falls_back_to_the_whole_batch_download(engine_factory, small_ref, pg_env):
    """The chunk-by-chunk delivery of pg_search_batch has room for three runs per read and list; a batch that needs more
    is downloaded the whole-batch way instead (forced here with room for half a run per read)."""
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 5000, seed=15)
    orc = run_oracle({}, small_ref, batch)
    pg_env.set("PG_TEST_TINY_DELIVERY", "1")
    compare_result(eng.search_batch(batch), orc, batch.n)
    compare_result(eng.close_end_batch(batch), orc, batch.n, check_far=False)


def test_one_block_delivery_equals_chunked_delivery(engine_factory, small_ref, pg_env):
    """A batch of one chunk (Pindel's own flush size) comes back in ONE device-to-host copy, the result's arrays being views
    into one pinned block; the same batch through the copy-per-array path and through seven small chunks gives the same
    result, and the views behave as results do (pg_far_end_batch extends a close result in place)."""
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 6500, seed=16)
    orc = run_oracle({}, small_ref, batch)
    compare_result(eng.search_batch(batch), orc, batch.n)
    close = eng.close_end_batch(batch)
    compare_result(close, orc, batch.n, check_far=False)
    compare_result(eng.far_end_batch(batch, close), orc, batch.n)
    pg_env.set("PG_NO_SINGLE_BLOCK", "1")
    compare_result(eng.search_batch(batch), orc, batch.n)
    pg_env.unset("PG_NO_SINGLE_BLOCK")
    pg_env.set("PG_HOST_CHUNK", "1000")
    compare_result(eng.search_batch(batch), orc, batch.n)
    close = eng.close_end_batch(batch)
    compare_result(eng.far_end_batch(batch, close), orc, batch.n)


def test_wide_cells_and_split_launches(engine_factory, small_ref, pg_env):
    """The 64-bit candidate ids and the two-launch form (close kernel, then far kernel) on a default
    workload give the same result as the default (32-bit ids, one fused launch)."""
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 3000, seed=15)
    orc = run_oracle({}, small_ref, batch)
    pg_env.set("PG_FORCE_WIDE_CELLS", "1")
    compare_result(eng.search_batch(batch), orc, batch.n)
    pg_env.unset("PG_FORCE_WIDE_CELLS")
    pg_env.set("PG_SPLIT_LAUNCH", "1")
    compare_result(eng.search_batch(batch), orc, batch.n)


@pytest.mark.parametrize("read_len", [100, 150, 250])
def test_generic_kernels_on_the_default_parameters(engine_factory, small_ref, pg_env, read_len):
    """Pindel's default parameter set runs kernels compiled with the five parameters as constants (pg_kernels.hip: PRM);
    PG_GENERIC_KERNELS=1 sends the same launch through the kernels every other parameter set uses.  Both equal the oracle,
    fused and as the two seams."""
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 3000, seed=150 + read_len, read_len=read_len)
    orc = run_oracle({}, small_ref, batch)
    for generic in (False, True):
        if generic:
            pg_env.set("PG_GENERIC_KERNELS", "1")
        compare_result(eng.search_batch(batch), orc, batch.n)
        close = eng.close_end_batch(batch)
        compare_result(close, orc, batch.n, check_far=False)
        compare_result(eng.far_end_batch(batch, close), orc, batch.n)


@pytest.mark.parametrize("change", [dict(max_range_index=1), dict(max_range_index=3), dict(additional_mismatch=2),
                                    dict(min_perfect_match=4), dict(min_close=9), dict(min_close=7)])
def test_one_parameter_off_the_defaults(engine_factory, small_ref, change):
    """Each of the parameters the default-parameter kernels hold as constants, changed alone: the launch must take the
    generic kernels (a constant left in would show here)."""
    eng = engine_factory(**change)
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 2500, seed=170 + sum(change.values()))
    compare_result(eng.search_batch(batch), run_oracle(change, small_ref, batch), batch.n)


def test_many_runs_per_search(engine_factory, small_ref):
    """Noisy reads (3 % errors) break the point lists into many runs: close-end lists where CleanUniquePoints
    has several runs to choose from, far-end lists of 3+ runs, runs in both 64-length rounds of a search."""
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 3000, seed=16, error_rate=0.03)
    gpu = eng.search_batch(batch)
    orc = run_oracle({}, small_ref, batch)
    multi = int(((gpu.close_off[1:] - gpu.close_off[:-1]) > 2).sum() + ((gpu.far_off[1:] - gpu.far_off[:-1]) > 2).sum())
    assert multi > 50, multi
    compare_result(gpu, orc, batch.n)
    long_batch = synth.make_reads(small_ref[0][1], 1500, seed=17, error_rate=0.04, read_len=250)
    compare_result(eng.search_batch(long_batch), run_oracle({}, small_ref, long_batch), long_batch.n)


def test_packed_reference_cache_roundtrip(engine_factory, small_ref, tmp_path):
    """pg_reference_save_packed / pg_reference_load_packed: the planes written by one context, loaded by
    another, give the same bases back and the same search results; wrong files are rejected."""
    from pindel_amd.binding import PgError
    chroms = [small_ref[0], ("second", synth.make_reference(300_000, seed=77))]
    a = engine_factory()
    a.load_reference(chroms)
    path = tmp_path / "ref.pgref"
    a.save_packed(path)
    b = engine_factory()
    b.load_packed(path)
    assert b.reference_info() == a.reference_info()
    for c, (_, seq) in enumerate(chroms):
        for start in (0, 99_990, len(seq) - 4000):
            assert b.reference_fetch(c, start, 3000) == bytes(seq[start:start + 3000])
    batch = synth.make_reads(chroms[0][1], 1500, seed=78)
    ra, rb = a.search_batch(batch), b.search_batch(batch)
    assert ra.close_runs.tobytes() == rb.close_runs.tobytes() and ra.far_runs.tobytes() == rb.far_runs.tobytes()
    assert np.array_equal(ra.close_off, rb.close_off) and np.array_equal(ra.far_off, rb.far_off)
    compare_result(rb, run_oracle({}, chroms, batch), batch.n)
    bad = tmp_path / "bad.pgref"
    bad.write_bytes(path.read_bytes()[:1000])
    with pytest.raises(PgError):
        engine_factory().load_packed(bad)
    with pytest.raises(PgError):
        engine_factory(spacer=50000).load_packed(path)


def test_many_chromosomes_150bp(engine_factory):
    """BASELINE configs[3]-shaped at small scale: 24 chromosomes, 150 bp reads spread over all of them
    (chr_id selects the reference planes, results carry the chromosome of every run)."""
    from pindel_amd.hostio import ReadBatch
    chroms = [(f"chr{c + 1}", synth.make_reference(250_000 + 10_000 * c, seed=500 + c, n_gaps=1, gap_len=5000))
              for c in range(24)]
    eng = engine_factory()
    eng.load_reference(chroms)
    parts = [synth.make_reads(chroms[c][1], 250, seed=600 + c, read_len=150, chr_id=c, max_del=3000) for c in range(24)]
    off = [np.zeros(1, dtype=np.uint64)]
    base = 0
    for p_ in parts:
        off.append(p_.seq_off[1:] + np.uint64(base))
        base += int(p_.seq_off[-1])
    batch = ReadBatch(seq=np.concatenate([p_.seq for p_ in parts]), seq_off=np.concatenate(off),
                      anchor_strand=np.concatenate([p_.anchor_strand for p_ in parts]),
                      anchor_pos=np.concatenate([p_.anchor_pos for p_ in parts]),
                      insert_size=np.concatenate([p_.insert_size for p_ in parts]),
                      chr_id=np.concatenate([p_.chr_id for p_ in parts]))
    # interleave the chromosomes so that neighbouring reads use different reference planes
    perm = np.random.default_rng(3).permutation(batch.n)
    lens = batch.lengths()
    o = batch.seq_off.astype(np.int64)
    seq = np.concatenate([batch.seq[o[i]:o[i + 1]] for i in perm])
    batch = ReadBatch(seq=seq, seq_off=np.concatenate([[0], np.cumsum(lens[perm])]).astype(np.uint64),
                      anchor_strand=batch.anchor_strand[perm], anchor_pos=batch.anchor_pos[perm],
                      insert_size=batch.insert_size[perm], chr_id=batch.chr_id[perm])
    gpu = eng.search_batch(batch)
    orc = run_oracle({}, chroms, batch)
    assert (orc["close_cnt"] > 0).sum() > 3000 and (orc["far_cnt"] > 0).sum() > 2000
    assert len(set(gpu.close_runs["chr_id"].tolist())) == 24
    compare_result(gpu, orc, batch.n)


def test_anchor_sweep_window_coincidences(engine_factory, small_ref):
    """Split reads with deletions of 150-1000 bases, each anchored at every offset of a +-700 base sweep for
    insert sizes 350 and 480: every alignment between the close-end windows (R = 0 and R = 1) and the innermost
    far-end chunk occurs -- including a R = 1 close-end window that starts exactly where the far-end chunk
    starts while the far end lies beyond what that window staged (fuzz seed 61026)."""
    from pindel_amd.hostio import ReadBatch
    eng = engine_factory()
    eng.load_reference(small_ref)
    ref = np.frombuffer(small_ref[0][1], dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8)
    for a_, b_ in zip(b"ACGTN", b"TGCAN"):
        comp[a_] = b_
    rng = np.random.default_rng(17)
    seqs, strand, pos, isz = [], [], [], []
    for i in range(10):
        while True:                                           # a place clear of the N gaps
            bp = int(rng.integers(400_000, 1_000_000))
            dsize = int(rng.integers(150, 1000))
            sp = int(rng.integers(25, 60))
            bases = np.concatenate([ref[bp - sp:bp], ref[bp + dsize:bp + dsize + 100 - sp]])
            if not np.any(ref[bp - 2000:bp + 3000] == ord("N")):
                break
        read = comp[bases[::-1]]                               # mate of a '+' anchor: reverse strand
        for ins in (350, 480):
            for d in range(-700, 701):
                seqs.append(read)
                strand.append(ord("+"))
                pos.append(bp - sp - 100000 + d)
                isz.append(ins)
    n = len(seqs)
    perm = rng.permutation(n)        # neighbours in the batch (= on the same CU) are unrelated reads
    batch = ReadBatch(seq=np.concatenate([seqs[i] for i in perm]), seq_off=(np.arange(n + 1, dtype=np.uint64) * 100),
                      anchor_strand=np.array(strand, dtype=np.uint8)[perm], anchor_pos=np.array(pos, dtype=np.int32)[perm],
                      insert_size=np.array(isz, dtype=np.int16)[perm], chr_id=np.zeros(n, dtype=np.int32))
    gpu = eng.search_batch(batch)
    orc = run_oracle({}, small_ref, batch)
    assert (orc["close_cnt"] > 0).sum() > n // 3 and (orc["far_cnt"] > 0).sum() > n // 4
    compare_result(gpu, orc, batch.n)


def test_iupac_and_lower_case_read_bases(engine_factory, small_ref):
    """Read characters other than ACGTN (IUPAC codes, lower case) never match: the reference compares characters,
    and Convert2RC4N maps everything but ACGTN to 0 (pindel.cpp:110, 966-970), so such a base is a mismatch in
    every orientation.  The kernel's "other" plane and the oracle agree on that."""
    eng = engine_factory()
    eng.load_reference(small_ref)
    batch = synth.make_reads(small_ref[0][1], 3000, seed=21)
    rng = np.random.default_rng(3)
    seq = batch.seq.copy()
    hit = rng.random(len(seq)) < 0.01
    seq[hit] = rng.choice(np.frombuffer(b"RYKMSWacgtn", dtype=np.uint8), int(hit.sum()))
    batch.seq = seq
    orc = run_oracle({}, small_ref, batch)
    assert (orc["close_cnt"] > 0).sum() > 1500
    compare_result(eng.search_batch(batch), orc, batch.n)


def test_rejections_are_loud(engine_factory, small_ref, tmp_path):
    """Everything outside what the device path handles is refused with a status code, never answered wrongly."""
    import ctypes as C
    import struct
    from pindel_amd import binding, hostio
    from pindel_amd.binding import PgError
    E_INVALID, E_NO_REFERENCE, E_READ_TOO_LONG, E_UNSUPPORTED = -1, -4, -5, -6

    def code(fn):
        with pytest.raises(PgError) as e:
            fn()
        return e.value.code

    # parameters outside the device path: -H > 64, negative -m
    assert code(lambda: binding.Engine(min_close=65)) == E_UNSUPPORTED
    assert code(lambda: binding.Engine(min_close=0)) == E_UNSUPPORTED
    assert code(lambda: binding.Engine(min_perfect_match=65)) == E_UNSUPPORTED
    # no reference yet
    batch = synth.make_reads(small_ref[0][1], 50, seed=22)
    assert code(lambda: engine_factory().search_batch(batch)) == E_NO_REFERENCE
    eng = engine_factory()
    eng.load_reference(small_ref)
    # a 500-base read (g_maxMismatch has 500 entries)
    long_read = hostio.batch_from_lists([b"A" * 500], [b"+"], [200000], [500], [0])
    assert code(lambda: eng.search_batch(long_read)) == E_READ_TOO_LONG
    ok_read = hostio.batch_from_lists([b"ACGT" * 25], [b"+"], [200000], [500], [0])
    eng.search_batch(ok_read).free()
    # unknown chromosome, anchor outside the padded chromosome
    assert code(lambda: eng.search_batch(hostio.batch_from_lists([b"ACGT" * 25], [b"+"], [200000], [500], [3]))) == E_INVALID
    assert code(lambda: eng.search_batch(hostio.batch_from_lists([b"ACGT" * 25], [b"+"], [5_000_000], [500], [0]))) == E_INVALID
    # more than 32 mismatch levels for the longest read of the batch (-e 0.08 at 450 bp; up to 32 are searched, see
    # test_more_than_16_mismatch_levels)
    noisy = engine_factory(seq_error_rate=0.08)
    noisy.load_reference(small_ref)
    assert code(lambda: noisy.search_batch(synth.make_reads(small_ref[0][1], 20, seed=23, read_len=450))) == E_UNSUPPORTED
    noisy.search_batch(synth.make_reads(small_ref[0][1], 20, seed=23, read_len=100)).free()
    # a device batch is bound to the reference loaded when it was uploaded (its records hold chromosome offsets and sizes)
    stale = engine_factory()
    stale.load_reference(small_ref)
    sdb = stale.upload(batch)
    stale.search_device(sdb)
    stale.load_reference(small_ref)
    assert code(lambda: stale.search_device(sdb)) == E_INVALID
    stale.free_device_batch(sdb)
    sdb = stale.upload(batch)                    # uploaded again: fine
    stale.search_device(sdb)
    stale.free_device_batch(sdb)
    # BreakDancer windows: an unknown chromosome, offsets that go backwards
    db = eng.upload(batch)
    off = np.zeros(batch.n + 1, dtype=np.uint64)
    off[1:] = 1
    big = np.zeros(1, dtype=binding.WINDOW_DTYPE)
    big["start"], big["end"], big["chr_id"] = 0, 1000, 7
    assert code(lambda: eng.set_windows(db, big, off)) == E_INVALID
    off[3] = 0
    big["chr_id"] = 0
    assert code(lambda: eng.set_windows(db, big, off)) == E_INVALID
    eng.free_device_batch(db)
    # a chromosome of 2^31 bases or more: refused before a single base is read (positions are signed 32-bit)
    L = binding.lib()
    names = (C.c_char_p * 1)(b"huge")
    buf = np.zeros(16, dtype=np.uint8)
    ptrs = (C.c_void_p * 1)(buf.ctypes.data)
    lens = (C.c_uint64 * 1)((1 << 31) + 5)
    assert L.pg_load_reference(eng._h, 1, names, ptrs, lens) == E_UNSUPPORTED
    assert b"2^31" in L.pg_last_error(eng._h)
    # ... and in a packed reference file
    path = tmp_path / "huge.pgref"
    with open(path, "wb") as fh:
        fh.write(b"PGREF01\0" + struct.pack("<II", 100000, 1) + struct.pack("<I", 4) + b"huge" +
                 struct.pack("<QQ", (1 << 31) + 5, 64))
    assert code(lambda: engine_factory().load_packed(path)) == E_INVALID
    # the engine still works after all of that
    eng.load_reference(small_ref)
    compare_result(eng.search_batch(batch), run_oracle({}, small_ref, batch), batch.n)


def _rc_bytes(a):
    lut = np.zeros(256, dtype=np.uint8)              # Convert2RC4N: everything but ACGTN -> 0
    for x, y in zip(b"ACGTN", b"TGCAN"):
        lut[x] = y
    return lut[a[::-1]]


def test_far_end_seam_on_the_filtered_union_of_flushes(engine_factory, small_ref):
    """Seam 2 where the reference calls it (src/pindel.cpp:1888): the close end runs in 50 000-read flushes
    (ReadBuffer, src/reader.cpp:55), every flush keeps only the reads with a close end, already reverse-complemented
    where GetCloseEnd did so (src/read_buffer.cpp:55-64), the kept reads of all flushes are concatenated, and
    pg_far_end_batch_from_close searches that vector from nothing but the reads and UP_Close.back()."""
    from pindel_amd import binding, hostio
    eng = engine_factory()
    eng.load_reference(small_ref)
    n, flush = 120_000, 50_000
    batch = synth.make_reads(small_ref[0][1], n, seed=77)
    orc = run_oracle({}, small_ref, batch)
    kept_idx, seqs, close_last, close_max = [], [], [], []
    off = batch.seq_off.astype(np.int64)
    for lo in range(0, n, flush):
        hi = min(n, lo + flush)
        res = eng.close_end_batch(batch.slice(lo, hi))
        has = np.diff(res.close_off.astype(np.int64)) > 0
        pts = binding.expand_runs(res.close_runs)
        last_run = res.close_runs[res.close_off[1:][has].astype(np.int64) - 1]
        # UP_Close.back(): the last point of the last run
        d = last_run["len_last"].astype(np.int64) - last_run["len_first"]
        back = (last_run["flags"] & 1) != 0
        cl = np.where(back, last_run["abs_loc_first"].astype(np.int64) - d, last_run["abs_loc_first"].astype(np.int64) + d)
        assert len(pts) == int((orc["close_cnt"][lo:hi]).sum())
        for k, i in enumerate(np.nonzero(has)[0]):
            g = lo + int(i)
            s = batch.seq[off[g]:off[g + 1]]
            seqs.append((_rc_bytes(s) if res.rc_flag[i] else s).tobytes())
            kept_idx.append(g)
        close_last.append(cl)
        close_max.append(last_run["len_last"].astype(np.int16))
    kept_idx = np.array(kept_idx)
    close_last = np.concatenate(close_last).astype(np.uint32)
    close_max = np.concatenate(close_max)
    np.testing.assert_array_equal(kept_idx, np.nonzero(orc["close_cnt"][:n] > 0)[0])
    assert 60_000 < len(kept_idx) < n                     # the vector seam 2 sees is NOT the vector seam 1 saw
    kept = hostio.batch_from_lists(seqs, [bytes([c]) for c in batch.anchor_strand[kept_idx]], batch.anchor_pos[kept_idx],
                                   batch.insert_size[kept_idx], batch.chr_id[kept_idx])
    far = eng.far_end_batch_from_close(kept, close_last, close_max)
    assert far.n == len(kept_idx) and far.close_off[-1] == 0 and not far.rc_flag.any()
    from tests.parity import oracle_points, points_per_read
    np.testing.assert_array_equal(points_per_read(far.far_off, far.far_runs), orc["far_cnt"][kept_idx])
    g_far = binding.expand_runs(far.far_runs)
    o_far = np.concatenate([oracle_points(orc, int(i), "far") for i in kept_idx])
    assert g_far.tobytes() == o_far.tobytes()
    assert (orc["far_cnt"][kept_idx] > 0).sum() > 40_000
    # reads without a close end (close_max <= 0) are passed through unsearched
    cm0 = close_max.copy()
    cm0[::2] = 0
    half = eng.far_end_batch_from_close(kept, close_last, cm0)
    cnt = points_per_read(half.far_off, half.far_runs)
    assert not cnt[::2].any()
    np.testing.assert_array_equal(cnt[1::2], orc["far_cnt"][kept_idx][1::2])


def test_more_than_127_windows_in_a_cluster(engine_factory):
    """BDData::getCorrespondingSearchWindowCluster has no cap on the windows of a cluster (src/bddata.cpp:949-979): 300
    windows per read, on two chromosomes, some with Start < 0 -- searched with 64-bit candidate ids -- == the oracle."""
    from pindel_amd import binding
    ref = [("chrA", synth.make_reference(700_000, seed=41)), ("chrB", synth.make_reference(500_000, seed=42))]
    eng = engine_factory()
    eng.load_reference(ref)
    batch = synth.make_reads(ref[0][1], 600, seed=43)
    rng = np.random.default_rng(44)
    per = 300
    bd = np.zeros(batch.n * per, dtype=binding.WINDOW_DTYPE)
    bd["chr_id"] = rng.integers(0, 2, len(bd))
    size = np.where(bd["chr_id"] == 0, 700_000, 500_000)
    st = rng.integers(100_000, size - 100_400)
    bd["start"] = st
    bd["end"] = st + rng.integers(50, 400, len(bd))
    # every read's own far-end neighbourhood is among its windows, so that the clusters do find far ends
    ap = batch.anchor_pos.astype(np.int64) + 100_000
    bd["chr_id"][::per] = 0
    bd["start"][::per] = np.clip(ap - 3000, 100_000, 590_000)
    bd["end"][::per] = bd["start"][::per] + 6000
    neg = rng.random(len(bd)) < 0.02
    bd["start"][neg] = -1                                   # Start < 0: the window is End - 1 (farend_searcher.cpp:69-71)
    bd_off = (np.arange(batch.n + 1) * per).astype(np.uint64)
    orc = run_oracle({}, ref, batch, bd=bd, bd_off=bd_off)
    close = eng.close_end_batch(batch)
    both = eng.far_end_batch(batch, close, bd, bd_off)
    compare_result(both, orc, batch.n)
    assert (orc["far_cnt"] > 0).sum() > 200
    # far-end points that lie in windows with an index beyond 127
    far = binding.expand_runs(both.far_runs)
    assert len(far) > 1000


@pytest.mark.parametrize("read_len", [50, 150])
def test_plane_layout_narrower_than_the_kernel_blocks(engine_factory, read_len):
    """The pack kernel lays the reads' bit planes out in pg_plane_blocks(longest read) 64-base blocks (1 for 50-base reads,
    3 for 150-base ones); the search kernel with 64-bit candidate ids (forced here by clusters of more than 127 windows)
    is compiled for 2 and 4 blocks: the missing blocks must read as zero.  == the oracle."""
    from pindel_amd import binding
    ref = [("chrA", synth.make_reference(400_000, seed=61))]
    eng = engine_factory()
    eng.load_reference(ref)
    batch = synth.make_reads(ref[0][1], 300, seed=62 + read_len, read_len=read_len)
    rng = np.random.default_rng(63)
    per = 140
    bd = np.zeros(batch.n * per, dtype=binding.WINDOW_DTYPE)
    st = rng.integers(100_000, 400_000 - 100_400, len(bd))
    bd["start"] = st
    bd["end"] = st + rng.integers(50, 400, len(bd))
    ap = batch.anchor_pos.astype(np.int64) + 100_000
    bd["start"][::per] = np.clip(ap - 3000, 100_000, 290_000)
    bd["end"][::per] = bd["start"][::per] + 6000
    bd_off = (np.arange(batch.n + 1) * per).astype(np.uint64)
    orc = run_oracle({}, ref, batch, bd=bd, bd_off=bd_off)
    close = eng.close_end_batch(batch)
    both = eng.far_end_batch(batch, close, bd, bd_off)
    compare_result(both, orc, batch.n)
    assert (orc["close_cnt"] > 0).sum() > 100 and (orc["far_cnt"] > 0).sum() > 50


def test_window_of_more_than_2_26_positions(engine_factory):
    """A search window wider than the 26-bit position field of a candidate id is searched as consecutive pieces (the
    reduction is additive over disjoint position sets): same points as the oracle on the whole window."""
    from pindel_amd import binding
    n_bases = (1 << 26) + 700_000
    ref = [("chrW", synth.make_reference(n_bases, seed=51))]
    eng = engine_factory()
    eng.load_reference(ref)
    batch = synth.make_reads(ref[0][1], 6, seed=52)
    bd = np.zeros(batch.n, dtype=binding.WINDOW_DTYPE)
    bd["start"] = 100_000 + 1000
    bd["end"] = 100_000 + 1000 + (1 << 26) + 5000
    bd_off = np.arange(batch.n + 1).astype(np.uint64)
    orc = run_oracle({}, ref, batch, bd=bd, bd_off=bd_off)
    close = eng.close_end_batch(batch)
    both = eng.far_end_batch(batch, close, bd, bd_off)
    compare_result(both, orc, batch.n)
    assert (orc["close_cnt"] > 0).sum() >= 3


def test_more_than_16_mismatch_levels(engine_factory, small_ref):
    """-e 0.05 on 300-base reads: g_maxMismatch[300] + ADDITIONAL_MISMATCH + 1 > 16 levels (the five-slice counter of the
    seed filter); the reference has no such limit below 500-base reads."""
    from oracle import pyoracle
    kw = dict(seq_error_rate=0.05)
    eng = engine_factory(**kw)
    eng.load_reference(small_ref)
    assert int(eng.max_mismatch_table()[300]) + 2 > 16
    batch = synth.make_reads(small_ref[0][1], 1200, seed=61, read_lens=[300, 250, 120], error_rate=0.04)
    gpu = eng.search_batch(batch)
    orc = run_oracle(kw, small_ref, batch)
    assert (orc["close_cnt"] > 0).sum() > 400 and (orc["far_cnt"] > 0).sum() > 200
    compare_result(gpu, orc, batch.n)


def test_host_pipeline_chunk_boundaries_equal_the_device_resident_path(engine_factory, small_ref):
    """pg_search_batch streams a batch through the GPU in 2^18-read chunks (copy / search on two kernel streams / deliver /
    download); whatever the batch size relative to the chunk -- one read, a chunk minus / plus one, several chunks with a
    ragged tail -- the result is the one of the device-resident entry points on the same reads, and so is the close-end
    only form (pg_close_end_batch).  One size is also checked against the oracle."""
    from pindel_amd import shard
    eng = engine_factory()
    eng.load_reference(small_ref)
    chunk = 1 << 18
    big = synth.make_reads(small_ref[0][1], 2 * chunk + 4097, seed=91, read_lens=[100, 76, 120])
    for n in (1, 255, chunk - 1, chunk, chunk + 1, big.n):
        b = big.slice(0, n)
        host = shard.result_arrays(eng.search_batch(b))
        db = eng.upload(b)
        eng.search_device(db)
        dev = shard.result_arrays(eng.download(db))
        eng.free_device_batch(db)
        for k in ("close_off", "far_off", "rc_flag"):
            assert np.array_equal(host[k], dev[k]), (n, k)
        for k in ("close_runs", "far_runs"):
            assert host[k].tobytes() == dev[k].tobytes(), (n, k)
        close = eng.close_end_batch(b)
        assert np.array_equal(close.close_off, dev["close_off"]) and close.close_runs.tobytes() == dev["close_runs"].tobytes()
        assert np.array_equal(close.rc_flag, dev["rc_flag"]) and int(close.far_off[-1]) == 0
    small = big.slice(chunk - 3000, chunk + 3000)              # (reads on both sides of a chunk boundary of the full batch)
    compare_result(eng.search_batch(small), run_oracle({}, small_ref, small), small.n)


def test_adapter_on_reference_shapes(engine_factory, tmp_path):
    """INTEGRATION.md's binding run against reference-shaped types (tests/ref_shapes.hpp = the public interface of
    src/pindel.h's SPLIT_READ / SortedUniquePoints / UniquePoint): seam 1 in 700-read flushes, the reads with a close end
    kept, seam 2 on their union (both overloads); every UniquePoint and the sequence left behind equal the oracle's."""
    import os
    import subprocess
    from pindel_amd import binding
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "adapter_ref_shapes"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(root, "include"),
                    "-I" + os.path.join(root, "pindel_amd", "csrc", "host"), "-I" + os.path.join(root, "tests"),
                    os.path.join(root, "tests", "adapter_ref_shapes.cpp"), "-L" + os.path.join(root, "pindel_amd"),
                    "-lpindel_pg", "-pthread", "-Wl,-rpath," + os.path.join(root, "pindel_amd"), "-o", str(exe)], check=True)
    spacer = 100_000
    chroms = [("chrA", synth.make_reference(400_000, seed=21)), ("chrB", synth.make_reference(300_000, seed=22))]
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as f:
        for name, padded in chroms:
            body = bytes(padded[spacer:-spacer]).decode()
            # Genome::loadChromosome repeats the final base of the last record (pindel.cpp:288-299): end it with a
            # throw-away record so that the real ones load as they are
            f.write(f">{name}\n{body}\n")
        f.write(">tail\nA\n")
    chroms_loaded = chroms + [("tail", b"N" * spacer + b"AA" + b"N" * spacer)]
    batch = synth.make_reads_genome(chroms, 2500, seed=23)
    # every 25th read arrives reverse-complemented behind a character outside ACGTN: GetCloseEnd's first
    # setUnmatchedSeq(ReverseComplement()) strips it, the read type's own setUnmatchedSeq (pg_adapter::apply_rc_flag) does so here
    from tests import shortening_cases as sc
    seqs = sc.seqs_of(batch)
    for i in range(0, batch.n, 25):
        seqs[i] = b"K" + sc.rc_ref(seqs[i])
    batch = sc.batch_of(seqs, batch.anchor_strand, batch.anchor_pos, batch.insert_size, batch.chr_id)
    tab = tmp_path / "reads.txt"
    with open(tab, "w") as f:
        for i in range(batch.n):
            s = bytes(batch.seq[batch.seq_off[i]:batch.seq_off[i + 1]]).decode()
            f.write(f"r{i} {chroms[batch.chr_id[i]][0]} {chr(batch.anchor_strand[i])} {batch.anchor_pos[i]} "
                    f"{batch.insert_size[i]} {s}\n")
    out = tmp_path / "out.txt"
    subprocess.run([str(exe), str(fa), str(tab), "700", str(out), "both"], check=True)
    orc = run_oracle({}, chroms_loaded, batch)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    lines = open(out).read().splitlines()
    n_first = 0
    for ln in lines:
        tok = ln.split()
        first = tok[0] == "first"
        if first:
            tok = tok[1:]
            n_first += 1
        i = int(tok[0][1:])
        k = 1
        if not first:
            a = int(batch.seq_off[i])
            want = bytes(orc["seq"][a:a + int(orc["len_out"][i])])      # UnmatchedSeq as GetCloseEnd left it
            if i % 25:
                s = bytes(batch.seq[batch.seq_off[i]:batch.seq_off[i + 1]])
                assert want == (s.translate(comp)[::-1] if orc["rc_flag"][i] else s)
            assert b"\0" not in want and tok[1].encode() == want, f"read {i}: UnmatchedSeq"
            k = 2
        while k < len(tok):
            which = {"C": "close", "F": "far"}[tok[k]]
            cnt = int(tok[k + 1])
            got = [tuple(t.split(",")) for t in tok[k + 2:k + 2 + cnt]]
            k += 2 + cnt
            o = orc[which + "_pts"][i][:orc[which + "_cnt"][i]]
            exp = [(str(int(p["length"])), str(int(p["abs_loc"])), p["direction"].decode(), p["strand"].decode(),
                    str(int(p["mismatches"])), str(int(p["chr_id"]))) for p in o]
            assert got == exp, f"read {i} {which}{' (one-flush overload)' if first else ''}"
    assert len(lines) - n_first == batch.n and n_first == 700
    assert (orc["far_cnt"] > 0).sum() > 800
    short = [i for i in range(0, batch.n, 25) if orc["rc_flag"][i] == 1]
    assert len(short) > 40 and all(orc["len_out"][i] == int(batch.seq_off[i + 1] - batch.seq_off[i]) - 1 for i in short)

// pg_kernels.hip -- split-read pattern growth for gfx950 (MI355X), one wavefront per read.
//
// What the reference does per read:
//   close end  GetCloseEnd / GetCloseEndInner      src/pindel.cpp:2531-2605, 2250-2326
//              CheckLeft_Close / CheckRight_Close  src/searcher.cpp:153-197, 247-286
//   far end    SearchFarEnd                        src/pindel.cpp:1001-1074
//              SearchFarEndAtPos                   src/farend_searcher.cpp:46-103
//              CheckBoth / ExtendMatch             src/pindel.cpp:2823-2902, 2673-2725
//   both       CategorizePositions, CheckMismatches src/searcher.cpp:48-63, 331-388
// The reference grows per-mismatch-level position lists one base at a time.  This
// kernel computes the same thing differently (DESIGN.md "kernel formulation"):
//
//   * The chromosome lives in HBM as three bit planes (2-bit code planar + N plane).
//     A window chunk is staged into LDS with coalesced dword loads.
//   * Each lane owns one window position p.  The mismatch pattern of the read placed
//     at p is obtained 64 bases at a time with funnel shifts + XOR on the planes
//     (no per-base loop); the read's planes are wave-uniform (built with ballots).
//   * A candidate's life is a short list of events "at length L it moves from
//     mismatch level k to k+1".  Lanes add these events (count and candidate id
//     packed in one 64-bit word) into an LDS difference histogram hist[level][L]
//     with ds_add_u64.  Candidates are order independent in the reference (a point
//     is only emitted when a level holds exactly one position), so per-level COUNTS
//     plus the identity of a singleton are all that is needed.
//   * Lanes then own lengths L: a wave prefix scan over L turns the differences
//     into cnt[level][L]; the reference's emission / abort rules are evaluated for
//     64 lengths at once, CheckMismatches is redone with the same plane arithmetic,
//     and consecutive points are emitted as run-length-encoded runs.
//   * Nested far-end ranges (128, 512, 2048 ... bases) only scan the new flanks:
//     the histogram is additive over disjoint position sets.
//
// No MFMA: this is bit/byte comparison work, not a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pg_device.h"

typedef unsigned long long u64;
typedef unsigned int u32;

#define WAVE 64

__device__ __forceinline__ u64 ballot64(bool p) { return __ballot(p); }
__device__ __forceinline__ u64 low_bits(int n)            // n in [0,64]
{
    return n >= 64 ? ~0ull : ((1ull << n) - 1ull);
}
__device__ __forceinline__ u64 bit_range(int lo, int hi)  // bits [lo,hi), clamped to [0,64]
{
    lo = lo < 0 ? 0 : (lo > 64 ? 64 : lo);
    hi = hi < 0 ? 0 : (hi > 64 ? 64 : hi);
    return hi > lo ? (low_bits(hi) & ~low_bits(lo)) : 0ull;
}
__device__ __forceinline__ u64 funnel64(u32 w0, u32 w1, u32 w2, u32 s)
{
    u32 a = __builtin_amdgcn_alignbit(w1, w0, s);
    u32 b = __builtin_amdgcn_alignbit(w2, w1, s);
    return (u64)a | ((u64)b << 32);
}

__device__ __forceinline__ u64 wave_incl_scan(u64 v, int lane)
{
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        u64 t = __shfl_up(v, d, WAVE);
        if (lane >= d) v += t;
    }
    return v;
}

// Wave-uniform description of the read for one orientation: bit planes in
// CONSUMPTION order (bit j of block b = base 64b+j the growth consumes).
template <int NB>
struct Planes {
    u64 lo[NB], hi[NB], nn[NB], oo[NB];   // code bit0, code bit1, is 'N', is other (never matches)
};

// Everything a search needs to know about the query.
template <int NB>
struct Query {
    Planes<NB> q;        // base planes (before complement)
    bool allowF, allowB; // candidate kinds searched
    bool cF, cB;         // complement flag per kind
    bool antisenseF, antisenseB;  // Strand reported for a point of that kind
    bool first_ok;       // first consumed base is one of ACGT
};

struct Search {
    int len, T, M, add_mm, bps, min_perfect, thr;
    int lh;
    u64 *hist;
    u64 *carry;
    uint4 *win;
};

// ---------------------------------------------------------------------------------
// mismatch / strict-inequality words of one 64-base block
template <int NB>
__device__ __forceinline__ void block_masks(const Planes<NB> &q, int b, bool comp,
                                            u64 rlo, u64 rhi, u64 rnn, u64 &mis, u64 &sne)
{
    u64 cm = comp ? ~0ull : 0ull;
    u64 x = rlo ^ q.lo[b] ^ cm;
    u64 y = rhi ^ q.hi[b] ^ cm;
    u64 d = x | y;
    // Matches(): read N matches any ACGT; reference N matches nothing (searcher.cpp:36-44)
    mis = (d & ~q.nn[b]) | rnn | q.oo[b];
    // exact character inequality (BP_On_Read != BP_On_Ref, searcher.cpp:349-364)
    sne = (d & ~(rnn | q.nn[b])) | (rnn ^ q.nn[b]) | q.oo[b];
}

// 64 reference bits of each plane starting at AbsLoc q, from the LDS window.
__device__ __forceinline__ void fetch_lds(const uint4 *win, long long wbase, long long q, bool rev,
                                          u64 &rlo, u64 &rhi, u64 &rnn)
{
    u32 rel = (u32)(q - wbase);
    u32 wi = rel >> 5, s = rel & 31u;
    uint4 w0 = win[wi], w1 = win[wi + 1], w2 = win[wi + 2];
    rlo = funnel64(w0.x, w1.x, w2.x, s);
    rhi = funnel64(w0.y, w1.y, w2.y, s);
    rnn = funnel64(w0.z, w1.z, w2.z, s);
    if (rev) { rlo = __brevll(rlo); rhi = __brevll(rhi); rnn = __brevll(rnn); }
}

// Same from HBM/L2 (used when re-checking a single candidate).
__device__ __forceinline__ void fetch_global(const PgDevRef &ref, int chr, long long q, bool rev,
                                             u64 &rlo, u64 &rhi, u64 &rnn)
{
    long long w = (long long)ref.chr_word_off[chr] + (q >> 5);   // arithmetic shift = floor
    u32 s = (u32)(q & 31);
    rlo = funnel64(ref.lo[w], ref.lo[w + 1], ref.lo[w + 2], s);
    rhi = funnel64(ref.hi[w], ref.hi[w + 1], ref.hi[w + 2], s);
    rnn = funnel64(ref.nn[w], ref.nn[w + 1], ref.nn[w + 2], s);
    if (rev) { rlo = __brevll(rlo); rhi = __brevll(rhi); rnn = __brevll(rnn); }
}

// ---------------------------------------------------------------------------------
// Scan window positions [s, e) of chromosome chr: every lane takes one position,
// decides whether it seeds a candidate, and adds the candidate's level events to
// the histogram.  Returns the number of seeds (NumberOfHits, farend_searcher.cpp:83).
template <int NB>
__device__ u32 scan_range(const PgDevRef &ref, const Search &S, const Query<NB> &Q, int chr,
                          long long s, long long e, long long origin, u32 region, int lane)
{
    u32 hits = 0;
    if (!Q.first_ok) return 0;
    const u32 q0lo = (u32)(Q.q.lo[0] & 1ull), q0hi = (u32)(Q.q.hi[0] & 1ull);
    for (long long cs = s; cs < e; cs += PG_CHUNK) {
        long long ce = cs + PG_CHUNK < e ? cs + PG_CHUNK : e;
        // ---- stage [cs - 64NB, ce + 64NB) into LDS, 32 bases per uint4 {lo,hi,nn,-}
        long long qlo = cs - 64 * NB, qhi = ce + 64 * NB;
        long long w0 = qlo >> 5;
        long long wbase = w0 << 5;
        int nwords = (int)(((qhi + 31) >> 5) - w0) + 2;
        __syncthreads();
        {
            long long g0 = (long long)ref.chr_word_off[chr] + w0;
            for (int i = lane; i < nwords; i += WAVE)
                S.win[i] = make_uint4(ref.lo[g0 + i], ref.hi[g0 + i], ref.nn[g0 + i], 0u);
        }
        __syncthreads();
        for (long long base = cs; base < ce; base += WAVE) {
            long long p = base + lane;
            bool act = p < ce;
            bool seedF = false, seedB = false;
            if (act) {
                u32 rel = (u32)(p - wbase);
                uint4 w = S.win[rel >> 5];
                u32 bit = rel & 31u;
                u32 bl = (w.x >> bit) & 1u, bh = (w.y >> bit) & 1u, bn = (w.z >> bit) & 1u;
                u32 xl = bl ^ q0lo, xh = bh ^ q0hi;
                seedF = Q.allowF && !bn && xl == (u32)Q.cF && xh == (u32)Q.cF;
                seedB = Q.allowB && !bn && xl == (u32)Q.cB && xh == (u32)Q.cB;
            }
            bool alive = seedF || seedB;
            hits += (u32)__popcll(ballot64(alive));
            if (!__any(alive)) continue;
            const bool isB = seedB;
            const bool comp = isB ? Q.cB : Q.cF;
            const u64 val = 1ull | (((u64)(p - origin) | ((u64)isB << PG_REL_BITS) |
                                     ((u64)region << (PG_REL_BITS + 1))) << PG_CNT_BITS);
            int level = 0;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if (64 * b >= S.len - 1) break;                 // uniform
                if (!__any(alive)) break;                        // uniform
                u64 bits = 0;
                if (alive) {
                    u64 rlo, rhi, rnn, mis, sne;
                    long long q = isB ? p - 64 * b - 63 : p + 64 * b;
                    fetch_lds(S.win, wbase, q, isB, rlo, rhi, rnn);
                    block_masks<NB>(Q.q, b, comp, rlo, rhi, rnn, mis, sne);
                    if (b == 0) {
                        // mismatches among the first bps bases decide the level at L = bps
                        level = __popcll(mis & low_bits(S.bps));
                        if (level >= S.T) alive = false;
                        else atomicAdd(&S.hist[level * S.lh + S.bps], val);
                    }
                    // events at consumed index j in [bps, len-2] -> L = j+1 in [bps+1, len-1]
                    bits = mis & bit_range(S.bps - 64 * b, S.len - 1 - 64 * b);
                }
                while (__any(alive && bits != 0)) {
                    if (alive && bits != 0) {
                        int j = __ffsll((long long)bits) - 1;
                        bits &= bits - 1;
                        int L = 64 * b + j + 1;
                        atomicAdd(&S.hist[level * S.lh + L], 0ull - val);
                        level++;
                        if (level >= S.T) alive = false;
                        else atomicAdd(&S.hist[level * S.lh + L], val);
                    }
                }
            }
        }
    }
    return hits;
}

// ---------------------------------------------------------------------------------
// Decoded candidate of a histogram id.
struct RegionInfo {
    // for range / close searches: one region on `chr` with `origin`
    // for BD searches: regions come from the per-read window list
    int chr;
    long long origin;
    const pg_window *bd;     // non-null: BD cluster search
    const PgDevRef *ref;
};

// Evaluate the reference's emission rules for every L (lanes own L) and write the
// resulting runs to `out`.  Returns the number of runs; max_len = LengthStr of the last
// emitted point (0 if none).
template <int NB>
__device__ int evaluate(const PgDevRef &ref, const PgDevParams &prm, const Search &S,
                        const Query<NB> &Q, const RegionInfo &R, pg_run *out, int &max_len,
                        int lane)
{
    int n_runs = 0;
    max_len = 0;
    if (lane < S.T) S.carry[lane] = 0;
    __syncthreads();
    bool aborted = false;
    for (int r0 = S.bps; r0 <= S.len - 1 && !aborted; r0 += WAVE) {
        const int L = r0 + lane;
        const bool valid = L <= S.len - 1;
        int lo = -1;
        u32 cnt_lo = 0, sumw = 0;
        u64 id_lo = 0;
        for (int i = 0; i < S.T; i++) {
            u64 d = valid ? S.hist[i * S.lh + L] : 0ull;
            u64 c = wave_incl_scan(d, lane) + S.carry[i];
            __syncthreads();
            if (lane == WAVE - 1) S.carry[i] = c;
            u32 cnt = (u32)(c & ((1ull << PG_CNT_BITS) - 1ull));
            if (lo < 0 && i <= S.M && cnt > 0) { lo = i; cnt_lo = cnt; id_lo = c >> PG_CNT_BITS; }
            if (lo >= 0 && i <= lo + S.add_mm) sumw += cnt;
        }
        __syncthreads();
        const int mmL = valid ? (int)prm.mm_tab[L] : 0;
        // "if (minimumNumberOfMismatches(...) > g_maxMismatch[L]) return;"
        const bool abortL = valid && ((lo < 0 ? S.M + 1 : lo) > mmL);
        const u64 ab = ballot64(abortL);
        const int first_abort = ab ? __ffsll((long long)ab) - 1 : WAVE;
        bool cand = valid && lane < first_abort && lo >= 0 && cnt_lo == 1 && L >= S.bps + lo &&
                    sumw == 1;
        // ---- CheckMismatches (searcher.cpp:331-388) on the singleton
        bool isB = false;
        long long p = 0;
        int chr = R.chr;
        if (cand) {
            u64 rel = id_lo & ((1ull << PG_REL_BITS) - 1ull);
            isB = (id_lo >> PG_REL_BITS) & 1ull;
            u32 region = (u32)(id_lo >> (PG_REL_BITS + 1));
            long long origin = R.origin;
            if (R.bd) {
                pg_window w = R.bd[region];
                chr = w.chr_id;
                int st = w.start < 0 ? w.end - 1 : w.start;
                origin = st;
            }
            p = origin + (long long)rel;
            const bool comp = isB ? Q.cB : Q.cF;
            int ham = 0;
            bool bad = false;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if (64 * b < S.len) {
                    u64 rlo, rhi, rnn, mis, sne;
                    long long q = isB ? p - 64 * b - 63 : p + 64 * b;
                    fetch_global(ref, chr, q, isB, rlo, rhi, rnn);
                    block_masks<NB>(Q.q, b, comp, rlo, rhi, rnn, mis, sne);
                    ham += __popcll(mis & low_bits(S.len - 64 * b));
                    bad |= (sne & bit_range(L - S.min_perfect - 64 * b, L - 64 * b)) != 0;
                }
            }
            bool len_ok = isB ? (L >= S.min_perfect) : (L > S.min_perfect);
            cand = len_ok && !bad && ham >= S.thr;
        }
        // ---- run-length encode consecutive points of the same candidate / level
        const u64 key = cand ? ((id_lo << 8) | (u64)(lo + 1)) : 0ull;
        u64 prev_key = __shfl_up(key, 1, WAVE);
        if (lane == 0) prev_key = 0ull;        // runs never span two 64-length rounds
        const bool start = cand && key != prev_key;
        const u64 starts = ballot64(start);
        const u64 brk = ballot64(start || !cand);
        if (start) {
            u64 higher = lane == 63 ? 0ull : (brk & ~low_bits(lane + 1));
            int end_lane = higher ? __ffsll((long long)higher) - 2 : WAVE - 1;
            int idx = n_runs + __popcll(starts & low_bits(lane));
            pg_run run;
            run.abs_loc_first = isB ? (u32)(p - L + 1) : (u32)(p + L - 1);
            run.len_first = (uint16_t)L;
            run.len_last = (uint16_t)(r0 + end_lane);
            run.mismatches = (uint8_t)lo;
            bool anti = isB ? Q.antisenseB : Q.antisenseF;
            run.flags = (uint8_t)((isB ? PG_RUN_BACKWARD : 0u) | (anti ? PG_RUN_ANTISENSE : 0u));
            run.chr_id = (int16_t)chr;
            out[idx] = run;
        }
        n_runs += __popcll(starts);
        const u64 em = ballot64(cand);
        if (em) max_len = r0 + 63 - __clzll((long long)em);
        if (ab) aborted = true;
    }
    __syncthreads();
    return n_runs;
}

__device__ __forceinline__ void zero_hist(const Search &S, int lane)
{
    __syncthreads();
    int n = S.T * S.lh;
    for (int i = lane; i < n; i += WAVE) S.hist[i] = 0ull;
    __syncthreads();
}

// ---------------------------------------------------------------------------------
template <int NB>
__device__ void load_planes(const uint8_t *seq, int len, int lane, Planes<NB> &fw, Planes<NB> &rv)
{
#pragma unroll
    for (int b = 0; b < NB; b++) {
        int idx = 64 * b + lane;
        bool in = idx < len;
        uint8_t cf = in ? seq[idx] : 0;
        uint8_t cr = in ? seq[len - 1 - idx] : 0;
        // code: A=0 C=1 G=2 T=3
        bool fA = cf == 'A', fC = cf == 'C', fG = cf == 'G', fT = cf == 'T', fN = cf == 'N';
        bool rA = cr == 'A', rC = cr == 'C', rG = cr == 'G', rT = cr == 'T', rN = cr == 'N';
        fw.lo[b] = ballot64(fC || fT);
        fw.hi[b] = ballot64(fG || fT);
        fw.nn[b] = ballot64(fN);
        fw.oo[b] = ballot64(in && !(fA || fC || fG || fT || fN));
        rv.lo[b] = ballot64(rC || rT);
        rv.hi[b] = ballot64(rG || rT);
        rv.nn[b] = ballot64(rN);
        rv.oo[b] = ballot64(in && !(rA || rC || rG || rT || rN));
    }
}

template <int NB>
__device__ __forceinline__ bool first_base_ok(const Planes<NB> &p)
{
    return ((p.nn[0] | p.oo[0]) & 1ull) == 0ull;
}

__device__ __forceinline__ void copy_runs(pg_run *dst, const pg_run *src, int n, int lane)
{
    for (int i = lane; i < n; i += WAVE) dst[i] = src[i];
}

// One read per 64-thread workgroup.
template <int NB>
__global__ __launch_bounds__(WAVE) void pg_search_kernel(PgDevRef ref, PgDevParams prm,
                                                         PgDevBatch B, int mode, uint32_t max_len,
                                                         uint32_t levels)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t rid = B.first_read + blockIdx.x;
    if (blockIdx.x >= B.n_reads) return;

    const PgLdsLayout lay = pg_lds_layout(max_len, levels, NB);
    Search S;
    S.hist = (u64 *)(smem + lay.hist_off);
    S.carry = (u64 *)(smem + lay.carry_off);
    S.win = (uint4 *)(smem + lay.win_off);
    pg_run *runs_tmp = (pg_run *)(smem + lay.runs_off);
    pg_run *runs_far = runs_tmp + lay.run_cap;
    pg_run *runs_close = runs_far + lay.run_cap;
    S.lh = (int)lay.lh;

    const u64 off = B.seq_off[rid];
    const int len = (int)(B.seq_off[rid + 1] - off);
    const uint8_t *seq = B.seq + off;
    const int chr = B.chr[rid];
    S.len = len;
    S.M = prm.mm_tab[len];
    S.add_mm = prm.add_mm;
    S.T = S.M + prm.add_mm + 1;
    S.min_perfect = prm.min_perfect;
    S.thr = prm.thr_tab[len];

    Planes<NB> A, Ar;   // original orientation: forward and reversed consumption order
    load_planes<NB>(seq, len, lane, A, Ar);

    float alg = (float)len;
    int flipped = 0;
    int n_close = 0, close_max = 0;
    u32 close_last = 0;

    // ============================ close end ======================================
    if (mode & PG_MODE_CLOSE) {
        const char strand = (char)B.strand[rid];
        const long long apos = (long long)B.pos[rid] + prm.spacer;
        const long long isz = B.isz[rid];
        S.bps = prm.min_close;
        long long wsize = 0;
        if (len - 1 >= S.bps && (strand == '+' || strand == '-')) {
            // attempts: (R0, seq) (R0, RC) (R1, RC) (R1, seq), pindel.cpp:2537-2575
            for (int att = 0; att < 4 && n_close == 0; att++) {
                const int Rg = att >> 1;
                flipped = (att == 1 || att == 2) ? 1 : 0;
                Query<NB> Q;
                long long s, e;
                if (strand == '+') {
                    // CurrentReadSeq = RC(cur), grown left to right (pindel.cpp:2271-2291)
                    Q.q = flipped ? A : Ar;
                    Q.cF = !flipped; Q.cB = false;
                    Q.allowF = true; Q.allowB = false;
                    s = apos - Rg * isz;
                    e = s + (2 * Rg + 1) * isz;
                } else {
                    // CurrentReadSeq = cur, grown right to left (pindel.cpp:2298-2319)
                    Q.q = flipped ? A : Ar;
                    Q.cB = flipped; Q.cF = false;
                    Q.allowF = false; Q.allowB = true;
                    e = apos + Rg * isz;
                    s = e - (2 * Rg + 1) * isz;
                }
                Q.antisenseF = true;    // CheckLeft_Close: FORWARD, ANTISENSE
                Q.antisenseB = false;   // CheckRight_Close: BACKWARD, SENSE
                Q.first_ok = first_base_ok<NB>(Q.q);
                wsize = e > s ? e - s : 0;
                zero_hist(S, lane);
                scan_range<NB>(ref, S, Q, chr, s, e, s, 0u, lane);
                __syncthreads();
                RegionInfo R = { chr, s, nullptr, &ref };
                n_close = evaluate<NB>(ref, prm, S, Q, R, runs_tmp, close_max, lane);
            }
            if (n_close == 0) flipped = 0;   // two flips: back to the original orientation
        }
        alg += 0.375f * (float)(wsize + 2 * len);
        // CleanUniquePoints (pindel.cpp:2904-2941): keep the points whose implied read
        // terminal equals the last point's, i.e. the runs of the last run's candidate.
        int kept = 0;
        if (n_close > 0) {
            pg_run last = runs_tmp[n_close - 1];
            u32 term_last = (last.flags & PG_RUN_BACKWARD) ? last.abs_loc_first + last.len_first
                                                           : last.abs_loc_first - last.len_first;
            for (int i0 = 0; i0 < n_close; i0 += WAVE) {
                int i = i0 + lane;
                bool keep = false;
                pg_run r;
                if (i < n_close) {
                    r = runs_tmp[i];
                    u32 term = (r.flags & PG_RUN_BACKWARD) ? r.abs_loc_first + r.len_first
                                                           : r.abs_loc_first - r.len_first;
                    keep = term == term_last && r.flags == last.flags && r.chr_id == last.chr_id;
                }
                u64 km = ballot64(keep);
                if (keep) runs_close[kept + __popcll(km & low_bits(lane))] = r;
                kept += __popcll(km);
            }
            __syncthreads();
            u32 span = (u32)(last.len_last - last.len_first);
            close_last = (last.flags & PG_RUN_BACKWARD) ? last.abs_loc_first - span
                                                        : last.abs_loc_first + span;
        }
        n_close = kept;
        if (lane == 0) {
            B.rc_flag[rid] = (uint8_t)flipped;
            B.close_last_abs[rid] = close_last;
            B.close_max_len[rid] = (uint16_t)close_max;
        }
        // publish UP_Close
        u32 base = 0;
        if (n_close > 0) {
            if (lane == 0) base = atomicAdd(B.pool_used, (u32)n_close);
            base = __shfl(base, 0, WAVE);
            if (base + (u32)n_close <= B.pool_cap)
                copy_runs(B.pool + base, runs_close, n_close, lane);
        }
        if (lane == 0) { B.close_run_off[rid] = base; B.close_run_cnt[rid] = (u32)n_close; }
        alg += 12.0f * (float)n_close;
    } else {
        flipped = B.rc_flag[rid];
        close_last = B.close_last_abs[rid];
        close_max = B.close_max_len[rid];
    }

    // ============================ far end ========================================
    if (mode & PG_MODE_FAR) {
        int n_far = 0, far_max = 0;
        S.bps = 10;                        // farend_searcher.cpp:90
        Query<NB> Q;
        // cur = flipped ? RC(orig) : orig.  Plus strand consumes cur left to right,
        // Minus strand consumes complement(cur) walking the reference right to left.
        Q.q = flipped ? Ar : A;
        Q.cF = flipped; Q.cB = !flipped;
        Q.allowF = Q.allowB = true;
        Q.antisenseF = false;              // FORWARD, SENSE
        Q.antisenseB = true;               // BACKWARD, ANTISENSE
        // "if (CurrentBase == 'N' || MaxLenCloseEnd() == 0) return;" -- any other
        // non-ACGT first base simply finds no seed.
        Q.first_ok = first_base_ok<NB>(Q.q);
        const bool searchable = close_max > 0 && Q.first_ok && len - 1 >= S.bps;
        float far_bases = 0.f;
        if (searchable) {
            const long long size = ref.chr_size[chr];
            bool done = false;
            // ---- BreakDancer cluster first (pindel.cpp:1006-1018)
            if (B.bd_off) {
                const u64 b0 = B.bd_off[rid], b1 = B.bd_off[rid + 1];
                const int nw = (int)(b1 - b0);
                if (nw > 0) {
                    const pg_window *bd = B.bd + b0;
                    zero_hist(S, lane);
                    u32 hits = 0;
                    for (int r = 0; r < nw; r++) {
                        pg_window w = bd[r];
                        long long st = w.start < 0 ? w.end - 1 : w.start;
                        long long en = w.end;
                        long long csz = ref.chr_size[w.chr_id];
                        long long s2 = st < 0 ? 0 : st, e2 = en > csz ? csz : en;
                        hits += scan_range<NB>(ref, S, Q, w.chr_id, s2, e2, st, (u32)r, lane);
                        far_bases += (float)(e2 > s2 ? e2 - s2 : 0) + 2.f * len;
                    }
                    __syncthreads();
                    if (hits > 0) {
                        RegionInfo R = { chr, 0, bd, &ref };
                        int mx;
                        int n = evaluate<NB>(ref, prm, S, Q, R, runs_tmp, mx, lane);
                        if (mx >= far_max) {           // NewUPFarIsBetter
                            copy_runs(runs_far, runs_tmp, n, lane);
                            n_far = n; far_max = mx;
                            __syncthreads();
                        }
                    }
                    done = far_max + close_max >= len; // goodFarEndFound
                }
            }
            // ---- ranges 64*4^(r-1) around the close end (pindel.cpp:1025-1070)
            if (!done) {
                const long long center = close_last;
                long long span = 64;
                long long maxspan = 64;
                for (int i = 0; i < prm.max_range_index; i++) maxspan *= 4;
                const long long origin = center - maxspan;
                long long ps = 0, pe = 0;          // previous (nested) window, empty if ps >= pe
                u32 hits = 0;
                long long reach = 0;
                zero_hist(S, lane);
                for (int r = 1; r <= prm.max_range_index + 1 && !done; r++, span *= 4) {
                    long long s, e;
                    if (center > span + prm.spacer) s = center - span; else s = prm.spacer;
                    if (center + span + prm.spacer < size) e = center + span; else e = size - prm.spacer;
                    if (s < e) {
                        if (ps < pe) {
                            // only the new flanks; the histogram is additive
                            long long le = e < ps ? e : ps;
                            if (s < le) hits += scan_range<NB>(ref, S, Q, chr, s, le, origin, 0u, lane);
                            long long rs = s > pe ? s : pe;
                            if (rs < e) hits += scan_range<NB>(ref, S, Q, chr, rs, e, origin, 0u, lane);
                            ps = s < ps ? s : ps;
                            pe = e > pe ? e : pe;
                        } else {
                            hits += scan_range<NB>(ref, S, Q, chr, s, e, origin, 0u, lane);
                            ps = s; pe = e;
                        }
                        reach = pe - ps;
                    }
                    __syncthreads();
                    if (hits > 0) {
                        RegionInfo R = { chr, origin, nullptr, &ref };
                        int mx;
                        int n = evaluate<NB>(ref, prm, S, Q, R, runs_tmp, mx, lane);
                        if (mx >= far_max) {
                            copy_runs(runs_far, runs_tmp, n, lane);
                            n_far = n; far_max = mx;
                            __syncthreads();
                        }
                    }
                    done = far_max + close_max >= len;
                }
                far_bases += (float)reach + 2.f * len;
            }
        }
        alg += 0.375f * far_bases;
        u32 base = 0;
        if (n_far > 0) {
            if (lane == 0) base = atomicAdd(B.pool_used, (u32)n_far);
            base = __shfl(base, 0, WAVE);
            if (base + (u32)n_far <= B.pool_cap) copy_runs(B.pool + base, runs_far, n_far, lane);
        }
        if (lane == 0) { B.far_run_off[rid] = base; B.far_run_cnt[rid] = (u32)n_far; }
        alg += 12.0f * (float)n_far;
    }
    if (B.alg_bytes && lane == 0) B.alg_bytes[rid] = (u32)(alg + 0.5f);
}

// ---------------------------------------------------------------------------------
extern "C" int pg_launch_search(const PgDevRef *ref, const PgDevParams *prm, const PgDevBatch *batch,
                                int mode, uint32_t max_len, uint32_t levels, void *stream)
{
    if (batch->n_reads == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(batch->n_reads), block(WAVE);
    if (max_len <= 128) {
        PgLdsLayout lay = pg_lds_layout(max_len, levels, 2);
        hipLaunchKernelGGL(pg_search_kernel<2>, grid, block, lay.total, st, *ref, *prm, *batch, mode,
                           max_len, levels);
    } else if (max_len <= 256) {
        PgLdsLayout lay = pg_lds_layout(max_len, levels, 4);
        hipLaunchKernelGGL(pg_search_kernel<4>, grid, block, lay.total, st, *ref, *prm, *batch, mode,
                           max_len, levels);
    } else {
        PgLdsLayout lay = pg_lds_layout(max_len, levels, 8);
        hipLaunchKernelGGL(pg_search_kernel<8>, grid, block, lay.total, st, *ref, *prm, *batch, mode,
                           max_len, levels);
    }
    return (int)hipGetLastError();
}

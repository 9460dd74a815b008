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
//   * PREFILTER: each lane owns one window position p.  One 32-base funnel extract +
//     XOR against the read's (wave-uniform, ballot-built) planes decides whether p
//     seeds a candidate and whether that candidate is still alive (fewer than
//     TOTAL_SNP_ERROR_CHECKED mismatches) at the first reportable length.  Survivors
//     (~10 % of positions) are compacted into an LDS queue with ballots.
//   * DENSE PASS: 64 queued candidates at a time, one per lane.  The mismatch pattern
//     of the read placed at p comes 64 bases per step (funnel shifts + XOR, no
//     per-base loop).  A candidate's life is the short list "at length L it leaves
//     mismatch level k".  Each such event is ONE LDS atomic add of -(1 | id<<CB) into
//     the cumulative difference histogram G[k][L] (G[k](L) = number of candidates
//     with level <= k at length L; count in the low CB bits, candidate id above).
//     Candidates are order independent in the reference (a point is only emitted
//     when a level holds exactly one position), so per-level COUNTS plus the identity
//     of a singleton are all that is needed.
//   * EVALUATE: a chunked prefix sum over L turns the differences into G[k](L); lanes
//     then own lengths L and apply the reference's abort / emission rules to 64
//     lengths at once, CheckMismatches is redone with the same plane arithmetic, and
//     consecutive points are emitted as run-length-encoded runs.
//   * Nested far-end ranges (128, 512, 2048 ... bases) only scan the new flanks:
//     the histogram is additive over disjoint position sets.
//
// The kernel is latency bound, so LDS per workgroup is kept small
// to run 8 waves per SIMD: histogram cells are 32-bit (16-bit count + 16-bit id) whenever
// every search window of the launch has at most 32 768 positions (Pindel defaults), and
// 64-bit otherwise (large -x, BreakDancer clusters).
//
// No MFMA: this is bit/byte comparison work, not a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pg_device.h"

typedef unsigned long long u64;
typedef unsigned int u32;

#define WAVE 64
// Optional per-phase cycle accounting (compile with -DPG_PHASE_TIMING; diagnostics only).
#ifdef PG_PHASE_TIMING
#define PT_DECL long long pt_t0 = clock64(); long long pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PT_MARK(k) { long long t_ = clock64(); pt_acc[k] += t_ - pt_t0; pt_t0 = t_; }
#else
#define PT_DECL
#define PT_MARK(k)
#endif
#ifndef PG_PF
#define PG_PF 1      // prefilter rounds (64 positions each) per loop iteration
#endif
#ifndef PG_WAVES_PER_EU
#define PG_WAVES_PER_EU 4   // register budget the kernel is compiled for (waves per SIMD)
#endif

__device__ __forceinline__ u64 ballot64(bool p) { return __ballot(p); }
// DPP lane shifts (no LDS round trip).  row_shr:n moves lane i-n -> i inside each 16-lane row and
// yields 0 where i-n leaves the row; wave_shr:1 moves lane i-1 -> i across the whole wave.
template <int N>
__device__ __forceinline__ u32 row_shr(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, true);
}
__device__ __forceinline__ u32 wave_shr1(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
template <int N>
__device__ __forceinline__ u64 row_shr(u64 v)
{
    return (u64)row_shr<N>((u32)v) | ((u64)row_shr<N>((u32)(v >> 32)) << 32);
}
__device__ __forceinline__ u64 wave_shr1(u64 v)
{
    return (u64)wave_shr1((u32)v) | ((u64)wave_shr1((u32)(v >> 32)) << 32);
}
// inclusive prefix sum inside aligned groups of G lanes (G = 4, 8 or 16); c = lane % G
template <typename C>
__device__ __forceinline__ C group_scan(C v, int c, int G)
{
    C t = row_shr<1>(v);
    if (c >= 1) v += t;
    t = row_shr<2>(v);
    if (c >= 2) v += t;
    if (G > 4) {
        t = row_shr<4>(v);
        if (c >= 4) v += t;
    }
    if (G > 8) {
        t = row_shr<8>(v);
        if (c >= 8) v += t;
    }
    return v;
}

// tells the compiler a value is wave-uniform (keeps it in SGPRs)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
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

// Histogram cell formats.
template <typename Cell> struct CellFmt;
template <> struct CellFmt<u32> {
    static constexpr int CB = PG_CNT_BITS_SMALL, RB = PG_REL_BITS_SMALL;
};
template <> struct CellFmt<u64> {
    static constexpr int CB = PG_CNT_BITS, RB = PG_REL_BITS;
};
template <typename Cell>
__device__ __forceinline__ Cell cell_pack(u64 rel, bool isB, u32 region)
{
    typedef CellFmt<Cell> F;
    u64 id = rel | ((u64)isB << F::RB) | ((u64)region << (F::RB + 1));
    return (Cell)(1ull | (id << F::CB));
}

// Wave-uniform description of the read for one orientation: bit planes in
// CONSUMPTION order (bit j of block b = base 64b+j the growth consumes).
template <int NB>
struct Planes {
    u64 lo[NB], hi[NB], nn[NB], oo[NB];   // code bit0, code bit1, is 'N', is other (never matches)
};

// Everything a search needs to know about the query.  The read's planes exist twice (original
// orientation: forward and reversed consumption order); a query selects one of them.
template <int NB>
struct Query {
    const Planes<NB> *fw, *rv;
    bool use_rv;         // base planes (before complement) = use_rv ? *rv : *fw
    bool allowF, allowB; // candidate kinds searched
    bool cF, cB;         // complement flag per kind
    bool antisenseF, antisenseB;  // Strand reported for a point of that kind
    bool first_ok;       // first consumed base is one of ACGT
};

template <typename Cell>
struct Search {
    int len, T, M, add_mm, bps, min_perfect, thr;
    int lh;
    Cell *hist;    // G difference histogram [T][lh]
    Cell *ginit;   // [PG_MAX_LEVELS] candidates entering at level k (at L = bps)
    Cell *carry;   // [PG_MAX_LEVELS] running prefix per level during evaluate
    Cell *pref;    // [T][64] absolute G of the current 64-length round (aliases win/queue)
    u32 *queue;    // [192] compacted survivors of the prefilter
    uint4 *win;    // staged window
    // what the LDS window currently holds: bases [win_lo, win_hi) of the chromosome whose AbsLoc 0 is
    // at word index win_wo; the first staged base is wbase (a multiple of 32)
    long long win_wo;
    int win_lo, win_hi, wbase;
    int nsurv;     // candidates added to the histogram since it was zeroed
};

// g_maxMismatch[L] from its breakpoints (the table is monotone): #{k : L >= mm_bp[k]}
__device__ __forceinline__ int max_mismatch_at(const PgDevParams &prm, int L)
{
    int m = 0;
#pragma unroll
    for (int k = 0; k < PG_MM_BREAKS; k++) m += (u32)L >= prm.mm_bp[k] ? 1 : 0;
    return m;
}

// ---------------------------------------------------------------------------------
// mismatch / strict-inequality words of one 64-base block
template <int NB>
__device__ __forceinline__ void block_masks(const Query<NB> &Q, int b, bool comp,
                                            u64 rlo, u64 rhi, u64 rnn, u64 &mis, u64 &sne)
{
    const u64 qlo = Q.use_rv ? Q.rv->lo[b] : Q.fw->lo[b];
    const u64 qhi = Q.use_rv ? Q.rv->hi[b] : Q.fw->hi[b];
    const u64 qnn = Q.use_rv ? Q.rv->nn[b] : Q.fw->nn[b];
    const u64 qoo = Q.use_rv ? Q.rv->oo[b] : Q.fw->oo[b];
    u64 cm = comp ? ~0ull : 0ull;
    u64 x = rlo ^ qlo ^ cm;
    u64 y = rhi ^ qhi ^ cm;
    u64 d = x | y;
    // Matches(): read N matches any ACGT; reference N matches nothing (searcher.cpp:36-44)
    mis = (d & ~qnn) | rnn | qoo;
    // exact character inequality (BP_On_Read != BP_On_Ref, searcher.cpp:349-364)
    sne = (d & ~(rnn | qnn)) | (rnn ^ qnn) | qoo;
}

// 64 reference bits of each plane starting at AbsLoc q, from the LDS window.
__device__ __forceinline__ void fetch_lds(const uint4 *win, int wbase, int q, bool rev,
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

// Same from HBM/L2 (used when re-checking a single candidate).  wo = word index of AbsLoc 0.
__device__ __forceinline__ void fetch_global(const PgDevRef &ref, long long wo, int q, bool rev,
                                             u64 &rlo, u64 &rhi, u64 &rnn)
{
    long long w = wo + (long long)(q >> 5);   // arithmetic shift = floor
    u32 s = (u32)(q & 31);
    rlo = funnel64(ref.lo[w], ref.lo[w + 1], ref.lo[w + 2], s);
    rhi = funnel64(ref.hi[w], ref.hi[w + 1], ref.hi[w + 2], s);
    rnn = funnel64(ref.nn[w], ref.nn[w + 1], ref.nn[w + 2], s);
    if (rev) { rlo = __brevll(rlo); rhi = __brevll(rhi); rnn = __brevll(rnn); }
}

// ---------------------------------------------------------------------------------
// Dense pass over n (<= 64) queued candidates, one per lane: full mismatch pattern, level at
// L = bps, then one histogram update per later mismatch until the candidate dies.
// MIXED = both candidate kinds in one search (far end); otherwise the kind is wave-uniform (close end)
// and the per-lane selects / bit reversals disappear.
template <int NB, typename Cell, bool MIXED>
__device__ __forceinline__ void dense_pass(const Search<Cell> &S, const Query<NB> &Q, int wbase,
                                           int origin, u32 region, int n, int lane)
{
    bool alive = lane < n;
    int p = 0;
    bool isB = MIXED ? false : Q.allowB;
    if (alive) {
        u32 e = S.queue[lane];
        if (MIXED) isB = e & 1u;
        p = wbase + (int)(e >> 1);
    }
    const bool comp = isB ? Q.cB : Q.cF;
    const Cell val = cell_pack<Cell>((u64)(u32)(p - origin), isB, region);
    const Cell neg = (Cell)0 - val;
    int cell = 0;            // level * lh
    const int cell_end = S.T * S.lh;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        if (64 * b >= S.len - 1) break;                 // uniform
        if (!__any(alive)) break;                        // uniform
        u32 blo = 0, bhi = 0;
        if (alive) {
            u64 rlo, rhi, rnn, mis, sne;
            int q = isB ? p - 64 * b - 63 : p + 64 * b;
            fetch_lds(S.win, wbase, q, isB, rlo, rhi, rnn);
            block_masks<NB>(Q, b, comp, rlo, rhi, rnn, mis, sne);
            if (b == 0) {
                // mismatches among the first bps bases decide the level at L = bps
                int level = __popcll(mis & low_bits(S.bps));
                if (level >= S.T) alive = false;
                else {
                    atomicAdd(&S.ginit[level], val);
                    cell = level * S.lh;
                }
            }
            // events at consumed index j in [bps, len-2] -> L = j+1 in [bps+1, len-1]
            u64 bits = mis & bit_range(S.bps - 64 * b, S.len - 1 - 64 * b);
            blo = (u32)bits;
            bhi = (u32)(bits >> 32);
        }
        // one histogram update per mismatch, low word first; selects instead of branches
#pragma unroll
        for (int half = 0; half < 2; half++) {
            u32 w = half ? bhi : blo;
            const int lbase = 64 * b + 32 * half + 1;
            while (__any(alive && w != 0u)) {
                const bool act = alive && w != 0u;
                const int j = __ffs((int)w) - 1;
                w &= w - 1u;
                if (act) atomicAdd(&S.hist[cell + lbase + j], neg);   // leaves its level at L = 64b+j+1
                cell += act ? S.lh : 0;
                alive = alive && cell < cell_end;
            }
        }
    }
}

// Stages bases [lo, hi) (hi - lo <= PG_CHUNK + 128 NB) of a chromosome into the LDS window.
template <int NB, typename Cell>
__device__ __forceinline__ void stage_window(const PgDevRef &ref, Search<Cell> &S, long long wo, int lo, int hi,
                                             int lane)
{
    const int w0 = lo >> 5;
    const int nwords = ((hi + 31) >> 5) - w0 + 2;
    __syncthreads();
    {
        const long long g0 = wo + (long long)w0;
        const u32 *glo = ref.lo + g0, *ghi = ref.hi + g0, *gnn = ref.nn + g0;
        for (int i = lane; i < nwords; i += WAVE) S.win[i] = make_uint4(glo[i], ghi[i], gnn[i], 0u);
    }
    __syncthreads();
    S.win_wo = wo;
    S.wbase = w0 << 5;
    S.win_lo = lo;
    S.win_hi = hi;
}

// Scan window positions [s, e) of a chromosome (wo = word index of its AbsLoc 0).
// Returns the number of seeds (NumberOfHits, farend_searcher.cpp:83).
// Scan of one window.  PREFILTER: lane = window position p; the first 32 consumed bases of the candidate
// at p (one funnel extract per plane) decide "seed" and "can still matter at the first reportable
// length"; survivors are compacted into the LDS queue and go through dense_pass 64 at a time.
template <int NB, typename Cell, bool MIXED>
__device__ __forceinline__ u32 scan_impl32(const PgDevRef &ref, const PgDevParams &prm, Search<Cell> &S,
                                           const Query<NB> &Q, long long wo, int s, int e, int origin, u32 region,
                                           int lane)
{
    u32 hits = 0;
    if (!Q.first_ok) return 0;
    const Planes<NB> &qp = Q.use_rv ? *Q.rv : *Q.fw;
    const u32 q0lo = (u32)qp.lo[0], q0hi = (u32)qp.hi[0], q0nn = (u32)qp.nn[0], q0oo = (u32)qp.oo[0];
    const u32 pre_mask = S.bps >= 32 ? 0xffffffffu : ((1u << S.bps) - 1u);
    // Which seeds matter (exact, DESIGN.md "relevance"): a candidate at level k at length L can only
    // influence the result if k <= g_maxMismatch[L] + ADD -- otherwise either a lower level exists (and
    // lo + ADD < k), or it is itself the lowest level and the search aborts at L with or without it.
    // So a seed is kept iff level(bps) <= g_maxMismatch[bps] + ADD, or it is still alive (< T mismatches)
    // after the 32 (or len-1) bases the prefilter sees and may become relevant further on.
    const int Wv = S.len - 1 < 32 ? S.len - 1 : 32;
    const u32 w_mask = Wv >= 32 ? 0xffffffffu : ((1u << Wv) - 1u);
    int cap0 = max_mismatch_at(prm, S.bps) + S.add_mm;
    for (int k = 0; k < PG_MM_BREAKS; k++)
        if ((int)prm.mm_bp[k] > S.bps && (int)prm.mm_bp[k] <= Wv) cap0 = S.T - 1;   // breakpoint inside the window
    if (cap0 > S.T - 1) cap0 = S.T - 1;
    for (int cs = s; cs < e; cs += (int)PG_CHUNK) {
        const int ce = cs + (int)PG_CHUNK < e ? cs + (int)PG_CHUNK : e;
        // the chunk plus 64 NB bases of overhang on both sides must be in LDS
        if (!(wo == S.win_wo && cs - 64 * NB >= S.win_lo && ce + 64 * NB <= S.win_hi))
            stage_window<NB, Cell>(ref, S, wo, cs - 64 * NB, ce + 64 * NB, lane);
        const int wbase = S.wbase;
        int qn = 0;     // queued survivors (uniform)
        for (int base = cs; base < ce; base += PG_PF * WAVE) {
            // ---- prefilter, two 64-position rounds per iteration, no branches: the first 32 consumed
            // bases of the candidate at p decide "seed" and "still alive at the first reportable length"
            bool surv[2] = {false, false}, seedv[2] = {false, false}, isBv[2] = {false, false};
            u32 relv[2] = {0u, 0u};
#pragma unroll
            for (int h = 0; h < PG_PF; h++) {
                const int pp = base + 64 * h + lane;
                const bool act = pp < ce;
                const int p = act ? pp : ce - 1;
                const u32 rel = (u32)(p - wbase);
                const u32 wi = rel >> 5, sh = rel & 31u;
                const uint4 wm = S.win[wi - 1], wc = S.win[wi], wp = S.win[wi + 1];
                const u32 bl = (wc.x >> sh) & 1u, bh = (wc.y >> sh) & 1u, bn = (wc.z >> sh) & 1u;
                const u32 xl = bl ^ (q0lo & 1u), xh = bh ^ (q0hi & 1u);
                const bool seedF = Q.allowF && !bn && xl == (u32)Q.cF && xh == (u32)Q.cF;
                const bool seedB = Q.allowB && !bn && xl == (u32)Q.cB && xh == (u32)Q.cB;
                const bool isB = MIXED ? seedB : Q.allowB;
                // forward: bits [p, p+32); backward: bits [p-31, p] reversed
                u32 rlo, rhi, rnn;
                if (MIXED) {
                    const u32 s2 = sh + (isB ? 1u : 0u);
                    rlo = (u32)((((u64)(isB ? wc.x : wp.x) << 32) | (isB ? wm.x : wc.x)) >> s2);
                    rhi = (u32)((((u64)(isB ? wc.y : wp.y) << 32) | (isB ? wm.y : wc.y)) >> s2);
                    rnn = (u32)((((u64)(isB ? wc.z : wp.z) << 32) | (isB ? wm.z : wc.z)) >> s2);
                    const u32 blo = __brev(rlo), bhi2 = __brev(rhi), bnn = __brev(rnn);
                    rlo = isB ? blo : rlo;
                    rhi = isB ? bhi2 : rhi;
                    rnn = isB ? bnn : rnn;
                } else if (Q.allowB) {          // wave-uniform kind: no selects
                    rlo = __brev((u32)((((u64)wc.x << 32) | wm.x) >> (sh + 1u)));
                    rhi = __brev((u32)((((u64)wc.y << 32) | wm.y) >> (sh + 1u)));
                    rnn = __brev((u32)((((u64)wc.z << 32) | wm.z) >> (sh + 1u)));
                } else {
                    rlo = __builtin_amdgcn_alignbit(wp.x, wc.x, sh);
                    rhi = __builtin_amdgcn_alignbit(wp.y, wc.y, sh);
                    rnn = __builtin_amdgcn_alignbit(wp.z, wc.z, sh);
                }
                const u32 cm = (isB ? Q.cB : Q.cF) ? 0xffffffffu : 0u;
                const u32 d = (rlo ^ q0lo ^ cm) | (rhi ^ q0hi ^ cm);
                const u32 mis = (d & ~q0nn) | rnn | q0oo;
                seedv[h] = act && (seedF || seedB);
                surv[h] = seedv[h] && (S.bps > 32 || __popc(mis & pre_mask) <= cap0 || __popc(mis & w_mask) < S.T);
                isBv[h] = isB;
                relv[h] = rel;
            }
            const u64 sd0 = ballot64(seedv[0]), sd1 = ballot64(seedv[1]);
            hits += (u32)(__popcll(sd0) + __popcll(sd1));
            const u64 sm0 = ballot64(surv[0]), sm1 = ballot64(surv[1]);
            if (sm0 | sm1) {
                const int n0 = __popcll(sm0);
                if (surv[0]) S.queue[qn + __popcll(sm0 & low_bits(lane))] = (relv[0] << 1) | (isBv[0] ? 1u : 0u);
                if (surv[1]) S.queue[qn + n0 + __popcll(sm1 & low_bits(lane))] = (relv[1] << 1) | (isBv[1] ? 1u : 0u);
                qn += n0 + __popcll(sm1);
                S.nsurv += n0 + __popcll(sm1);
                while (qn >= WAVE) {
                    __syncthreads();
#ifndef PG_ABL_NODENSE
                    dense_pass<NB, Cell, MIXED>(S, Q, wbase, origin, region, WAVE, lane);
#endif
                    __syncthreads();
                    // move the remainder (< 128 entries) to the front
                    const int rem = qn - WAVE;
                    u32 m0 = (lane < rem) ? S.queue[WAVE + lane] : 0u;
                    u32 m1 = (WAVE + lane < rem) ? S.queue[2 * WAVE + lane] : 0u;
                    __syncthreads();
                    if (lane < rem) S.queue[lane] = m0;
                    if (WAVE + lane < rem) S.queue[WAVE + lane] = m1;
                    qn = rem;
                }
            }
        }
        if (qn > 0) {
            __syncthreads();
#ifndef PG_ABL_NODENSE
            dense_pass<NB, Cell, MIXED>(S, Q, wbase, origin, region, qn, lane);
#endif
        }
    }
    __syncthreads();
    return hits;
}

template <int NB, typename Cell>
__device__ __forceinline__ u32 scan_range(const PgDevRef &ref, const PgDevParams &prm, Search<Cell> &S,
                                          const Query<NB> &Q, long long wo, int s, int e, int origin, u32 region,
                                          int lane)
{
    if (Q.allowF && Q.allowB) return scan_impl32<NB, Cell, true>(ref, prm, S, Q, wo, s, e, origin, region, lane);
    return scan_impl32<NB, Cell, false>(ref, prm, S, Q, wo, s, e, origin, region, lane);
}

// ---------------------------------------------------------------------------------
// Where the candidates of a search live.
struct RegionInfo {
    int chr;                 // range / close searches: one region on `chr` ...
    long long wo;            // ... whose AbsLoc 0 is at word index wo, positions relative to `origin`
    int origin;
    const pg_window *bd;     // non-null: BreakDancer cluster search, regions from the window list
};

// Evaluate the reference's emission rules for every L (lanes own L).  The runs with index in
// [skip, skip+cap) are written to out[0..cap); the total number of runs is returned, so a caller
// whose buffer is too small can come back for the next chunk.  max_len = LengthStr of the last
// emitted point (0 if none).  The histogram is not modified.
template <int NB, typename Cell>
__device__ __forceinline__ int evaluate(const PgDevRef &ref, const PgDevParams &prm, const Search<Cell> &S,
                                        const Query<NB> &Q, const RegionInfo &R, pg_run *out, int skip,
                                        int cap, int &max_len, int lane)
{
    typedef CellFmt<Cell> F;
    int n_runs = 0;
    max_len = 0;
    // G[k](bps) = sum over l <= k of the candidates that entered at level l
    __syncthreads();
    {
        Cell g = lane < S.T ? S.ginit[lane] : (Cell)0;
        g = group_scan<Cell>(g, lane & 15, 16);
        if (lane < S.T) S.carry[lane] = g;
    }
    __syncthreads();
    // lanes as (level, chunk) for the prefix over L: 8 chunks of 8 cells per level (T <= 8), else
    // 4 chunks of 16 cells; a level's lanes sit inside one 16-lane row so the scan is pure DPP
    const int CPL = S.T <= 8 ? 8 : 4;            // chunks per level
    const int pk = lane / CPL, pc = lane & (CPL - 1);
    const bool pact = pk < S.T;
    bool aborted = false;
    for (int r0 = S.bps; r0 <= S.len - 1 && !aborted; r0 += WAVE) {
        // ---- phase 1: absolute G[k](L) for L in [r0, r0+64) into pref[k][L-r0]
        {
            const int nvalid = S.len - r0 < WAVE ? S.len - r0 : WAVE;   // lengths r0 .. len-1
            const int CS = (nvalid + CPL - 1) / CPL;                     // cells per chunk
            const int j0 = pc * CS, j1 = (j0 + CS < WAVE) ? j0 + CS : WAVE;
            const Cell *row = S.hist + pk * S.lh + r0;
            Cell tot = 0;
            if (pact)
                for (int j = j0; j < j1; j++)
                    if (r0 + j <= S.len - 1) tot += row[j];
            const Cell inc = group_scan<Cell>(tot, pc, CPL);
            Cell acc = pact ? (Cell)(S.carry[pk] + inc - tot) : (Cell)0;
            if (pact)
                for (int j = j0; j < j1; j++) {
                    if (r0 + j <= S.len - 1) acc += row[j];
                    S.pref[pk * WAVE + j] = acc;
                }
            __syncthreads();
            if (pact && pc == CPL - 1) S.carry[pk] = acc;
            __syncthreads();
        }
        // ---- phase 2: lanes own L
        const int L = r0 + lane;
        const bool valid = L <= S.len - 1;
        int lo = -1;
        u32 cnt_lo = 0, sumw = 0;
        u64 id_lo = 0;
        for (int i = 0; i < S.T; i++) {
            Cell c = valid ? S.pref[i * WAVE + lane] : (Cell)0;
            u32 cnt = (u32)(c & (Cell)((1ull << F::CB) - 1ull));
            if (lo < 0 && i <= S.M && cnt > 0) { lo = i; cnt_lo = cnt; }
            if (lo >= 0 && i == lo + S.add_mm) { sumw = cnt; id_lo = (u64)(c >> F::CB); }
        }
        const int mmL = valid ? max_mismatch_at(prm, L) : 0;
        // "if (minimumNumberOfMismatches(...) > g_maxMismatch[L]) return;"
        const bool abortL = valid && ((lo < 0 ? S.M + 1 : lo) > mmL);
        const u64 ab = ballot64(abortL);
        const int first_abort = ab ? __ffsll((long long)ab) - 1 : WAVE;
        // cumulative counts: G[lo] == 1 and G[lo+ADD] == 1 <=> the level-lo position is the only one
        // within ADDITIONAL_MISMATCH extra mismatches; its id is the id field of G[lo+ADD]
        bool cand = valid && lane < first_abort && lo >= 0 && cnt_lo == 1 && L >= S.bps + lo &&
                    sumw == 1;
        // ---- CheckMismatches (searcher.cpp:331-388) on the singleton
        bool isB = false;
        int p = 0;
        int chr = R.chr;
        if (cand) {
            u32 rel = (u32)(id_lo & ((1ull << F::RB) - 1ull));
            isB = (id_lo >> F::RB) & 1ull;
            u32 region = (u32)(id_lo >> (F::RB + 1));
            int origin = R.origin;
            long long wo = R.wo;
            if (R.bd) {
                pg_window w = R.bd[region];
                chr = w.chr_id;
                wo = (long long)ref.chr_word_off[chr];
                int st = w.start < 0 ? w.end - 1 : w.start;
                origin = st;
            }
            p = origin + (int)rel;
            const bool comp = isB ? Q.cB : Q.cF;
            // the whole read placed at the candidate lies inside the staged window?
            const bool in_lds = wo == S.win_wo && (isB ? (p - 64 * NB + 1 >= S.win_lo && p < S.win_hi)
                                                       : (p >= S.win_lo && p + 64 * NB <= S.win_hi));
            int ham = 0;
            bool bad = false;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if (64 * b < S.len) {
                    u64 rlo, rhi, rnn, mis, sne;
                    int q = isB ? p - 64 * b - 63 : p + 64 * b;
                    if (in_lds) fetch_lds(S.win, S.wbase, q, isB, rlo, rhi, rnn);
                    else fetch_global(ref, wo, q, isB, rlo, rhi, rnn);
                    block_masks<NB>(Q, b, comp, rlo, rhi, rnn, mis, sne);
                    ham += __popcll(mis & low_bits(S.len - 64 * b));
                    bad |= (sne & bit_range(L - S.min_perfect - 64 * b, L - 64 * b)) != 0;
                }
            }
            bool len_ok = isB ? (L >= S.min_perfect) : (L > S.min_perfect);
            cand = len_ok && !bad && ham >= S.thr;
        }
        // ---- run-length encode consecutive points of the same candidate / level
        const u64 key = cand ? ((id_lo << 8) | (u64)(lo + 1)) : 0ull;
        const u64 prev_key = wave_shr1(key);   // lane 0 gets 0: runs never span two 64-length rounds
        const bool start = cand && key != prev_key;
        const u64 starts = ballot64(start);
        const u64 brk = ballot64(start || !cand);
        if (start) {
            u64 higher = lane == 63 ? 0ull : (brk & ~low_bits(lane + 1));
            int end_lane = higher ? __ffsll((long long)higher) - 2 : WAVE - 1;
            int idx = n_runs + __popcll(starts & low_bits(lane)) - skip;
            if (idx >= 0 && idx < cap) {
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
        }
        n_runs += __popcll(starts);
        const u64 em = ballot64(cand);
        if (em) max_len = r0 + 63 - __clzll((long long)em);
        if (ab) aborted = true;
    }
    __syncthreads();
    return n_runs;
}

template <typename Cell>
__device__ __forceinline__ void zero_hist(const Search<Cell> &S, int lane)
{
    __syncthreads();
    // hist is 16-byte aligned; its padded tail belongs to it (pg_lds_layout)
    const int n16 = (int)(((size_t)S.T * S.lh * sizeof(Cell) + 15) / 16);
    uint4 *h = (uint4 *)S.hist;
    for (int i = lane; i < n16; i += WAVE) h[i] = make_uint4(0u, 0u, 0u, 0u);
    if (lane < PG_MAX_LEVELS) S.ginit[lane] = (Cell)0;
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

// Bump-allocates n runs in this workgroup's pool shard (one atomic per wave); returns the pool
// offset.  fits = the allocation lies inside the shard (otherwise the host repeats the launch with a
// larger pool).
__device__ __forceinline__ u32 pool_alloc(const PgDevBatch &B, int n, int lane, bool &fits)
{
    const u32 shard = blockIdx.x & (PG_POOL_SHARDS - 1u);
    u32 off = 0;
    if (n > 0 && lane == 0) off = atomicAdd(B.pool_used + shard * 16u, (u32)n);
    off = __shfl(off, 0, WAVE);
    fits = (u64)off + (u64)n <= (u64)B.pool_shard_cap;
    return shard * B.pool_shard_cap + off;
}

template <int NB>
__device__ __forceinline__ bool first_base_ok(const Query<NB> &Q)
{
    const u64 x = Q.use_rv ? (Q.rv->nn[0] | Q.rv->oo[0]) : (Q.fw->nn[0] | Q.fw->oo[0]);
    return (x & 1ull) == 0ull;
}

// One read per 64-thread workgroup.  The read goes through a sequence of search STEPS that share
// one scan site and one evaluate site:
//   steps 0..3  close-end attempts (R0,seq) (R0,RC) (R1,RC) (R1,seq)   pindel.cpp:2537-2575
//   step  4     far end, BreakDancer cluster                            pindel.cpp:1006-1018
//   steps 5..   far end, ranges r = 1 .. MaxRangeIndex+1                pindel.cpp:1025-1070
template <int NB, typename Cell, int mode>
__global__ __launch_bounds__(WAVE, PG_WAVES_PER_EU) void pg_search_kernel(PgDevRef ref, PgDevParams prm,
                                                         PgDevBatch B, uint32_t max_len, uint32_t levels)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t rid = B.first_read + blockIdx.x;
    if (blockIdx.x >= B.n_reads) return;

    const PgLdsLayout lay = pg_lds_layout(max_len, levels, NB, (uint32_t)sizeof(Cell));
    Search<Cell> S;
    S.hist = (Cell *)(smem + lay.hist_off);
    S.ginit = (Cell *)(smem + lay.carry_off);
    S.carry = S.ginit + PG_MAX_LEVELS;
    S.pref = (Cell *)(smem + lay.pref_off);
    S.queue = (u32 *)(smem + lay.queue_off);
    S.win = (uint4 *)(smem + lay.win_off);
    pg_run *runs_tmp = (pg_run *)(smem + lay.runs_off);
    S.lh = (int)lay.lh;
    S.win_wo = -1;
    S.win_lo = S.win_hi = S.wbase = 0;
    S.nsurv = 0;

    const u64 off = B.seq_off[rid];
    const int len = (int)(B.seq_off[rid + 1] - off);
    const uint8_t *seq = B.seq + off;
    const int chr = B.chr[rid];
    const long long chr_wo = (long long)ref.chr_word_off[chr];
    const int chr_size = (int)ref.chr_size[chr];
    S.len = len;
    S.M = max_mismatch_at(prm, len);
    S.add_mm = prm.add_mm;
    S.T = S.M + prm.add_mm + 1;
    S.min_perfect = prm.min_perfect;
    S.thr = prm.thr_tab[len];

    PT_DECL
    Planes<NB> A, Ar;   // original orientation: forward and reversed consumption order
    load_planes<NB>(seq, len, lane, A, Ar);
    PT_MARK(0)
#if defined(PG_STOP_AFTER) && PG_STOP_AFTER == 0
    if (lane == 0) B.rc_flag[rid] = (uint8_t)(A.lo[0] ^ Ar.hi[0]);
    return;
#endif

    float alg = (mode & PG_MODE_CLOSE) ? (float)len : 0.f;   // the read itself is counted once
    int flipped = 0, close_max = 0, n_close = 0, n_far = 0, far_max = 0;
    u32 close_last = 0, close_base = 0, far_base = 0;

    const bool do_close = (mode & PG_MODE_CLOSE) != 0, do_far = (mode & PG_MODE_FAR) != 0;
    const char strand = do_close ? (char)B.strand[rid] : '+';
    const int apos = do_close ? (int)(B.pos[rid] + (int)prm.spacer) : 0;
    const int isz = do_close ? (int)B.isz[rid] : 0;
    if (!do_close) {
        flipped = B.rc_flag[rid];
        close_last = B.close_last_abs[rid];
        close_max = B.close_max_len[rid];
    }
    int nbd = 0;
    const pg_window *bd = nullptr;
    if (do_far && B.bd_off) {
        const u64 b0 = B.bd_off[rid];
        nbd = (int)(B.bd_off[rid + 1] - b0);
        bd = B.bd + b0;
    }
    int maxspan = 64;
    for (int i = 0; i < prm.max_range_index; i++) maxspan *= 4;
    const int last_step = 5 + prm.max_range_index;

    // far-range bookkeeping (nested windows)
    int ps = 0, pe = 0, span = 64, reach = 0;
    u32 hits = 0;
    float close_bases = 0.f, far_bases = 0.f;

    int step = do_close ? 0 : 4;
    if (do_close && !(len - 1 >= prm.min_close && (strand == '+' || strand == '-')))
        step = 4;                                    // no close end possible
    bool far_ready = false;                          // far-end query configured
    int nsurv_eval = -1;                             // S.nsurv when the histogram was last evaluated
    while (step <= last_step) {
        const bool is_close = step < 4;
        if (!is_close && !do_far) break;
        if (!is_close && !far_ready) {
            // entering the far end: "if (CurrentBase == 'N' || MaxLenCloseEnd() == 0) return;"
            if (!(close_max > 0 && len - 1 >= 10)) break;
            far_ready = true;
        }
        // ---------------- configure the step
        Query<NB> Q;
        Q.fw = &A;
        Q.rv = &Ar;
        int nwin = 0;                 // windows to scan this step (<= 2, or nbd)
        int s1 = 0, e1 = 0, s2 = 0, e2 = 0;
        int origin = 0;
        bool zero = false;
        if (is_close) {
            const int Rg = step >> 1;
            flipped = (step == 1 || step == 2) ? 1 : 0;
            S.bps = prm.min_close;
            // '+' anchor: CurrentReadSeq = RC(cur), grown left to right (pindel.cpp:2271-2291)
            // '-' anchor: CurrentReadSeq = cur, grown right to left     (pindel.cpp:2298-2319)
            Q.use_rv = !flipped;
            if (strand == '+') {
                Q.cF = !flipped; Q.cB = false; Q.allowF = true; Q.allowB = false;
                s1 = apos - Rg * isz;
                e1 = s1 + (2 * Rg + 1) * isz;
            } else {
                Q.cB = flipped; Q.cF = false; Q.allowF = false; Q.allowB = true;
                e1 = apos + Rg * isz;
                s1 = e1 - (2 * Rg + 1) * isz;
            }
            Q.antisenseF = true;      // CheckLeft_Close: FORWARD, ANTISENSE
            Q.antisenseB = false;     // CheckRight_Close: BACKWARD, SENSE
            origin = s1;
            nwin = 1;
            zero = true;
            hits = 0;
            close_bases = (float)(e1 > s1 ? e1 - s1 : 0);
        } else {
            S.bps = 10;               // farend_searcher.cpp:90
            // cur = flipped ? RC(orig) : orig.  Plus strand consumes cur left to right, Minus strand
            // consumes complement(cur) walking the reference right to left.
            Q.use_rv = flipped;
            Q.cF = flipped; Q.cB = !flipped;
            Q.allowF = Q.allowB = true;
            Q.antisenseF = false;     // FORWARD, SENSE
            Q.antisenseB = true;      // BACKWARD, ANTISENSE
            if (step == 4) {
                if (nbd == 0) { step++; continue; }
                nwin = nbd;
                zero = true;
                hits = 0;
            } else {
                const int center = (int)close_last;
                origin = center - maxspan;
                if (step == 5) {
                    zero = true; hits = 0; ps = pe = 0; span = 64;
                    // one LDS fill serves the nested ranges up to 2048 bases (all of them at -x <= 2)
                    const int half = maxspan < (int)PG_CHUNK / 2 ? maxspan : (int)PG_CHUNK / 2;
                    stage_window<NB, Cell>(ref, S, chr_wo, center - half - 64 * NB, center + half + 64 * NB, lane);
                }
                // window of this range, clipped to the non-spacer part (pindel.cpp:1034-1043)
                int s, e;
                if ((u32)center > (u32)span + prm.spacer) s = center - span; else s = (int)prm.spacer;
                if ((u32)center + (u32)span + prm.spacer < (u32)chr_size) e = center + span;
                else e = chr_size - (int)prm.spacer;
                if (s < e) {
                    if (ps < pe) {
                        // only the new flanks; the histogram is additive
                        const int le = e < ps ? e : ps;
                        const int rs = s > pe ? s : pe;
                        if (s < le) { s1 = s; e1 = le; nwin = 1; }
                        if (rs < e) {
                            if (nwin == 0) { s1 = rs; e1 = e; } else { s2 = rs; e2 = e; }
                            nwin++;
                        }
                        ps = s < ps ? s : ps;
                        pe = e > pe ? e : pe;
                    } else {
                        s1 = s; e1 = e; nwin = 1;
                        ps = s; pe = e;
                    }
                    reach = pe - ps;
                }
                span *= 4;
            }
        }
        Q.first_ok = first_base_ok<NB>(Q);
        if (!is_close && !Q.first_ok) break;         // far end: first base N (or not ACGT): nothing to find
        PT_MARK(5)
        if (zero) { zero_hist(S, lane); S.nsurv = 0; nsurv_eval = -1; }
        PT_MARK(4)
        // ---------------- scan
#ifdef PG_ABL_NOSCAN
        nwin = 0;
#endif
        for (int w = 0; w < nwin; w++) {
            long long wo = chr_wo;
            int s, e, org = origin;
            u32 region = 0;
            if (step == 4) {
                const pg_window bw = bd[w];
                const int st = bw.start < 0 ? bw.end - 1 : bw.start;
                const int csz = (int)ref.chr_size[bw.chr_id];
                wo = (long long)ref.chr_word_off[bw.chr_id];
                s = st < 0 ? 0 : st;
                e = bw.end > csz ? csz : bw.end;
                org = st;
                region = (u32)w;
                far_bases += (float)(e > s ? e - s : 0) + 2.f * len;
            } else {
                s = w == 0 ? s1 : s2;
                e = w == 0 ? e1 : e2;
            }
            hits += scan_range<NB, Cell>(ref, prm, S, Q, wo, s, e, org, region, lane);
        }
        PT_MARK(1)
#if defined(PG_STOP_AFTER) && PG_STOP_AFTER == 1
        if (lane == 0) B.rc_flag[rid] = (uint8_t)hits;
        return;
#endif
        // ---------------- evaluate (NumberOfHits == 0 leaves UP_Far untouched, farend_searcher.cpp:87)
        // One evaluate site inside a small pass machine.  Normally a search yields <= PG_RUN_TMP runs
        // and every pass works on the LDS copy; with more runs the later passes re-evaluate chunk by
        // chunk (skip = first run of the chunk).
        //   pass 0  evaluate; decide whether the result is kept (close: any point; far: NewUPFarIsBetter)
        //   pass 1  close end with > PG_RUN_TMP runs: fetch the last run
        //   pass 2  close end: count the runs CleanUniquePoints keeps (pindel.cpp:2904-2941: points whose
        //           implied read terminal equals the last point's = runs of the last run's candidate)
        //   pass 3  write the (kept) runs to the pool
        // An evaluation can only differ from the previous one of the same histogram if candidates were
        // added since; an empty histogram yields no point (and "replaces" an empty UP_Far by itself).
        const bool fresh = S.nsurv != nsurv_eval && S.nsurv > 0;
        if (!fresh && is_close) close_max = 0;
        nsurv_eval = S.nsurv;
        if ((is_close || hits > 0) && fresh) {
            const RegionInfo R = { chr, chr_wo, origin, step == 4 ? bd : nullptr };
            const u32 *tmp32 = (const u32 *)runs_tmp;
            int n = 0, mx = 0, kept = 0, wr = 0, pass = 0, skip = 0;
            u32 base = 0, last0 = 0, last1 = 0, last2 = 0;
            bool fits = true;
            for (;;) {
                if (pass == 0 || n > PG_RUN_TMP) {
                    int mm;
#ifdef PG_ABL_NOEVAL
                    int nn = 0; mm = 0;
#else
                    int nn = evaluate<NB, Cell>(ref, prm, S, Q, R, runs_tmp, skip, PG_RUN_TMP, mm, lane);
#endif
                    if (pass == 0) { n = uni(nn); mx = uni(mm); }
                    PT_MARK(2)
#if defined(PG_STOP_AFTER) && PG_STOP_AFTER == 2
                    if (lane == 0) B.rc_flag[rid] = (uint8_t)(n + mx);
                    return;
#endif
                }
                if (pass == 0) {
                    if (is_close) {
                        close_max = mx;
                        if (n == 0) break;
                        if (n > PG_RUN_TMP) { pass = 1; skip = ((n - 1) / PG_RUN_TMP) * PG_RUN_TMP; }
                        else pass = 2;
                    } else {
                        if (mx < far_max) break;              // the earlier UP_Far stays
                        far_max = mx;
                        n_far = n;
                        far_base = 0;
                        if (n == 0) break;
                        base = (u32)uni((int)pool_alloc(B, n, lane, fits));
                        far_base = base;
                        pass = 3;
                    }
                    continue;
                }
                if (pass == 1 || (pass == 2 && skip == 0 && n <= PG_RUN_TMP)) {
                    const int li = (n - 1) - (pass == 1 ? skip : 0);
                    last0 = tmp32[3 * li];
                    last1 = tmp32[3 * li + 1];
                    last2 = tmp32[3 * li + 2];
                    if (pass == 1) { pass = 2; skip = 0; continue; }
                }
                // one chunk of runs, one run per lane
                const int cn = n - skip < PG_RUN_TMP ? n - skip : PG_RUN_TMP;
                u32 r0 = 0, r1 = 0, r2 = 0;
                bool keep = lane < cn;
                if (keep) {
                    r0 = tmp32[3 * lane];
                    r1 = tmp32[3 * lane + 1];
                    r2 = tmp32[3 * lane + 2];
                    if (is_close) {
                        const bool back = (r2 >> 8) & PG_RUN_BACKWARD, lback = (last2 >> 8) & PG_RUN_BACKWARD;
                        const u32 term = back ? r0 + (r1 & 0xffffu) : r0 - (r1 & 0xffffu);
                        const u32 lterm = lback ? last0 + (last1 & 0xffffu) : last0 - (last1 & 0xffffu);
                        keep = term == lterm && (r2 >> 8) == (last2 >> 8);   // same flags and chromosome
                    }
                }
                const u64 km = ballot64(keep);
                if (pass == 2) {
                    kept += __popcll(km);
                    skip += PG_RUN_TMP;
                    if (skip >= n) {
                        base = (u32)uni((int)pool_alloc(B, kept, lane, fits));
                        pass = 3;
                        skip = 0;
                    }
                    continue;
                }
                // pass 3
                if (keep && fits) {
                    u32 *dst = (u32 *)(B.pool + base + wr + __popcll(km & low_bits(lane)));
                    dst[0] = r0; dst[1] = r1; dst[2] = r2;
                }
                wr += __popcll(km);
                skip += PG_RUN_TMP;
                if (skip >= n) break;
            }
            PT_MARK(3)
            if (is_close && n > 0) {
                n_close = kept;
                close_base = base;
                const u32 sp = (last1 >> 16) - (last1 & 0xffffu);
                close_last = ((last2 >> 8) & PG_RUN_BACKWARD) ? last0 - sp : last0 + sp;
            }
        }
        // ---------------- next step
        if (is_close) {
            if (n_close > 0 || step == 3) step = 4;          // found, or all four attempts failed
            else step++;
        } else {
            if (far_max + close_max >= len) break;           // goodFarEndFound (pindel.cpp:480-483)
            step++;
        }
    }
    if (do_close && n_close == 0) { flipped = 0; close_max = 0; }   // back to the original orientation

    if (do_close) {
        alg += 0.375f * (close_bases + 2.f * len) + 12.0f * (float)n_close;
        if (lane == 0) {
            B.rc_flag[rid] = (uint8_t)flipped;
            B.close_last_abs[rid] = close_last;
            B.close_max_len[rid] = (uint16_t)close_max;
            B.close_run_off[rid] = close_base;
            B.close_run_cnt[rid] = (u32)n_close;
        }
    }
    if (do_far) {
        if (far_ready) far_bases += (float)reach + 2.f * len;
        alg += 0.375f * far_bases + 12.0f * (float)n_far;
        if (lane == 0) { B.far_run_off[rid] = far_base; B.far_run_cnt[rid] = (u32)n_far; }
    }
#ifdef PG_PHASE_TIMING
    PT_MARK(6)
    if (B.alg_bytes && lane == 0 && rid < 65536) {      // overwrites the alg-bytes of the first reads
        u32 *d = B.alg_bytes + (size_t)B.n_reads - 65536 * 16 + (size_t)rid * 16 + (do_close ? 0 : 8);
        for (int k = 0; k < 8; k++) d[k] = (u32)pt_acc[k];
    }
    return;
#endif
    if (B.alg_bytes && lane == 0) {
        if (do_close) B.alg_bytes[rid] = (u32)(alg + 0.5f);
        else B.alg_bytes[rid] += (u32)(alg + 0.5f);        // the far-end launch adds to the close-end launch
    }
}

// ---------------------------------------------------------------------------------
template <int NB, typename Cell>
static void launch(const PgDevRef *ref, const PgDevParams *prm, const PgDevBatch *batch, int mode,
                   uint32_t max_len, uint32_t levels, hipStream_t st, unsigned lds_pad)
{
    PgLdsLayout lay = pg_lds_layout(max_len, levels, NB, (uint32_t)sizeof(Cell));
    dim3 grid(batch->n_reads), block(WAVE);
    // close end + far end in one launch; PG_SPLIT_LAUNCH=1 runs the two seams as separate launches
    const bool fused = getenv("PG_SPLIT_LAUNCH") == nullptr;
    if (mode == PG_MODE_BOTH && fused) {
        hipLaunchKernelGGL((pg_search_kernel<NB, Cell, PG_MODE_BOTH>), grid, block, lay.total + lds_pad, st,
                           *ref, *prm, *batch, max_len, levels);
        return;
    }
    if (mode & PG_MODE_CLOSE)
        hipLaunchKernelGGL((pg_search_kernel<NB, Cell, PG_MODE_CLOSE>), grid, block, lay.total + lds_pad, st,
                           *ref, *prm, *batch, max_len, levels);
    if (mode & PG_MODE_FAR)
        hipLaunchKernelGGL((pg_search_kernel<NB, Cell, PG_MODE_FAR>), grid, block, lay.total + lds_pad, st,
                           *ref, *prm, *batch, max_len, levels);
}

// Diagnostics: streams n dwords with the staging access pattern (one dword per lane, coalesced) so the
// FETCH_SIZE counter can be calibrated against a known byte count (MI355X_MICROARCH.md, HBM section).
__global__ void pg_calib_stream(const u32 *src, size_t n, u32 *sink)
{
    u32 acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc ^= src[i];
    if (acc == 0x12345678u) *sink = acc;
}

extern "C" int pg_debug_calib_stream(const void *src, size_t n_dwords, void *sink)
{
    hipLaunchKernelGGL(pg_calib_stream, dim3(4096), dim3(256), 0, 0, (const u32 *)src, n_dwords, (u32 *)sink);
    return (int)hipDeviceSynchronize();
}

// Debug/diagnostics: resident workgroups per CU the runtime predicts for the close/far kernels.
extern "C" int pg_debug_occupancy(uint32_t max_len, uint32_t levels, int small_cells, int *close_blocks,
                                  int *far_blocks, unsigned *lds_bytes)
{
    const int nb = max_len <= 128 ? 2 : (max_len <= 256 ? 4 : 8);
    PgLdsLayout lay = pg_lds_layout(max_len, levels, nb, small_cells ? 4u : 8u);
    *lds_bytes = lay.total;
    hipError_t e1, e2;
    if (small_cells && nb == 2) {
        e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(close_blocks, pg_search_kernel<2, u32, PG_MODE_CLOSE>, WAVE, lay.total);
        e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(far_blocks, pg_search_kernel<2, u32, PG_MODE_FAR>, WAVE, lay.total);
    } else {
        e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(close_blocks, pg_search_kernel<2, u64, PG_MODE_CLOSE>, WAVE, lay.total);
        e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(far_blocks, pg_search_kernel<2, u64, PG_MODE_FAR>, WAVE, lay.total);
    }
    return (int)e1 | (int)e2;
}

extern "C" int pg_launch_search(const PgDevRef *ref, const PgDevParams *prm, const PgDevBatch *batch,
                                int mode, uint32_t max_len, uint32_t levels, int small_cells, void *stream)
{
    if (batch->n_reads == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // experiment knob: extra dynamic LDS per workgroup (lowers occupancy), bytes
    static const unsigned lds_pad = getenv("PG_LDS_PAD") ? (unsigned)atoi(getenv("PG_LDS_PAD")) : 0u;
    const int nb = max_len <= 128 ? 2 : (max_len <= 256 ? 4 : 8);
    if (small_cells) {
        if (nb == 2) launch<2, u32>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
        else if (nb == 4) launch<4, u32>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
        else launch<8, u32>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
    } else {
        if (nb == 2) launch<2, u64>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
        else if (nb == 4) launch<4, u64>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
        else launch<8, u64>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
    }
    return (int)hipGetLastError();
}

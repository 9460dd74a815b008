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
//     A window chunk (2048 positions + overhang) is staged into LDS with coalesced dword
//     loads, as code planes and as one-hot planes (is-A/C/G/T/not-N).
//   * SEED FILTER, bit sliced: each lane owns the 32 window positions of one LDS word.
//     The match mask of consumed base j for all 32 positions is one v_alignbit of the
//     one-hot plane of that read symbol; mismatch counts live in a 4-bit ripple counter
//     of 32-bit slices.  It keeps exactly the seeds that can matter (DESIGN.md
//     "relevance"); survivors (~2 % of positions) are enumerated into an LDS queue.
//     At the far end one filter pass per strand serves all nested ranges.
//   * DENSE PASS: 64 queued candidates at a time, one per lane.  The mismatch pattern
//     of the read placed at p comes 64 bases per step (funnel shifts + XOR, no
//     per-base loop).  A candidate's life is the short list "at length L it leaves
//     mismatch level k".  Each such event is ONE LDS atomic add of -(1 | id<<CB) into
//     the cumulative difference histogram G[k][L] (G[k](L) = number of candidates
//     with level <= k at length L; count in the low CB bits, candidate id above).
//     Candidates are order independent in the reference (a point is only emitted
//     when a level holds exactly one position), so per-level COUNTS plus the identity
//     of a singleton are all that is needed.
//   * EVALUATE: lanes own lengths L; a DPP wave scan per level turns the differences
//     into G[k](L); the reference's abort / emission rules are applied to 64 lengths at
//     once, CheckMismatches is redone with the same plane arithmetic, and consecutive
//     points are emitted as run-length-encoded runs.
//   * Nested far-end ranges (128, 512, 2048 ... bases) only add the new flanks:
//     the histogram is additive over disjoint position sets.
//
// The kernel is VALU-issue bound: what counts is the number of
// vector instructions per read and the resident waves per SIMD (LDS per workgroup and
// VGPRs).  Histogram cells are 32-bit (16-bit count + 16-bit id) whenever every search
// window of the launch has at most 32 768 positions (Pindel defaults), and 64-bit
// otherwise (large -x, BreakDancer clusters).  The read's own bit planes live in LDS:
// as SGPRs they get spilled to VGPR lanes and every use costs a v_readlane on the VALU.
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
#ifndef PG_N_XCD
#define PG_N_XCD 8u        // MI355X: 8 accelerator complex dies, 32 CUs and one L2 each
#endif
#ifndef PG_WAVES_PER_EU
#define PG_WAVES_PER_EU 5   // register budget the kernel is compiled for (waves per SIMD): 96 VGPRs
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

// inclusive prefix sum over the 64 lanes of the wave, all DPP: Hillis-Steele inside the 16-lane rows
// (row_shr shifts zeros in), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3
__device__ __forceinline__ u32 wave_scan(u32 v)
{
    v += row_shr<1>(v);
    v += row_shr<2>(v);
    v += row_shr<4>(v);
    v += row_shr<8>(v);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

__device__ __forceinline__ u64 wave_scan(u64 v)
{
    v += row_shr<1>(v);
    v += row_shr<2>(v);
    v += row_shr<4>(v);
    v += row_shr<8>(v);
    v += (u64)(u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0x142, 0xa, 0xf, false) |
         ((u64)(u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0x142, 0xa, 0xf, false) << 32);
    v += (u64)(u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0x143, 0xc, 0xf, false) |
         ((u64)(u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0x143, 0xc, 0xf, false) << 32);
    return v;
}
__device__ __forceinline__ u32 read_lane(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ u64 read_lane(u64 v, int l)
{
    return (u64)read_lane((u32)v, l) | ((u64)read_lane((u32)(v >> 32), l) << 32);
}

// tells the compiler a value is wave-uniform (keeps it in SGPRs)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// An opaque copy of a per-lane value: expressions built on it cannot be hoisted out of the enclosing
// loop, so lane-derived addresses and masks are recomputed per phase instead of each pinning a VGPR for
// the whole kernel (occupancy matters more than the few VALU instructions).
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}
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

// The read's bit planes live in LDS (not in SGPRs: 32+ SGPRs of planes would be spilled to VGPR lanes and
// come back through v_readlane, which issues on the VALU this kernel is bound by; an LDS broadcast
// read does not).  Layout: u64 qp[2 orientations][4 planes][NB blocks]; orientation 0 = the read left
// to right, 1 = reversed; planes in CONSUMPTION order (bit j of block b = base 64b+j the growth
// consumes): code bit0, code bit1, is 'N', is other (never matches).
enum { QP_LO = 0, QP_HI = 1, QP_NN = 2, QP_OO = 3 };

// Everything a search needs to know about the query.
template <int NB>
struct Query {
    const u64 *qp;       // planes of the base orientation (before complement): qp[plane * NB + block]
    bool allowF, allowB; // candidate kinds searched
    bool cF, cB;         // complement flag per kind
    bool antisenseF, antisenseB;  // Strand reported for a point of that kind
    bool first_ok;       // first consumed base is one of ACGT
};
template <int NB> __device__ __forceinline__ u64 q_lo(const Query<NB> &Q, int b) { return Q.qp[QP_LO * NB + b]; }
template <int NB> __device__ __forceinline__ u64 q_hi(const Query<NB> &Q, int b) { return Q.qp[QP_HI * NB + b]; }
template <int NB> __device__ __forceinline__ u64 q_nn(const Query<NB> &Q, int b) { return Q.qp[QP_NN * NB + b]; }
template <int NB> __device__ __forceinline__ u64 q_oo(const Query<NB> &Q, int b) { return Q.qp[QP_OO * NB + b]; }

template <typename Cell>
struct Search {
    int len, T, M, add_mm, bps, min_perfect, thr;
    int lh;
    Cell *hist;    // G difference histogram [T][lh]
    Cell *ginit;   // [PG_MAX_LEVELS] candidates entering at level k (at L = bps)
    Cell *carry;   // [PG_MAX_LEVELS] running prefix per level during evaluate
    u32 *queue;    // [192] compacted survivors of the prefilter
    uint4 *win;    // staged window: code planes (lo, hi, N)
    u32 *eq;       // staged window: one-hot planes [5][eq_stride]
    int eq_stride;
    // what the LDS window currently holds: bases [win_lo, win_hi) of the chromosome whose AbsLoc 0 is
    // at word index win_wo; the first staged base is wbase = win_lo (any alignment)
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
    const u64 qlo = q_lo<NB>(Q, b), qhi = q_hi<NB>(Q, b), qnn = q_nn<NB>(Q, b), qoo = q_oo<NB>(Q, b);
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
                                           int origin, u32 region, int n, int lane, bool undo = false)
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
#ifdef PG_DUP
    const Cell val0 = cell_pack<Cell>((u64)(u32)(p - origin), isB, region);
    const Cell val = undo ? (Cell)0 - val0 : val0;      // diagnostics: the second pass takes the first one back
#else
    const Cell val = cell_pack<Cell>((u64)(u32)(p - origin), isB, region);
#endif
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

// Stages bases [lo, hi) (hi - lo <= PG_CHUNK + 128 NB) of a chromosome into LDS: word i of the window
// holds bases [lo + 32 i, lo + 32 i + 32) whatever the alignment of lo (funnel shift of two HBM words),
// once as the code planes the dense pass / CheckMismatches use and once as one-hot planes (is-A, is-C,
// is-G, is-T, is-not-N) for the bit-sliced seed filter.
template <int NB, typename Cell>
__device__ __forceinline__ void stage_window(const PgDevRef &ref, Search<Cell> &S, long long wo, int lo, int hi,
                                             int lane)
{
    const int nw = ((hi - lo + 31) >> 5) + 2;
    const u32 sh = (u32)(lo & 31);
    __syncthreads();
    {
        const long long g0 = wo + (long long)(lo >> 5);     // arithmetic shift = floor
        const u32 *glo = ref.lo + g0, *ghi = ref.hi + g0, *gnn = ref.nn + g0;
        constexpr int st = (int)PG_WIN_WORDS(NB);
        for (int i = lane; i < nw; i += WAVE) {
            const u32 x = __builtin_amdgcn_alignbit(glo[i + 1], glo[i], sh);
            const u32 y = __builtin_amdgcn_alignbit(ghi[i + 1], ghi[i], sh);
            const u32 z = __builtin_amdgcn_alignbit(gnn[i + 1], gnn[i], sh);
            S.win[i] = make_uint4(x, y, z, 0u);
            const u32 ok = ~z;
            S.eq[i] = ~x & ~y & ok;
            S.eq[st + i] = x & ~y & ok;
            S.eq[2 * st + i] = ~x & y & ok;
            S.eq[3 * st + i] = x & y & ok;
            S.eq[4 * st + i] = ok;
        }
    }
    __syncthreads();
    S.win_wo = wo;
    S.wbase = lo;
    S.win_lo = lo;
    S.win_hi = hi;
}

__device__ __forceinline__ u32 bfi32(u32 m, u32 a, u32 b) { return (m & a) | (~m & b); }
__device__ __forceinline__ u32 bits32(int lo, int hi)      // bits [lo,hi), clamped to [0,32]
{
    lo = lo < 0 ? 0 : lo;
    hi = hi > 32 ? 32 : hi;
    const u32 hm = hi >= 32 ? 0xffffffffu : ((1u << (hi & 31)) - 1u);
    return lo < hi ? (hm & ~((1u << (lo & 31)) - 1u)) : 0u;
}

#ifndef PG_SEED_J
#define PG_SEED_J(T) (2 * (T) + 4)     // consumed bases the seed filter looks at
#endif

// SEED FILTER, bit sliced: the lane owns the 32 window positions of word `lane` of the chunk and returns
// the mask of positions whose candidate (of kind F or B) can matter.  Position bit i, consumed base j
// reads reference base p+j (F) / p-j (B): one alignbit of the one-hot plane of read base j.  Mismatch
// counts are kept bit sliced (a 4-bit ripple counter per position + overflow), 9 VALU per base for 32
// positions.  Which candidates matter (exact, DESIGN.md "relevance"): a candidate at level k at
// length L can only influence the result if k <= g_maxMismatch[L] + ADD -- otherwise either a lower
// level exists (lo + ADD < k) or it is the lowest level itself and the search aborts at L with or
// without it.  With J > bps bases inspected: relevant at some L in [bps, J] implies
// c(bps) <= g_maxMismatch[J] + ADD (the table is monotone); relevant later implies alive after J bases,
// c(J) <= T-1.  With J <= bps only the second test applies.  Anything kept beyond that is harmless.
// positions whose bit-sliced mismatch count (c3 c2 c1 c0, ov = overflowed) is <= thr (wave-uniform)
__device__ __forceinline__ u32 count_le(u32 c0, u32 c1, u32 c2, u32 c3, u32 ov, int thr)
{
    const u32 c[4] = { c0, c1, c2, c3 };
    u32 eq = ~ov, lt = 0u;
#pragma unroll
    for (int i = 3; i >= 0; i--) {
        const u32 ti = ((thr >> i) & 1) ? ~0u : 0u;
        lt |= eq & ~c[i] & ti;
        eq &= ~(c[i] ^ ti);
    }
    return thr >= 15 ? ~ov : (lt | eq);
}

template <int NB, typename Cell>
__device__ __forceinline__ u32 seed_filter(const PgDevParams &prm, const Search<Cell> &S, const Query<NB> &Q,
                                           bool kindB, int lane)
{
    u32 lo = (u32)uni((int)(u32)q_lo<NB>(Q, 0)), hi = (u32)uni((int)(u32)q_hi<NB>(Q, 0));
    const u32 nn = (u32)uni((int)(u32)q_nn<NB>(Q, 0)), oo = (u32)uni((int)(u32)q_oo<NB>(Q, 0));
    if (kindB ? Q.cB : Q.cF) { lo = ~lo; hi = ~hi; }
    const u32 acgt = ~(nn | oo);
    const int T = S.T;
    int J = S.len - 1 < 32 ? S.len - 1 : 32;
    if (J > PG_SEED_J(T)) J = PG_SEED_J(T);
    const int jb = S.bps < J ? S.bps : J;
    const u32 jmask = bits32(1, J), g0mask = bits32(1, jb);
    int cap0 = max_mismatch_at(prm, J) + S.add_mm;        // min(T-1, g_maxMismatch[J] + ADD)
    if (cap0 > T - 1) cap0 = T - 1;
    const int a = 2 * NB + lane - (kindB ? 1 : 0);         // LDS word holding the low half of the pair
    constexpr int st = (int)PG_WIN_WORDS(NB);              // compile-time row stride: immediate LDS offsets

    const int x0 = (int)((lo & 1u) | ((hi & 1u) << 1));    // first base is ACGT (first_ok)
    const u32 seed = S.eq[x0 * st + 2 * NB + lane];
    // read symbols: A C G T N other; bases [1, jb) first, snapshot, then bases [jb, J)
    const u32 sym[6] = { ~lo & ~hi & acgt, lo & ~hi & acgt, ~lo & hi & acgt, lo & hi & acgt, nn, oo };
    u32 wl[6], wh[6];
#pragma unroll
    for (int X = 0; X < 5; X++) {
        wl[X] = S.eq[X * st + a];
        wh[X] = S.eq[X * st + a + 1];
    }
    wl[5] = wh[5] = 0u;                                     // symbol 5 (not ACGTN) never matches
    u32 c0 = 0u, c1 = 0u, c2 = 0u, c3 = 0u, ov = 0u;        // mismatch count per position, bit sliced
    u32 snap = 0u;
#pragma unroll
    for (int it = 0; it < 12; it++) {
        const int X = it >= 6 ? it - 6 : it;
        if (it == 6) snap = count_le(c0, c1, c2, c3, ov, cap0);
        u32 pm = sym[X] & (it >= 6 ? (jmask & ~g0mask) : g0mask);
        while (pm != 0u) {
            const int j = __ffs((int)pm) - 1;
            pm &= pm - 1u;
            const u32 m = __builtin_amdgcn_alignbit(wh[X], wl[X], (u32)(kindB ? 32 - j : j));
            // count += mismatch (= ~m), in place: 8 VALU instructions.  Written as asm because the
            // compiler's version of the same ripple rotates the counter through three extra v_mov.
            u32 k0, k1;
            asm("v_bfi_b32 %5, %7, 0, %0\n\t"          // k0 = ~m & c0
                "v_xnor_b32 %0, %0, %7\n\t"            // c0 ^= ~m
                "v_and_b32 %6, %1, %5\n\t"             // k1 = c1 & k0
                "v_xor_b32 %1, %1, %5\n\t"             // c1 ^= k0
                "v_and_b32 %5, %2, %6\n\t"             // k2 = c2 & k1
                "v_xor_b32 %2, %2, %6\n\t"             // c2 ^= k1
                "v_and_or_b32 %4, %3, %5, %4\n\t"      // ov |= c3 & k2
                "v_xor_b32 %3, %3, %5"                   // c3 ^= k2
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(ov), "=&v"(k0), "=&v"(k1)
                : "v"(m));
        }
    }
    return seed & (snap | count_le(c0, c1, c2, c3, ov, T - 1));
}

// Scan the positions of [s, e) outside [xs, xe) (wo = word index of AbsLoc 0 of the chromosome).
// The window is cut into 2048-position chunks on the grid g0 + 2048 k; a chunk is staged into LDS
// (up to e_max, so that later nested ranges find it resident), every lane filters one 32-position
// word per candidate kind (seed_filter), the surviving positions are compacted into the LDS queue
// and go through dense_pass 64 at a time.  cache*: filter masks of chunk 0 of the far-end window,
// computed once and reused by the nested ranges.
template <int NB, typename Cell, bool MIXED>
__device__ __forceinline__ void scan_impl(const PgDevRef &ref, const PgDevParams &prm, Search<Cell> &S,
                                          const Query<NB> &Q, long long wo, int g0, int s, int e, int e_max,
                                          int xs, int xe, int origin, u32 region, int lane,
                                          bool use_cache, u32 &cacheF, u32 &cacheB, bool &cache_valid)
{
    if (!Q.first_ok || s >= e) return;
    const int k0 = (s - g0) >> PG_CHUNK_SHIFT, k1 = (e - 1 - g0) >> PG_CHUNK_SHIFT;   // floor
    for (int k = k0; k <= k1; k++) {
        const int cs = g0 + (k << PG_CHUNK_SHIFT);
        const int ns = s > cs ? s : cs;
        const int ne = e < cs + (int)PG_CHUNK ? e : cs + (int)PG_CHUNK;
        if (ns >= xs && ne <= xe) continue;                 // nothing new in this chunk
        const int wb = cs - 64 * NB;
        // The chunk must be resident up to e_max, not just up to this call's end: the filter masks of the
        // innermost far-end chunk are computed once for all nested ranges, and an earlier fill with the same
        // base (a close-end window that happens to start where this chunk starts) may be shorter.
        const int se = e_max < cs + (int)PG_CHUNK ? e_max : cs + (int)PG_CHUNK;
        if (!(wo == S.win_wo && S.wbase == wb && se + 64 * NB <= S.win_hi))
        {
            stage_window<NB, Cell>(ref, S, wo, wb, se + 64 * NB, lane);
#if defined(PG_DUP) && PG_DUP == 2
            stage_window<NB, Cell>(ref, S, wo, wb, se + 64 * NB, opaque(lane));
#endif
        }
        const int pbase = cs + 32 * lane;
        const u32 rmask = bits32(ns - pbase, ne - pbase) & ~bits32(xs - pbase, xe - pbase);
        const bool cached = use_cache && k == 0 && cache_valid;
        int qn = 0;                                          // queued survivors (uniform)
#pragma unroll 1
        for (int kb = 0; kb < 3; kb++) {                     // kind F, kind B, drain
            u32 pm = 0u;
            if (kb < 2 && (kb ? Q.allowB : Q.allowF)) {
                u32 m;
                if (cached) m = kb ? cacheB : cacheF;
                else {
                    m = seed_filter<NB, Cell>(prm, S, Q, kb != 0, lane);
#if defined(PG_DUP) && PG_DUP == 3
                    m &= seed_filter<NB, Cell>(prm, S, Q, kb != 0, opaque(lane)) | (u32)opaque(0);
#endif
                    if (use_cache && k == 0) { if (kb) cacheB = m; else cacheF = m; }
                }
                pm = m & rmask;
            }
            for (;;) {
                const bool more = __any(pm != 0u);
                if (more) {
                    const bool act = pm != 0u;
                    const int bit = __ffs((int)pm) - 1;
                    pm &= pm - 1u;
                    const u64 bm = ballot64(act);
                    if (act)
                        S.queue[qn + __popcll(bm & low_bits(lane))] =
                            ((u32)(64 * NB + 32 * lane + bit) << 1) | (u32)kb;
                    const int add = __popcll(bm);
                    qn += add;
                    S.nsurv += add;
                }
                const bool drain = !more && kb == 2 && qn > 0;
                if (qn >= WAVE || drain) {
                    const int n = qn < WAVE ? qn : WAVE;
                    __syncthreads();
#ifndef PG_ABL_NODENSE
                    dense_pass<NB, Cell, MIXED>(S, Q, wb, origin, region, n, lane);
#if defined(PG_DUP) && PG_DUP == 4
                    __syncthreads();
                    dense_pass<NB, Cell, MIXED>(S, Q, wb, origin, region, n, opaque(lane), true);
                    __syncthreads();
                    dense_pass<NB, Cell, MIXED>(S, Q, wb, origin, region, n, opaque(lane));
#endif
#endif
                    __syncthreads();
                    // move the remainder (< 128 entries) to the front
                    const int rem = qn - n;
                    const u32 m0 = (lane < rem) ? S.queue[WAVE + lane] : 0u;
                    const u32 m1 = (WAVE + lane < rem) ? S.queue[2 * WAVE + lane] : 0u;
                    __syncthreads();
                    if (lane < rem) S.queue[lane] = m0;
                    if (WAVE + lane < rem) S.queue[WAVE + lane] = m1;
                    qn = rem;
                } else if (!more) break;
            }
        }
        if (use_cache && k == 0) cache_valid = true;
    }
    __syncthreads();
}

template <int NB, typename Cell>
__device__ __forceinline__ void scan_range(const PgDevRef &ref, const PgDevParams &prm, Search<Cell> &S,
                                           const Query<NB> &Q, long long wo, int g0, int s, int e, int e_max,
                                           int xs, int xe, int origin, u32 region, int lane,
                                           bool use_cache, u32 &cacheF, u32 &cacheB, bool &cache_valid)
{
    // window coordinates come out of LDS / per-read loads: tell the compiler they are wave-uniform
    g0 = uni(g0); s = uni(s); e = uni(e); e_max = uni(e_max); xs = uni(xs); xe = uni(xe); origin = uni(origin);
    if (Q.allowF && Q.allowB)
        scan_impl<NB, Cell, true>(ref, prm, S, Q, wo, g0, s, e, e_max, xs, xe, origin, region, lane,
                                  use_cache, cacheF, cacheB, cache_valid);
    else
        scan_impl<NB, Cell, false>(ref, prm, S, Q, wo, g0, s, e, e_max, xs, xe, origin, region, lane,
                                   use_cache, cacheF, cacheB, cache_valid);
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
    // Lanes own L and one DPP wave scan per level turns the differences into G[k](L); lane k of cv
    // carries G[k] at the end of the previous 64-length round.
    Cell cv = lane < S.T ? S.carry[lane] : (Cell)0;
    bool aborted = false;
    for (int r0 = S.bps; r0 <= S.len - 1 && !aborted; r0 += WAVE) {
        const int L = r0 + lane;
        const bool valid = L <= S.len - 1;
        int lo = -1;
        u32 cnt_lo = 0, sumw = 0;
        u64 id_lo = 0;
        {
            u32 zc = 0u, n1 = 0u;
            Cell gid = 0;
            for (int k = 0; k < S.T; k++) {
                const Cell d = valid ? S.hist[k * S.lh + L] : (Cell)0;
                const Cell g = wave_scan(d) + read_lane(cv, k);
                const Cell g63 = read_lane(g, 63);
                cv = lane == k ? g63 : cv;
                // G[k](L) is non-decreasing in k, so three counters say everything the rules need:
                // zc = levels <= M with G = 0 (= the lowest non-empty level, M+1 if none), n1 = levels
                // with G = 1 (a block starting at zc when G[zc] = 1, hence G[zc] = G[zc+ADD] = 1 <=>
                // n1 > ADD) and the candidate id of any level with G = 1 (the same single candidate)
                const u32 cnt = (u32)(g & (Cell)((1ull << F::CB) - 1ull));
                zc += (k <= S.M && cnt == 0u) ? 1u : 0u;
                n1 += cnt == 1u ? 1u : 0u;
                gid = cnt == 1u ? g : gid;
            }
            lo = (int)zc <= S.M ? (int)zc : -1;
            cnt_lo = sumw = n1 > (u32)S.add_mm ? 1u : 0u;
            id_lo = (u64)(gid >> F::CB);
        }
        int mmL = 0;                                  // g_maxMismatch[L] (<= M for L <= len)
        for (int k = 0; k < S.M; k++) mmL += (valid && (u32)L >= prm.mm_bp[k]) ? 1 : 0;
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
__device__ __forceinline__ void load_planes(const uint8_t *seq, int len, int lane, u64 *qp)
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
        const u64 flo = ballot64(fC || fT), fhi = ballot64(fG || fT), fnn = ballot64(fN);
        const u64 foo = ballot64(in && !(fA || fC || fG || fT || fN));
        const u64 rlo = ballot64(rC || rT), rhi = ballot64(rG || rT), rnn = ballot64(rN);
        const u64 roo = ballot64(in && !(rA || rC || rG || rT || rN));
        if (lane == 0) {
            qp[QP_LO * NB + b] = flo; qp[QP_HI * NB + b] = fhi; qp[QP_NN * NB + b] = fnn; qp[QP_OO * NB + b] = foo;
            u64 *qr = qp + 4 * NB;
            qr[QP_LO * NB + b] = rlo; qr[QP_HI * NB + b] = rhi; qr[QP_NN * NB + b] = rnn; qr[QP_OO * NB + b] = roo;
        }
    }
    __syncthreads();
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
    const u32 x = (u32)uni((int)(u32)(q_nn<NB>(Q, 0) | q_oo<NB>(Q, 0)));
    return (x & 1u) == 0u;
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
    if (blockIdx.x >= B.n_reads) return;
    // XCD-aware read order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
    // workgroup b runs on XCD b % 8.  Give every XCD a CONTIGUOUS eighth of the reads: the per-read input
    // and output fields of neighbouring reads share cache lines, which then live in one L2 instead of
    // being fetched and written back by up to eight.
    uint32_t local = blockIdx.x;
    {
        const uint32_t per = B.n_reads / PG_N_XCD;
        if (local < per * PG_N_XCD) local = (local % PG_N_XCD) * per + local / PG_N_XCD;
    }
    const uint32_t rid = B.first_read + local;

    const PgLdsLayout lay = pg_lds_layout(max_len, levels, NB, (uint32_t)sizeof(Cell));
    Search<Cell> S;
    S.hist = (Cell *)(smem + lay.hist_off);
    S.ginit = (Cell *)(smem + lay.carry_off);
    S.carry = S.ginit + PG_MAX_LEVELS;
    S.queue = (u32 *)(smem + lay.queue_off);
    S.win = (uint4 *)(smem + lay.win_off);
    S.eq = (u32 *)(smem + lay.eq_off);
    S.eq_stride = (int)lay.win_words;
    pg_run *runs_tmp = (pg_run *)(smem + lay.runs_off);
    S.lh = (int)lay.lh;
    S.win_wo = -1;
    S.win_lo = S.win_hi = S.wbase = 0;
    S.nsurv = 0;

    const u64 off = B.seq_off[rid];
    const int len = (int)(B.seq_off[rid + 1] - off);
    const uint8_t *seq = B.seq + off;
    const int chr = uni((int)B.chr[rid]);
    const long long chr_wo = (long long)ref.chr_word_off[chr];
    const int chr_size = (int)ref.chr_size[chr];
    S.len = len;
    S.M = max_mismatch_at(prm, len);
    S.add_mm = prm.add_mm;
    S.T = S.M + prm.add_mm + 1;
    S.min_perfect = prm.min_perfect;
    S.thr = prm.thr_tab[len];

    PT_DECL
    u64 *qplanes = (u64 *)(smem + lay.qp_off);   // [0]: forward, [1]: reversed consumption order
    load_planes<NB>(seq, len, lane, qplanes);
#if defined(PG_DUP) && PG_DUP == 1
    load_planes<NB>(seq, len, opaque(lane), qplanes);
#endif
    PT_MARK(0)
#if defined(PG_STOP_AFTER) && PG_STOP_AFTER == 0
    if (lane == 0) B.rc_flag[rid] = (uint8_t)(qplanes[0] ^ qplanes[4 * NB + NB]);
    return;
#endif

    int alg8 = (mode & PG_MODE_CLOSE) ? 8 * len : 0;         // algorithmic bytes x 8; the read itself is counted once
    int flipped = 0, close_max = 0, n_close = 0, n_far = 0, far_max = 0;
    u32 close_last = 0, close_base = 0, far_base = 0;

    const bool do_close = (mode & PG_MODE_CLOSE) != 0, do_far = (mode & PG_MODE_FAR) != 0;
    const char strand = do_close ? (char)uni((int)B.strand[rid]) : '+';   // byte loads go through VMEM
    const int apos = do_close ? uni((int)(B.pos[rid] + (int)prm.spacer)) : 0;
    const int isz = do_close ? uni((int)B.isz[rid]) : 0;
    if (!do_close) {
        flipped = uni((int)B.rc_flag[rid]);
        close_last = (u32)uni((int)B.close_last_abs[rid]);
        close_max = uni((int)B.close_max_len[rid]);
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
    int close_bases = 0, far_bases = 0;
    u32 cacheF = 0u, cacheB = 0u;                    // seed-filter masks of the innermost far-end chunk
    bool cache_valid = false;

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
        int nwin = 0;                 // windows to scan this step (<= 1, or nbd)
        int s1 = 0, e1 = 0;           // positions [s1, e1) minus [xs, xe) are new in this step
        int xs = 0, xe = 0, g0 = 0, emax = 0;
        int origin = 0;
        bool zero = false;
        if (is_close) {
            const int Rg = step >> 1;
            flipped = (step == 1 || step == 2) ? 1 : 0;
            S.bps = prm.min_close;
            // '+' anchor: CurrentReadSeq = RC(cur), grown left to right (pindel.cpp:2271-2291)
            // '-' anchor: CurrentReadSeq = cur, grown right to left     (pindel.cpp:2298-2319)
            Q.qp = qplanes + (!flipped ? 4 * NB : 0);
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
            g0 = s1;
            emax = e1;
            nwin = 1;
            zero = true;
            close_bases = e1 > s1 ? e1 - s1 : 0;
        } else {
            S.bps = 10;               // farend_searcher.cpp:90
            // cur = flipped ? RC(orig) : orig.  Plus strand consumes cur left to right, Minus strand
            // consumes complement(cur) walking the reference right to left.
            Q.qp = qplanes + (flipped ? 4 * NB : 0);
            Q.cF = flipped; Q.cB = !flipped;
            Q.allowF = Q.allowB = true;
            Q.antisenseF = false;     // FORWARD, SENSE
            Q.antisenseB = true;      // BACKWARD, ANTISENSE
            if (step == 4) {
                if (nbd == 0) { step++; continue; }
                nwin = nbd;
                zero = true;
            } else {
                const int center = (int)close_last;
                origin = center - maxspan;
                if (step == 5) { zero = true; ps = pe = 0; span = 64; cache_valid = false; }
                // window of this range, clipped to the non-spacer part (pindel.cpp:1034-1043); the
                // histogram is additive, so only the flanks the previous ranges did not cover are
                // scanned.  Chunk grid: the innermost 2048 positions are one chunk (one LDS fill and
                // one seed-filter pass serve the ranges up to 1024).
                int s, e;
                if ((u32)center > (u32)span + prm.spacer) s = center - span; else s = (int)prm.spacer;
                if ((u32)center + (u32)span + prm.spacer < (u32)chr_size) e = center + span;
                else e = chr_size - (int)prm.spacer;
                g0 = center - (int)PG_CHUNK / 2;
                if ((u32)center + (u32)maxspan + prm.spacer < (u32)chr_size) emax = center + maxspan;
                else emax = chr_size - (int)prm.spacer;
                if (s < e) {
                    s1 = s; e1 = e; nwin = 1;
                    xs = ps; xe = pe;
                    if (ps < pe) {
                        ps = s < ps ? s : ps;
                        pe = e > pe ? e : pe;
                    } else {
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
        if (zero) { zero_hist(S, opaque(lane)); S.nsurv = 0; nsurv_eval = -1; }
#if defined(PG_DUP) && PG_DUP == 6
        if (zero) zero_hist(S, opaque(lane));
#endif
        PT_MARK(4)
        // ---------------- scan
#ifdef PG_ABL_NOSCAN
        nwin = 0;
#endif
        for (int w = 0; w < nwin; w++) {
            long long wo = chr_wo;
            int s = s1, e = e1, org = origin;
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
                g0 = s;
                emax = e;
                far_bases += (e > s ? e - s : 0) + 2 * len;
            }
            scan_range<NB, Cell>(ref, prm, S, Q, wo, g0, s, e, emax, xs, xe, org, region, opaque(lane),
                                 step >= 5, cacheF, cacheB, cache_valid);
        }
        PT_MARK(1)
#if defined(PG_STOP_AFTER) && PG_STOP_AFTER == 1
        if (lane == 0) B.rc_flag[rid] = (uint8_t)S.nsurv;
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
        if (fresh) {
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
                    int nn = evaluate<NB, Cell>(ref, prm, S, Q, R, runs_tmp, skip, PG_RUN_TMP, mm, opaque(lane));
#if defined(PG_DUP) && PG_DUP == 5
                    { int mm2; nn = evaluate<NB, Cell>(ref, prm, S, Q, R, runs_tmp, skip, PG_RUN_TMP, mm2, opaque(lane)); mm = mm2; }
#endif
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
                    last0 = (u32)uni((int)tmp32[3 * li]);
                    last1 = (u32)uni((int)tmp32[3 * li + 1]);
                    last2 = (u32)uni((int)tmp32[3 * li + 2]);
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
        alg8 += 3 * (close_bases + 2 * len) + 96 * n_close;   // 3 bits per base, 12 bytes per run
        if (lane == 0) {
            B.rc_flag[rid] = (uint8_t)flipped;
            B.close_last_abs[rid] = close_last;
            B.close_max_len[rid] = (uint16_t)close_max;
            B.close_run_off[rid] = close_base;
            B.close_run_cnt[rid] = (u32)n_close;
        }
    }
    if (do_far) {
        if (far_ready) far_bases += reach + 2 * len;
        alg8 += 3 * far_bases + 96 * n_far;
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
        if (do_close) B.alg_bytes[rid] = (u32)(alg8 + 4) >> 3;
        else B.alg_bytes[rid] += (u32)(alg8 + 4) >> 3;        // the far-end launch adds to the close-end launch
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

// ---------------------------------------------------------------------------------
// Results to CSR on the device: the kernel leaves each read's runs somewhere in its pool shard; before
// the copy to the host they are gathered in read order (offsets = exclusive prefix sums of the per-read
// counts), so that only the compact lists cross PCIe and the host does no per-read work.
#include <hipcub/hipcub.hpp>

__global__ void pg_gather_runs_kernel(const pg_run *pool, const uint32_t *off, const uint32_t *cnt,
                                      const uint32_t *csr, pg_run *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = cnt[i];
    const u32 *src = (const u32 *)(pool + off[i]);
    u32 *dst = (u32 *)(out + csr[i]);
    for (uint32_t k = 0; k < 3u * c; k++) dst[k] = src[k];
}

// csr[0..n] = exclusive prefix sums of cnt[0..n) (cnt[n] must be readable; it is ignored: the scan
// runs over n + 1 items so that csr[n] = total).  tmp / tmp_bytes: scratch from pg_scan_tmp_bytes.
extern "C" size_t pg_scan_tmp_bytes(uint32_t n)
{
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)(n + 1));
    return bytes;
}

extern "C" int pg_compact_runs(const pg_run *pool, const uint32_t *off, const uint32_t *cnt, uint32_t *csr,
                               pg_run *out, uint32_t n, void *tmp, size_t tmp_bytes, int gather, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!gather) {
        hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, cnt, csr, (int)(n + 1), st);
        return (int)e;
    }
    if (n) pg_gather_runs_kernel<<<(n + 255u) / 256u, 256, 0, st>>>(pool, off, cnt, csr, out, n);
    return (int)hipGetLastError();
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
    // 64-base blocks per read: 1/2/3/4/8 with 32-bit cells (the common case), 2/4/8 with 64-bit cells
    const int nb = max_len <= 128 ? 2 : (max_len <= 256 ? 4 : 8);
    if (small_cells) {
        if (max_len <= 64) launch<1, u32>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
        else if (nb == 2) launch<2, u32>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
        else if (max_len <= 192) launch<3, u32>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
        else if (nb == 4) launch<4, u32>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
        else launch<8, u32>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
    } else {
        if (nb == 2) launch<2, u64>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
        else if (nb == 4) launch<4, u64>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
        else launch<8, u64>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
    }
    return (int)hipGetLastError();
}

// pg_kernels.hip -- split-read pattern growth for gfx950 (MI355X), one wavefront per read.
//
// What the reference does per read:
//   close end  GetCloseEnd / GetCloseEndInner      src/pindel.cpp:2531-2605, 2250-2326
//              CheckLeft_Close / CheckRight_Close  src/searcher.cpp:153-197, 247-286
//   far end    SearchFarEnd                        src/pindel.cpp:1001-1074
//              SearchFarEndAtPos                   src/farend_searcher.cpp:46-103
//              CheckBoth / ExtendMatch             src/pindel.cpp:2823-2902, 2673-2725
//   both       CategorizePositions, CheckMismatches src/searcher.cpp:48-63, 331-388
// The reference grows per-mismatch-level position lists one base at a time.  This kernel computes
// the same function differently (DESIGN.md "kernel formulation"):
//
//   * The chromosome lives in HBM as three bit planes (2-bit code planar + N plane).  A window chunk
//     (2048 positions + overhang) is staged into LDS with coalesced dword loads as code planes; the seed filter derives
//     the one-hot planes (is-A/C/G/T/not-N) of its words from them.  The reads arrive as bit planes too (built by
//     pg_pack_kernel), one 8-byte load per lane and read.
//   * SEED FILTER, bit sliced: each lane owns the 32 window positions of one LDS word.  The match mask of
//     consumed base j for all 32 positions is one v_alignbit of the one-hot plane of that read symbol;
//     mismatch counts live in a bit-sliced carry-save counter (3 to 5 slices of 32 bits).  It keeps exactly the seeds that
//     can matter (DESIGN.md "relevance"); survivors (~2 % of positions) get queue slots from a wave prefix
//     sum of the per-lane popcounts.
//   * CANDIDATES, one per lane, 64 at a time: the mismatch pattern of the read placed at p comes 64 bases
//     per step (funnel shifts + XOR).  A candidate is fully described by its mismatch bitmap `mis`
//     (level after L bases = popcount(mis & lowbits(L))), its exact-inequality bitmap `sne` (for
//     CheckMismatches' perfect-match window) and its whole-read Hamming count.
//   * ACCUMULATE, lanes own lengths L: the reference's emission rule at length L only needs the lowest
//     level, whether a second candidate lies within ADDITIONAL_MISMATCH of it, and the identity of the
//     lowest one.  That is a running (min1, min2, argmin) reduction over candidates -- associative, so
//     candidates stream through it in any order and nothing like a per-level position list or histogram
//     is ever stored.  Two tiers: candidates that die within 16 bases of the first evaluated length
//     (almost all of them) are folded four at a time by 16-lane quarters that cover only those 16
//     lengths; the few long-lived ones are folded with all 64 lanes per 64-length round.
//     CheckMismatches is evaluated for every (candidate, L) inside the fold (three bit operations) and
//     travels with the argmin.
//   * EVALUATE: abort rule, emission rule and run-length encoding of consecutive points with ballots;
//     runs go straight to the pooled output.
//   * Nested far-end ranges (128, 512, 2048 ... bases) only add the candidates of the new flanks: the
//     reduction is additive over disjoint position sets.
//   * PERSISTENT waves: a launch has a few workgroups per CU; each claims chunks of reads from per-XCD
//     counters (every XCD works through a contiguous eighth of the batch: neighbouring reads share cache
//     lines and reference windows in ONE L2).
//
// The kernel is bound by instruction issue, not by HBM -- vector and scalar together, the CU's one scalar unit first: in situ a
// scalar instruction costs 1.65 ns of SIMD time, a vector one 0.8-1.1 (profiles/r03/ubench_issue_rates.txt,
// profiles/r04/issue_calibration.txt, profiles/r04/component_instruction_counts.txt).  No MFMA: this is bit/byte comparison
// work, not a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pg_device.h"

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));       // (scalar-load destinations)
typedef u32 u32x8 __attribute__((ext_vector_type(8)));

// KERNEL ARGUMENTS ON DEMAND.  The kernel's by-value arguments (reference planes, parameters, batch: ~45 dwords of pointers
// and sizes) are used a handful of times per read, but as ordinary arguments they sit in scalar registers from the first
// instruction to the last -- a third of the 102 the wave has, in a kernel whose every loop nest wants all of them: 200 SGPR
// spills, 117 of the 480 reloads (v_readlane, a slow-rate VALU instruction) four or five loops deep.  KA(obj, member) /
// KAP(member) fetch an argument from the kernarg segment where it is used (s_load + wait: the scalar cache holds the segment
// after the first wave) and nothing stays live: 102 spills, 5 reloads at depth >= 4, configs[2] -4.3 %, -x 5 -7.6 %
// (profiles/r04/kernel_experiments.txt).  PgKArgs mirrors pg_search_kernel's parameter list (natural alignment = the layout
// of the kernarg segment).
struct PgKArgs { PgDevRef ref; PgDevParams prm; PgDevBatch B; uint32_t max_len, levels; };
// A pointer out of the kernarg segment points to global memory.  Said so (a cast through address space 1, which the compiler
// propagates to the uses), its loads and stores are global_* instructions; left generic they are flat_*, which count on the LDS
// counter too -- every wait for an LDS read then also waits for the HBM loads in flight, and "request early, use late" is lost.
template <typename T> struct KaGlobal { static __device__ __forceinline__ T of(unsigned long long v) { return (T)v; } };
template <typename E> struct KaGlobal<E *> {
#ifdef PG_FLAT_KARGS       // (ablation)
    static __device__ __forceinline__ E *of(unsigned long long v) { return (E *)v; }
#else
    static __device__ __forceinline__ E *of(unsigned long long v) { return (E *)(__attribute__((address_space(1))) E *)v; }
#endif
};
template <typename T>
__device__ __forceinline__ T karg_load(int off)
{
    const auto k = __builtin_amdgcn_kernarg_segment_ptr();
    if (sizeof(T) == 8) {
        unsigned long long v;
        asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(k), "n"(off));   // (early clobber: never the base's registers)
        return KaGlobal<T>::of(v);
    } else {
        unsigned int v;
        asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(k), "n"(off));
        return (T)v;
    }
}
// three consecutive 64-bit kernel arguments with one wait
template <int OFF, typename T>
__device__ __forceinline__ void karg_load3(T &a, T &b, T &c)
{
    static_assert(sizeof(T) == 8, "pointers");
    const auto k = __builtin_amdgcn_kernarg_segment_ptr();
    unsigned long long x, y, z;
    asm volatile("s_load_dwordx2 %0, %3, %4\n\ts_load_dwordx2 %1, %3, %5\n\ts_load_dwordx2 %2, %3, %6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(x), "=&s"(y), "=&s"(z) : "s"(k), "n"(OFF), "n"(OFF + 8), "n"(OFF + 16));
    a = KaGlobal<T>::of(x); b = KaGlobal<T>::of(y); c = KaGlobal<T>::of(z);
}
#define KA(obj, member) karg_load<decltype(obj.member)>((int)offsetof(PgKArgs, obj.member))
#define KAP(member) karg_load<decltype(PgDevParams::member)>((int)offsetof(PgKArgs, prm.member))
// A search parameter inside search_read<..., DEF>: out of the kernarg segment, or -- in the kernels built for Pindel's default parameter
// set (-x 2 -a 1 -m 3 -H 8, spacer 100000: pg_default_params) -- a constant.  With the five of them known the compiler folds the close
// end's masks and tier switches, unrolls the range schedule and drops the wide-window bookkeeping: SGPR spills 69 -> 20,
// 1251 -> 1130 scalar and 1647 -> 1563 vector instructions per read, configs[2] -6.9 %, 150 bp -7.6 % (results identical).  Any other
// parameter set runs the generic kernels.
#define PRM(member, dflt) (DEF ? (decltype(PgDevParams::member))(dflt) : KA(prm, member))
#define PG_DEF_MAX_RANGE_INDEX 2
#define PG_DEF_ADD_MM 1
#define PG_DEF_MIN_PERFECT 3
#define PG_DEF_MIN_CLOSE 8
#define PG_DEF_SPACER 100000u

#define WAVE 64
#ifndef PG_N_XCD
#define PG_N_XCD 8u        // MI355X: 8 accelerator complex dies, 32 CUs and one L2 each
#endif
// Register budget the kernel is compiled for, in waves per SIMD (= resident single-wave workgroups per CU / 4), decided by the
// size of the kernel's LDS object -- LDS comes in granules of 1280 bytes:
//   seven (72 VGPRs)  up to 5120 B = four granules, 28 workgroups per CU: reads of up to 128 bases with 32-bit candidate ids
//   six   (80 VGPRs)  up to 6400 B = five granules, 24 workgroups: 129..192 bases; 64-bit ids
//   five  (96 VGPRs)  longer reads (7.6 KB and up)
// History of the measurement (100 bp, ms per 2 M reads): four waves 6.69, five 6.03, six 5.60 once the kernel's arguments had
// stopped occupying scalar registers (round 4), seven then 6.02 (50 VGPR spills); with the default-parameter kernels and without
// machine LICM seven fit 72 VGPRs with 3 spills: 4.97 -> 4.91 (generic kernels 5.35 -> 5.24).
#ifdef PG_WAVES_PER_EU
#define PG_WAVES(NB, Id) PG_WAVES_PER_EU
#else
#define PG_WAVES(NB, Id) (sizeof(Lds<NB, Id>) <= 5120 ? 7 : (sizeof(Lds<NB, Id>) <= 6400 ? 6 : (sizeof(Lds<NB, Id>) <= 7680 ? 5 : 4)))
#endif
#ifndef PG_CLAIM
#define PG_CLAIM 8u         // reads claimed per atomic
#endif
#define PG_BIG 0xffffu      // "no candidate" level
#define PG_CHR_TAB 24       // chromosomes whose word offset / size are kept in LDS (window clusters on other chromosomes)
// Candidates per pass.  Reads over 128 bases (NB >= 3) take 32: the tier B entries (PASS x NB mismatch words), the queue and the
// chromosome table are what decides between six and seven (NB = 3: 6032 -> 5008 B), five and six (NB = 4: 7.2 -> 6.0 KB) and three
// and four waves per SIMD (NB = 8: 12.5 -> 10.2 KB = eight 1280-byte granules, sixteen workgroups per CU) -- and a pass rarely
// holds more than a dozen candidates (more than 32 survivors in the innermost far-end chunk: the ranges one by one, as for 64).
#define PG_PASS(nb) ((nb) >= 3 ? 32 : 64)
#define PG_CHR_TAB_N(nb) ((nb) >= 3 ? 0 : PG_CHR_TAB)
#define PG_MM_IN_WIN(nb) ((nb) <= 4)

// Lane masks.  ballot64 of ONE compare is that compare's result register; of a compound condition the compiler first builds the
// condition per lane and then turns it into a mask with a 0 / 1 select and a second compare -- so compound conditions are written
// as scalar ANDs of single-compare ballots, and lane_bit() turns a mask back into a per-lane condition for free.
__device__ __forceinline__ u64 ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool lane_bit(u64 m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
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
__device__ __forceinline__ u64 wave_shr1(u64 v)
{
    return (u64)wave_shr1((u32)v) | ((u64)wave_shr1((u32)(v >> 32)) << 32);
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
__device__ __forceinline__ u32 read_lane(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ u64 read_lane(u64 v, int l)
{
    return (u64)read_lane((u32)v, l) | ((u64)read_lane((u32)(v >> 32), l) << 32);
}

// LDS hand-over between the lanes of the workgroup's ONE wave: the LDS unit executes a wave's instructions in order, so all
// this has to do is keep the compiler from moving LDS reads above LDS writes of other lanes.  (__syncthreads() costs
// nothing as a barrier here -- the compiler drops s_barrier for a 64-thread workgroup -- but its workgroup-scope fences
// also wait for every outstanding global load and STORE.)
#ifdef PG_SYNCTHREADS
#define PG_SYNC() __syncthreads()
#else
#define PG_SYNC()                                              \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#endif
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
// The lane's index, recomputed (v_mbcnt: two instructions) where a fresh copy is wanted: the compiler otherwise keeps threadIdx.x in
// one VGPR for the whole kernel and, short of registers, parks it in scratch and reloads it at the start of every read.
__device__ __forceinline__ int lane_now()
{
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (u32)opaque(0)));
}
__device__ __forceinline__ u64 low_bits(int n)            // clamped to [0,64]
{
    return n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull));
}
__device__ __forceinline__ u32 low32(int n)               // clamped to [0,32]
{
    return n >= 32 ? 0xffffffffu : (n <= 0 ? 0u : ((1u << n) - 1u));
}
// the same for a per-lane n, without the two compare / select pairs (v_cmp + s_nop + v_cndmask each): v_med3 + v_bfm
// (width n & 31: 0 for n = 32) + the n = 32 case.  (Measured: -1 % where the masks are per read, +0.5 % in the wide
// far-end windows' loop, which keeps low32.)
__device__ __forceinline__ u32 low32_lane(int n)
{
    int t;
    asm("v_med3_i32 %0, %1, 0, 32" : "=v"(t) : "v"(n));
    u32 m;
    asm("v_bfm_b32 %0, %1, 0" : "=v"(m) : "v"(t));
    return m | (0u - ((u32)t >> 5));
}
// low_bits for a per-lane n: two halves of four instructions (v_med3, v_bfm, v_bfe_i32 for the width-32 case, v_or), no compare /
// select pairs and their wait states
__device__ __forceinline__ u32 low32_lane4(int n)
{
    int t, f;
    asm("v_med3_i32 %0, %1, 0, 32" : "=v"(t) : "v"(n));
    u32 m;
    asm("v_bfm_b32 %0, %1, 0" : "=v"(m) : "v"(t));
    asm("v_bfe_i32 %0, %1, 5, 1" : "=v"(f) : "v"(t));      // -1 for t = 32
    return m | (u32)f;
}
__device__ __forceinline__ u64 low_bits_lane(int n)
{
    return (u64)low32_lane4(n) | ((u64)low32_lane4(n - 32) << 32);
}
// bits of a (wave-uniform) mask below the calling lane
__device__ __forceinline__ int count_below(u64 m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}
__device__ __forceinline__ u64 bit_range(int lo, int hi)  // bits [lo,hi), clamped to [0,64]
{
    return low_bits(hi) & ~low_bits(lo);
}
__device__ __forceinline__ u64 funnel64(u32 w0, u32 w1, u32 w2, u32 s)
{
    u32 a = __builtin_amdgcn_alignbit(w1, w0, s);
    u32 b = __builtin_amdgcn_alignbit(w2, w1, s);
    return (u64)a | ((u64)b << 32);
}
__device__ __forceinline__ u32 med3(u32 a, u32 b, u32 c)
{
    u32 r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Address of a read's input / output record: base + index * 2^SHIFT, done on the scalar unit by hand.  Left to itself the compiler
// does this 64-bit arithmetic next to the per-lane addresses on the vector unit, keeps the zero-extended read index in a VGPR pair
// for the whole read -- and spills it to scratch (the round's only VGPR spills: 8 bytes of scratch written and read per read).
template <int SHIFT, typename T>
__device__ __forceinline__ T *record_ptr(T *base, uint32_t rid)
{
    const u64 b = (u64)(uintptr_t)base;
    const u32 blo = (u32)uni((int)(u32)b), bhi = (u32)uni((int)(u32)(b >> 32)), r = (u32)uni((int)rid);
    u32 lo, hi;
    asm("s_lshl_b32 %0, %4, %5\n\ts_lshr_b32 %1, %4, %6\n\ts_add_u32 %0, %0, %2\n\ts_addc_u32 %1, %1, %3"
        : "=&s"(lo), "=&s"(hi) : "s"(blo), "s"(bhi), "s"(r), "n"(SHIFT), "n"(32 - SHIFT) : "scc");
    return KaGlobal<T *>::of((u64)lo | ((u64)hi << 32));
}

// Candidate ids: position relative to the search's origin, kind (F/B) and window index (BreakDancer cluster).
//   32-bit: rel(24) | kind << 24 | region << 25   -- every window of the launch has <= 2^24 positions, clusters <= 127 windows
//   64-bit: rel(26) | kind << 26 | region << 27   -- anything the ABI accepts (56 bits are kept: clusters of up to 2^29
//                                                    windows; windows of 2^26 positions or more are split by the host)
template <typename Id> struct IdFmt;
template <> struct IdFmt<u32> { static constexpr int RB = PG_REL_BITS_SMALL; };
template <> struct IdFmt<u64> { static constexpr int RB = PG_REL_BITS; };
template <typename Id>
__device__ __forceinline__ Id make_id(u32 rel, bool isB, u32 region)
{
    return (Id)rel | ((Id)(isB ? 1u : 0u) << IdFmt<Id>::RB) | ((Id)region << (IdFmt<Id>::RB + 1));
}

// The read's bit planes live in LDS (not in SGPRs: 32+ SGPRs of planes would be spilled to VGPR lanes and
// come back through v_readlane; an LDS broadcast read costs no vector issue slot).  Layout:
// u64 qp[2 orientations][4 planes][NB blocks]; orientation 0 = the read left to right, 1 = reversed;
// planes in CONSUMPTION order (bit j of block b = base 64b+j the growth consumes): code bit0, code bit1,
// is 'N', is other (never matches).
enum { QP_LO = 0, QP_HI = 1, QP_NN = 2, QP_OO = 3 };

// Everything a search needs to know about the query.  (Plain bools on purpose: they are compile-time constants along each
// path of search_read once everything is inlined; as bits of one flags word -- tried, to save scalar registers -- they turn
// into run-time data and the fused kernel's SGPR spills went from 96 to 130.)
template <int NB>
struct Query {
    const u64 *qp;       // planes of the base orientation (before complement): qp[plane * NB + block]
    bool allowF_, allowB_; // candidate kinds searched
    bool cF_, cB_;         // complement flag per kind
    bool first_ok_;        // first consumed base is one of ACGT
    bool o1_;              // the planes in hand are those of orientation 1 (the read from its last base): which symbol program (seed_filter_ro)
    __device__ __forceinline__ bool allowF() const { return allowF_; }
    __device__ __forceinline__ bool allowB() const { return allowB_; }
    __device__ __forceinline__ bool cF() const { return cF_; }
    __device__ __forceinline__ bool cB() const { return cB_; }
    __device__ __forceinline__ bool first_ok() const { return first_ok_; }
};
template <int NB> __device__ __forceinline__ u64 q_lo(const Query<NB> &Q, int b) { return Q.qp[QP_LO * NB + b]; }
template <int NB> __device__ __forceinline__ u64 q_hi(const Query<NB> &Q, int b) { return Q.qp[QP_HI * NB + b]; }
template <int NB> __device__ __forceinline__ u64 q_nn(const Query<NB> &Q, int b) { return Q.qp[QP_NN * NB + b]; }
template <int NB> __device__ __forceinline__ u64 q_oo(const Query<NB> &Q, int b) { return Q.qp[QP_OO * NB + b]; }

// Static LDS of a workgroup (one wave).  Static, not dynamic: every address below is a compile-time constant,
// which keeps the five base pointers out of the scalar registers the kernel is short of.
// The reduction state of one length in LDS (rounds >= 1): 8 bytes with 32-bit ids -- m1 | min(m2, 0x7fff) << 16 |
// ok << 31, id (levels are mismatch counts <= the read length; PG_BIG only means "none") -- 16 bytes with 64-bit ids.
template <typename Id> struct AccB;
template <> struct AccB<u32> {
    typedef uint2 T;
    static __device__ __forceinline__ void load(const void *base, int i, u32 &m1, u32 &m2, u32 &ok, u32 &id)
    {
        const uint2 st = ((const uint2 *)base)[i];
        m1 = st.x & 0xffffu; m2 = (st.x >> 16) & 0x7fffu; ok = st.x >> 31; id = st.y;
    }
    static __device__ __forceinline__ void store(void *base, int i, u32 m1, u32 m2, u32 ok, u32 id)
    {
        const u32 c2 = m2 < 0x7fffu ? m2 : 0x7fffu;
        ((uint2 *)base)[i] = make_uint2(m1 | (c2 << 16) | (ok << 31), id);
    }
};
template <> struct AccB<u64> {
    typedef uint4 T;
    static __device__ __forceinline__ void load(const void *base, int i, u32 &m1, u32 &m2, u32 &ok, u64 &id)
    {
        const uint4 st = ((const uint4 *)base)[i];
        m1 = st.x; m2 = st.y & 0xffffu; ok = st.y >> 16; id = (u64)st.z | ((u64)st.w << 32);
    }
    static __device__ __forceinline__ void store(void *base, int i, u32 m1, u32 m2, u32 ok, u64 id)
    {
        ((uint4 *)base)[i] = make_uint4(m1, m2 | (ok << 16), (u32)id, (u32)(id >> 32));
    }
};

template <int NB, typename Id>
struct Lds {
    uint4 bufA[68];                           // tier A entries {mis0 lo, sne0 lo, id lo, meta}; scratch for the quarter merge
    uint2 bufB[PG_PASS(NB) * NB];             // tier B entries: the mismatch bitmap, NB x {mis lo, mis hi} per candidate
    uint2 hdrB[64];                           // ... and their {id lo, meta}
    u64 ringB[3 * NB];                        // fused far-end ranges: the long-lived candidates of a pass per ring and round
    typename AccB<Id>::T accB[NB > 1 ? 64 * (NB - 1) : 1];   // reduction state of the rounds >= 1 (AccB)
    u64 qp[2 * 4 * NB];                       // the read's bit planes, two orientations
    uint16_t queue[PG_PASS(NB)];              // survivors of the prefilter for one candidate pass: (window position << 1) | kind
    // g_maxMismatch[L] for every length a lane can own (filled once per workgroup).  Up to 256-base reads the table
    // lives in the fourth dword of the window entries (four lengths per dword; the fills write three dwords): the LDS
    // it would take costs a resident workgroup per CU at NB = 3 and 4.
    uint8_t mm_tab[PG_MM_IN_WIN(NB) ? 4 : 64 * NB + 64];
    uint2 chr_tab[PG_CHR_TAB_N(NB)];          // word offset and size of the first PG_CHR_TAB chromosomes (size 0: not in the table --
                                              // its word offset does not fit 32 bits)
#ifdef PG_TIMING
    u64 t_last;
    u32 t_acc[12];
    u32 t_pad[2];
#endif
    // staged window: code planes (lo, hi, N).  LAST member: for NB = 3 its second chunk's words are the launch's dynamic LDS,
    // which begins where this object ends (PG_WIN_STATIC_WORDS, pg_device.h; checked at the start of the kernel)
    uint4 win[PG_WIN_STATIC_WORDS(NB)];
};

struct Search {
    int len, T, M, add_mm, bps, min_perfect, thr;     // (add_mm / min_perfect: fetched once per read)
    uint16_t *queue;
    uint4 *win;
    uint4 *bufA;
    uint2 *bufB;
    uint2 *hdrB;
    u64 *ringB;
    void *accB;
    const uint8_t *mm_tab;
    const uint2 *chr_tab;
    // the seed filter's two depths of this read (plain, wide windows), from the read's record: `depth` = J plain | J wide << 8 |
    // bound plain << 16 | bound wide << 24 (bound = min(T - 1, g_maxMismatch[J] + ADD)), jmask = bits [1, J)
    u32 depth, jmask[2];
    u32 ro;              // PgInRec::ro (read-order filter: groups, bounds, PG_RO_OK)
    u32 rid;             // the read's index
    u32 rp_lo, rp_hi;    // address of the read's record, as two wave-uniform words (the filter runs fetch its symbol programs; as a
                         // POINTER member the compiler kept it in a VGPR pair and parked that in scratch)
    // what the LDS window currently holds: bases [win_lo, win_hi) of the chromosome whose AbsLoc 0 is
    // at word index win_wo; the first staged base is wbase = win_lo (any alignment)
    long long win_wo;
    int win_lo, win_hi, wbase;
    int nsurv;           // candidates folded since the state was reset
    u32 nsurv_total;     // ... since the read started (diagnostics: survivors of the seed filter)
#ifdef PG_TIMING
    u64 *t_last;         // diagnostics build (LDS): s_memtime at the last phase boundary, cycles per phase so far
    u32 *t_acc;
    int t_base;
#define PG_T(S, k) do { if (threadIdx.x == 0) { const u64 t_ = __builtin_readcyclecounter(); (S).t_acc[k] += (u32)(t_ - *(S).t_last); *(S).t_last = t_; } } while (0)
#else
#define PG_T(S, k) ((void)0)
#endif
#ifdef PG_STOP
    // diagnostics ladder (wrong results): -DPG_STOP=k returns from the read the first time it reaches point k; the PMC difference
    // between two builds = the instructions of the code between the two points (profiles/r05/everything_else_breakdown.txt)
    u32 stopped;
// (every point leaves a marker in the ISA -- s_nop 14, s_nop k & 7, s_nop k >> 3 -- so that the code between two points can be read)
#define PG_STOP_AT(S, k) do { asm volatile("s_nop 14\n\ts_nop %0\n\ts_nop %1" : : "n"((k) & 7), "n"((k) >> 3)); \
                              if (PG_STOP == (k) && uni(opaque(1))) { (S).stopped = 1u; return; } } while (0)
#define PG_STOPPED(S) do { if ((S).stopped) return; } while (0)
// (inside a lambda that returns a value)
#define PG_STOP_AT_V(S, k, v) do { asm volatile("s_nop 14\n\ts_nop %0\n\ts_nop %1" : : "n"((k) & 7), "n"((k) >> 3)); \
                                 if (PG_STOP == (k) && uni(opaque(1))) { (S).stopped = 1u; return (v); } } while (0)
#define PG_STOPPED_V(S, v) do { if ((S).stopped) return (v); } while (0)
#else
#define PG_STOP_AT(S, k) ((void)0)
#define PG_STOPPED(S) ((void)0)
#define PG_STOP_AT_V(S, k, v) ((void)0)
#define PG_STOPPED_V(S, v) ((void)0)
#endif
#ifdef PG_DIAG
    u32 dg;              // diagnostics build: fills | seed-filter runs << 8 | candidate passes << 16 | evaluations << 24
#define PG_DG(S, sh) ((S).dg += 1u << (sh))
#else
#define PG_DG(S, sh) ((void)0)
#endif
    int cap_state;       // state-dependent relevance bound for seeds (see evaluate); T - 1 = none
    // bits of one word (a wave-uniform bool costs two scalar registers):
    //   SF_WANT_CAP   more window chunks will be filtered after the next evaluation: keep cap_state up to date
    //   SF_TIER_A     the short-lived tier is usable: bps + 16 <= 32 and CheckMismatches' "L > m" test cannot fail
    //   SF_LEN_CHECK  Min_Perfect_Match_Around_BP >= bps: the "L > m" test of CheckMismatches can fail
    u32 sf;
    __device__ __forceinline__ bool want_cap() const { return (sf & 1u) != 0u; }
    __device__ __forceinline__ bool tierA() const { return (sf & 2u) != 0u; }
    __device__ __forceinline__ bool len_check() const { return (sf & 4u) != 0u; }
};

// Running reduction of a search.  Tier B: lane owns L = bps + 64 r + lane in round r; round 0 lives in
// registers, the rounds >= 1 (touched only by candidates that match 64+ bases) in LDS (Lds::accB), `dirty`
// says which of them hold anything.  Tier A: lane = 16 q + j owns L = bps + j; quarter q holds a partial
// reduction (registers).
template <int NB, typename Id>
struct Acc {
    u32 m1, m2, ok;
    Id id;
    u32 a1, a2, aok;
    Id aid;
    u32 dirty;
    __device__ __forceinline__ void reset()
    {
        m1 = m2 = PG_BIG; ok = 0u; id = 0;
        a1 = a2 = PG_BIG; aok = 0u; aid = 0;
        dirty = 0u;
    }
};

// fold one candidate (level k, id cid, CheckMismatches result okc) into (min1, min2, argmin)
template <typename Id>
__device__ __forceinline__ void fold(u32 &m1, u32 &m2, Id &id, u32 &ok, u32 k, Id cid, u32 okc)
{
    const bool lt = k < m1;
    m2 = med3(m1, m2, k);           // second smallest of {m1 <= m2, k}
    id = lt ? cid : id;
    ok = lt ? okc : ok;
    m1 = k < m1 ? k : m1;
}
// merge two partial reductions (a <- a + b)
template <typename Id>
__device__ __forceinline__ void merge(u32 &a1, u32 &a2, Id &aid, u32 &aok, u32 b1, u32 b2, Id bid, u32 bok)
{
    const bool lt = b1 < a1;
    const u32 hi1 = a1 > b1 ? a1 : b1, lo2 = a2 < b2 ? a2 : b2;
    a2 = hi1 < lo2 ? hi1 : lo2;
    aid = lt ? bid : aid;
    aok = lt ? bok : aok;
    a1 = lt ? b1 : a1;
}

// g_maxMismatch[L] from its breakpoints (the table is monotone): #{k : L >= mm_bp[k]}
__device__ __forceinline__ int max_mismatch_at(const u32 *mm_bp, int L)
{
    int m = 0;
#pragma unroll
    for (int k = 0; k < PG_MM_BREAKS; k++) m += (u32)L >= mm_bp[k] ? 1 : 0;
    return m;
}

// word offset / size of a chromosome: from LDS for the first PG_CHR_TAB ones (c is wave-uniform)
template <int NB>
__device__ __forceinline__ long long chr_word_off_of(const PgDevRef &ref, const Search &S, int c)
{
    if (c < PG_CHR_TAB_N(NB)) {
        const uint2 e = S.chr_tab[c];
        if (uni((int)e.y) != 0) return (long long)(u64)(u32)uni((int)e.x);
    }
    const u64 w = KA(ref, chr_word_off)[c];
    return (long long)((u64)(u32)uni((int)(u32)w) | ((u64)(u32)uni((int)(u32)(w >> 32)) << 32));
}
template <int NB>
__device__ __forceinline__ int chr_size_of(const PgDevRef &ref, const Search &S, int c)
{
    if (c < PG_CHR_TAB_N(NB)) {
        const int sz = uni((int)S.chr_tab[c].y);
        if (sz != 0) return sz;
    }
    return uni((int)KA(ref, chr_size)[c]);
}

// ---------------------------------------------------------------------------------
// Mismatch word of one 64-base block, and the read's N word.  Matches(): read N matches any ACGT; reference N
// matches nothing (searcher.cpp:36-44).  The exact character inequality CheckMismatches' perfect-match window
// needs (BP_On_Read != BP_On_Ref, searcher.cpp:349-364) differs from it exactly where the read has an N
// (N vs ACGT: match but unequal; N vs N: mismatch but equal): sne = mis ^ qnn.
template <int NB>
__device__ __forceinline__ void block_masks(const Query<NB> &Q, int b, bool comp,
                                            u64 rlo, u64 rhi, u64 rnn, u64 &mis, u64 &qn)
{
    const u64 qlo = q_lo<NB>(Q, b), qhi = q_hi<NB>(Q, b), qnn = q_nn<NB>(Q, b), qoo = q_oo<NB>(Q, b);
    u64 cm = comp ? ~0ull : 0ull;
    u64 x = rlo ^ qlo ^ cm;
    u64 y = rhi ^ qhi ^ cm;
    u64 d = x | y;
    mis = (d & ~qnn) | rnn | qoo;
    qn = qnn;
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
// ... when the kind is wave-uniform (close end): a branch instead of six reversals + six selects per block
__device__ __forceinline__ void fetch_lds_uniform(const uint4 *win, int wbase, int q, bool rev,
                                                  u64 &rlo, u64 &rhi, u64 &rnn)
{
    u32 rel = (u32)(q - wbase);
    u32 wi = rel >> 5, s = rel & 31u;
    uint4 w0 = win[wi], w1 = win[wi + 1], w2 = win[wi + 2];
    rlo = funnel64(w0.x, w1.x, w2.x, s);
    rhi = funnel64(w0.y, w1.y, w2.y, s);
    rnn = funnel64(w0.z, w1.z, w2.z, s);
    if (uni((int)rev)) {
        asm volatile("" : "+v"(rlo), "+v"(rhi), "+v"(rnn));       // (keeps the compiler from turning the branch into selects)
        rlo = __brevll(rlo); rhi = __brevll(rhi); rnn = __brevll(rnn);
    }
}

// ---------------------------------------------------------------------------------
// One pass over n (<= 64) queued candidates.
//  1. lanes = candidates: mismatch / inequality bitmaps of the read placed at the candidate, Hamming count;
//     short-lived candidates (dead before L = bps + 16) write a 16-byte entry to bufA, the others a full
//     entry to bufB.
//  2. tier A, lanes = 4 candidates x 16 lengths: fold the short-lived ones.
//  3. tier B, lanes = 64 lengths of a round: fold the long-lived ones that are still alive when the round
//     starts.
// MIXED = both candidate kinds in one search (far end); otherwise the kind is wave-uniform (close end)
// and the per-lane selects / bit reversals disappear.
// The nested far-end ranges (128, 512, 2048 positions around the close end) that lie in one LDS chunk go through ONE
// candidate pass: `on` = the candidates are kept apart by ring (ring 0 = [s0, e0), ring 1 = [s1, e1) minus ring 0,
// ring 2 = the rest).  Tier A: quarter 0 folds ring 0, quarter 1 ring 1, quarters 2 and 3 share ring 2, so the state of
// range r is the merge of the quarters up to r (evaluate's qmask).  Tier B: the masks of the long-lived candidates are
// left in Search::ringB per ring and round; the caller folds them ring by ring (fold_tier_b) before it evaluates a range.
struct Rings {
    bool on;
    int s0, e0, s1, e1;
};

// tier B: the long-lived candidates `longm[r]` (entries in bufB / hdrB) into round r of the reduction, lanes = lengths
template <int NB, typename Id>
__device__ __forceinline__ void fold_tier_b(const Search &S, const Query<NB> &Q, Acc<NB, Id> &A, const u64 *longm, int lane)
{
    constexpr int EW = NB;
    {
        u64 any = 0ull;
#pragma unroll
        for (int r = 0; r < NB; r++) any |= longm[r];
        if (any == 0ull) return;                          // uniform
    }
    // The per-lane masks of round r and block b depend on b - r only (L - 64 b = bps + lane + 64 (r - b)): five masks serve
    // every round.  W = the bases consumed at length L, P = those before CheckMismatches' window of min_perfect bases.
    const int lB = opaque(lane);
    const int n0 = S.bps + lB, p0 = n0 - S.min_perfect;
    const u64 W0 = low_bits_lane(n0), W1 = low_bits_lane(n0 - 64);               // b = r, b = r + 1  (b < r: every base)
    const u64 BP0 = W0 & ~low_bits_lane(p0), BP1 = W1 & ~low_bits_lane(p0 - 64), BPm = ~low_bits_lane(p0 + 64);
#pragma unroll
    for (int r = 0; r < NB; r++) {
        const int L0 = S.bps + 64 * r;
        if (L0 > S.len - 1) break;                        // uniform
        u64 mask = longm[r];
        if (mask == 0ull) continue;                       // uniform
        const int L = L0 + lB;
        u64 Mk[NB], BP[NB], QN[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            Mk[b] = b == r ? W0 : (b == r + 1 ? W1 : ~0ull);                      // (b > r + 1 is not read)
            BP[b] = b == r ? BP0 : (b == r + 1 ? BP1 : BPm);                      // (b < r - 1 is not read)
            QN[b] = q_nn<NB>(Q, b);
        }
        u32 m1 = A.m1, m2 = A.m2, ok = A.ok;
        Id wid = A.id;
        if (r > 0) {
            m1 = m2 = PG_BIG; ok = 0u; wid = 0;
            if ((A.dirty >> r) & 1u) {
                AccB<Id>::load(S.accB, (r - 1) * 64 + lB, m1, m2, ok, wid);
            }
        }
        while (mask != 0ull) {
            const int i = __ffsll((long long)mask) - 1;
            mask &= mask - 1ull;
            const uint2 *e = S.bufB + i * EW;
            const uint2 h = S.hdrB[i];
            u32 k = 0u;
            u64 bad = 0ull;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if (b > r + 1) continue;                  // L <= bps + 64 r + 63 < 64 (r + 2)
                const uint2 w = e[b];
                const u64 m = (u64)w.x | ((u64)w.y << 32);
                k += (u32)__popcll(b < r ? m : (m & Mk[b]));
                // exact inequality = mismatch with the read's N bits flipped (block_masks); the window lies inside the read
                if (b + 1 >= r) bad |= (m ^ QN[b]) & BP[b];        // L - m >= 64 r - 64 (m <= 64 <= 64 + bps)
            }
            u32 okc = bad == 0ull ? (h.y >> 31) : 0u;
            if (S.len_check()) okc = (u32)L >= ((h.y >> 24) & 0x7fu) ? okc : 0u;   // uniform branch
            const Id cid = sizeof(Id) == 8 ? (Id)((u64)h.x | ((u64)(h.y & 0xffffffu) << 32)) : (Id)h.x;
            fold<Id>(m1, m2, wid, ok, k, cid, okc);
        }
        if (r == 0) { A.m1 = m1; A.m2 = m2; A.ok = ok; A.id = wid; }
        else {
            AccB<Id>::store(S.accB, (r - 1) * 64 + lB, m1, m2, ok, wid);
            A.dirty |= 1u << r;
        }
    }
}

// ring_n (rings only): queued candidates per ring
template <int NB, typename Id, bool MIXED>
__device__ __forceinline__ void fold_candidates(const Search &S, const Query<NB> &Q, Acc<NB, Id> &A, int wbase,
                                                int origin, u32 region, int n, int lane, const Rings &R, int *ring_n)
{
    constexpr int EW = NB;                                // 64-base blocks per tier B entry
    PG_DG(const_cast<Search &>(S), 16);
    // (conditions as lane masks in scalar registers: see ballot64)
    const u64 inm = low_bits(n);                          // lanes with a candidate
    int p = 0;
    bool isB = MIXED ? false : Q.allowB();
    if (lane_bit(inm)) {
        u32 e = S.queue[lane];
        if (MIXED) isB = e & 1u;
        p = wbase + (int)(e >> 1);
    }
    const bool comp = isB ? Q.cB() : Q.cF();
    // Per 64-base block: the mismatch word goes straight into the candidate's tier B entry (LDS);
    // only popcounts stay in registers.  Words of blocks that are not computed keep stale bits: they can
    // only add mismatches to a candidate that is dead there anyway.
    int cum = 0, lvl0 = 0, kA = 0;
    int kk[NB];
    u32 m0lo = 0u, s0lo = 0u;
#pragma unroll
    for (int r = 0; r < NB; r++) kk[r] = 0;
    u64 needm = inm;
    const int alive = S.T > S.thr ? S.T : S.thr;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        if (64 * b >= S.len) break;                      // uniform
        if (needm == 0ull) break;
        if (lane_bit(needm)) {
            u64 rlo, rhi, rnn, m, s;
            int q = isB ? p - 64 * b - 63 : p + 64 * b;
            if (MIXED) fetch_lds(S.win, wbase, q, isB, rlo, rhi, rnn);
            else fetch_lds_uniform(S.win, wbase, q, isB, rlo, rhi, rnn);
            block_masks<NB>(Q, b, comp, rlo, rhi, rnn, m, s);
            const u64 lm = low_bits(S.len - 64 * b);
            m &= lm;
            s = (m ^ s) & lm;                             // exact inequality (the read's N bits flipped)
            S.bufB[lane * EW + b] = make_uint2((u32)m, (u32)(m >> 32));
            cum += __popcll(m);
            if (b == 0) {
                lvl0 = __popcll(m & low_bits(S.bps));
                kA = __popc((u32)m & low32(S.bps + 16));
                m0lo = (u32)m;
                s0lo = (u32)s;
            }
#pragma unroll
            for (int r = 1; r < NB; r++)
                if (b <= r) kk[r] += __popcll(m & low_bits(S.bps + 64 * r - 64 * b));
        }
        // later blocks matter only while the candidate is alive (fewer than T mismatches so far) or its
        // whole-read Hamming count has not reached CheckMismatches' threshold yet
        needm &= ballot64(cum < alive);
    }
    const u32 hamok = cum >= S.thr ? 0x80000000u : 0u;
    const u64 validm = inm & ballot64(lvl0 < S.T);         // dead before the first length: never counts
    u64 lngm = validm;
    if (S.tierA()) lngm &= ballot64(kA < S.T);
    const bool lng = lane_bit(lngm);
    const Id id = make_id<Id>((u32)(p - origin), isB, region);
    const u32 lenthr = (u32)(S.min_perfect + (isB ? 0 : 1));           // FORWARD: L > m, BACKWARD: L >= m
    const u32 meta = (u32)((u64)id >> 32) | (lenthr << 24) | hamok;     // id bits 32..55 | CheckMismatches' length bound | Hamming verdict
    const u64 shortm = validm & ~lngm;
    const bool sht = lane_bit(shortm);
    u64 longm[NB];
    // tier A lists: the short-lived candidates sit in bufA sorted by ring (one list without rings); a quarter walks its
    // list from qb in steps of qs up to qe
    const int nA = __popcll(shortm);
    int qb, qs, qe, n_iter;
    {
        const int qd = opaque(lane) >> 4;
        if (MIXED && R.on) {
            // (the rings are nested intervals around one centre: ring 0 inside ring 1)
            const u32 w0 = R.e0 > R.s0 ? (u32)(R.e0 - R.s0) : 0u, w1 = R.e1 > R.s1 ? (u32)(R.e1 - R.s1) : 0u;
            const u64 in0 = inm & ballot64((u32)(p - R.s0) < w0);
            const u64 in1 = inm & ~in0 & ballot64((u32)(p - R.s1) < w1);
            const int ring = lane_bit(in0) ? 0 : (lane_bit(in1) ? 1 : 2);
            ring_n[0] = __popcll(in0);
            ring_n[1] = __popcll(in1);
            ring_n[2] = n - ring_n[0] - ring_n[1];
            const u64 sh0 = shortm & in0, sh1 = shortm & in1, sh2 = shortm & ~(in0 | in1);
            const int n0 = __popcll(sh0), n1 = __popcll(sh1), n2 = nA - n0 - n1;
            if (sht) {
                const u64 mine = ring == 0 ? sh0 : (ring == 1 ? sh1 : sh2);
                const int rank = count_below(mine) + (ring == 0 ? 0 : (ring == 1 ? n0 : n0 + n1));
                S.bufA[rank] = make_uint4(m0lo, s0lo, (u32)id, meta);
            }
            qb = qd == 0 ? 0 : (qd == 1 ? n0 : n0 + n1 + (qd - 2));
            qs = qd < 2 ? 1 : 2;
            qe = qd == 0 ? n0 : (qd == 1 ? n0 + n1 : nA);
            const int h2 = (n2 + 1) >> 1;
            n_iter = n0 > n1 ? n0 : n1;
            n_iter = n_iter > h2 ? n_iter : h2;
            const u64 inl[3] = { in0, in1, ~(in0 | in1) };
#pragma unroll
            for (int x = 0; x < 3; x++) {
                const u64 l0 = lngm & inl[x];
                if (lane == 0) S.ringB[x * NB] = l0;
#pragma unroll
                for (int r = 1; r < NB; r++) {
                    const u64 lr = lngm & ballot64(kk[r] < S.T) & inl[x];
                    if (lane == 0) S.ringB[x * NB + r] = lr;
                }
            }
        } else {
            if (sht) {
                const int rank = count_below(shortm);
                S.bufA[rank] = make_uint4(m0lo, s0lo, (u32)id, meta);
            }
            qb = qd;
            qs = 4;
            qe = nA;
            n_iter = (nA + 3) >> 2;
            longm[0] = lngm;
#pragma unroll
            for (int r = 1; r < NB; r++) longm[r] = lngm & ballot64(kk[r] < S.T);
        }
    }
    if (lng) S.hdrB[lane] = make_uint2((u32)id, meta);
    PG_STOP_AT(const_cast<Search &>(S), (MIXED && R.on) ? 28 : 17);
    PG_SYNC();
    // ---- tier A
    if (nA > 0) {
        const int lA = opaque(lane);
        const int L = S.bps + (lA & 15);
        const u32 mk = low32_lane(L);
        const u32 bpm = mk & ~low32_lane(L - S.min_perfect);               // bits [L - m, L)
        int idx = qb;
        for (int it = 0; it < n_iter; it++, idx += qs) {
            const uint4 e = S.bufA[idx < 67 ? idx : 67];                // (an index past the quarter's list is not used)
            u32 k = (u32)__popc(e.x & mk);
            k = idx < qe ? k : PG_BIG;
            const u32 okc = (e.y & bpm) == 0u ? (e.w >> 31) : 0u;    // (no "L > m" test: tier A is off when it can fail)
            const Id cid = sizeof(Id) == 8 ? (Id)((u64)e.z | ((u64)(e.w & 0xffffffu) << 32)) : (Id)e.z;
            fold<Id>(A.a1, A.a2, A.aid, A.aok, k, cid, okc);
        }
    }
    // ---- tier B (with rings: left to the caller, ring by ring)
    if (!(MIXED && R.on)) fold_tier_b<NB, Id>(S, Q, A, longm, lane);
    PG_SYNC();
    PG_STOP_AT(const_cast<Search &>(S), (MIXED && R.on) ? 29 : 18);
}

// Stages bases [lo, hi) (hi - lo <= PG_CHUNK + 128 NB) of a chromosome into LDS: word i of the window
// holds bases [lo + 32 i, lo + 32 i + 32) whatever the alignment of lo (funnel shift of two HBM words), as
// the code planes (lo, hi, N) that both the candidate pass and the seed filter read.
// next_rec (first fill of a read only): the NEXT read's record is touched with a one-dword scalar load issued beside the window's
// HBM loads and waited for with them -- its 64-byte line is then in the scalar cache (or at least in L2) when the next read
// starts with it.  (Anywhere else an outstanding scalar load would stall the next LDS wait: lgkmcnt counts both.)
template <int NB, bool WIDE = false>
__device__ __forceinline__ void stage_window(const PgDevRef &ref, Search &S, long long wo, int lo, int hi, int lane,
                                             const PgInRec *next_rec = nullptr)
{
    const int nw = ((hi - lo + 31) >> 5) + 2;
    const u32 sh = (u32)(lo & 31);
    PG_DG(S, 0);
    PG_SYNC();
    {
        const long long g0 = wo + (long long)(lo >> 5);     // arithmetic shift = floor
        const u32 *glo, *ghi, *gnn;
        karg_load3<(int)offsetof(PgKArgs, ref.lo)>(glo, ghi, gnn);   // (one wait for the three plane pointers, not three)
        glo += g0; ghi += g0; gnn += g0;
        // two window words per lane and pass, all six loads in flight before the first is used: a 2048-base window with its
        // overhangs is 74-78 words, and 64 + 10 in two dependent passes was two HBM round trips.  (Hand-written: the compiler
        // sinks the second set of loads behind the first set's wait however the source is arranged.)
#ifndef PG_NO_WIDE3
        if (WIDE && nw > 2 * WAVE) {
            // TWO chunks of a wide far-end window with their overhangs are 134-142 words: 128 + 10 in two dependent passes was two HBM
            // round trips per fill, twelve fills per read at -x 5.  Three words per lane, all nine loads in flight together.
            const int i = lane, j = lane + WAVE, k = lane + 2 * WAVE;
            const bool three = k < nw;
            const u32 oi = 4u * (u32)i, oj = 4u * (u32)j, ok = 4u * (u32)(three ? k : i);
            u64 a, b, c, d, e, f, g, h, m;
            asm volatile("global_load_dwordx2 %0, %9, %12\n\tglobal_load_dwordx2 %1, %9, %13\n\tglobal_load_dwordx2 %2, %9, %14\n\t"
                         "global_load_dwordx2 %3, %10, %12\n\tglobal_load_dwordx2 %4, %10, %13\n\tglobal_load_dwordx2 %5, %10, %14\n\t"
                         "global_load_dwordx2 %6, %11, %12\n\tglobal_load_dwordx2 %7, %11, %13\n\tglobal_load_dwordx2 %8, %11, %14\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h), "=&v"(m)
                         : "v"(oi), "v"(oj), "v"(ok), "s"(glo), "s"(ghi), "s"(gnn) : "memory");
            *(uint3 *)__builtin_assume_aligned(&S.win[i], 16) =
                make_uint3(__builtin_amdgcn_alignbit((u32)(a >> 32), (u32)a, sh), __builtin_amdgcn_alignbit((u32)(b >> 32), (u32)b, sh),
                           __builtin_amdgcn_alignbit((u32)(c >> 32), (u32)c, sh));
            *(uint3 *)__builtin_assume_aligned(&S.win[j], 16) =
                make_uint3(__builtin_amdgcn_alignbit((u32)(d >> 32), (u32)d, sh), __builtin_amdgcn_alignbit((u32)(e >> 32), (u32)e, sh),
                           __builtin_amdgcn_alignbit((u32)(f >> 32), (u32)f, sh));
            if (three)
                *(uint3 *)__builtin_assume_aligned(&S.win[k], 16) =
                    make_uint3(__builtin_amdgcn_alignbit((u32)(g >> 32), (u32)g, sh), __builtin_amdgcn_alignbit((u32)(h >> 32), (u32)h, sh),
                               __builtin_amdgcn_alignbit((u32)(m >> 32), (u32)m, sh));
        } else
#endif
        for (int i = lane; i < nw; i += 2 * WAVE) {
            const int j = i + WAVE;
            const bool two = j < nw;
            const u32 oi = 4u * (u32)i, oj = 4u * (u32)(two ? j : i);
            u64 a, b, c, d, e, f;
            if (next_rec) {
                u32 touched;
                asm volatile("global_load_dwordx2 %[a], %[oi], %[glo]\n\tglobal_load_dwordx2 %[b], %[oi], %[ghi]\n\tglobal_load_dwordx2 %[c], %[oi], %[gnn]\n\t"
                             "global_load_dwordx2 %[d], %[oj], %[glo]\n\tglobal_load_dwordx2 %[e], %[oj], %[ghi]\n\tglobal_load_dwordx2 %[f], %[oj], %[gnn]\n\t"
                             "s_load_dword %[t], %[nr], 0x0\n\t"
                             "s_waitcnt vmcnt(0) lgkmcnt(0)"
                             : [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [d] "=&v"(d), [e] "=&v"(e), [f] "=&v"(f), [t] "=&s"(touched)
                             : [oi] "v"(oi), [oj] "v"(oj), [glo] "s"(glo), [ghi] "s"(ghi), [gnn] "s"(gnn), [nr] "s"(next_rec) : "memory");
            } else
            asm volatile("global_load_dwordx2 %0, %6, %8\n\tglobal_load_dwordx2 %1, %6, %9\n\tglobal_load_dwordx2 %2, %6, %10\n\t"
                         "global_load_dwordx2 %3, %7, %8\n\tglobal_load_dwordx2 %4, %7, %9\n\tglobal_load_dwordx2 %5, %7, %10\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f)
                         : "v"(oi), "v"(oj), "s"(glo), "s"(ghi), "s"(gnn) : "memory");
            *(uint3 *)__builtin_assume_aligned(&S.win[i], 16) =
                make_uint3(__builtin_amdgcn_alignbit((u32)(a >> 32), (u32)a, sh), __builtin_amdgcn_alignbit((u32)(b >> 32), (u32)b, sh),
                           __builtin_amdgcn_alignbit((u32)(c >> 32), (u32)c, sh));          // (the fourth dword holds the table above)
            if (two)
                *(uint3 *)__builtin_assume_aligned(&S.win[j], 16) =
                    make_uint3(__builtin_amdgcn_alignbit((u32)(d >> 32), (u32)d, sh), __builtin_amdgcn_alignbit((u32)(e >> 32), (u32)e, sh),
                               __builtin_amdgcn_alignbit((u32)(f >> 32), (u32)f, sh));
        }
    }
    PG_SYNC();
    S.win_wo = wo;
    S.wbase = lo;
    S.win_lo = lo;
    S.win_hi = hi;
}

__device__ __forceinline__ u32 bits32(int lo, int hi)      // bits [lo,hi), clamped to [0,32]
{
    return low32(hi) & ~low32(lo);
}

// (consumed bases the seed filter inspects: pg_seed_depth, pg_device.h -- the pack kernel puts them into the read's record)

// positions whose bit-sliced mismatch count (c[NS-1] .. c0, ov = overflowed) is <= thr (wave-uniform, >= 0).
// "count <= constant" is a fixed boolean function of the slices: the low three slices against thr & 7 are ONE v_bitop3
// whose truth table is picked by a scalar switch, the upper slice(s) a second one, the overflow bit a third --
// v_bitop3 on VGPR operands issues at the fast VALU rate (scripts/ubench_issue.hip), the compare-free generic form
// (selects on scalar conditions, v_or3) at the slow one.
__host__ __device__ constexpr u32 le3_table(int t)          // truth table of [4 c + 2 b + a <= t] for v_bitop3(a, b, c)
{
    u32 tt = 0u;
    for (int i = 0; i < 8; i++) {
        const int v = 4 * (i & 1) + 2 * ((i >> 1) & 1) + ((i >> 2) & 1);
        if (v <= t) tt |= 1u << i;
    }
    return tt;
}
#define PG_BO3(a, b, c, tt) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt))
// (two counters against the same threshold share the scalar switch: TWO = the far end's second kind)
template <int NS, bool TWO>
__device__ __forceinline__ void count_le(const u32 *c, u32 ov, const u32 *d, u32 ovd, int thr, u32 &rc, u32 &rd)
{
    thr = uni(thr);
    rd = 0u;
    if (thr < 0) { rc = 0u; return; }
    if (thr >= (1 << NS) - 1) { rc = ~ov; if (TWO) rd = ~ovd; return; }
    u32 g, h = 0u;
#define PG_LE3(t) g = PG_BO3(c[0], c[1], c[2], le3_table(t)); if (TWO) h = PG_BO3(d[0], d[1], d[2], le3_table(t)); break;
    switch (thr & 7) {
    case 0: PG_LE3(0)
    case 1: PG_LE3(1)
    case 2: PG_LE3(2)
    case 3: PG_LE3(3)
    case 4: PG_LE3(4)
    case 5: PG_LE3(5)
    case 6: PG_LE3(6)
    default: g = h = ~0u; break;
    }
#undef PG_LE3
    if (NS == 3) {                                                 // g & ~ov
        rc = PG_BO3(g, ov, ov, 0x30);
        if (TWO) rd = PG_BO3(h, ovd, ovd, 0x30);
    } else if (NS == 4) {
        if (thr >> 3) {                                            // (g | ~c3) & ~ov
            rc = PG_BO3(g, c[3], ov, 0x51);
            if (TWO) rd = PG_BO3(h, d[3], ovd, 0x51);
        } else {                                                   // g & ~c3 & ~ov
            rc = PG_BO3(g, c[3], ov, 0x10);
            if (TWO) rd = PG_BO3(h, d[3], ovd, 0x10);
        }
    } else {
        u32 u, v = 0u;
#define PG_LE5(tt) u = PG_BO3(g, c[3], c[4], tt); if (TWO) v = PG_BO3(h, d[3], d[4], tt); break;
        switch (thr >> 3) {                                        // (c4 c3) against thr >> 3, g breaks the tie
        case 0: PG_LE5(0x10)                                       // ~c4 & ~c3 & g
        case 1: PG_LE5(0x51)                                       // ~c4 & (~c3 | g)
        case 2: PG_LE5(0x75)                                       // ~c4 | (~c3 & g)
        default: PG_LE5(0xf7)                                      // ~c4 | ~c3 | g
        }
#undef PG_LE5
        rc = PG_BO3(u, ov, ov, 0x30);
        if (TWO) rd = PG_BO3(v, ovd, ovd, 0x30);
    }
}

// Bit-sliced mismatch counter of the seed filter: NS slices (3: up to 8 levels, 4: up to 16, 5: up to 32) + a sticky
// overflow bit per position.  Bases are
// added one, two or three at a time (carry-save: the sum bits of two or three match masks first, then one
// ripple through the slices -- 7, 4.5 and 3.7 VALU instructions per base).  Each add is ONE asm statement that
// also takes the base(s) off the wave-uniform set (s_ff1 / s_bitset0) and shifts the planes (v_alignbit):
// the compiler's version of the ripple rotates the counter through extra v_mov, and it pads every asm
// statement with an s_nop, so fewer, larger statements are cheaper.
#define PG_TAKE(j) "s_ff1_i32_b32 %[" j "], %[pm]\n\ts_bitset0_b32 %[pm], %[" j "]\n\t"
#define PG_SHIFT(m, j) "v_alignbit_b32 %[" m "], %[hi], %[lo], %[" j "]\n\t"
#define PG_MIRROR(t, j) "s_sub_i32 %[" t "], 32, %[" j "]\n\t"      // (writes SCC: the statements using it clobber "scc")
// one base: k0 = ~ma & c0, c0 ^= ~ma, k1 = c1 & k0, c1 ^= k0
#define PG_ADD1 "v_bitop3_b32 %[k0], %[ma], %[c0], %[c0] bitop3:0x0c\n\tv_bitop3_b32 %[c0], %[c0], %[ma], %[ma] bitop3:0xc3\n\t" \
                "v_and_b32 %[k1], %[c1], %[k0]\n\tv_xor_b32 %[c1], %[c1], %[k0]\n\t"
// two bases: sum of the two mismatch bits = low bit ma ^ mb, high bit ~(ma | mb); the high bit and the carry out
// of slice 0 exclude each other, so slice 1 adds t = high | carry
#define PG_ADD2 "v_bitop3_b32 %[k0], %[c0], %[ma], %[mb] bitop3:0x60\n\tv_bitop3_b32 %[c0], %[c0], %[ma], %[mb] bitop3:0x96\n\t" \
                "v_bitop3_b32 %[s0], %[ma], %[mb], %[k0] bitop3:0xab\n\t" \
                "v_and_b32 %[k1], %[c1], %[s0]\n\tv_xor_b32 %[c1], %[c1], %[s0]\n\t"
// three bases: s0 = ~(ma ^ mb ^ mc), s1 = ~maj(ma, mb, mc), then a full adder per slice
#define PG_ADD3 "v_bitop3_b32 %[s0], %[ma], %[mb], %[mc] bitop3:0x69\n\tv_bitop3_b32 %[s1], %[ma], %[mb], %[mc] bitop3:0x17\n\t" \
                "v_and_b32 %[k0], %[c0], %[s0]\n\tv_xor_b32 %[c0], %[c0], %[s0]\n\t" \
                "v_bitop3_b32 %[k1], %[c1], %[s1], %[k0] bitop3:0xe8\n\tv_bitop3_b32 %[c1], %[c1], %[s1], %[k0] bitop3:0x96\n\t"
// the carry out of slice 1 (k1) through the upper slice(s)
#define PG_UP3 "v_bitop3_b32 %[ov], %[c2], %[k1], %[ov] bitop3:0xea\n\tv_xor_b32 %[c2], %[c2], %[k1]"
#define PG_UP4 "v_and_b32 %[k0], %[c2], %[k1]\n\tv_xor_b32 %[c2], %[c2], %[k1]\n\t" \
               "v_bitop3_b32 %[ov], %[c3], %[k0], %[ov] bitop3:0xea\n\tv_xor_b32 %[c3], %[c3], %[k0]"
#define PG_UP5 "v_and_b32 %[k0], %[c2], %[k1]\n\tv_xor_b32 %[c2], %[c2], %[k1]\n\t" \
               "v_and_b32 %[k1], %[c3], %[k0]\n\tv_xor_b32 %[c3], %[c3], %[k0]\n\t" \
               "v_bitop3_b32 %[ov], %[c4], %[k1], %[ov] bitop3:0xea\n\tv_xor_b32 %[c4], %[c4], %[k1]"
template <int NS>
struct Counter {
    u32 c0, c1, c2, c3, c4, ov;
    __device__ __forceinline__ void reset() { c0 = c1 = c2 = c3 = c4 = ov = 0u; }
    // positions with count <= thr in this counter (and, TWO, in `o`)
    template <bool TWO>
    __device__ __forceinline__ void le(const Counter &o, int thr, u32 &mine, u32 &others) const
    {
        const u32 c[5] = { c0, c1, c2, c3, c4 }, d[5] = { o.c0, o.c1, o.c2, o.c3, o.c4 };
        count_le<NS, TWO>(c, ov, d, o.ov, thr, mine, others);
    }
#define PG_CTR3 [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [ov] "+v"(ov)
#define PG_CTR4 [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [ov] "+v"(ov)
#define PG_CTR5 [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [c4] "+v"(c4), [ov] "+v"(ov)
#define PG_PLANES [lo] "v"(lo), [hi] "v"(hi)
    // --- the lowest base(s) of pm (removed from it), planes shifted by the base's bit: returns the bit(s)
    __device__ __forceinline__ void take1(u32 &pm, u32 lo, u32 hi, u32 &ja)
    {
        u32 ma, k0, k1;
        if (NS == 3)
            asm(PG_TAKE("ja") PG_SHIFT("ma", "ja") PG_ADD1 PG_UP3
                : PG_CTR3, [pm] "+s"(pm), [ja] "=&s"(ja), [ma] "=&v"(ma), [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES);
        else if (NS == 4)
            asm(PG_TAKE("ja") PG_SHIFT("ma", "ja") PG_ADD1 PG_UP4
                : PG_CTR4, [pm] "+s"(pm), [ja] "=&s"(ja), [ma] "=&v"(ma), [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES);
        else
            asm(PG_TAKE("ja") PG_SHIFT("ma", "ja") PG_ADD1 PG_UP5
                : PG_CTR5, [pm] "+s"(pm), [ja] "=&s"(ja), [ma] "=&v"(ma), [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES);
    }
    __device__ __forceinline__ void take2(u32 &pm, u32 lo, u32 hi, u32 &ja, u32 &jb)
    {
        u32 ma, mb, s0, k0, k1;
        if (NS == 3)
            asm(PG_TAKE("ja") PG_TAKE("jb") PG_SHIFT("ma", "ja") PG_SHIFT("mb", "jb") PG_ADD2 PG_UP3
                : PG_CTR3, [pm] "+s"(pm), [ja] "=&s"(ja), [jb] "=&s"(jb), [ma] "=&v"(ma), [mb] "=&v"(mb), [s0] "=&v"(s0),
                  [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES);
        else if (NS == 4)
            asm(PG_TAKE("ja") PG_TAKE("jb") PG_SHIFT("ma", "ja") PG_SHIFT("mb", "jb") PG_ADD2 PG_UP4
                : PG_CTR4, [pm] "+s"(pm), [ja] "=&s"(ja), [jb] "=&s"(jb), [ma] "=&v"(ma), [mb] "=&v"(mb), [s0] "=&v"(s0),
                  [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES);
        else
            asm(PG_TAKE("ja") PG_TAKE("jb") PG_SHIFT("ma", "ja") PG_SHIFT("mb", "jb") PG_ADD2 PG_UP5
                : PG_CTR5, [pm] "+s"(pm), [ja] "=&s"(ja), [jb] "=&s"(jb), [ma] "=&v"(ma), [mb] "=&v"(mb), [s0] "=&v"(s0),
                  [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES);
    }
    __device__ __forceinline__ void take3(u32 &pm, u32 lo, u32 hi, u32 &ja, u32 &jb, u32 &jc)
    {
        u32 ma, mb, mc, s0, s1, k0, k1;
        if (NS == 3)
            asm(PG_TAKE("ja") PG_TAKE("jb") PG_TAKE("jc") PG_SHIFT("ma", "ja") PG_SHIFT("mb", "jb") PG_SHIFT("mc", "jc")
                PG_ADD3 PG_UP3
                : PG_CTR3, [pm] "+s"(pm), [ja] "=&s"(ja), [jb] "=&s"(jb), [jc] "=&s"(jc), [ma] "=&v"(ma), [mb] "=&v"(mb),
                  [mc] "=&v"(mc), [s0] "=&v"(s0), [s1] "=&v"(s1), [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES);
        else if (NS == 4)
            asm(PG_TAKE("ja") PG_TAKE("jb") PG_TAKE("jc") PG_SHIFT("ma", "ja") PG_SHIFT("mb", "jb") PG_SHIFT("mc", "jc")
                PG_ADD3 PG_UP4
                : PG_CTR4, [pm] "+s"(pm), [ja] "=&s"(ja), [jb] "=&s"(jb), [jc] "=&s"(jc), [ma] "=&v"(ma), [mb] "=&v"(mb),
                  [mc] "=&v"(mc), [s0] "=&v"(s0), [s1] "=&v"(s1), [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES);
        else
            asm(PG_TAKE("ja") PG_TAKE("jb") PG_TAKE("jc") PG_SHIFT("ma", "ja") PG_SHIFT("mb", "jb") PG_SHIFT("mc", "jc")
                PG_ADD3 PG_UP5
                : PG_CTR5, [pm] "+s"(pm), [ja] "=&s"(ja), [jb] "=&s"(jb), [jc] "=&s"(jc), [ma] "=&v"(ma), [mb] "=&v"(mb),
                  [mc] "=&v"(mc), [s0] "=&v"(s0), [s1] "=&v"(s1), [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES);
    }
    // --- the same base(s) for the other kind: planes shifted by 32 - bit
    __device__ __forceinline__ void mirror1(u32 lo, u32 hi, u32 ja)
    {
        u32 ta, ma, k0, k1;
        if (NS == 3)
            asm(PG_MIRROR("ta", "ja") PG_SHIFT("ma", "ta") PG_ADD1 PG_UP3
                : PG_CTR3, [ta] "=&s"(ta), [ma] "=&v"(ma), [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES, [ja] "s"(ja) : "scc");
        else if (NS == 4)
            asm(PG_MIRROR("ta", "ja") PG_SHIFT("ma", "ta") PG_ADD1 PG_UP4
                : PG_CTR4, [ta] "=&s"(ta), [ma] "=&v"(ma), [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES, [ja] "s"(ja) : "scc");
        else
            asm(PG_MIRROR("ta", "ja") PG_SHIFT("ma", "ta") PG_ADD1 PG_UP5
                : PG_CTR5, [ta] "=&s"(ta), [ma] "=&v"(ma), [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES, [ja] "s"(ja) : "scc");
    }
    __device__ __forceinline__ void mirror2(u32 lo, u32 hi, u32 ja, u32 jb)
    {
        u32 ta, tb, ma, mb, s0, k0, k1;
        if (NS == 3)
            asm(PG_MIRROR("ta", "ja") PG_MIRROR("tb", "jb") PG_SHIFT("ma", "ta") PG_SHIFT("mb", "tb") PG_ADD2 PG_UP3
                : PG_CTR3, [ta] "=&s"(ta), [tb] "=&s"(tb), [ma] "=&v"(ma), [mb] "=&v"(mb), [s0] "=&v"(s0), [k0] "=&v"(k0),
                  [k1] "=&v"(k1) : PG_PLANES, [ja] "s"(ja), [jb] "s"(jb) : "scc");
        else if (NS == 4)
            asm(PG_MIRROR("ta", "ja") PG_MIRROR("tb", "jb") PG_SHIFT("ma", "ta") PG_SHIFT("mb", "tb") PG_ADD2 PG_UP4
                : PG_CTR4, [ta] "=&s"(ta), [tb] "=&s"(tb), [ma] "=&v"(ma), [mb] "=&v"(mb), [s0] "=&v"(s0), [k0] "=&v"(k0),
                  [k1] "=&v"(k1) : PG_PLANES, [ja] "s"(ja), [jb] "s"(jb) : "scc");
        else
            asm(PG_MIRROR("ta", "ja") PG_MIRROR("tb", "jb") PG_SHIFT("ma", "ta") PG_SHIFT("mb", "tb") PG_ADD2 PG_UP5
                : PG_CTR5, [ta] "=&s"(ta), [tb] "=&s"(tb), [ma] "=&v"(ma), [mb] "=&v"(mb), [s0] "=&v"(s0), [k0] "=&v"(k0),
                  [k1] "=&v"(k1) : PG_PLANES, [ja] "s"(ja), [jb] "s"(jb) : "scc");
    }
    __device__ __forceinline__ void mirror3(u32 lo, u32 hi, u32 ja, u32 jb, u32 jc)
    {
        u32 ta, tb, tc, ma, mb, mc, s0, s1, k0, k1;
        if (NS == 3)
            asm(PG_MIRROR("ta", "ja") PG_MIRROR("tb", "jb") PG_MIRROR("tc", "jc") PG_SHIFT("ma", "ta") PG_SHIFT("mb", "tb")
                PG_SHIFT("mc", "tc") PG_ADD3 PG_UP3
                : PG_CTR3, [ta] "=&s"(ta), [tb] "=&s"(tb), [tc] "=&s"(tc), [ma] "=&v"(ma), [mb] "=&v"(mb), [mc] "=&v"(mc),
                  [s0] "=&v"(s0), [s1] "=&v"(s1), [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES, [ja] "s"(ja), [jb] "s"(jb), [jc] "s"(jc) : "scc");
        else if (NS == 4)
            asm(PG_MIRROR("ta", "ja") PG_MIRROR("tb", "jb") PG_MIRROR("tc", "jc") PG_SHIFT("ma", "ta") PG_SHIFT("mb", "tb")
                PG_SHIFT("mc", "tc") PG_ADD3 PG_UP4
                : PG_CTR4, [ta] "=&s"(ta), [tb] "=&s"(tb), [tc] "=&s"(tc), [ma] "=&v"(ma), [mb] "=&v"(mb), [mc] "=&v"(mc),
                  [s0] "=&v"(s0), [s1] "=&v"(s1), [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES, [ja] "s"(ja), [jb] "s"(jb), [jc] "s"(jc) : "scc");
        else
            asm(PG_MIRROR("ta", "ja") PG_MIRROR("tb", "jb") PG_MIRROR("tc", "jc") PG_SHIFT("ma", "ta") PG_SHIFT("mb", "tb")
                PG_SHIFT("mc", "tc") PG_ADD3 PG_UP5
                : PG_CTR5, [ta] "=&s"(ta), [tb] "=&s"(tb), [tc] "=&s"(tc), [ma] "=&v"(ma), [mb] "=&v"(mb), [mc] "=&v"(mc),
                  [s0] "=&v"(s0), [s1] "=&v"(s1), [k0] "=&v"(k0), [k1] "=&v"(k1) : PG_PLANES, [ja] "s"(ja), [jb] "s"(jb), [jc] "s"(jc) : "scc");
    }
};

// Every base of the (wave-uniform) set pm goes into the counter(s).  Kind F: the base at bit j reads the pair
// (own word, next word) shifted by j.  Kind B: the pair (previous word, own word) shifted by 32 - j.  DUAL: both
// kinds in one loop (the scalar bookkeeping is shared; the caller passes kind B's planes of the complementary
// symbol).  Single kind B: pm arrives bit-reversed and moved up one bit, so that its bit IS the shift.
// group: three or two bases at a time (not for the rare N bases, to keep the code small).
template <int NS, bool DUAL>
__device__ __forceinline__ void add_bases(bool group, u32 pm, Counter<NS> &C, u32 lo, u32 hi,
                                          Counter<NS> &C2, u32 lo2, u32 hi2)
{
    u32 ja, jb, jc;
    if (group) {
        int n = __popc(pm);
        while (n >= 3) {
            C.take3(pm, lo, hi, ja, jb, jc);
            if (DUAL) C2.mirror3(lo2, hi2, ja, jb, jc);
            n -= 3;
        }
        if (n == 2) {
            C.take2(pm, lo, hi, ja, jb);
            if (DUAL) C2.mirror2(lo2, hi2, ja, jb);
        }
    }
    while (pm != 0u) {
        C.take1(pm, lo, hi, ja);
        if (DUAL) C2.mirror1(lo2, hi2, ja);
    }
}

// DUAL (far end, both kinds wanted, kind B reading the complement of what kind F reads): one pass yields
// both masks (mF, mB).  Otherwise the mask of the one kind (kindB) comes back in mF.
template <int NB, int NS, bool DUAL>
__device__ __forceinline__ void seed_filter_run(const Search &S, const Query<NB> &Q, bool kindB, int lane,
                                                u32 jmask, u32 g0mask, int cap0, u32 &mF, u32 &mB)
{
    // jmask = the inspected bases, bits [1, J); g0mask = those before the first evaluated length, bits [1, min(bps, J))
    u32 lo = (u32)uni((int)(u32)q_lo<NB>(Q, 0)), hi = (u32)uni((int)(u32)q_hi<NB>(Q, 0));
    const u32 nn = (u32)uni((int)(u32)q_nn<NB>(Q, 0)), oo = (u32)uni((int)(u32)q_oo<NB>(Q, 0));
    if (DUAL ? Q.cF() : (kindB ? Q.cB() : Q.cF())) { lo = ~lo; hi = ~hi; }
    const u32 acgt = ~(nn | oo);
    // read symbols A C G T N (in the orientation of the single kind / of kind F); bases [1, jb) first,
    // snapshot, then bases [jb, J)
    const u32 sym[5] = { ~lo & ~hi & acgt, lo & ~hi & acgt, ~lo & hi & acgt, lo & hi & acgt, nn };
    // one-hot planes (is-A, is-C, is-G, is-T, is-not-N) of the lane's window words, from the code planes:
    // w0 = the word of the lane's 32 positions, w1 = the next one (kind F), wm = the previous one (kind B)
    const int a = 2 * NB + lane;
    const bool useF = DUAL || !kindB, useB = DUAL || kindB;
    const uint4 p0 = S.win[a];
    u32 w0[5], w1[5], wm[5];
    w0[0] = ~(p0.x | p0.y | p0.z);
    w0[1] = p0.x & ~(p0.y | p0.z);
    w0[2] = p0.y & ~(p0.x | p0.z);
    w0[3] = p0.x & p0.y & ~p0.z;
    w0[4] = ~p0.z;
#pragma unroll
    for (int X = 0; X < 5; X++) w1[X] = wm[X] = 0u;
    if (useF) {
        const uint4 p1 = S.win[a + 1];
        w1[0] = ~(p1.x | p1.y | p1.z);
        w1[1] = p1.x & ~(p1.y | p1.z);
        w1[2] = p1.y & ~(p1.x | p1.z);
        w1[3] = p1.x & p1.y & ~p1.z;
        w1[4] = ~p1.z;
    }
    if (useB) {
        const uint4 pm1 = S.win[a - 1];
        wm[0] = ~(pm1.x | pm1.y | pm1.z);
        wm[1] = pm1.x & ~(pm1.y | pm1.z);
        wm[2] = pm1.y & ~(pm1.x | pm1.z);
        wm[3] = pm1.x & pm1.y & ~pm1.z;
        wm[4] = ~pm1.z;
    }
    // the seed: the position's own base equals the first read base (first_ok: it is one of ACGT)
    const u32 l0 = (lo & 1u) ? ~0u : 0u, h0 = (hi & 1u) ? ~0u : 0u;
    const u32 seed = ~((p0.x ^ l0) | (p0.y ^ h0) | p0.z);
    const u32 seed2 = ~((p0.x ^ ~l0) | (p0.y ^ ~h0) | p0.z);          // DUAL: kind B reads the complement
    // a read base that is none of ACGTN matches nothing: a constant for every position, taken off the thresholds
    const int o_pre = __popc(oo & g0mask), o_all = __popc(oo & jmask);
    Counter<NS> C, C2;
    C.reset();
    C2.reset();
    u32 snap = 0u, snap2 = 0u;
#pragma unroll
    for (int it = 0; it < 10; it++) {
        const int X = it >= 5 ? it - 5 : it;
        const int X2 = X < 4 ? 3 - X : X;                              // the complementary symbol
        if (it == 5) C.template le<DUAL>(C2, cap0 - o_pre, snap, snap2);
        u32 pm = sym[X] & (it >= 5 ? (jmask & ~g0mask) : g0mask);
        if (DUAL) add_bases<NS, true>(X < 4, pm, C, w0[X], w1[X], C2, wm[X2], w0[X2]);
        else if (kindB) add_bases<NS, false>(X < 4, __brev(pm) << 1, C, wm[X], w0[X], C2, 0u, 0u);
        else add_bases<NS, false>(X < 4, pm, C, w0[X], w1[X], C2, 0u, 0u);
    }
    u32 fin, fin2;
    C.template le<DUAL>(C2, S.T - 1 - o_all, fin, fin2);
    mF = seed & (snap | fin);
    if (DUAL) mB = seed2 & (snap2 | fin2);
}

// ---------------------------------------------------------------------------------
// SEED FILTER IN READ ORDER (round 6).  The formulation above walks the read's bases symbol by symbol and pays for that on the
// scalar unit: ten (symbol, phase) groups per run, each with its popcount, loop tests and branches, two scalar instructions per base
// to take it off the set -- 113 to 130 scalar instructions per run on the pipe that bounds the kernel (profiles/r05/pmc_sq.txt,
// everything_else_breakdown.txt).  Here the bases are taken in READ order, three per carry-save add, and what selects the one-hot
// plane of a base's symbol is the VGPR INDEX MODE of gfx9 (s_set_gpr_idx_on / M0): the five planes (A, C, G, T, not-N) of a window
// word sit in five consecutive VGPRs, the instruction names the first one, the symbol is the index.  The symbols come as a PROGRAM
// worked out once per read by the pack kernel (PgInRec::prog: one dword per group of three bases, each byte 0x30 | symbol), so a
// base costs ONE scalar instruction (s_set_gpr_idx_on takes byte 0; s_bfe_u32 m0 moves bytes 1|2 and 2|3 into M0 -- the upper
// byte's 0x3 nibble lands on M0[15:12] = "index src0 and src1") and one v_alignbit with an IMMEDIATE shift; there is no loop: the
// groups are unrolled, the run leaves after the read's group count.  scripts/ubench_gpridx.hip: the mode works on gfx950, an index
// written by s_set_gpr_idx_* or by s_bfe_u32 m0 is seen by the very next vector instruction.
//   * kind B reads the complement of what kind F reads, from the pair (previous word, own word) shifted by 32 - j: its planes are
//     laid out in COMPLEMENT order (T, G, C, A, not-N), so one index serves both kinds of the far end.
//   * the counter starts at 7 - (T - 1): "count <= T - 1" is then "no overflow" -- the final test costs nothing; the snapshot after
//     PRE groups ("count <= bound", bound <= T - 1) is "adding T - 1 - bound does not overflow": three carries.
//   * depth: bases 1 .. 3 G (G from the record, >= 4); the snapshot after 3 PRE <= bps - 1 bases.  Any depth keeps every seed that
//     can matter (relevant at some L in [bps, 3 G + 1] => count(bps - 1 bases) <= g_maxMismatch[3 G + 1] + ADD,
//     and a count over FEWER bases is smaller still; relevant later => alive after 3 G + 1 bases), it only moves the survivor count.
// ONLY v_alignbit and v_mov run in index mode: an indexed v_bitop3 (the seed's plane straight into the final combine) gave correct
// masks in every single-wave test and memory faults with seven waves per SIMD -- delta-debugged with variants of tests/rofilter_unit.hip:
// the fault follows the index of that one instruction, not the program fetch, not M0[11:8], not the register numbers.
// Everything lives in fixed registers (v32 .. v43 [+ v44, v45 with four slices], v48 .. v57, s84 .. s88): the index needs consecutive
// planes at known numbers (v48 + symbol: the 0x30 bias), and nothing inside is spilled, copied or padded by the compiler.  s88 is the
// highest scalar register on purpose: with VCC, FLAT_SCRATCH and XNACK_MASK that is 95 of the 96 scalar registers a wave may own at
// seven waves per SIMD.  Reads with more than 16 mismatch levels (NS = 5 launches), fewer than four groups, a base outside ACGTN among
// the first 25 of either orientation, or g_MinClose < 8 keep the filter above.
#ifndef PG_NO_RO_FILTER
#define PG_RO 1
// planes of the window word in (x, y, z) = (code bit 0, code bit 1, N): A, C, G, T, not-N into the five registers given
#define RO_ONEHOT(dA, dC, dG, dT, dN, x, y, z)                                                                       \
    "v_bitop3_b32 " dA ", " x ", " y ", " z " bitop3:0x01\n\tv_bitop3_b32 " dC ", " x ", " y ", " z " bitop3:0x10\n\t" \
    "v_bitop3_b32 " dG ", " x ", " y ", " z " bitop3:0x04\n\tv_bitop3_b32 " dT ", " x ", " y ", " z " bitop3:0x40\n\t" \
    "v_not_b32 " dN ", " z "\n\t"
// three match masks into the counter (c0 c1 c2 [c3], sticky overflow ov): the adds of PG_ADD3 / PG_UP3 / PG_UP4; temporaries v35 .. v38
#define RO_ADD3_CORE(c0, c1, ma, mb, mc)                                                                              \
    "v_bitop3_b32 v35, " ma ", " mb ", " mc " bitop3:0x69\n\tv_bitop3_b32 v36, " ma ", " mb ", " mc " bitop3:0x17\n\t"   \
    "v_and_b32 v37, " c0 ", v35\n\tv_xor_b32 " c0 ", " c0 ", v35\n\t"                                                   \
    "v_bitop3_b32 v38, " c1 ", v36, v37 bitop3:0xe8\n\tv_bitop3_b32 " c1 ", " c1 ", v36, v37 bitop3:0x96\n\t"
#define RO_UP3(c2, ov) "v_bitop3_b32 " ov ", " c2 ", v38, " ov " bitop3:0xea\n\tv_xor_b32 " c2 ", " c2 ", v38\n\t"
#define RO_UP4(c2, c3, ov) "v_and_b32 v37, " c2 ", v38\n\tv_xor_b32 " c2 ", " c2 ", v38\n\t"                            \
                           "v_bitop3_b32 " ov ", " c3 ", v37, " ov " bitop3:0xea\n\tv_xor_b32 " c3 ", " c3 ", v37\n\t"
// (the index changes of a group as macros of their own: scripts/test_rofilter.hip builds variants of them)
#ifndef RO_IDX_ON
#define RO_IDX_ON(P) "s_set_gpr_idx_on " P ", 0x3\n\t"
#define RO_IDX_1(P) "s_bfe_u32 m0, " P ", 0x100008\n\t"
#define RO_IDX_2(P) "s_bfe_u32 m0, " P ", 0x100010\n\t"
#define RO_IDX_OFF "s_set_gpr_idx_off\n\t"
#endif
// one group, in two halves: the three match masks (v32 .. v34) of the bases with shifts s1 .. s3 -- low planes v48 .., high planes
// v53 .. (the instruction names register - 48) --, then their sum into the counter v40 .. v42 + sticky overflow v43
#define RO_SH(P, s1, s2, s3)                                                                                           \
    RO_IDX_ON(P) "v_alignbit_b32 v32, v5, v0, " #s1 "\n\t"                                                              \
    RO_IDX_1(P) "v_alignbit_b32 v33, v5, v0, " #s2 "\n\t"                                                               \
    RO_IDX_2(P) "v_alignbit_b32 v34, v5, v0, " #s3 "\n\t" RO_IDX_OFF
// counter: slices v40 v41 v42 (+ v44 when a read may have up to 16 mismatch levels), sticky overflow v43
#define RO_AD3 RO_ADD3_CORE("v40", "v41", "v32", "v33", "v34") RO_UP3("v42", "v43")
#define RO_AD4 RO_ADD3_CORE("v40", "v41", "v32", "v33", "v34") RO_UP4("v42", "v44", "v43")
// The program comes in two halves into the SAME four registers (groups 1 .. 4, then 5 .. 8: the second fetch is issued when group 4's
// index changes are done and is waited for behind that group's add), the seed's index is kept in s88: five scalar registers, the
// highest s88 -- with VCC, FLAT_SCRATCH and XNACK_MASK that is 95 of the 96 a wave may have at seven waves per SIMD (the first
// version held all eight dwords in s84 .. s91 + s92: 99, and ran at six waves per SIMD).
#ifndef RO_PROG_LOAD
#define RO_PROG_LOAD "s_load_dwordx4 s[84:87], %[rp], %[off]\n\t"
#define RO_PROG_LOAD2 "s_load_dwordx4 s[84:87], %[rp], %[off] offset:0x10\n\t"
#endif
// kind F: base j against the pair (own word, next word) shifted by j; kind B: (previous word, own word) shifted by 32 - j
#define RO_SH_F(P, j1, j2, j3, n1, n2, n3) RO_SH(P, j1, j2, j3)
#define RO_SH_B(P, j1, j2, j3, n1, n2, n3) RO_SH(P, n1, n2, n3)
#define RO_G1(M) M("s84", 1, 2, 3, 31, 30, 29)
#define RO_G2(M) M("s85", 4, 5, 6, 28, 27, 26)
#define RO_G3(M) M("s86", 7, 8, 9, 25, 24, 23)
#define RO_G4(M) M("s87", 10, 11, 12, 22, 21, 20)
#define RO_G5(M) M("s84", 13, 14, 15, 19, 18, 17)
#define RO_G6(M) M("s85", 16, 17, 18, 16, 15, 14)
#define RO_G7(M) M("s86", 19, 20, 21, 13, 12, 11)
#define RO_G8(M) M("s87", 22, 23, 24, 10, 9, 8)
#define RO_EXIT(k) "s_cmp_eq_u32 %[G], " #k "\n\ts_cbranch_scc1 9f\n\t"
// counter start (2^NS - 1) - (T - 1) = ib, as bit slices  (%[t] = ib | d << 4: one scalar operand)
#define RO_INIT3 "v_bfe_i32 v40, %[t], 0, 1\n\tv_bfe_i32 v41, %[t], 1, 1\n\tv_bfe_i32 v42, %[t], 2, 1\n\tv_mov_b32 v43, 0\n\t"
#define RO_INIT4 RO_INIT3 "v_bfe_i32 v44, %[t], 3, 1\n\t"
// snapshot into v39: positions whose count so far is <= bound -- adding d = (T - 1) - bound (slices v35 .. v37 [, v45]) does not
// overflow.  RO_SNAP1: ... whose count INCLUDING the next base (its match mask is v32: the group's shifts are done, its add is not)
// is <= bound: the mismatch of that base is the carry into the lowest slice.
#define RO_DMASK3 "v_bfe_i32 v35, %[t], 4, 1\n\tv_bfe_i32 v36, %[t], 5, 1\n\tv_bfe_i32 v37, %[t], 6, 1\n\t"
#define RO_DMASK4 RO_DMASK3 "v_bfe_i32 v45, %[t], 7, 1\n\t"
#define RO_CARRY0 "v_and_b32 v38, v40, v35\n\t"
#define RO_CARRY1 "v_bitop3_b32 v38, v40, v35, v32 bitop3:0xd4\n\t"
#define RO_REST3 "v_bitop3_b32 v38, v41, v36, v38 bitop3:0xe8\n\tv_bitop3_b32 v38, v42, v37, v38 bitop3:0xe8\n\t" \
                 "v_bitop3_b32 v39, v38, v43, v43 bitop3:0x01\n\t"
#define RO_REST4 "v_bitop3_b32 v38, v41, v36, v38 bitop3:0xe8\n\tv_bitop3_b32 v38, v42, v37, v38 bitop3:0xe8\n\t" \
                 "v_bitop3_b32 v38, v44, v45, v38 bitop3:0xe8\n\tv_bitop3_b32 v39, v38, v43, v43 bitop3:0x01\n\t"
// a whole run: SH = RO_SH_F / RO_SH_B, AD = RO_AD3 / RO_AD4, DM / RS = the snapshot's pieces for the counter width.  Close end (first
// evaluated length 8): the snapshot after SEVEN bases = two groups and the first base of the third; far end (first evaluated length
// 10): after nine = three groups.
#define RO_TAIL(SH, AD) RO_G4(SH) RO_PROG_LOAD2 AD RO_EXIT(4) "s_waitcnt lgkmcnt(0)\n\t"                                \
    RO_G5(SH) AD RO_EXIT(5) RO_G6(SH) AD RO_EXIT(6) RO_G7(SH) AD RO_EXIT(7) RO_G8(SH) AD
#define RO_RUN_CLOSE(SH, AD, DM, RS) RO_G1(SH) AD RO_G2(SH) AD RO_G3(SH) DM RO_CARRY1 RS AD RO_TAIL(SH, AD)
#define RO_RUN_FAR(SH, AD, DM, RS) RO_G1(SH) AD RO_G2(SH) AD RO_G3(SH) AD DM RO_CARRY0 RS RO_TAIL(SH, AD)
// the seed's plane (index = byte 3 of the first program dword) by an indexed v_mov -- SEEDREG: "v0" (kind F: the own word is the
// low one) or "v5" (kind B: the high one) --, then seed & (snapshot | no overflow)
// (a run that leaves after four groups still has the second fetch in flight: waited for here)
#define RO_FINAL(SEEDREG) "9:\n\ts_waitcnt lgkmcnt(0)\n\ts_set_gpr_idx_on s88, 0x1\n\tv_mov_b32 v32, " SEEDREG "\n\ts_set_gpr_idx_off\n\t" \
                          "v_bitop3_b32 %[m], v32, v39, v43 bitop3:0xd0"
#define RO_CLOBBERS4 "v44", "v45",
#define RO_CLOBBERS "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43",                   \
                    "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57",                               \
                    "s84", "s85", "s86", "s87", "s88", "m0", "scc", "memory"
// One kind.  KIND 0: F, 1: B; FAR: the far end's snapshot depth.  o1 = the read's orientation in hand (0: left to right, 1: from
// its last base) = which program.  word: the lane's window word (scan_impl).
template <int NB, int KIND, bool FAR, int NS>
__device__ __forceinline__ u32 seed_filter_ro1(const Search &S, bool o1, bool wide, int word)
{
    static_assert(NS == 3 || NS == 4, "three slices (up to 8 mismatch levels) or four (up to 16)");
    const u32 ro = S.ro;
    const u32 G = (ro >> (wide ? 4 : 0)) & 15u;
    const int bound = (int)((ro >> (wide ? 16 : 8)) & 0xffu);
    const int thrA = bound < S.cap_state ? bound : S.cap_state;          // (see seed_filter: the state's bound once candidates are folded)
    const u32 t = (u32)((1 << NS) - S.T) | ((u32)(S.T - 1 - thrA) << 4);  // counter start (2^NS - 1) - (T - 1) | snapshot distance (T - 1) - bound
    const u32 off = o1 ? 0x60u : 0x40u;                                   // PgInRec::prog[o1]
    // kind F reads the words (own, next), kind B (previous, own)
    const u32 wa = (u32)(uintptr_t)(const __attribute__((address_space(3))) uint4 *)(S.win + (2 * NB + word - (KIND == 1 ? 1 : 0)));
#ifdef PG_RO_RP_RECOMPUTE     // (ablation: the record's address worked out again at every run: a kernarg fetch + wait + four scalar instructions)
    const PgInRec *rp = record_ptr<7>(karg_load<const PgInRec *>((int)offsetof(PgKArgs, B.in)), S.rid);
#else
    const PgInRec *rp = KaGlobal<const PgInRec *>::of((u64)S.rp_lo | ((u64)S.rp_hi << 32));
#endif
    u32 m;
#define RO_HEAD(INIT) RO_PROG_LOAD "ds_read_b128 v[32:35], %[wa]\n\tds_read_b128 v[36:39], %[wa] offset:16\n\t" INIT "s_waitcnt lgkmcnt(0)\n\ts_lshr_b32 s88, s84, 24\n\t"
#define RO_OPERANDS(MORE) : [m] "=&v"(m) : [rp] "s"(rp), [off] "s"(off), [wa] "v"(wa), [t] "s"(t), [G] "s"(G) : MORE RO_CLOBBERS
// planes in natural order (A, C, G, T, not-N): low = the own word, high = the next one (kind F); in COMPLEMENT order (T, G, C, A,
// not-N): low = the previous word, high = the own one (kind B)
#define RO_PLANES_F RO_ONEHOT("v48", "v49", "v50", "v51", "v52", "v32", "v33", "v34") RO_ONEHOT("v53", "v54", "v55", "v56", "v57", "v36", "v37", "v38")
#define RO_PLANES_B RO_ONEHOT("v51", "v50", "v49", "v48", "v52", "v32", "v33", "v34") RO_ONEHOT("v56", "v55", "v54", "v53", "v57", "v36", "v37", "v38")
    if (NS == 3) {
        if (KIND == 0 && FAR) asm volatile(RO_HEAD(RO_INIT3) RO_PLANES_F RO_RUN_FAR(RO_SH_F, RO_AD3, RO_DMASK3, RO_REST3) RO_FINAL("v0") RO_OPERANDS());
        else if (KIND == 0) asm volatile(RO_HEAD(RO_INIT3) RO_PLANES_F RO_RUN_CLOSE(RO_SH_F, RO_AD3, RO_DMASK3, RO_REST3) RO_FINAL("v0") RO_OPERANDS());
        else if (FAR) asm volatile(RO_HEAD(RO_INIT3) RO_PLANES_B RO_RUN_FAR(RO_SH_B, RO_AD3, RO_DMASK3, RO_REST3) RO_FINAL("v5") RO_OPERANDS());
        else asm volatile(RO_HEAD(RO_INIT3) RO_PLANES_B RO_RUN_CLOSE(RO_SH_B, RO_AD3, RO_DMASK3, RO_REST3) RO_FINAL("v5") RO_OPERANDS());
    } else {
        if (KIND == 0 && FAR) asm volatile(RO_HEAD(RO_INIT4) RO_PLANES_F RO_RUN_FAR(RO_SH_F, RO_AD4, RO_DMASK4, RO_REST4) RO_FINAL("v0") RO_OPERANDS(RO_CLOBBERS4));
        else if (KIND == 0) asm volatile(RO_HEAD(RO_INIT4) RO_PLANES_F RO_RUN_CLOSE(RO_SH_F, RO_AD4, RO_DMASK4, RO_REST4) RO_FINAL("v0") RO_OPERANDS(RO_CLOBBERS4));
        else if (FAR) asm volatile(RO_HEAD(RO_INIT4) RO_PLANES_B RO_RUN_FAR(RO_SH_B, RO_AD4, RO_DMASK4, RO_REST4) RO_FINAL("v5") RO_OPERANDS(RO_CLOBBERS4));
        else asm volatile(RO_HEAD(RO_INIT4) RO_PLANES_B RO_RUN_CLOSE(RO_SH_B, RO_AD4, RO_DMASK4, RO_REST4) RO_FINAL("v5") RO_OPERANDS(RO_CLOBBERS4));
    }
    return m;
}
// KIND 0: kind F alone (close end of a '+' anchor), 1: kind B alone ('-' anchor), 2: both (far end: two runs -- one pass with both
// kinds' planes and counters resident is 40 fixed registers, and the compiler spilled 30 of its own around it)
template <int NB, int KIND, int NS = 3>
__device__ __forceinline__ void seed_filter_ro(const Search &S, bool o1, bool wide, int word, u32 &mF, u32 &mB)
{
    if (KIND == 2) {
        mF = seed_filter_ro1<NB, 0, true, NS>(S, o1, wide, word);
        mB = seed_filter_ro1<NB, 1, true, NS>(S, o1, wide, word);
    } else
        mF = seed_filter_ro1<NB, KIND, false, NS>(S, o1, wide, word);
}
#endif

template <int NB, int NS, bool DUAL>
__device__ __forceinline__ void seed_filter(const Search &S, const Query<NB> &Q, bool kindB, bool wide, int lane,
                                            u32 &mF, u32 &mB)
{
    const int T = S.T;
    PG_DG(const_cast<Search &>(S), 8);
#ifdef PG_RO
#ifndef PG_RO_KINDS
#define PG_RO_KINDS 7      // (ablation: bit 0 = kind F alone, bit 1 = kind B alone, bit 2 = both kinds)
#endif
#ifdef PG_RO_CHECK
    // diagnostics: both filters on every run; every seed the symbol-by-symbol filter keeps must survive the read-order one (plain
    // depth: same bases in the final count, a shorter prefix in the snapshot).  Counters behind the launch's read counters
    // (pg_debug_read_phase_cycles): [0] runs, [1 + kind] runs with a seed lost, [4] lanes with a seed lost.
    // PG_RO_CHECK = 1: the search goes on with the OLD masks, 2: with the new ones.
    if (NS <= 4 && (S.ro & PG_RO_OK) && !wide) {
        constexpr int RNS = NS <= 3 ? 3 : 4;
        u32 nF = 0u, nB = 0u, oF = 0u, oB = 0u;
        {   // the program as the run will fetch it: every byte 0x30 | symbol <= 4 ?  ([5] counts the runs with another)
            const PgInRec *rp = record_ptr<7>(karg_load<const PgInRec *>((int)offsetof(PgKArgs, B.in)), S.rid);
            u32x8 P;
            asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(P) : "s"(rp), "s"(Q.o1_ ? 0x60u : 0x40u));
            u32 badp = 0u;
#pragma unroll
            for (int k = 0; k < 8; k++) badp |= ((P[k] & 0xf8f8f8f8u) ^ 0x30303030u) | ((P[k] & 0x07070707u) + 0x03030303u) & 0x08080808u;
            if (badp != 0u) {
                if (threadIdx.x == 0) atomicAdd((unsigned long long *)(karg_load<uint32_t *>((int)offsetof(PgKArgs, B.work_ctr)) + PG_WORK_CTRS * 16u) + 5, 1ull);
                const u32 jm = S.jmask[0], g0 = jm & low32(S.bps);
                int c0 = (int)((S.depth >> 16) & 0xffu);
                if (c0 > S.cap_state) c0 = S.cap_state;
                seed_filter_run<NB, NS, DUAL>(S, Q, kindB, lane, jm, g0, c0, mF, mB);
                return;
            }
        }
#if PG_RO_CHECK == 3
        {   // (the old filter alone: does the check code itself disturb anything?)
            const u32 jm = S.jmask[0], g0 = jm & low32(S.bps);
            int c0 = (int)((S.depth >> 16) & 0xffu);
            if (c0 > S.cap_state) c0 = S.cap_state;
            seed_filter_run<NB, NS, DUAL>(S, Q, kindB, lane, jm, g0, c0, mF, mB);
            return;
        }
#endif
        if (DUAL) seed_filter_ro<NB, 2, RNS>(S, Q.o1_, wide, lane, nF, nB);
        else if (kindB) seed_filter_ro<NB, 1, RNS>(S, Q.o1_, wide, lane, nF, nB);
        else seed_filter_ro<NB, 0, RNS>(S, Q.o1_, wide, lane, nF, nB);
        {
            const u32 jm = S.jmask[0], g0 = jm & low32(S.bps);
            int c0 = (int)((S.depth >> 16) & 0xffu);
            if (c0 > S.cap_state) c0 = S.cap_state;
            seed_filter_run<NB, NS, DUAL>(S, Q, kindB, lane, jm, g0, c0, oF, oB);
        }
        const u64 bad = ballot64(((oF & ~nF) | (DUAL ? (oB & ~nB) : 0u)) != 0u);
        if (threadIdx.x == 0) {
            unsigned long long *dg = (unsigned long long *)(karg_load<uint32_t *>((int)offsetof(PgKArgs, B.work_ctr)) + PG_WORK_CTRS * 16u);
            atomicAdd(dg, 1ull);
            if (bad) { atomicAdd(dg + 1 + (DUAL ? 2 : (kindB ? 1 : 0)), 1ull); atomicAdd(dg + 4, (unsigned long long)__popcll(bad)); }
        }
        mF = PG_RO_CHECK == 1 ? oF : nF;
        mB = PG_RO_CHECK == 1 ? oB : nB;
        return;
    }
#endif
    if (NS <= 4 && (S.ro & PG_RO_OK) && ((PG_RO_KINDS >> (DUAL ? 2 : (kindB ? 1 : 0))) & 1)) {
        constexpr int RNS = NS <= 3 ? 3 : 4;          // (the counter of the launch: three slices up to 8 mismatch levels, four up to 16)
        if (DUAL) seed_filter_ro<NB, 2, RNS>(S, Q.o1_, wide, lane, mF, mB);
        else if (kindB) seed_filter_ro<NB, 1, RNS>(S, Q.o1_, wide, lane, mF, mB);       // (kindB is a constant at every call site)
        else seed_filter_ro<NB, 0, RNS>(S, Q.o1_, wide, lane, mF, mB);
        return;
    }
#endif
#if defined(PG_PAD_S) || defined(PG_PAD_VF) || defined(PG_PAD_VS)
    {   // diagnostics: what do 128 more scalar / fast-rate vector / slow-rate vector instructions per filter run cost?
        u32 pa = (u32)lane;
#define PG_R16(x) x x x x x x x x x x x x x x x x
#if defined(PG_PAD_S)
        asm volatile("s_mov_b32 s100, 1\n\t"
                     PG_R16("s_add_u32 s100, s100, 3\n\ts_xor_b32 s100, s100, 5\n\ts_add_u32 s100, s100, 7\n\ts_xor_b32 s100, s100, 9\n\t"
                            "s_add_u32 s100, s100, 3\n\ts_xor_b32 s100, s100, 5\n\ts_add_u32 s100, s100, 7\n\ts_xor_b32 s100, s100, 9\n\t")
                     "v_xor_b32 %0, s100, %0"
                     : "+v"(pa) : : "scc", "s100");
#elif defined(PG_PAD_VF)
        asm volatile(PG_R16("v_xor_b32 %0, 3, %0\n\tv_add_u32 %0, 5, %0\n\tv_xor_b32 %0, 7, %0\n\tv_add_u32 %0, 9, %0\n\t"
                            "v_xor_b32 %0, 3, %0\n\tv_add_u32 %0, 5, %0\n\tv_xor_b32 %0, 7, %0\n\tv_add_u32 %0, 9, %0\n\t")
                     : "+v"(pa));
#else
        asm volatile(PG_R16("v_alignbit_b32 %0, %0, %0, 3\n\tv_alignbit_b32 %0, %0, %0, 5\n\tv_alignbit_b32 %0, %0, %0, 7\n\tv_alignbit_b32 %0, %0, %0, 9\n\t"
                            "v_alignbit_b32 %0, %0, %0, 3\n\tv_alignbit_b32 %0, %0, %0, 5\n\tv_alignbit_b32 %0, %0, %0, 7\n\tv_alignbit_b32 %0, %0, %0, 9\n\t")
                     : "+v"(pa));
#endif
        asm volatile("" : : "v"(pa));
    }
#endif
    // bases inspected: two more in the chunks of wide far-end windows, where survivors cost a whole pass of
    // fold_candidates for a handful of candidates (measured: -4 % time at -x 5, +2 % if used everywhere).  Depths, base masks and
    // bounds come with the read's record (pg_pack_kernel): bits [1, J), and of those the ones below the first evaluated length
    const u32 jmask = S.jmask[wide ? 1 : 0];
    const u32 g0mask = jmask & low32(S.bps);
    int cap0 = (int)((S.depth >> (wide ? 24 : 16)) & 0xffu);      // min(T - 1, g_maxMismatch[J] + ADD)
    // ... and, once candidates have been folded, min with the state's bound: a seed whose level at bps exceeds
    // (lowest level present at L) + ADD for every L in [bps, J] cannot be the lowest there nor within ADD of it
    // (levels only grow with L); later lengths are covered by the "alive after J bases" test
    if (cap0 > S.cap_state) cap0 = S.cap_state;
    // counts up to 7 decide everything when T <= 8 (cap0 <= T - 1 <= 7): three slices + overflow; four up to 16 levels,
    // five up to 32 (-e 0.05 on 300-base reads).  NS is a parameter of the launch (the largest T of the batch, `levels`):
    // with all three counter widths behind a run-time switch at every call site the headline kernel was 246 KB of code
    // with 219 SGPR spills and scratch; one width: 100 KB, 200, none.
    seed_filter_run<NB, NS, DUAL>(S, Q, kindB, lane, jmask, g0mask, cap0, mF, mB);
}

// Scan the positions of [s, e) outside [xs, xe) (wo = word index of AbsLoc 0 of the chromosome).
// The window is cut into 2048-position chunks on the grid g0 + 2048 k; a chunk is staged into LDS
// (up to e_max, so that later nested ranges find it resident), every lane filters one 32-position
// word per candidate kind (seed_filter); the survivors get queue slots from a wave prefix sum of the
// per-lane popcounts and go through fold_candidates 64 at a time.  cache*: filter masks of chunk 0 of
// the far-end window, computed once and reused by the nested ranges.
template <int NB, int NS, typename Id, bool MIXED, bool W3>
__device__ __forceinline__ void scan_impl(const PgDevRef &ref, Search &S,
                                          const Query<NB> &Q, Acc<NB, Id> &A, long long wo, int g0, int s, int e,
                                          int e_max, int xs, int xe, int origin, u32 region, int lane,
                                          u32 use_cache, u32 &cacheF, u32 &cacheB, u32 &cache_valid)
{
    if (!Q.first_ok() || s >= e) return;
    const int k0 = (s - g0) >> PG_CHUNK_SHIFT, k1 = (e - 1 - g0) >> PG_CHUNK_SHIFT;   // floor
    for (int k = k0; k <= k1; k++) {
        const int cs = g0 + (k << PG_CHUNK_SHIFT);
        const int ns = s > cs ? s : cs;
        {
            const int ne = e < cs + (int)PG_CHUNK ? e : cs + (int)PG_CHUNK;
            if (ns >= xs && ne <= xe) continue;             // nothing new in this chunk
        }
        const int wb = cs - 64 * NB;
        if (MIXED && !(use_cache && k == 0)) {
            // WIDE FAR-END WINDOWS (every chunk but the cached innermost one).  Two chunks at a time (PG_PAIR_CHUNKS)
            // when the next one has work too: one LDS fill and, more importantly, one pass of fold_candidates for the handful
            // of survivors of both.  (Its own loop: kept apart from the common path below, which it slowed down.)
            int nh = 1;
            // (129..192-base reads: only when the launch brought the second chunk's LDS, i.e. -x >= 3 -- which is when the
            // ranges set want_cap; cluster windows of more than a chunk at -x <= 2 go chunk by chunk)
            const bool pair_ok = PG_PAIR_CHUNKS(NB) && (PG_WIN_DYN_BYTES(NB) == 0u || S.want_cap());
            if (pair_ok && k + 1 <= k1 && !(use_cache && k + 1 == 0)) {
                const int c1 = cs + (int)PG_CHUNK;
                const int ns1 = s > c1 ? s : c1, ne1 = e < c1 + (int)PG_CHUNK ? e : c1 + (int)PG_CHUNK;
                if (!(ns1 >= xs && ne1 <= xe)) nh = 2;
            }
            const int ce = cs + nh * (int)PG_CHUNK;
            const int ne = e < ce ? e : ce;
            const int se = e_max < ce ? e_max : ce;
            if (!(wo == S.win_wo && S.wbase == wb && se + 64 * NB <= S.win_hi))
                stage_window<NB, W3>(ref, S, wo, wb, se + 64 * NB, lane);
            // a half's survivors get queue slots behind what is already waiting; a full queue is folded at once,
            // the rest after the last half
            int h = 0, end = 0, slot = 0;
            u32 mF = 0u, mB = 0u, pos0 = 0u;
            for (;;) {
                while (mF != 0u && slot < PG_PASS(NB)) {
                    const int bit = __ffs((int)mF) - 1;
                    mF &= mF - 1u;
                    S.queue[slot] = (uint16_t)((pos0 + (u32)bit) << 1);
                    slot++;
                }
                while (mF == 0u && mB != 0u && slot < PG_PASS(NB)) {
                    const int bit = __ffs((int)mB) - 1;
                    mB &= mB - 1u;
                    S.queue[slot] = (uint16_t)(((pos0 + (u32)bit) << 1) | 1u);
                    slot++;
                }
                int n;
                if (end >= PG_PASS(NB)) n = PG_PASS(NB);
                else if (h < nh) {                            // every survivor so far is queued: the next half
                    const int word = 64 * h + lane;
                    seed_filter<NB, NS, true>(S, Q, false, true, word, mF, mB);
#if defined(PG_DUP) && PG_DUP == 3
                    u32 dF, dB;
                    seed_filter<NB, NS, true>(S, Q, false, true, opaque(word), dF, dB);
                    mF &= dF | (u32)opaque(0);
                    mB &= dB | (u32)opaque(0);
#endif
                    const int pbase = cs + 32 * word;
                    const u32 rmask = bits32(ns - pbase, ne - pbase) & ~bits32(xs - pbase, xe - pbase);
                    mF &= rmask;
                    mB &= rmask;
                    h++;
#ifndef PG_NO_WIDE_SKIP
                    if (ballot64((mF | mB) != 0u) == 0ull) continue;      // (most halves of a wide window: nobody passed the filter -- no prefix sum)
#endif
                    const u32 cnt = (u32)(__popc(mF) + __popc(mB));
                    const u32 incl = wave_scan(cnt);
                    slot = end + (int)(incl - cnt);
                    end += (int)read_lane(incl, 63);
                    pos0 = (u32)(64 * NB + 32 * word);
                    continue;
                }
                else if (end > 0) n = end;
                else break;
                S.nsurv += n;
                S.nsurv_total += (u32)n;
                PG_SYNC();
                PG_T(S, S.t_base);
                fold_candidates<NB, Id, MIXED>(S, Q, A, wb, origin, region, n, lane, Rings{ false, 0, 0, 0, 0 }, nullptr);
                PG_STOPPED(S);
                PG_T(S, S.t_base + 1);
                slot -= n;
                end -= n;
                if (end == 0 && h == nh) break;
            }
            k += nh - 1;
            continue;
        }
        const int ne = e < cs + (int)PG_CHUNK ? e : cs + (int)PG_CHUNK;
        // The chunk must be resident up to e_max, not just up to this call's end: the filter masks of the
        // innermost far-end chunk are computed once for all nested ranges, and an earlier fill with the same
        // base (a close-end window that happens to start where this chunk starts) may be shorter.
        const int se = e_max < cs + (int)PG_CHUNK ? e_max : cs + (int)PG_CHUNK;
        if (!(wo == S.win_wo && S.wbase == wb && se + 64 * NB <= S.win_hi))
        {
            stage_window<NB>(ref, S, wo, wb, se + 64 * NB, lane);
#if defined(PG_DUP) && PG_DUP == 2
            stage_window<NB>(ref, S, wo, wb, se + 64 * NB, opaque(lane));
#endif
        }
        PG_STOP_AT(S, 13);
        const int pbase = cs + 32 * lane;
        const u32 rmask = (low32_lane(ne - pbase) & ~low32_lane(ns - pbase)) & ~(low32_lane(xe - pbase) & ~low32_lane(xs - pbase));
        const bool cached = use_cache && k == 0 && cache_valid != 0u;
        u32 mF = 0u, mB = 0u;
        if (cached) {
            mF = cacheF;
            mB = cacheB;
        } else {
            if (MIXED) {
                // far end: kind B reads the complement of what kind F reads (cF != cB), both kinds in one pass
                seed_filter<NB, NS, true>(S, Q, false, false, lane, mF, mB);
            } else {
                u32 unused;
                if (Q.allowF()) seed_filter<NB, NS, false>(S, Q, false, false, lane, mF, unused);
                if (Q.allowB()) seed_filter<NB, NS, false>(S, Q, true, false, lane, mB, unused);
#if defined(PG_DUP) && PG_DUP == 3
                u32 d = 0u;
                if (Q.allowF()) { seed_filter<NB, NS, false>(S, Q, false, false, opaque(lane), d, unused); mF &= d | (u32)opaque(0); }
                if (Q.allowB()) { seed_filter<NB, NS, false>(S, Q, true, false, opaque(lane), d, unused); mB &= d | (u32)opaque(0); }
#endif
            }
            if (use_cache && k == 0) { cacheF = mF; cacheB = mB; }
        }
        PG_STOP_AT(S, 14);
        mF &= rmask;
        mB &= rmask;
        if (use_cache && k == 0) cache_valid = 1u;
        // queue slots: exclusive prefix sum of the per-lane survivor counts (a lane's F survivors first)
        const u32 cnt = (u32)(__popc(mF) + __popc(mB));
        const u32 incl = wave_scan(cnt);
        const int total = (int)read_lane(incl, 63);
        int slot = (int)(incl - cnt);
        PG_STOP_AT(S, 15);
        for (int base = 0; base < total; base += PG_PASS(NB)) {
            PG_SYNC();
            const int top = base + PG_PASS(NB);
            while (mF != 0u && slot < top) {
                const int bit = __ffs((int)mF) - 1;
                mF &= mF - 1u;
                S.queue[slot - base] = (uint16_t)((u32)(64 * NB + 32 * lane + bit) << 1);
                slot++;
            }
            while (mF == 0u && mB != 0u && slot < top) {
                const int bit = __ffs((int)mB) - 1;
                mB &= mB - 1u;
                S.queue[slot - base] = (uint16_t)(((u32)(64 * NB + 32 * lane + bit) << 1) | 1u);
                slot++;
            }
            const int n = total - base < PG_PASS(NB) ? total - base : PG_PASS(NB);
            S.nsurv += n;
            S.nsurv_total += (u32)n;
            PG_SYNC();
#if defined(PG_DUP) && PG_DUP == 4
            {   // diagnostics: the same pass into a throw-away copy of the state
                Acc<NB, Id> A2 = A;
                fold_candidates<NB, Id, MIXED>(S, Q, A2, wb, origin, region, n, opaque(lane), Rings{ false, 0, 0, 0, 0 }, nullptr);
                if (A2.m1 == 0x12345u) A.m1 = A2.m2;
            }
#endif
            PG_T(S, S.t_base);
            PG_STOP_AT(S, 16);
            fold_candidates<NB, Id, MIXED>(S, Q, A, wb, origin, region, n, lane, Rings{ false, 0, 0, 0, 0 }, nullptr);
            PG_STOPPED(S);
            PG_T(S, S.t_base + 1);
        }
    }
}

// W3: the three-words-per-lane fill of two wide chunks exists in this instantiation (stage_window).  Not in the default-parameter
// kernels, which never fill two chunks at once -- and whose 150-base class paid for the mere presence of the code with a vector
// register parked in scratch inside a hot loop (+22 scratch loads per read, 24.1 -> 28.2 ms per 10 M reads).
template <int NB, int NS, typename Id, bool W3 = false>
__device__ __forceinline__ void scan_range(const PgDevRef &ref, Search &S,
                                           const Query<NB> &Q, Acc<NB, Id> &A, long long wo, int g0, int s, int e,
                                           int e_max, int xs, int xe, int origin, u32 region, int lane,
                                           u32 use_cache, u32 &cacheF, u32 &cacheB, u32 &cache_valid)
{
    // window coordinates come out of LDS / per-read loads: tell the compiler they are wave-uniform
    g0 = uni(g0); s = uni(s); e = uni(e); e_max = uni(e_max); xs = uni(xs); xe = uni(xe); origin = uni(origin);
    if (Q.allowF() && Q.allowB())
        scan_impl<NB, NS, Id, true, W3>(ref, S, Q, A, wo, g0, s, e, e_max, xs, xe, origin, region, lane,
                                    use_cache, cacheF, cacheB, cache_valid);
    else
        scan_impl<NB, NS, Id, false, false>(ref, S, Q, A, wo, g0, s, e, e_max, xs, xe, origin, region, lane,
                                        use_cache, cacheF, cacheB, cache_valid);
}

// ---------------------------------------------------------------------------------
// Where the candidates of a search live.
struct RegionInfo {
    int chr;                 // range / close searches: one region on `chr`, positions relative to `origin`
    int origin;
    const pg_window *bd;     // non-null: BreakDancer cluster search, regions from the window list
};

// What evaluate leaves behind for the emission of the runs (registers): per round the lane's winner and
// the ballots that delimit the runs.
template <int NB, typename Id>
struct Eval {
    Id id[NB];
    u32 lo[NB];
    u64 startm[NB], brkm[NB];
    int n_runs, max_len;     // max_len = LengthStr of the last emitted point (0 if none)
    Id id_last;              // winner of the last emitted point
};

// g_maxMismatch[L] for the lane's L (<= M for L <= len)
template <int NB>
__device__ __forceinline__ u32 mm_of(const Search &S, int L)
{
    // (L <= bps + 64 NB - 1 < the table's size; lanes past the read are masked by the caller)
    if (PG_MM_IN_WIN(NB)) return (S.win[L >> 2].w >> (8 * (L & 3))) & 0xffu;
    return S.mm_tab[L];
}

// The reference's rules for every L (lanes own L): "if (minimumNumberOfMismatches > g_maxMismatch[L]) return"
// (searcher.cpp:167, pindel.cpp:2836), emission iff the lowest level holds exactly one position and the
// levels up to +ADDITIONAL_MISMATCH hold no other (searcher.cpp:171-191, pindel.cpp:2849-2893), after
// CheckMismatches (already folded into the state).  The
// reduction itself is not modified.
// qmask: the tier A quarters that belong to the state (bit q; all four unless the pass kept the rings of the nested
// far-end ranges apart, see Rings).
template <int NB, typename Id>
__device__ __forceinline__ void evaluate(Search &S, const Acc<NB, Id> &A, Eval<NB, Id> &E, int lane, int qmask = 15)
{
    const u32 mm0 = mm_of<NB>(S, S.bps + lane);       // (one LDS read per evaluation; a VGPR per phase otherwise: the register budget of six and seven waves per SIMD)
    E.n_runs = 0;
    E.max_len = 0;
    E.id_last = 0;
    PG_DG(S, 24);
    // tier A lives in four 16-lane quarters: bring quarters 1..3 to quarter 0 through LDS and merge
    u32 t1 = A.m1, t2 = A.m2, tok = A.ok;
    Id tid = A.id;
    if (S.tierA()) {
        u32 a1 = PG_BIG, a2 = PG_BIG, aok = 0u;
        Id aid = 0;
        if (lane < 16) { a1 = A.a1; a2 = A.a2; aok = A.aok; aid = A.aid; }
        if (qmask != 1) {                                     // uniform (quarter 0 alone: nothing to fetch)
            S.bufA[lane] = make_uint4(A.a1, A.a2 | (A.aok << 16), (u32)A.aid, (u32)((u64)A.aid >> 32));
            PG_SYNC();
            if (lane < 16) {
#pragma unroll
                for (int q = 1; q < 4; q++) {
                    if (!((qmask >> q) & 1)) continue;        // uniform
                    const uint4 o = S.bufA[lane + 16 * q];
                    const Id oid = sizeof(Id) == 8 ? (Id)((u64)o.z | ((u64)o.w << 32)) : (Id)o.z;
                    merge<Id>(a1, a2, aid, aok, o.x, o.y & 0xffffu, oid, o.y >> 16);
                }
            }
            PG_SYNC();
        }
        merge<Id>(t1, t2, tid, tok, a1, a2, aid, aok);
    }
    if (S.want_cap()) {
        // max over L in [bps, J] of the lowest level present (none present: no bound), + ADD; J as in seed_filter
        // for the chunks of wide windows (the only ones filtered after an evaluation)
        const int J = (int)((S.depth >> 8) & 0xffu);
        u32 v = (S.bps + lane <= J) ? t1 : 0u;
        // (each shift is taken once, outside the select: a DPP read under a diverged EXEC mask sees 0 in the
        // lanes that are switched off)
        u32 o;
        o = row_shr<1>(v); v = v > o ? v : o;
        o = row_shr<2>(v); v = v > o ? v : o;
        o = row_shr<4>(v); v = v > o ? v : o;
        o = row_shr<8>(v); v = v > o ? v : o;
        u32 mx = read_lane(v, 15);
        const u32 m2_ = read_lane(v, 31);               // (J - bps < 32: rows 0 and 1 hold every lane of interest)
        mx = mx > m2_ ? mx : m2_;
        const int cap = (int)mx + S.add_mm;
        S.cap_state = cap < S.T - 1 ? cap : S.T - 1;
    }
    u32 aborted = 0u;
#pragma unroll
    for (int r = 0; r < NB; r++) {
        E.startm[r] = 0ull;
        E.brkm[r] = ~0ull;
        E.id[r] = 0;
        E.lo[r] = 0u;
        const int r0 = S.bps + 64 * r;
        if (r0 > S.len - 1 || aborted) continue;          // uniform
        const int L = r0 + lane;
        const bool valid = L <= S.len - 1;
        u32 m1 = t1, m2 = t2, ok = tok, mmL = mm0;
        Id wid = tid;
        if (r > 0) {
            m1 = m2 = PG_BIG; ok = 0u; wid = 0;
            mmL = mm_of<NB>(S, L);
            if ((A.dirty >> r) & 1u) {                    // uniform
                AccB<Id>::load(S.accB, (r - 1) * 64 + lane, m1, m2, ok, wid);
            }
        }
        const u32 lo = m1 <= (u32)S.M ? m1 : (u32)S.M + 1u;
        const u64 vm = ballot64(valid);
        const u64 ab = vm & ballot64(lo > mmL);
        // a point at L: below the first abort, lowest level <= M and alone up to + ADDITIONAL_MISMATCH, L >= bps + level, CheckMismatches
        const u64 cm = vm & (~ab & (ab - 1ull)) & ballot64(m1 <= (u32)S.M) & ballot64(m2 > m1 + (u32)S.add_mm) &
                       ballot64((u32)L >= (u32)S.bps + m1) & ballot64(ok != 0u);
        // run-length encode consecutive points of the same candidate / level (a run never spans two rounds)
        const u32 lok = lane_bit(cm) ? lo + 1u : 0u;       // 0 = no point here
        const u32 plok = wave_shr1(lok);                   // lane 0 gets 0
        u64 samem = ballot64(plok == lok) & ballot64(wave_shr1((u32)wid) == (u32)wid);
        if (sizeof(Id) == 8) samem &= ballot64(wave_shr1((u32)((u64)wid >> 32)) == (u32)((u64)wid >> 32));
        E.id[r] = wid;
        E.lo[r] = lo;
        E.startm[r] = cm & ~samem;
        E.brkm[r] = E.startm[r] | ~cm;
        E.n_runs += __popcll(E.startm[r]);
        if (cm) {
            const int top = 63 - __clzll((long long)cm);
            E.max_len = r0 + top;
            E.id_last = read_lane(wid, top);
        }
        if (ab) aborted = 1u;
    }
}

// Writes the runs of the evaluation to out[0..): all of them (far end), or those of the candidate of the
// last point (close end: CleanUniquePoints, pindel.cpp:2904-2941, keeps the points whose implied read
// terminal equals the last point's = the runs of the last run's candidate).  kept = masks of the run starts
// that are written.
template <int NB, typename Id>
__device__ __forceinline__ int count_kept(const Eval<NB, Id> &E, bool only_last, u64 *kept)
{
    int n = 0;
#pragma unroll
    for (int r = 0; r < NB; r++) {
        kept[r] = E.startm[r];
        if (only_last && E.startm[r]) kept[r] &= ballot64(E.id[r] == E.id_last);
        n += __popcll(kept[r]);
    }
    return n;
}

template <int NB, typename Id>
__device__ __forceinline__ void emit_runs(const Search &S, bool antiF, bool antiB, int chr0, int origin0,
                                          const pg_window *bd, const Eval<NB, Id> &E, const u64 *kept,
                                          pg_run *out, int lane)
{
    typedef IdFmt<Id> F;
    int written = 0;
#pragma unroll
    for (int r = 0; r < NB; r++) {
        if (kept[r] == 0ull) continue;                    // uniform
        if ((kept[r] >> lane) & 1ull) {
            const int L = S.bps + 64 * r + lane;
            const u64 higher = E.brkm[r] & ~low_bits_lane(lane + 1);
            const int end_lane = higher ? __ffsll((long long)higher) - 2 : WAVE - 1;
            const u64 id = (u64)E.id[r];
            const u32 rel = (u32)(id & ((1ull << F::RB) - 1ull));
            const bool isB = (id >> F::RB) & 1ull;
            int chr = chr0, origin = origin0;
            if (bd) {
                const pg_window w = bd[(u32)(id >> (F::RB + 1))];
                chr = w.chr_id;
                origin = w.start < 0 ? w.end - 1 : w.start;
            }
            const int p = origin + (int)rel;
            const bool anti = isB ? antiB : antiF;
            // pg_run as three dwords: abs_loc_first | len_first, len_last | mismatches, flags, chr_id
            u32 *dst = (u32 *)(out + written + count_below(kept[r]));
            dst[0] = isB ? (u32)(p - L + 1) : (u32)(p + L - 1);
            dst[1] = (u32)L | ((u32)(S.bps + 64 * r + end_lane) << 16);
            dst[2] = E.lo[r] | ((isB ? PG_RUN_BACKWARD : 0u) << 8) | ((anti ? PG_RUN_ANTISENSE : 0u) << 8) |
                     ((u32)(chr & 0xffff) << 16);
        }
        written += __popcll(kept[r]);
    }
}

// ---------------------------------------------------------------------------------
// The read's bit planes (PgDevBatch::planes, built by pg_pack_reads) into LDS.  In two steps, so that the caller can put
// other loads between the request and the first use.  Lane l < 8 plane_blocks holds one u64; blocks the batch's layout
// does not have (plane_blocks < NB) stay zero in LDS from the start of the kernel.
template <int NB>
__device__ __forceinline__ u64 request_planes(const PgDevBatch &B, uint32_t rid, int lane)
{
    const u32 pb = KA(B, plane_blocks);
    return (u32)lane < 8u * pb ? KA(B, planes)[(size_t)rid * 8u * pb + (u32)lane] : 0ull;
}
template <int NB>
__device__ __forceinline__ void store_planes(const PgDevBatch &B, u64 v, int lane, u64 *qp)
{
    const u32 pb = KA(B, plane_blocks);
    PG_SYNC();
    if ((u32)lane < 8u * pb) qp[pb == (u32)NB ? (u32)lane : ((u32)lane / pb) * NB + (u32)lane % pb] = v;
    PG_SYNC();
}

// Bump-allocates n runs in this workgroup's pool shard (one atomic per wave); returns the pool
// offset.  fits = the allocation lies inside the shard (otherwise the host repeats the launch with a
// larger pool).
__device__ __forceinline__ u32 pool_alloc(const PgDevBatch &B, int n, int lane, u32 &fits)
{
    const u32 shard = blockIdx.x & (PG_POOL_SHARDS - 1u);
    u32 off = 0;
    if (n > 0 && lane == 0) off = atomicAdd(KA(B, pool_used) + shard * 16u, (u32)n);
    off = (u32)uni((int)off);
    fits = (u64)off + (u64)n <= (u64)KA(B, pool_shard_cap) ? 1u : 0u;
    return shard * KA(B, pool_shard_cap) + off;
}

// One read: close end, then far end.
//   close end   attempts (R0,seq) (R0,RC) (R1,RC) (R1,seq) until one yields points    pindel.cpp:2537-2575
//   far end     BreakDancer cluster (if the read has one), then the ranges
//               r = 1 .. MaxRangeIndex+1 until goodFarEndFound                          pindel.cpp:1006-1070
// EXACT (round 6): the instantiation for reads that hold a character outside ACGTN (pg_search_exact_kernel, a handful of reads if
// any: the pack kernel lists them).  When an attempt fails the reference does "setUnmatchedSeq(ReverseComplement(seq))"
// (pindel.cpp:2545): Convert2RC4N turns every such character into NUL (pindel.cpp:966-970), setUnmatchedSeq strips the NULs that
// end up at the END (pindel.cpp:142-157) -- the characters the read BEGAN with -- and recomputes ReadLength, MAX_SNP_ERROR and
// TOTAL_SNP_ERROR_CHECKED.  The read stays short: attempts 1 and 2 and a far end after them see RC(read) without its last `ja`
// characters (ja = leading characters outside ACGTN); the second reverse complement before attempt 3 strips what the read ENDED
// with (jb): attempt 3 and a far end after it see the read without either.  In plane terms: orientation 0 loses its first ja bits,
// orientation 1 its first jb, the length and everything that follows from it (levels, thresholds, filter depths) shrink.
template <int NB, int NS, typename Id, int mode, bool DEF, bool EXACT = false>
__device__ __forceinline__ void search_read(const PgDevRef &ref, const PgDevParams &prm, const PgDevBatch &B,
                                            Search &S, u64 *qplanes, const uint32_t rid, const int touch_next, const int lane,
                                            const u32 res_base, const u32 res_fits)
{
    // The lane index where the rest of a read needs it.  Kept from the start of the read it is a VGPR the compiler parks in scratch in
    // the kernels for reads of up to 128 bases (8 bytes per lane: a store per read and a reload per use -- 190 bytes of HBM writes per
    // read, profiles/r06): worked out again instead (two v_mbcnt; +18 vector instructions per read, +0.3 % time, no scratch).  The
    // longer classes keep the value (recomputing cost them 1 %).
#define PG_LANE (NB <= 2 ? lane_now() : opaque(lane))
    S.win_wo = -1;
    S.win_hi = S.wbase = 0;
    S.nsurv_total = 0u;
#ifdef PG_TIMING
    S.t_base = 1;
#endif
#ifdef PG_DIAG
    S.dg = 0u;
#endif
    S.cap_state = 255;
    S.sf = 0u;
#ifdef PG_STOP
    S.stopped = 0u;
#endif
    // The read's packed record (pg_device.h: everything that depends on the read and the parameters alone, worked out by the pack
    // kernel), by scalar loads straight into SGPRs: dwords 0 .. 11 now, 12 .. 15 at the start of the far end.  (rid is wave-uniform.)
    // The read's bit planes are requested first (they need nothing but the read's index) and arrive while the record is
    // waited for; the window of the first close-end attempt follows (the scan below finds it resident).
    const u64 planes_of_read = request_planes<NB>(B, rid, lane);
    const PgInRec *rp = record_ptr<7>(KA(B, in), rid);
    u32x8 ra;
    u32x4 rb;
    // (load and wait in ONE statement: between two statements the compiler may spill or reuse the destination registers -- it
    // does not know that a scalar load is still in flight -- and the late data then lands on whatever lives there)
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(ra), "=&s"(rb) : "s"(rp));
    int len = (int)(ra[6] & 0xffffu);                       // (changes only in the EXACT instantiation)
    u32 flags = ra[6] >> 16;                                // PG_RF_*
    const int chr = (int)rb[3];
    const long long chr_wo = (long long)((u64)ra[4] | ((u64)ra[5] << 32));
    S.len = len;
    S.thr = (int)(ra[7] & 0xffffu);
    S.M = (int)((ra[7] >> 16) & 0xffu);
    S.T = (int)(ra[7] >> 24);
    S.add_mm = PRM(add_mm, PG_DEF_ADD_MM);
    S.min_perfect = PRM(min_perfect, PG_DEF_MIN_PERFECT);
    S.depth = rb[0];
    S.jmask[0] = rb[1];
    S.jmask[1] = 0u;                                        // (the wide depth's mask comes with the far end's part of the record)
    S.ro = rb[2];
    S.rid = rid;
    S.rp_lo = (u32)uni((int)(u32)(u64)(uintptr_t)rp);
    S.rp_hi = (u32)uni((int)(u32)((u64)(uintptr_t)rp >> 32));
    PG_STOP_AT(S, 10);
    if ((mode & PG_MODE_CLOSE) && (flags & PG_RF_CLOSE_OK)) {
        // attempt 0's window -- the whole R = 1 window when that fits one chunk (PG_RF_SHARED_GRID): every attempt then runs on its grid
        const int ss = (int)ra[2], se = (int)ra[3];
        // (touch_next = 0: the next record is another claim's and is still to be written -- pack in place; the read's own record then)
        if (ss < se) stage_window<NB>(ref, S, chr_wo, ss - 64 * NB, se + 64 * NB, lane, rp + uni(touch_next));
    }
    store_planes<NB>(B, planes_of_read, lane, qplanes);

    PG_T(S, 0);
#if defined(PG_STOP) && PG_STOP == 1       // diagnostics (wrong results): what do the claim and the start of a read cost?
    if (uni(opaque(1))) return;
#endif
    // ---- EXACT: the read's two shortenings
    int ja = 0, jb = 0, ex_stage = 0;      // leading characters outside ACGTN of orientation 0 / 1; reverse complements applied so far
    auto ex_apply = [&](int to) __attribute__((always_inline)) {
        // one more "setUnmatchedSeq(ReverseComplement())": stage 1 drops the first ja bases of orientation 0, stage 2 the first jb of
        // orientation 1 (what the reverse complement left at the end as NULs); then everything that depends on ReadLength
        while (ex_stage < to) {
            ex_stage++;
            const int cut = ex_stage == 1 ? ja : jb;
            u64 *pl = qplanes + (ex_stage == 1 ? 0 : 4 * NB);
            if (cut > 0) {
                const int l = PG_LANE;
                u64 v = 0ull;
                if (l < 4 * NB) {
                    const int w = l % NB, wi = w + (cut >> 6), sh = cut & 63;
                    const u64 lo_ = wi < NB ? pl[(l / NB) * NB + wi] : 0ull, hi_ = wi + 1 < NB ? pl[(l / NB) * NB + wi + 1] : 0ull;
                    v = sh ? (lo_ >> sh) | (hi_ << (64 - sh)) : lo_;
                }
                PG_SYNC();
                if (l < 4 * NB) pl[l] = v;
                PG_SYNC();
                len = len > cut ? len - cut : 0;
            }
        }
        S.len = len;
        S.M = max_mismatch_at(prm.mm_bp, len);
        S.T = S.M + S.add_mm + 1;
        S.thr = (int)uni((int)KA(B, thr_tab)[len]);
        const int J0 = pg_seed_depth(len, S.T, 0), J1 = pg_seed_depth(len, S.T, 1);
        const u32 b0 = (u32)min(S.T - 1, max_mismatch_at(prm.mm_bp, J0 < 0 ? 0 : J0) + S.add_mm),
                  b1 = (u32)min(S.T - 1, max_mismatch_at(prm.mm_bp, J1 < 0 ? 0 : J1) + S.add_mm);
        S.depth = (u32)(J0 < 0 ? 0 : J0) | ((u32)(J1 < 0 ? 0 : J1) << 8) | (b0 << 16) | (b1 << 24);
        S.jmask[0] = J0 > 1 ? (low32(J0) & ~1u) : 0u;
        S.jmask[1] = J1 > 1 ? (low32(J1) & ~1u) : 0u;
        S.ro = 0u;
        // is the first consumed base of either orientation one of ACGT ?  (bit 0 of the N / other planes)
        const u32 f0 = (u32)uni((int)(u32)(qplanes[QP_NN * NB] | qplanes[QP_OO * NB])) & 1u,
                  f1 = (u32)uni((int)(u32)(qplanes[4 * NB + QP_NN * NB] | qplanes[4 * NB + QP_OO * NB])) & 1u;
        flags &= ~(PG_RF_FIRST_OK_FWD | PG_RF_FIRST_OK_REV);
        if (!f0 && len > 0) flags |= PG_RF_FIRST_OK_FWD;
        if (!f1 && len > 0) flags |= PG_RF_FIRST_OK_REV;
    };
    if (EXACT) {
        // leading characters outside ACGTN = the run of set bits from bit 0 of the "other" plane
        auto lead = [&](const u64 *oo) {
            int n = 0;
            for (int w = 0; w < NB; w++) {
                const u64 x = (u64)(u32)uni((int)(u32)oo[w]) | ((u64)(u32)uni((int)(u32)(oo[w] >> 32)) << 32);
                if (x == ~0ull) { n += 64; continue; }
                n += __ffsll((long long)~x) - 1;
                break;
            }
            return n < len ? n : len;
        };
        ja = lead(qplanes + QP_OO * NB);
        jb = lead(qplanes + 4 * NB + QP_OO * NB);
        S.add_mm = PRM(add_mm, PG_DEF_ADD_MM);
    }
    const bool do_close = (mode & PG_MODE_CLOSE) != 0, do_far = (mode & PG_MODE_FAR) != 0;
    int flipped = 0, close_max = 0, n_close = 0;
    u32 close_last = 0, close_base = 0, alg = 0u;
    u32 unused0 = 0u, unused1 = 0u;
    u32 unused_valid = 0u;     // (wave-uniform flags as words: a bool is a 64-bit lane mask, two scalar registers and mask arithmetic)
    Acc<NB, Id> A;
    u32 fits = 1u;

    // ------------------------------------------------------------------------------- close end
    if (do_close) {
        int close_bases = 0;
        if (flags & PG_RF_CLOSE_OK) {                     // len - 1 >= g_MinClose, MatchedD '+' or '-' (pindel.cpp:2258, 2271, 2298)
            const int w1s = (int)ra[0], isz = (int)ra[1];
            const u32 plus = flags & PG_RF_PLUS;
            S.bps = PRM(min_close, PG_DEF_MIN_CLOSE);
            // Min_Perfect_Match_Around_BP >= the first evaluated length: CheckMismatches' length test can fail, no short-lived tier
            S.sf = S.min_perfect >= S.bps ? 4u : (S.bps + 16 <= 32 ? 2u : 0u);
            // Attempt 0 (the one that succeeds for most reads) stages and filters exactly its own window.  The
            // retries share work: the window of the attempts with R = 1 contains the one with R = 0, so from
            // attempt 1 on the chunk grid is anchored at the R = 1 window, which is staged once; attempts 1 and 2
            // use the same orientation of the read, so the seed-filter masks of the first chunk are computed
            // once for both and attempt 2 only adds the flanks to attempt 1's state (the reduction is additive).
            // (whichever the strand, R = 1 is [w1s, w1s + 3 isz) and R = 0 its middle third: the record carries w1s)
            const int w1e = w1s + 3 * isz;
            // When the R = 1 window fits one chunk (3 InsertSize <= 2048, the usual case), attempt 0 runs on that grid too:
            // the stage above already brought the whole R = 1 window (one pass of the fill loop either way), and
            // attempts 0 and 3 -- same orientation, nested windows -- share their seed-filter masks the way attempts 1
            // and 2 do: a read without a close end costs two filter runs and one fill instead of three and two.
            const u32 shared_grid = (flags / PG_RF_SHARED_GRID) & 1u;       // isz > 0 && 3 isz <= PG_CHUNK
            u32 cr0 = 0u, cr1 = 0u;                       // cached masks of the orientation in hand ...
            u32 vr = 0u;
            u32 co0 = 0u, co1 = 0u;                       // ... and of the other one (swapped at attempts 1 and 3)
            u32 vo = 0u;
            int ps = 0, pe = 0, nsurv_eval = 0;
            PG_STOP_AT(S, 11);
            // The attempts as one body instantiated twice: attempt 0 -- where four reads in five stop -- with the attempt number a
            // compile-time constant (no orientation swap, no continued state, R = 0 folded into the window arithmetic: -44 vector and
            // -20 scalar instructions per read, -1.1 %) and, for that instance, the anchor's strand too (the candidate kind: another
            // -11 / -6, -0.6 %); attempts 1..3 as a loop.  Returns true when the attempt found points.
            auto attempt = [&](const int att, const bool plus_k) __attribute__((always_inline)) -> bool {
                const int Rg = att >> 1;
#ifdef PG_TIMING
                PG_T(S, S.t_base + 2);
                S.t_base = att == 0 ? 1 : 4;
#endif
                flipped = (att == 1 || att == 2) ? 1 : 0;
                if (EXACT && (att == 1 || att == 3)) {
                    // the failed attempt's "setUnmatchedSeq(ReverseComplement())": the read may be shorter now, and nothing worked out for
                    // the other length is valid any more
                    ex_apply(att == 1 ? 1 : 2);
                    cr0 = cr1 = co0 = co1 = 0u;
                    vr = vo = 0u;
                }
                if (EXACT && len - 1 < PRM(min_close, PG_DEF_MIN_CLOSE)) return false;      // BP_End < BP_Start: no point (pindel.cpp:2268-2269)
                // '+' anchor: CurrentReadSeq = RC(cur), grown left to right (pindel.cpp:2271-2291)
                // '-' anchor: CurrentReadSeq = cur, grown right to left     (pindel.cpp:2298-2319)
                Query<NB> Q;
                Q.qp = qplanes + (!flipped ? 4 * NB : 0);
                Q.o1_ = !flipped;
                if (plus_k) {
                    Q.cF_ = !flipped; Q.cB_ = false; Q.allowF_ = true; Q.allowB_ = false;
                } else {
                    Q.cB_ = flipped; Q.cF_ = false; Q.allowF_ = false; Q.allowB_ = true;
                }
                const int s1 = Rg ? w1s : w1s + isz, e1 = Rg ? w1e : w1s + 2 * isz;
                // (points: CheckLeft_Close FORWARD / ANTISENSE, CheckRight_Close BACKWARD / SENSE: emit_runs' arguments)
                Q.first_ok_ = (flags & (flipped ? PG_RF_FIRST_OK_FWD : PG_RF_FIRST_OK_REV)) != 0u;   // (orientation 1 unless flipped)
                close_bases = e1 > s1 ? e1 - s1 : 0;
                if (att != 2) {           // attempt 2 continues attempt 1's reduction
                    A.reset();
                    S.cap_state = 255;
                    S.nsurv = 0;
                    nsurv_eval = 0;
                    ps = pe = 0;
                }
                if (att == 1 || att == 3) {               // the read turns round: the other orientation's masks
                    const u32 t0 = cr0, t1 = cr1;
                    const u32 tv = vr;
                    cr0 = co0; cr1 = co1; vr = vo;
                    co0 = t0; co1 = t1; vo = tv;
                }
                // one call site: attempt 0 on its own grid unless the R = 1 window fits a chunk, the retries on the grid
                // of the R = 1 window
                const bool own_grid = att == 0 && !shared_grid;
                PG_STOP_AT_V(S, 12, true);
                scan_range<NB, NS, Id, !DEF>(ref, S, Q, A, chr_wo, own_grid ? s1 : w1s, s1, e1, own_grid ? e1 : w1e, ps, pe, w1s, 0u,
                                   PG_LANE, (att == 1 || att == 2 ? 1u : 0u) | shared_grid, cr0, cr1, vr);
                PG_STOPPED_V(S, true);
                PG_STOP_AT_V(S, 19, true);
                ps = s1;
                pe = e1;
                if (S.nsurv != nsurv_eval) {
                    nsurv_eval = S.nsurv;
                    Eval<NB, Id> E;
#if defined(PG_DUP) && PG_DUP == 5
                    {   // diagnostics: the evaluation twice (on an opaque copy of the state: two calls on the same state are merged)
                        Acc<NB, Id> A2 = A;
                        A2.m1 = (u32)opaque((int)A.m1); A2.a1 = (u32)opaque((int)A.a1);
                        Eval<NB, Id> E2;
                        evaluate<NB, Id>(S, A2, E2, PG_LANE);
                        {
                            u32 chk = (u32)E2.n_runs ^ (u32)E2.max_len ^ (u32)E2.id_last;
#pragma unroll
                            for (int r = 0; r < NB; r++) chk ^= (u32)E2.id[r] ^ E2.lo[r] ^ (u32)E2.startm[r] ^ (u32)(E2.brkm[r] >> 32);
                            if (__builtin_amdgcn_ballot_w64(chk == 0x12345u) == ~0ull) A.m1 = chk;
                        }
                    }
#endif
                    evaluate<NB, Id>(S, A, E, PG_LANE);
                    PG_STOP_AT_V(S, 20, true);
                    close_max = uni(E.max_len);
                    if (uni(E.n_runs) > 0) {
                        u64 kept[NB];
                        n_close = uni(count_kept<NB, Id>(E, true, kept));
                        // the read's reserved slots, or (a list longer than that) an allocation of its own
                        if (n_close <= (int)PG_RES_CLOSE) {
                            close_base = res_base;
                            fits = res_fits;
                        } else
                            close_base = pool_alloc(B, n_close, lane, fits);
#if defined(PG_DUP) && PG_DUP == 7
                        if (fits) emit_runs<NB, Id>(S, true, false, chr, opaque(w1s), nullptr, E, kept, KA(B, pool) + close_base, PG_LANE);
#endif
                        if (fits) emit_runs<NB, Id>(S, true, false, chr, w1s, nullptr, E, kept, KA(B, pool) + close_base, PG_LANE);
                        // AbsLoc of the last point (getLastAbsLocCloseEnd)
                        const u64 idl = (u64)E.id_last;
                        const int pl = w1s + (int)(u32)(idl & ((1ull << IdFmt<Id>::RB) - 1ull));
                        close_last = ((idl >> IdFmt<Id>::RB) & 1ull) ? (u32)(pl - close_max + 1) : (u32)(pl + close_max - 1);
                        PG_STOP_AT_V(S, 21, true);
                        return true;
                    }
                }
                return false;
            };
            if (!(plus ? attempt(0, true) : attempt(0, false)))
                for (int att = 1; att < 4; att++)
                    if (attempt(att, plus != 0u)) break;
            PG_STOPPED(S);
        }
        if (n_close == 0) { flipped = 0; close_max = 0; }       // back to the original orientation
        if (EXACT) {
            // what GetCloseEnd left: 1 = reverse-complemented once; 2 = twice (attempt 3 ran, or nothing was found) -- NOT the original
            // again for a read of this kernel: its characters outside ACGTN are NUL now, those at either end gone
            if (n_close == 0) ex_stage = 2;       // (GetCloseEnd reverse-complements twice whether or not GetCloseEndInner could search)
            flipped = ex_stage == 1 ? 1 : (ex_stage == 2 ? 2 : 0);
        }
        alg = (u32)(8 * len + 3 * (close_bases + 2 * len) + 96 * n_close);   // x 8: the read once, 3 bits per base, 12 bytes per run
    } else {
        // far-end launch: the close-end summary of the earlier launch
        const uint4 o1 = ((const uint4 *)record_ptr<5>(KA(B, out), rid))[1];
        close_last = (u32)uni((int)o1.x);
        close_max = uni((int)(o1.y & 0xffffu));
        flipped = uni((int)((o1.y >> 16) & 0xffu));
        alg = (u32)uni((int)o1.z) << 3;
        if (EXACT) ex_apply(flipped);                          // the read as the close end left it (rc_flag 1 / 2)
    }
    const int rc_out = flipped;                                // PgOutRec::rc_flag
    if (EXACT) flipped &= 1;                                   // (orientation in hand for the far end: 2 = the original orientation)

    // ------------------------------------------------------------------------------- far end
#ifdef PG_TIMING
    PG_T(S, S.t_base + 2);
    S.t_base = 7;
#endif
    PG_STOP_AT(S, 22);
    int n_far = 0, far_max = 0;
    u32 far_base = 0;
    // "if (CurrentBase == 'N' || MaxLenCloseEnd() == 0) return;" (farend_searcher.cpp:60-66)
#if defined(PG_STOP) && PG_STOP == 2       // diagnostics (wrong results): no far end
    if (uni(opaque(1))) close_max = 0;
#endif
    if (do_far && close_max > 0 && len - 1 >= 10) {
        S.bps = 10;               // farend_searcher.cpp:90
        S.sf = S.min_perfect >= 10 ? 4u : 2u;
        // cur = flipped ? RC(orig) : orig.  Plus strand consumes cur left to right, Minus strand
        // consumes complement(cur) walking the reference right to left.
        Query<NB> Q;
        Q.qp = qplanes + (flipped ? 4 * NB : 0);
        Q.o1_ = flipped != 0;
        Q.cF_ = flipped; Q.cB_ = !flipped;                // (points: FORWARD / SENSE, BACKWARD / ANTISENSE)
        Q.allowF_ = Q.allowB_ = true;
        Q.first_ok_ = (flags & (flipped ? PG_RF_FIRST_OK_REV : PG_RF_FIRST_OK_FWD)) != 0u;       // (orientation 1 if flipped)
        if (Q.first_ok_) {
            // the rest of the record (chromosome size, window cluster); its address again rather than two scalar registers
            // held through the close end
            u32x4 rc;
            asm volatile("s_load_dwordx4 %0, %1, 0x30\n\ts_waitcnt lgkmcnt(0)" : "=&s"(rc) : "s"(record_ptr<7>(KA(B, in), rid)));
            const int chr_size = (int)rc[0];
            S.jmask[1] = rc[3];
            int far_bases = 0;
            // a search window's result replaces UP_Far if its MaxLen is >= (NewUPFarIsBetter, farend_searcher.cpp:30-44)
            auto far_update = [&](int origin, const pg_window *bdw, int qmask) {
                Eval<NB, Id> E;
#if defined(PG_DUP) && PG_DUP == 5
                {
                    Acc<NB, Id> A2 = A;
                    A2.m1 = (u32)opaque((int)A.m1); A2.a1 = (u32)opaque((int)A.a1);
                    Eval<NB, Id> E2;
                    evaluate<NB, Id>(S, A2, E2, PG_LANE, qmask);
                    {
                        u32 chk = (u32)E2.n_runs ^ (u32)E2.max_len ^ (u32)E2.id_last;
#pragma unroll
                        for (int r = 0; r < NB; r++) chk ^= (u32)E2.id[r] ^ E2.lo[r] ^ (u32)E2.startm[r] ^ (u32)(E2.brkm[r] >> 32);
                        if (__builtin_amdgcn_ballot_w64(chk == 0x12345u) == ~0ull) A.m1 = chk;
                    }
                }
#endif
                evaluate<NB, Id>(S, A, E, PG_LANE, qmask);
                const int mx = uni(E.max_len);
                if (mx >= far_max) {
                    far_max = mx;
                    n_far = uni(E.n_runs);
                    far_base = 0;
                    if (n_far > 0) {
                        u64 kept[NB];
                        (void)count_kept<NB, Id>(E, false, kept);
                        if (n_far <= (int)PG_RES_FAR) {          // (a later range's result overwrites an earlier one's slots)
                            far_base = res_base + PG_RES_CLOSE;
                            fits &= res_fits;
                        } else {
                            u32 f2 = 1u;
                            far_base = pool_alloc(B, n_far, lane, f2);
                            fits &= f2;
                        }
#if defined(PG_DUP) && PG_DUP == 7
                        if (fits) emit_runs<NB, Id>(S, false, true, chr, opaque(origin), bdw, E, kept, KA(B, pool) + far_base, PG_LANE);
#endif
                        if (fits) emit_runs<NB, Id>(S, false, true, chr, origin, bdw, E, kept, KA(B, pool) + far_base, PG_LANE);
                    }
                }
            };
            bool done = false;
            // BreakDancer / read-pair cluster of this read first (pindel.cpp:1006-1018)
            const pg_window *bd_all = KA(B, bd);
            if (bd_all && rc[1] != 0u) {
                const int nbd = (int)rc[1];
                const pg_window *bd = bd_all + rc[2];
                A.reset();
                S.cap_state = 255;
                S.nsurv = 0;
                for (int w = 0; w < nbd; w++) {
                    const pg_window bw = bd[w];
                    const int st = bw.start < 0 ? bw.end - 1 : bw.start;
                    const bool own_chr = uni(bw.chr_id) == chr;          // (the usual case: the record's offset and size)
                    const int csz = own_chr ? chr_size : chr_size_of<NB>(ref, S, uni(bw.chr_id));
                    const int s = st < 0 ? 0 : st, e = bw.end > csz ? csz : bw.end;
                    far_bases += (e > s ? e - s : 0) + 2 * len;
                    scan_range<NB, NS, Id, !DEF>(ref, S, Q, A, own_chr ? chr_wo : chr_word_off_of<NB>(ref, S, uni(bw.chr_id)), s, s, e, e, 0, 0, st,
                                       (u32)w, PG_LANE, false, unused0, unused1, unused_valid);
                    PG_STOPPED(S);
                }
                if (S.nsurv > 0) far_update(0, bd, 15);
                done = far_max + close_max >= len;           // goodFarEndFound (pindel.cpp:480-483)
            }
            if (!done) {
                // ranges 128 * 4^r around the last close-end point, clipped to the non-spacer part
                // (pindel.cpp:1025-1070).  The reduction is additive, so only the flanks the previous ranges
                // did not cover are scanned.  Chunk grid: the innermost 2048 positions are one chunk (one LDS
                // fill and one seed-filter pass serve the ranges up to 1024).
                const int center = (int)close_last;
#ifdef PG_DEF_X_RUNTIME      // (experiment: -x a launch argument of the default-parameter kernels too)
                const int k_mri = KA(prm, max_range_index);
#else
                const int k_mri = PRM(max_range_index, PG_DEF_MAX_RANGE_INDEX);
#endif
                const u32 k_spacer = PRM(spacer, PG_DEF_SPACER);
                const int maxspan = 64 << (2 * k_mri);
                const int origin = center - maxspan;
                const int g0 = center - (int)PG_CHUNK / 2;
                int emax;
                if ((u32)center + (u32)maxspan + k_spacer < (u32)chr_size) emax = center + maxspan;
                else emax = chr_size - (int)k_spacer;
                u32 cacheF = 0u, cacheB = 0u;                    // seed-filter masks of the innermost chunk
                u32 cache_valid = 0u;
                int ps = 0, pe = 0, nsurv_eval = 0;
                A.reset();
                S.cap_state = 255;
                if (k_mri >= 3) S.sf |= 1u;     // ranges beyond the cached innermost chunk will be filtered
                S.nsurv = 0;
                auto range_of = [&](int span, int &s, int &e) {
                    if ((u32)center > (u32)span + k_spacer) s = center - span; else s = (int)k_spacer;
                    if ((u32)center + (u32)span + k_spacer < (u32)chr_size) e = center + span;
                    else e = chr_size - (int)k_spacer;
                };
                int r_first = 0;
#ifndef PG_NO_FUSED_RANGES
                // FUSED RANGES.  The ranges that lie in the innermost chunk (spans 64, 256, 1024) share ONE candidate pass:
                // the survivors of the whole chunk are queued at once, the pass keeps the rings apart (Rings), and the
                // ranges are then evaluated one after the other exactly as the reference walks them -- ring r's long-lived
                // candidates are folded just before range r is evaluated, its short-lived ones sit in their own tier A
                // quarter(s).  A read that needs all three ranges pays one pass instead of three.
                {
                    const int R = k_mri < 2 ? k_mri : 2;
                    int rs[3], re[3];
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        range_of(64 << (2 * (r < R ? r : R)), rs[r], re[r]);
                        rs[r] = uni(rs[r]);
                        re[r] = uni(re[r]);
                    }
                    PG_STOP_AT(S, 23);
                    if (rs[R] < re[R]) {
                        const int wb = g0 - 64 * NB;
                        const int se = emax < g0 + (int)PG_CHUNK ? emax : g0 + (int)PG_CHUNK;
                        if (!(chr_wo == S.win_wo && S.wbase == wb && se + 64 * NB <= S.win_hi))
                            stage_window<NB>(ref, S, chr_wo, wb, se + 64 * NB, lane);
                        PG_STOP_AT(S, 24);
                        u32 mF = 0u, mB = 0u;
                        seed_filter<NB, NS, true>(S, Q, false, false, lane, mF, mB);
                        PG_STOP_AT(S, 25);
#if defined(PG_DUP) && PG_DUP == 8
                        {
                            u32 dF, dB;
                            seed_filter<NB, NS, true>(S, Q, false, false, PG_LANE, dF, dB);
                            mF &= dF | (u32)opaque(0);
                            mB &= dB | (u32)opaque(0);
                        }
#endif
                        cacheF = mF;
                        cacheB = mB;
                        cache_valid = 1u;
                        const int pbase = g0 + 32 * lane;
                        const u32 rmask = low32_lane(re[R] - pbase) & ~low32_lane(rs[R] - pbase);
                        mF &= rmask;
                        mB &= rmask;
                        const u32 cnt = (u32)(__popc(mF) + __popc(mB));
                        const u32 incl = wave_scan(cnt);
                        const int total = (int)read_lane(incl, 63);
                        PG_STOP_AT(S, 26);
                        if (total <= PG_PASS(NB)) {
                            r_first = R + 1;
                            ps = rs[R];
                            pe = re[R];
                            if (total > 0) {
                                int slot = (int)(incl - cnt);
                                PG_SYNC();
                                while (mF != 0u) {
                                    const int bit = __ffs((int)mF) - 1;
                                    mF &= mF - 1u;
                                    S.queue[slot] = (uint16_t)((u32)(64 * NB + 32 * lane + bit) << 1);
                                    slot++;
                                }
                                while (mB != 0u) {
                                    const int bit = __ffs((int)mB) - 1;
                                    mB &= mB - 1u;
                                    S.queue[slot] = (uint16_t)(((u32)(64 * NB + 32 * lane + bit) << 1) | 1u);
                                    slot++;
                                }
                                S.nsurv += total;
                                S.nsurv_total += (u32)total;
                                PG_SYNC();
                                PG_STOP_AT(S, 27);
                                int ring_n[3];
                                PG_T(S, 7);
#if defined(PG_DUP) && PG_DUP == 9
                                {
                                    Acc<NB, Id> A2 = A;
                                    int rn2[3];
                                    fold_candidates<NB, Id, true>(S, Q, A2, wb, origin, 0u, total, PG_LANE,
                                                                  Rings{ true, rs[0], re[0], rs[1], re[1] }, rn2);
                                    if (A2.m1 == 0x12345u) A.m1 = A2.m2;
                                }
#endif
                                fold_candidates<NB, Id, true>(S, Q, A, wb, origin, 0u, total, lane,
                                                              Rings{ true, rs[0], re[0], rs[1], re[1] }, ring_n);
                                PG_STOPPED(S);
                                PG_STOP_AT(S, 30);
                                PG_T(S, 8);
                                for (int r = 0; r <= R; r++) {
                                    if (uni(ring_n[r]) > 0) {       // (no new candidate: the evaluation would repeat the previous one)
                                        u64 longm[NB];
#pragma unroll
                                        for (int k = 0; k < NB; k++) longm[k] = read_lane(S.ringB[r * NB + k], 0);
#if defined(PG_DUP) && PG_DUP == 6
                                        {
                                            Acc<NB, Id> A2 = A;
                                            fold_tier_b<NB, Id>(S, Q, A2, longm, PG_LANE);
                                            if (A2.m1 == 0x12345u) A.m1 = A2.m2;
                                        }
#endif
                                        fold_tier_b<NB, Id>(S, Q, A, longm, lane);
                                        far_update(origin, nullptr, r == 0 ? 1 : (r == 1 ? 3 : 15));
                                    }
                                    if (far_max + close_max >= len) {        // goodFarEndFound
                                        done = true;
                                        ps = rs[r];
                                        pe = re[r];
                                        break;
                                    }
                                }
                                nsurv_eval = S.nsurv;
                            }
                        }
                    }
                    PG_STOP_AT(S, 31);
                }
#endif
                int span = 64 << (2 * r_first);
                for (int r = r_first; r <= k_mri && !done; r++, span *= 4) {
                    int s, e;
                    range_of(span, s, e);
                    if (s < e) {
                        scan_range<NB, NS, Id, !DEF>(ref, S, Q, A, chr_wo, g0, s, e, emax, ps, pe, origin, 0u, PG_LANE, true,
                                           cacheF, cacheB, cache_valid);
                        PG_STOPPED(S);
                        if (ps < pe) {
                            ps = s < ps ? s : ps;
                            pe = e > pe ? e : pe;
                        } else {
                            ps = s; pe = e;
                        }
                    }
                    // an evaluation can only differ from the previous one if candidates were folded since; an
                    // empty state yields no point ("NumberOfHits == 0" leaves UP_Far untouched, farend_searcher.cpp:87)
                    if (S.nsurv != nsurv_eval) {
                        nsurv_eval = S.nsurv;
                        far_update(origin, nullptr, 15);
                    }
                    if (far_max + close_max >= len) break;       // goodFarEndFound
                }
                far_bases += (pe - ps) + 2 * len;
            }
            alg += (u32)(3 * far_bases + 96 * n_far);
        }
    }
    PG_STOP_AT(S, 32);
    PG_T(S, 9);
    alg = (alg + 4u) >> 3;
    if (PG_LANE == 0) {      // (the lane index again: kept from the start of the read it is a VGPR the compiler parks in scratch)
        uint4 *op = (uint4 *)record_ptr<5>(KA(B, out), rid);
        if (do_close) {
            op[0] = make_uint4(close_base, (u32)n_close, far_base, (u32)n_far);
#ifdef PG_DIAG
            op[1] = make_uint4(close_last, (u32)close_max | ((u32)rc_out << 16), alg, S.dg);
#else
            op[1] = make_uint4(close_last, (u32)close_max | ((u32)rc_out << 16), alg, (u32)S.nsurv_total);
#endif
        } else {
            u32 *o = (u32 *)op;
            o[2] = far_base;
            o[3] = (u32)n_far;
            o[6] = alg;                                  // the far-end launch adds to the close-end launch
            o[7] += (u32)S.nsurv_total;
        }
    }
}
#undef PG_LANE

template <int PB>
__device__ __forceinline__ void pack_block(const PgSoaIn &a, PgInRec *in, const u32 lo, const u32 r0, const u32 nb, const u32 lane);

// Persistent 64-thread workgroups.  The reads of the launch are split into PG_N_XCD contiguous parts; a
// workgroup (which the dispatcher places on XCD blockIdx % 8) claims PG_CLAIM reads at a time from its own
// part's counter and moves on to the next part when that one is exhausted.
template <int NB, int NS, typename Id, int mode, bool DEF>
__global__ __launch_bounds__(WAVE, PG_WAVES(NB, Id)) void pg_search_kernel(PgDevRef ref, PgDevParams prm,
                                                         PgDevBatch B, uint32_t max_len, uint32_t levels)
{
    __shared__ Lds<NB, Id> lds;
    const int lane = threadIdx.x;
    if (PG_WIN_DYN_BYTES(NB) != 0u) {
        // the window's dynamic tail must begin where the static LDS object ends (see Lds::win)
        extern __shared__ uint4 pg_dyn_lds[];
        if ((const char *)pg_dyn_lds != (const char *)&lds + sizeof(lds)) __builtin_trap();
    }
    if (PG_MM_IN_WIN(NB)) {
        for (int w = lane; w < 16 * NB + 16; w += WAVE) {
            u32 v = 0u;
#pragma unroll
            for (int k = 0; k < 4; k++) v |= (u32)max_mismatch_at(prm.mm_bp, 4 * w + k) << (8 * k);
            lds.win[w].w = v;
        }
    } else
        for (int L = lane; L < 64 * NB + 64; L += WAVE) lds.mm_tab[L] = (uint8_t)max_mismatch_at(prm.mm_bp, L);
    if (lane < PG_CHR_TAB_N(NB) && lane < ref.n_chr) {
        const u64 wo = ref.chr_word_off[lane];
        lds.chr_tab[lane] = make_uint2((u32)wo, (wo >> 32) == 0ull ? ref.chr_size[lane] : 0u);
    }
    PG_SYNC();
    Search S;
    S.queue = lds.queue;
    S.win = lds.win;
    S.bufA = lds.bufA;
    S.bufB = lds.bufB;
    S.hdrB = lds.hdrB;
    S.ringB = lds.ringB;
    S.accB = lds.accB;
    S.mm_tab = lds.mm_tab;
    S.chr_tab = lds.chr_tab;
    u64 *qplanes = lds.qp;                        // [0]: forward, [1]: reversed consumption order
    if (lane < 8 * NB) qplanes[lane] = 0ull;      // (blocks beyond the batch's plane layout are never written)
#ifdef PG_TIMING
    S.t_acc = lds.t_acc;
    S.t_last = &lds.t_last;
    if (lane == 0) {
        for (int k = 0; k < 12; k++) S.t_acc[k] = 0u;
        *S.t_last = __builtin_readcyclecounter();
    }
    S.t_base = 1;
#endif

    // Reads claimed per atomic (PgDevBatch::claim, worked out by pg_launch_search): a workgroup's share of the launch in the fewest
    // equal claims of at most PG_CLAIM reads.  The host
    // launches one workgroup per PG_CLAIM reads up to the chip's resident slots, so up to 57 k reads every wave takes exactly one claim
    // of eight; between that and a few hundred thousand reads the share is 8..64 reads and claims of exactly eight would leave some
    // waves a whole claim more than others (100 000 reads: 14 per wave = two claims of seven).  Measured, seven waves per SIMD:
    // 50 000 reads 0.262 ms with claims of two, 0.23 with eight; 100 000: 0.403 -> 0.36; 5000 reads on 633 workgroups 0.106 either way,
    // and 0.125 on 5008 workgroups of one read each -- a small launch pays for the NUMBER of workgroups.  A wave that looks at the
    // other parts' counters before claiming from them (the walk over the eight parts at the end of a launch is one atomic per wave
    // and address) gained nothing.
    // (Claims of ONE read for the last round and a half of a launch, to shorten its tail, were measured and rejected: a claim
    // is a dependent chain atomic -> records -> first window, 2 us that eight reads share -- 262 144 reads 0.94 -> 0.98 ms.)
    // Nothing but `part` and `tried` lives from one claim to the next: the launch's size comes from the kernarg segment again.
    // (Quarter claims for the last rounds of a launch that packs in place -- the waves finish spread over one claim's duration -- were
    // measured and rejected, round 6: a claim is a dependent chain atomic -> pack (three HBM round trips) -> records, and the short
    // claims cost more than the shorter tail gives back: 2 M reads at -x 5 22.16 -> 22.24 ms, 2 M at -x 2 4.55 -> 4.58 / 4.70 ms
    // for two / four rounds.  Claims of eight, as without the pack: sixteen and more lose at -x 5, where a read takes 79 us.)
    uint32_t part = blockIdx.x % PG_N_XCD, tried = 0;
    while (tried < PG_N_XCD) {
        const uint32_t n = KA(B, n_reads);
#ifdef PG_FORCE_CLAIM
        const uint32_t claim = PG_FORCE_CLAIM;
#else
        const uint32_t claim = KA(B, claim);              // (pg_launch_search: the share of a workgroup in the fewest equal claims <= PG_CLAIM)
#endif
        const uint32_t per = n / PG_N_XCD;
        const uint32_t lo = part * per, hi = part + 1 == PG_N_XCD ? n : lo + per;
        // (the single-lane atomics by hand: the compiler wraps an atomicAdd in its wave-reduction form -- exec juggling, mbcnt,
        // bcnt, a multiply -- some 25 instructions each)
        uint32_t got = 0;
        uint32_t *ctr = KA(B, work_ctr) + part * 16u;
        if (lane == 0) asm volatile("global_atomic_add %0, %1, %2, %3 sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(got) : "v"(0u), "v"(claim), "s"(ctr) : "memory");
        got = (u32)uni((int)got);
        if (got >= hi - lo) {                             // this part is exhausted
            part = part + 1 == PG_N_XCD ? 0 : part + 1;
            tried++;
            continue;
        }
        const uint32_t first = lo + got, end = hi - first < claim ? hi : first + claim;
        PG_T(S, 11);
        uint32_t no_touch = ~0u;                          // the read that must not touch its successor's record (none)
        {
            // PACK IN PLACE (PgDevBatch::soa set; any mode -- the two seams' launches pack their reads too): the wave builds the records and bit planes of its claim from the SoA arrays
            // before it searches them -- the pack kernel's body on the claim's reads.  A streaming transpose (HBM-bound on its own,
            // 3 TB/s) inside a kernel that is bound by instruction issue and leaves 90 % of the HBM bandwidth idle: ~30 instructions
            // per read instead of a launch of its own in front of this one.  What the wave wrote it reads back itself, through the
            // caches of its own CU and XCD (scalar loads of the records, vector loads of the planes): the stores' completion is all
            // there is to wait for.  (The host sets soa only when the batch's plane layout is this kernel's: plane_blocks == NB.)
            const PgSoaIn *soa = KA(B, soa);
            if (soa) {
                const PgSoaIn a = *soa;
                pack_block<NB>(a, const_cast<PgInRec *>(KA(B, in)), KA(B, first_read) + first, 0u, end - first, (u32)lane_now());
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // (a read touches the next read's record while it waits for its first window: the claim's last read would bring
                // the line of a record nobody has written yet into the scalar cache, where its owner might then find it)
                no_touch = end - 1u;
            }
        }
        // run-pool slots of the claim's reads: one atomic per claim
        u32 res = 0u;
        {
            const u32 shard = blockIdx.x & (PG_POOL_SHARDS - 1u);
            uint32_t *cur = KA(B, pool_used) + shard * 16u;
            if (lane == 0) asm volatile("global_atomic_add %0, %1, %2, %3 sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(res) : "v"(0u), "v"(claim * PG_RESERVE), "s"(cur) : "memory");
            res = (u32)uni((int)res);
            const u32 res_fits = (u64)res + (u64)(claim * PG_RESERVE) <= (u64)KA(B, pool_shard_cap) ? 1u : 0u;
            res += shard * KA(B, pool_shard_cap);
            for (uint32_t i = first; i < end; i++)
                search_read<NB, NS, Id, mode, DEF>(ref, prm, B, S, qplanes, KA(B, first_read) + i, i != no_touch ? 1 : 0, lane_now(),
                                          res + (i - first) * PG_RESERVE, res_fits);
            PG_T(S, 10);
        }
    }
#ifdef PG_TIMING
    if (lane == 0) {
        u64 *dg = (u64 *)(KA(B, work_ctr) + PG_WORK_CTRS * 16u);
        for (int k = 0; k < 12; k++) atomicAdd((unsigned long long *)(dg + k), (unsigned long long)S.t_acc[k]);
    }
#endif
}

// EXACT kernel (search_read<..., EXACT = true>): the reads the pack kernel listed because they hold a character outside ACGTN
// (PgDevBatch::exact_list / exact_count), searched again with the reference's read-shortening semantics; their output records and runs
// replace what the launch before wrote for them.  Same parameter list as pg_search_kernel (KA()).  One instantiation per mode: eight
// blocks (any read the ABI accepts), five counter slices (any number of levels), 64-bit candidate ids (any window), generic
// parameters -- speed is no concern for a handful of reads.  A small fixed grid walks the list; entries outside the launch's read
// range [first_read, first_read + n_reads) are another chunk's.
template <int mode>
__global__ __launch_bounds__(WAVE, PG_WAVES(8, u64)) void pg_search_exact_kernel(PgDevRef ref, PgDevParams prm, PgDevBatch B,
                                                                               uint32_t max_len, uint32_t levels)
{
    constexpr int NB = 8;
    typedef u64 Id;
    __shared__ Lds<NB, Id> lds;
    const int lane = threadIdx.x;
    for (int L = lane; L < 64 * NB + 64; L += WAVE) lds.mm_tab[L] = (uint8_t)max_mismatch_at(prm.mm_bp, L);
    PG_SYNC();
    Search S;
    S.queue = lds.queue;
    S.win = lds.win;
    S.bufA = lds.bufA;
    S.bufB = lds.bufB;
    S.hdrB = lds.hdrB;
    S.ringB = lds.ringB;
    S.accB = lds.accB;
    S.mm_tab = lds.mm_tab;
    S.chr_tab = lds.chr_tab;
    u64 *qplanes = lds.qp;
    if (lane < 8 * NB) qplanes[lane] = 0ull;
#ifdef PG_TIMING
    S.t_acc = lds.t_acc;
    S.t_last = &lds.t_last;
    S.t_base = 1;
#endif
    const uint32_t n_list = *KA(B, exact_count), first = KA(B, first_read), n = KA(B, n_reads);
    const uint32_t *list = KA(B, exact_list);
    for (uint32_t i = blockIdx.x; i < n_list; i += gridDim.x) {
        const uint32_t rid = (u32)uni((int)list[i]);
        if (rid - first >= n) continue;
        // (a read that was listed has rows of its own in the run pool: one reservation per read)
        u32 res = 0u;
        const u32 shard = blockIdx.x & (PG_POOL_SHARDS - 1u);
        uint32_t *cur = KA(B, pool_used) + shard * 16u;
        if (lane == 0) res = atomicAdd(cur, (u32)PG_RESERVE);
        res = (u32)uni((int)res);
        const u32 res_fits = (u64)res + (u64)PG_RESERVE <= (u64)KA(B, pool_shard_cap) ? 1u : 0u;
        res += shard * KA(B, pool_shard_cap);
        search_read<NB, 5, Id, mode, false, true>(ref, prm, B, S, qplanes, rid, 0, lane_now(), res, res_fits);
    }
}
extern "C" int pg_launch_search_exact(const PgDevRef *ref, const PgDevParams *prm, const PgDevBatch *batch, int mode,
                                      uint32_t max_len, uint32_t levels, void *stream)
{
    if (batch->n_reads == 0 || !batch->exact_list) return 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(64), block(WAVE);
    if (mode == PG_MODE_BOTH)
        hipLaunchKernelGGL((pg_search_exact_kernel<PG_MODE_BOTH>), grid, block, 0, st, *ref, *prm, *batch, max_len, levels);
    else if (mode == PG_MODE_CLOSE)
        hipLaunchKernelGGL((pg_search_exact_kernel<PG_MODE_CLOSE>), grid, block, 0, st, *ref, *prm, *batch, max_len, levels);
    else
        hipLaunchKernelGGL((pg_search_exact_kernel<PG_MODE_FAR>), grid, block, 0, st, *ref, *prm, *batch, max_len, levels);
    return (int)hipGetLastError();
}

// Self-check of KA(): a kernel with pg_search_kernel's parameter list compares what KA() / KAP() fetch from the kernarg segment at
// PgKArgs' offsets with the by-value arguments the compiler passes (one launch per context, pg_debug_kargs_check); a mismatch --
// a changed parameter list, an ABI that lays the segment out differently -- is reported instead of searched with.
static_assert(offsetof(PgKArgs, ref) == 0 && offsetof(PgKArgs, prm) == sizeof(PgDevRef) && alignof(PgDevRef) == 8 && alignof(PgDevParams) == 4 &&
              offsetof(PgKArgs, B) == ((sizeof(PgDevRef) + sizeof(PgDevParams) + 7) & ~(size_t)7) &&
              offsetof(PgKArgs, max_len) == offsetof(PgKArgs, B) + sizeof(PgDevBatch) && offsetof(PgKArgs, levels) == offsetof(PgKArgs, max_len) + 4,
              "PgKArgs mirrors the kernarg segment of pg_search_kernel(PgDevRef, PgDevParams, PgDevBatch, uint32_t, uint32_t)");
__global__ void pg_kargs_check_kernel(PgDevRef ref, PgDevParams prm, PgDevBatch B, uint32_t max_len, uint32_t levels)
{
    u32 bad = 0u;
#define PG_CHK(obj, m) bad += (u64)KA(obj, m) != (u64)obj.m ? 1u : 0u
    PG_CHK(ref, lo); PG_CHK(ref, hi); PG_CHK(ref, nn); PG_CHK(ref, chr_word_off); PG_CHK(ref, chr_size); PG_CHK(ref, n_chr);
    PG_CHK(prm, max_range_index); PG_CHK(prm, add_mm); PG_CHK(prm, min_perfect); PG_CHK(prm, min_close); PG_CHK(prm, spacer);
    PG_CHK(B, n_reads); PG_CHK(B, first_read); PG_CHK(B, in); PG_CHK(B, out); PG_CHK(B, seq); PG_CHK(B, planes); PG_CHK(B, plane_blocks);
    PG_CHK(B, bd); PG_CHK(B, pool); PG_CHK(B, pool_shard_cap); PG_CHK(B, pool_used); PG_CHK(B, work_ctr); PG_CHK(B, claim);
    PG_CHK(B, exact_list); PG_CHK(B, exact_count); PG_CHK(B, thr_tab); PG_CHK(B, soa);
#undef PG_CHK
    {
        const u32 *glo, *ghi, *gnn;
        karg_load3<(int)offsetof(PgKArgs, ref.lo)>(glo, ghi, gnn);
        bad += (glo != ref.lo ? 1u : 0u) + (ghi != ref.hi ? 1u : 0u) + (gnn != ref.nn ? 1u : 0u);
    }
    bad += karg_load<uint32_t>((int)offsetof(PgKArgs, max_len)) != max_len ? 1u : 0u;
    bad += karg_load<uint32_t>((int)offsetof(PgKArgs, levels)) != levels ? 1u : 0u;
    if (threadIdx.x == 0) *B.work_ctr = bad;      // (B.work_ctr: the caller's scratch word)
}
extern "C" int pg_debug_kargs_check(const PgDevRef *ref, const PgDevParams *prm, const PgDevBatch *batch, uint32_t max_len, uint32_t levels,
                                    uint32_t *scratch_dev, void *stream)
{
    PgDevBatch b = *batch;
    b.work_ctr = scratch_dev;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pg_kargs_check_kernel, dim3(1), dim3(WAVE), 0, st, *ref, *prm, b, max_len, levels);
    uint32_t bad = ~0u;
    if (hipMemcpyAsync(&bad, scratch_dev, sizeof bad, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
    return (int)bad;
}

// ---------------------------------------------------------------------------------
template <int NB, int NS, typename Id, bool DEF>
static void launch_modes(const PgDevRef *ref, const PgDevParams *prm, const PgDevBatch *batch, int mode,
                         uint32_t max_len, uint32_t levels, hipStream_t st, unsigned lds_pad, dim3 grid, dim3 block)
{
    // close end + far end in one launch; PG_SPLIT_LAUNCH=1 runs the two seams as separate launches
    const bool fused = !pg_env_switches()->split_launch;
    if (mode == PG_MODE_BOTH && fused) {
        hipLaunchKernelGGL((pg_search_kernel<NB, NS, Id, PG_MODE_BOTH, DEF>), grid, block, lds_pad, st,
                           *ref, *prm, *batch, max_len, levels);
        return;
    }
#ifdef PG_ONLY_BENCH
    abort();          // experiment builds (scripts/build_variant.sh -DPG_ONLY_BENCH): only the fused kernel exists
#else
    if (mode & PG_MODE_CLOSE)
        hipLaunchKernelGGL((pg_search_kernel<NB, NS, Id, PG_MODE_CLOSE, DEF>), grid, block, lds_pad, st,
                           *ref, *prm, *batch, max_len, levels);
    if (mode & PG_MODE_FAR) {
        if (mode & PG_MODE_CLOSE) (void)hipMemsetAsync(batch->work_ctr, 0, PG_N_XCD * 16u * sizeof(uint32_t), st);
        hipLaunchKernelGGL((pg_search_kernel<NB, NS, Id, PG_MODE_FAR, DEF>), grid, block, lds_pad, st,
                           *ref, *prm, *batch, max_len, levels);
    }
#endif
}

template <int NB, int NS, typename Id>
static void launch_ns(const PgDevRef *ref, const PgDevParams *prm, const PgDevBatch *batch, int mode,
                   uint32_t max_len, uint32_t levels, hipStream_t st, unsigned lds_pad)
{
    // a few resident workgroups per CU (the launch is persistent); more than fit simply queue up and find
    // the remaining chunks
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n_cu = p.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    const uint32_t chunks = (batch->n_reads + PG_CLAIM - 1) / PG_CLAIM + PG_N_XCD;
    const uint32_t want = (uint32_t)n_cu * 4u * (uint32_t)PG_WAVES(NB, Id);
    dim3 grid(chunks < want ? chunks : want), block(WAVE);
    // Reads claimed per atomic: a workgroup's share of the launch in the fewest equal claims of at most PG_CLAIM reads (the kernel
    // used to divide twice per claim for this: ~50 instructions)
    PgDevBatch with_claim = *batch;
    {
        const uint32_t per_wg = (batch->n_reads + grid.x - 1u) / grid.x, n_claims = (per_wg + PG_CLAIM - 1u) / PG_CLAIM;
        with_claim.claim = per_wg >= 8u * PG_CLAIM ? PG_CLAIM : (per_wg + n_claims - 1u) / n_claims;
        if (with_claim.claim == 0u) with_claim.claim = 1u;
        // pack in place: the pack's per-read lane (phase A) costs the same ~100 instructions for a claim of 8 or 16 reads, but a longer
        // claim lengthens the launch's tail where a wave is busy with it for long.  Measured (ms, pack launch + search launch -> one
        // launch with claims of 8 / 16): 10 M x 100 bp 22.25 -> 21.91 / 21.68; 10 M x 150 bp 25.58 -> 25.24 / 25.07; 2 M 4.60 -> 4.53 / 4.52;
        // 1 M 2.43 -> 2.38 / 2.42; 2 M at -x 5 (79 us per read) 21.95 -> 21.88 / 22.29.  Sixteen where a wave takes at least 32 such
        // claims and the ranges are the default ones, eight otherwise.  With claims of eight the one launch was never slower than
        // the two at any size measured (20 000 reads 0.222 -> 0.197 ms, 50 000 0.303 -> 0.280, 262 144 0.814 -> 0.782, 500 000 1.345 -> 1.331).
        if (batch->soa) {
            const uint32_t dflt = prm->max_range_index <= 2 && per_wg >= 32u * PG_PACK_CLAIM_LONG ? PG_PACK_CLAIM_LONG : PG_PACK_CLAIM;
            const uint32_t pc = pg_env_switches()->pack_claim ? pg_env_switches()->pack_claim : dflt;
            with_claim.claim = pc > 64u ? 64u : pc;
        }
    }
    batch = &with_claim;
    if (PG_WIN_DYN_BYTES(NB) != 0u && prm->max_range_index >= 3) lds_pad += PG_WIN_DYN_BYTES(NB);   // two chunks per fill need their LDS
    // Pindel's default parameters have kernels of their own (see PRM): up to 16 mismatch levels, 32-bit candidate ids
    constexpr bool HAS_DEF = NS <= 4 && sizeof(Id) == 4;
#ifdef PG_DEF_X_RUNTIME
    const bool def = HAS_DEF && !pg_env_switches()->generic_kernels &&
#else
    const bool def = HAS_DEF && !pg_env_switches()->generic_kernels && prm->max_range_index == PG_DEF_MAX_RANGE_INDEX &&
#endif
                     prm->add_mm == PG_DEF_ADD_MM && prm->min_perfect == PG_DEF_MIN_PERFECT && prm->min_close == PG_DEF_MIN_CLOSE &&
                     prm->spacer == PG_DEF_SPACER;
    if constexpr (HAS_DEF) {
        if (def) {
            launch_modes<NB, NS, Id, true>(ref, prm, batch, mode, max_len, levels, st, lds_pad, grid, block);
            return;
        }
    }
    launch_modes<NB, NS, Id, false>(ref, prm, batch, mode, max_len, levels, st, lds_pad, grid, block);
}

// the seed filter's counter width follows the batch's largest number of mismatch levels (validate_and_measure)
template <int NB, typename Id>
static void launch(const PgDevRef *ref, const PgDevParams *prm, const PgDevBatch *batch, int mode,
                   uint32_t max_len, uint32_t levels, hipStream_t st, unsigned lds_pad)
{
    if (levels <= 8) launch_ns<NB, 3, Id>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
#ifdef PG_ONLY_BENCH
    else abort();     // experiment builds: default parameters only
#else
    else if (levels <= 16) launch_ns<NB, 4, Id>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
    else launch_ns<NB, 5, Id>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
#endif
}

// ---------------------------------------------------------------------------------
// Packed per-read records <-> the SoA arrays of the C ABI (pg_device.h).
//
// pg_pack_kernel: ASCII reads -> the 32-byte input record + the bit planes the search kernel reads (PgDevBatch::planes:
// u64[2 orientations][4 planes][PB blocks] per read; code A=0 C=1 G=2 T=3, N and "other" (matches nothing) as planes of
// their own; orientation 1 = the read from its last base).  A streaming transpose, HBM-bound by nature
// (len + 27 bytes in, 64 PB + 32 out per read), so it is laid out for bandwidth, not for a wave per block:
//   * 8 PB consecutive lanes own one read, lane L its bases [8 L, 8 L + 8): the wave's loads walk the concatenated
//     sequence buffer in 8-byte steps (three aligned dword loads + v_alignbyte per lane; any read alignment), the
//     bases are classified four at a time (v_perm + one SWAR compare, exact for every byte value), no ballots;
//   * a 4 x 4 byte transpose inside every lane quad (two quad shuffles) turns "8 bases x 4 planes" per lane into
//     "32 bases of ONE plane" per lane: lane (D, p) = dword D of plane p, so the read's 8 PB lanes write its forward
//     planes as 8 PB dwords that tile 32 PB contiguous bytes;
//   * the reversed orientation is the same bit string mirrored: dword D of it is a 32-bit window of the forward plane
//     at bit len - 32 - 32 D, bit-reversed -- two lane shuffles + v_alignbit + v_bfrev, no second pass over the bases.
// A wave takes 64 CONSECUTIVE reads at a time.  What bounds this kernel is neither ALU (-8 % with the classification removed)
// nor bytes in flight per wave (unrolling does nothing) nor the number of memory instructions, but the number of memory
// REQUESTS (cache lines touched) per byte moved -- measured on 10 M x 100 bp, profiles/r04/pack_kernel.txt: four reads
// per wave with every lane fetching its own read's fields touched ~20 lines for 476 useful bytes and stopped at 2.6 TB/s.
//   phase A  lane j = read j of the 64: its two offsets (one 16-byte load), position / chromosome / insert size / strand
//            (four full-width coalesced loads), the two table entries of its length -> the whole 32-byte record from one
//            lane (the wave writes 2 KB contiguous)
//   phase B  RPW reads per step (8 PB lanes each, lane L = bases [8 L, 8 L + 8)): offset and length of the step's reads come
//            from phase A's lanes by ds_bpermute, the bases as one 12-byte load per lane (the wave walks the concatenated
//            sequence buffer), PG_PACK_UNROLL steps' loads in flight together.
#ifndef PG_PACK_UNROLL
#define PG_PACK_UNROLL 2
#endif
#ifndef PG_PACK_GRID
#define PG_PACK_GRID 65536u
#endif
static_assert(sizeof(PgOutRec) == 32, "record_ptr<5>");
static_assert(sizeof(PgInRec) == 128 && offsetof(PgInRec, prog) == 64 && offsetof(PgInRec, w1s) == 0 && offsetof(PgInRec, isz) == 4 && offsetof(PgInRec, stage_s) == 8 &&
              offsetof(PgInRec, stage_e) == 12 && offsetof(PgInRec, chr_wo_lo) == 16 && offsetof(PgInRec, lenf) == 24 && offsetof(PgInRec, lvl) == 28 &&
              offsetof(PgInRec, depth) == 32 && offsetof(PgInRec, jmask0) == 36 && offsetof(PgInRec, ro) == 40 && offsetof(PgInRec, chr) == 44 &&
              offsetof(PgInRec, chr_size) == 48 && offsetof(PgInRec, bd_cnt) == 52 && offsetof(PgInRec, bd_off) == 56 && offsetof(PgInRec, jmask1) == 60,
              "pg_pack_kernel writes PgInRec as four uint4; search_read reads it as dwords 0..7, 8..11, 12..15");
struct PgDw3 { u32 x, y, z; };
// One wave, reads [lo + r0, lo + r0 + nb) of the SoA arrays, nb <= 64 (pg_pack_kernel: 64 at a time; pg_search_kernel with
// PgDevBatch::soa set: the reads of a claim, just before it searches them).
template <int PB>
__device__ __forceinline__ void pack_block(const PgSoaIn &a, PgInRec *in, const u32 lo, const u32 r0, const u32 nb, const u32 lane)
{
    constexpr u32 LPR = 8u * PB;                       // lanes per read in phase B
    constexpr u32 RPW = 64u / LPR;                     // reads per step (PB = 3: two reads on 48 lanes)
    constexpr u32 STEPS = (64u + RPW - 1u) / RPW;
    constexpr u32 U = PG_PACK_UNROLL < STEPS ? PG_PACK_UNROLL : STEPS;
    static_assert(STEPS % U == 0, "unroll must divide the steps");
    const u32 slot = lane / LPR, L = lane % LPR;
    const u32 D = L >> 2, pl = L & 3u, idx0 = 8u * L;
    {
        // ---- phase A: lane = read.  The whole 64-byte record from one lane (the wave writes 4 KB contiguous): everything the search
        // kernel would otherwise derive per read on its scalar unit (PgInRec, pg_device.h)
        u32 so_lo = 0u, so_hi = 0u, len = 0u;
        if (lane < nb) {
            const u32 ii = lo + r0 + lane;
            uint4 o;
            __builtin_memcpy(&o, __builtin_assume_aligned(a.seq_off + ii, 8), 16);
            so_lo = o.x;
            so_hi = o.y;
            len = o.z - o.x;                           // (a read is shorter than 2^32 bases: the low words suffice)
            const u32 lc = len < 512u ? len : 511u;
            const int isz = (int)a.isz[ii];
            const int apos = a.pos[ii] + (int32_t)a.spacer;
            const u32 strand = a.strand[ii];
            const int chr = a.chr[ii];
            const u64 wo = a.chr_word_off[chr];
            // what follows from the read's length alone: mismatch levels, CheckMismatches' threshold, the seed filter's depths, bounds
            // and masks (pg_len_rec, one table per context; two 16-byte loads)
            uint4 lt0, lt1;
            __builtin_memcpy(&lt0, __builtin_assume_aligned(a.len_tab + lc, 16), 16);
            __builtin_memcpy(&lt1, __builtin_assume_aligned((const char *)(a.len_tab + lc) + 16, 16), 16);
            u32 flags = 0u;
            const bool plus = strand == (u32)'+';
            if ((int)len - 1 >= a.min_close && (plus || strand == (u32)'-')) flags |= PG_RF_CLOSE_OK;
            if (plus) flags |= PG_RF_PLUS;
            const bool shared = isz > 0 && 3 * isz <= (int)PG_CHUNK;
            if (shared) flags |= PG_RF_SHARED_GRID;
            if (len) {
                const u64 at = (u64)so_lo | ((u64)so_hi << 32);
                const u32 c0 = a.seq[at], c1 = a.seq[at + len - 1u];
                if (c0 == 'A' || c0 == 'C' || c0 == 'G' || c0 == 'T') flags |= PG_RF_FIRST_OK_FWD;
                if (c1 == 'A' || c1 == 'C' || c1 == 'G' || c1 == 'T') flags |= PG_RF_FIRST_OK_REV;
            }
            const int w1s = plus ? apos - isz : apos - 2 * isz;
            // the first window fill: attempt 0's window (R = 0), or the whole R = 1 window when that fits one chunk
            int ss = 0, se = 0;
            if (flags & PG_RF_CLOSE_OK) {
                const int s1 = shared ? w1s : w1s + isz, e1 = shared ? w1s + 3 * isz : w1s + 2 * isz;
                if (s1 < e1) {
                    ss = s1;
                    se = e1 < s1 + (int)PG_CHUNK ? e1 : s1 + (int)PG_CHUNK;
                }
            }
            uint4 q0, q1, q2, q3;
            q0.x = (u32)w1s;
            q0.y = (u32)isz;
            q0.z = (u32)ss;
            q0.w = (u32)se;
            q1.x = (u32)wo;
            q1.y = (u32)(wo >> 32);
            q1.z = (len & 0xffffu) | (flags << 16);
            q1.w = lt0.x;               // PgLenRec: lvl, depth, jmask0, ro | jmask1
            q2.x = lt0.y;
            q2.y = lt0.z;
            q2.z = lt0.w;
            q2.w = (u32)chr;
            q3.x = a.chr_size[chr];
            q3.y = q3.z = 0u;
            q3.w = lt1.x;
            if (a.bd_off) {
                uint4 w;
                __builtin_memcpy(&w, __builtin_assume_aligned(a.bd_off + ii, 8), 16);
                q3.z = w.x;
                q3.y = w.z - w.x;
            }
            uint4 *dst = (uint4 *)(in + ii);
            dst[0] = q0;
            dst[1] = q1;
            dst[2] = q2;
            dst[3] = q3;
        }
        // ---- phase B: 8 PB lanes = one read
        for (u32 s = 0; s < STEPS && s * RPW < nb; s += U) {
            u32 d0[U], d1[U], d2[U], ln[U], sh[U];
            bool act[U];
#pragma unroll
            for (u32 u = 0; u < U; u++) {
                const u32 src = (s + u) * RPW + slot;
                act[u] = slot < RPW && src < nb;
                const u32 olo = (u32)__shfl((int)so_lo, (int)(src & 63u)), ohi = (u32)__shfl((int)so_hi, (int)(src & 63u));
                ln[u] = (u32)__shfl((int)len, (int)(src & 63u));
                if (!act[u]) ln[u] = 0u;
                const u64 A = ((u64)olo | ((u64)ohi << 32)) + idx0;
                sh[u] = (u32)A & 3u;
                d0[u] = d1[u] = d2[u] = 0u;
                if (idx0 < ln[u]) {
                    // (the buffer is padded: up to 11 bytes past the read are touched; a multi-dword load only needs dword alignment)
                    PgDw3 t;
                    __builtin_memcpy(&t, __builtin_assume_aligned((const u32 *)a.seq + (A >> 2), 4), 12);
                    d0[u] = t.x;
                    d1[u] = t.y;
                    d2[u] = t.z;
                }
            }
#pragma unroll
            for (u32 u = 0; u < U; u++) {
                const u32 w0 = __builtin_amdgcn_alignbyte(d1[u], d0[u], sh[u]), w1 = __builtin_amdgcn_alignbyte(d2[u], d1[u], sh[u]);
                const u32 nv = ln[u] > idx0 ? (ln[u] - idx0 < 8u ? ln[u] - idx0 : 8u) : 0u;
                const u32 vm0 = nv >= 4u ? 0x80808080u : (0x80808080u & ((1u << (8u * nv)) - 1u));
                const u32 vm1 = nv >= 8u ? 0x80808080u : (nv > 4u ? (0x80808080u & ((1u << (8u * (nv - 4u))) - 1u)) : 0u);
                // Eight bases -> P: byte j = plane j (code bit 0, code bit 1, N, other) of bases [8 L, 8 L + 8), bit k = base k.
                // (Round 6, when this code began to run inside the search kernel -- pack in place -- where every issue slot counts:
                // ten SWAR compares and eight multiply-gathers, 106 instructions with eight quarter-rate multiplies, became 55.)
                // Per word of four characters: bits 1-2 of the ASCII code tell A / C / T / G apart (0 1 2 3); v_perm turns them back
                // into the letter they stand for, and a character IS one of ACGT iff it equals that letter -- exact for all 256 byte
                // values; N by one compare; the rest is "other".  The four flags of a base make a nibble (code bit 0 = C|T =
                // bit 0 ^ bit 1 of the two-bit code, code bit 1 = G|T = its bit 1), two words make "byte k = bases k and k + 4", and
                // three delta swaps (distances 7, 14, 21) transpose both 4 x 4 bit matrices at once.
                u32 P;
                {
                    auto nibbles = [](u32 w, u32 vm) {
                        const u32 code = (w >> 1) & 0x03030303u;
                        const u32 x = w ^ __builtin_amdgcn_perm(0x47544341u, 0x47544341u, code);
                        const u32 V = ~((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x)) & vm;           // 0x80 where the byte is one of ACGT (and valid)
                        const u32 xn = w ^ 0x4e4e4e4eu;
                        const u32 N = ~((((xn & 0x7f7f7f7fu) + 0x7f7f7f7fu) | xn)) & vm;
                        const u32 O = vm & ~(V | N);
                        const u32 c = code & ((V >> 7) | (V >> 6));
                        return ((c ^ (c >> 1)) & 0x01010101u) | (c & 0x02020202u) | (N >> 5) | (O >> 4);
                    };
                    u32 Z = nibbles(w0, vm0) | (nibbles(w1, vm1) << 4);
                    u32 t;
                    t = (Z ^ (Z >> 7)) & 0x884422u;  Z ^= t | (t << 7);
                    t = (Z ^ (Z >> 14)) & 0x8844u;   Z ^= t | (t << 14);
                    t = (Z ^ (Z >> 21)) & 0x88u;     Z ^= t | (t << 21);
                    P = Z;
                }
                // ---- 4 x 4 byte transpose in the quad: lane (D = L >> 2, p = L & 3) <- dword D of plane p
                const u32 t = (u32)__shfl_xor((int)P, 2);
                const u32 y = (L & 2u) ? ((t >> 16) | (P & 0xffff0000u)) : ((P & 0xffffu) | (t << 16));
                const u32 x = (u32)__shfl_xor((int)y, 1);
                const u32 Fd = (L & 1u) ? (((x >> 8) & 0x00ff00ffu) | (y & 0xff00ff00u)) : ((y & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8));
                // ---- the mirrored orientation: dword D = bits [q, q + 32) of the forward plane, reversed; q = len - 32 - 32 D
                const int q = (int)ln[u] - 32 - 32 * (int)D;
                const int dF = q >> 5;                                        // (arithmetic: floor)
                const u32 sb = (u32)q & 31u;
                const int j0 = dF < 0 ? 0 : (dF >= 2 * PB ? 2 * PB - 1 : dF), j1 = dF + 1 < 0 ? 0 : (dF + 1 >= 2 * PB ? 2 * PB - 1 : dF + 1);
                u32 fl = (u32)__shfl((int)Fd, (int)((slot * LPR + 4u * (u32)j0 + pl) & 63u));
                u32 fh = (u32)__shfl((int)Fd, (int)((slot * LPR + 4u * (u32)j1 + pl) & 63u));
                if (dF < 0 || dF >= 2 * PB) fl = 0u;
                if (dF + 1 < 0 || dF + 1 >= 2 * PB) fh = 0u;
                const u32 Rd = __brev(__builtin_amdgcn_alignbit(fh, fl, sb));
                if (act[u]) {
                    u32 *dst = (u32 *)a.planes + (size_t)(lo + r0 + (s + u) * RPW + slot) * (16u * PB);
                    dst[2u * PB * pl + D] = Fd;
                    dst[2u * PB * (4u + pl) + D] = Rd;
                }
                // ---- the read-order seed filter's symbol programs (PgInRec::prog): the quad D = 0 of a read holds the first 32
                // bases of its planes (lane p = plane p: code bit 0, code bit 1, N, other).  The sixteen program dwords of a read
                // (two orientations x eight groups) are spread over the read's lanes -- lane L builds dword L (and L + 8 when a read
                // has only eight lanes) -- so that the read's 64 bytes go out with ONE store instruction, contiguous.
                {
                    const u32 q0 = slot * LPR;
                    const u32 flo = (u32)__shfl((int)Fd, (int)(q0 & 63u)), fhi = (u32)__shfl((int)Fd, (int)((q0 + 1u) & 63u)),
                              fnn = (u32)__shfl((int)Fd, (int)((q0 + 2u) & 63u));
                    const u32 rlo = (u32)__shfl((int)Rd, (int)(q0 & 63u)), rhi = (u32)__shfl((int)Rd, (int)((q0 + 1u) & 63u)),
                              rnn = (u32)__shfl((int)Rd, (int)((q0 + 2u) & 63u));
                    u32 *rec = (u32 *)(in + (lo + r0 + (s + u) * RPW + slot));
#pragma unroll
                    for (u32 h = 0; h < (LPR >= 16u ? 1u : 2u); h++) {
                        const u32 q = L + 8u * h;                        // program dword: orientation q >> 3, group q & 7
                        if (act[u] && q < 16u) {
                            const u32 g = q & 7u;
                            const bool o1 = q >= 8u;
                            const u32 plo = o1 ? rlo : flo, phi = o1 ? rhi : fhi, pnn = o1 ? rnn : fnn;
                            const u32 j = 3u * g + 1u;
                            // three symbols at once: bit k of (plane >> j) to bit 8 k (x 0x4081, masked: bits 0 / 7 / 14 of the factor)
                            const u32 l3 = (((plo >> j) & 7u) * 0x4081u) & 0x010101u, h3 = (((phi >> j) & 7u) * 0x4081u) & 0x010101u,
                                      n3 = (((pnn >> j) & 7u) * 0x4081u) & 0x010101u;
                            u32 w = l3 | (h3 << 1);
                            if (o1) w ^= 0x030303u;                      // orientation 1's program holds the complement
                            w = (w & ~(n3 * 3u)) | (n3 << 2);            // N: symbol 4
                            w |= 0x30303030u;
                            if (g == 0u) {                               // the seed: base 0
                                u32 x = (plo & 1u) | ((phi & 1u) << 1);
                                if (o1) x ^= 3u;
                                if (pnn & 1u) x = 4u;
                                w |= x << 24;
                            }
                            rec[16u + q] = w;
                        }
                    }
                    // a base that is none of ACGTN among those a program can cover (lane 3 of the quad holds the "other" plane of
                    // both orientations): the read keeps the symbol-by-symbol filter.  Rare; phase A's store of this dword has
                    // completed (the loads of this phase were waited for behind it, and the vector memory counter retires in order).
                    if (act[u] && L == 3u && ((Fd | Rd) & 0x1ffffffu)) atomicAnd(rec + 10, ~PG_RO_OK);
                    // ANY character outside ACGTN anywhere in the read (this lane: 32 bases of the "other" plane): the read goes on the
                    // list of the exact kernel (pg_search_exact_kernel), once -- the flag in the record says who was first
                    if (act[u] && pl == 3u && Fd != 0u && a.exact_list) {
                        const u32 old = atomicOr(rec + 6, (u32)PG_RF_EXACT << 16);
                        if (!(old & ((u32)PG_RF_EXACT << 16))) a.exact_list[atomicAdd(a.exact_count, 1u)] = lo + r0 + (s + u) * RPW + slot;
                    }
                }
            }
        }
    }
}
template <int PB>
__global__ __launch_bounds__(256) void pg_pack_kernel(PgSoaIn a, PgInRec *in, uint32_t lo, uint32_t cnt)
{
    const u32 n_blocks = (cnt + 63u) >> 6, n_waves = gridDim.x * 4u;
    for (u32 g = blockIdx.x * 4u + (threadIdx.x >> 6); g < n_blocks; g += n_waves) {
        const u32 r0 = g << 6;
        pack_block<PB>(a, in, lo, r0, cnt - r0 < 64u ? cnt - r0 : 64u, threadIdx.x & 63u);
    }
}

__global__ void pg_pack_close_kernel(PgSoaOut a, PgOutRec *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PgOutRec r;
    r.close_off = r.close_cnt = r.far_off = r.far_cnt = 0;
    r.close_last = a.close_last[i];
    r.close_max = a.close_max[i];
    r.rc_flag = a.rc_flag[i];
    r.pad = 0;
    r.alg = 0;
    r.reserved = 0;
    out[i] = r;
}

__global__ void pg_unpack_kernel(const PgOutRec *out, PgSoaOut a, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PgOutRec r = out[i];
    a.rc_flag[i] = r.rc_flag;
    a.close_last[i] = r.close_last;
    a.close_max[i] = r.close_max;
    a.close_off[i] = r.close_off;
    a.close_cnt[i] = r.close_cnt;
    a.far_off[i] = r.far_off;
    a.far_cnt[i] = r.far_cnt;
    a.alg[i] = r.alg;
    if (a.cand) a.cand[i] = r.reserved;
}

template <int PB>
static void launch_pack(const PgSoaIn *soa, PgInRec *in, uint32_t lo, uint32_t cnt, hipStream_t st)
{
    const uint32_t want = (cnt + 255u) / 256u, cap = PG_PACK_GRID;       // 64 reads per wave and loop iteration, 4 waves per workgroup
    pg_pack_kernel<PB><<<want < cap ? want : cap, 256, 0, st>>>(*soa, in, lo, cnt);
}
extern "C" int pg_pack_reads(const PgSoaIn *soa, PgInRec *in, uint32_t lo, uint32_t cnt, void *stream)
{
    if (cnt) {
        hipStream_t st = (hipStream_t)stream;
        switch (soa->plane_blocks) {
        case 1: launch_pack<1>(soa, in, lo, cnt, st); break;
        case 2: launch_pack<2>(soa, in, lo, cnt, st); break;
        case 3: launch_pack<3>(soa, in, lo, cnt, st); break;
        case 4: launch_pack<4>(soa, in, lo, cnt, st); break;
        case 8: launch_pack<8>(soa, in, lo, cnt, st); break;
        default: return (int)hipErrorInvalidValue;
        }
    }
    return (int)hipGetLastError();
}

extern "C" int pg_pack_close_summary(const PgSoaOut *soa, PgOutRec *out, uint32_t n, void *stream)
{
    if (n) pg_pack_close_kernel<<<(n + 255u) / 256u, 256, 0, (hipStream_t)stream>>>(*soa, out, n);
    return (int)hipGetLastError();
}

extern "C" int pg_unpack_results(const PgOutRec *out, const PgSoaOut *soa, uint32_t n, void *stream)
{
    if (n) pg_unpack_kernel<<<(n + 255u) / 256u, 256, 0, (hipStream_t)stream>>>(out, *soa, n);
    return (int)hipGetLastError();
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
extern "C" int pg_debug_occupancy(uint32_t max_len, uint32_t levels, int small_ids, int *close_blocks,
                                  int *far_blocks, unsigned *lds_bytes)
{
    (void)max_len;
    (void)levels;
    *lds_bytes = (unsigned)sizeof(Lds<2, u32>);
    hipError_t e1, e2;
#ifdef PG_ONLY_BENCH
    (void)small_ids;
    e1 = e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(far_blocks, pg_search_kernel<2, 3, u32, PG_MODE_BOTH, false>, WAVE, 0);
    *close_blocks = *far_blocks;
    return (int)e1;
#endif
    if (small_ids) {
        e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(close_blocks, pg_search_kernel<2, 3, u32, PG_MODE_CLOSE, false>, WAVE, 0);
        e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(far_blocks, pg_search_kernel<2, 3, u32, PG_MODE_BOTH, false>, WAVE, 0);
    } else {
        e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(close_blocks, pg_search_kernel<2, 3, u64, PG_MODE_CLOSE, false>, WAVE, 0);
        e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(far_blocks, pg_search_kernel<2, 3, u64, PG_MODE_BOTH, false>, WAVE, 0);
    }
    return (int)e1 | (int)e2;
}

// 64-base blocks per read of the kernels pg_launch_search picks for a batch
static int search_class_blocks(uint32_t max_len, int small_ids)
{
    const int nb = max_len <= 128 ? 2 : (max_len <= 256 ? 4 : 8);
    if (!small_ids) return nb;
    return max_len <= 64 ? 1 : (nb == 2 ? 2 : (max_len <= 192 ? 3 : nb));
}
// May a launch of `mode` over n_reads reads build its records itself (PgDevBatch::soa) ?  One launch per call, the batch's plane
// layout that of the kernels' class (PG_PACK_IN_PLACE_MIN reads at least: 1, see pg_device.h; the environment can raise it).
extern "C" int pg_pack_in_place_ok(int mode, uint32_t max_len, int small_ids, uint32_t n_reads, uint32_t plane_blocks)
{
    // (one launch per call: close end + far end together, or one seam alone; PG_SPLIT_LAUNCH's two launches over one batch keep the pack launch)
    if ((mode == PG_MODE_BOTH && pg_env_switches()->split_launch) || pg_env_switches()->no_pack_in_place) return 0;
    if (mode != PG_MODE_BOTH && mode != PG_MODE_CLOSE && mode != PG_MODE_FAR) return 0;
    if ((uint32_t)search_class_blocks(max_len, small_ids) != plane_blocks) return 0;
    return n_reads >= pg_env_switches()->pack_in_place_min ? 1 : 0;
}

extern "C" int pg_launch_search(const PgDevRef *ref, const PgDevParams *prm, const PgDevBatch *batch,
                                int mode, uint32_t max_len, uint32_t levels, int small_ids, void *stream)
{
    if (batch->n_reads == 0) return 0;
    // a launch that packs in place must be one pg_pack_in_place_ok admits (the kernels run the pack of THEIR class on the batch's planes)
    if (batch->soa && !pg_pack_in_place_ok(mode, max_len, small_ids, batch->n_reads, batch->plane_blocks)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    // experiment knob: extra dynamic LDS per workgroup (lowers occupancy), bytes
    const unsigned lds_pad = pg_env_switches()->lds_pad;
    // 64-base blocks per read: 1/2/3/4/8 with 32-bit candidate ids (the common case), 2/4/8 with 64-bit ids
    const int nb = max_len <= 128 ? 2 : (max_len <= 256 ? 4 : 8);
#ifdef PG_ONLY_BENCH
    // experiment builds: only the instantiations of the bench workloads (100 / 150-base reads, 32-bit ids), compiled in a
    // fraction of the time
    if (!small_ids || max_len > 192) abort();
    if (nb == 2 && max_len > 64) launch<2, u32>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
    else launch<3, u32>(ref, prm, batch, mode, max_len, levels, st, lds_pad);
    return (int)hipGetLastError();
#else
    if (small_ids) {
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
#endif
}

// ---------------------------------------------------------------------------------
// Delivery of a searched read range to the host, chunk by chunk (pg_search_batch & co): three small kernels turn the
// pooled runs of reads [0, cnt) of a chunk (cnt <= PG_DELIVER_CHUNK) into their slice of the batch-wide CSR -- 64-bit
// offsets exactly as the C ABI hands them out, runs gathered in read order behind the runs of the earlier chunks -- so
// that a chunk's result can cross PCIe while the next chunk is still being searched and the host does no per-read work.
//   scan1   per 256-read block: per-read summaries to SoA, block-local exclusive sums of the two run counts, block sums
//   scan2   one block: exclusive sums of the block sums; takes the chunk's base from the running totals and advances them
//   gather  offsets = chunk base + block base + local sum; copies the runs; raises *overflow if a list outgrows `cap`
#include <hipcub/block/block_scan.hpp>
#include <hipcub/block/block_reduce.hpp>

__global__ __launch_bounds__(256) void pg_deliver_scan1(const PgOutRec *out, uint32_t cnt, uint8_t *rc_flag, uint32_t *close_last,
                                                        uint16_t *close_max, uint2 *local, uint2 *blk)
{
    typedef hipcub::BlockScan<uint32_t, 256> Scan;
    __shared__ typename Scan::TempStorage tc, tf;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t c = 0, f = 0;
    if (i < cnt) {
        const PgOutRec r = out[i];
        rc_flag[i] = r.rc_flag;
        close_last[i] = r.close_last;
        close_max[i] = r.close_max;
        c = r.close_cnt;
        f = r.far_cnt;
    }
    uint32_t ec, ef, sc, sf;
    Scan(tc).ExclusiveSum(c, ec, sc);
    Scan(tf).ExclusiveSum(f, ef, sf);
    if (i < cnt) local[i] = make_uint2(ec, ef);
    if (threadIdx.x == 0) blk[blockIdx.x] = make_uint2(sc, sf);
}

// run_tot[0..1]: runs (close, far) of the chunks delivered so far; info[0..3] = {close base, far base, close runs, far runs} of this chunk
// ... info[4] = the fullest run-pool shard's cursor so far (pool overflow check without another copy), info[5] = 0 (the
// gather sets it if a list outgrows the delivery buffers)
__global__ __launch_bounds__(1024) void pg_deliver_scan2(uint2 *blk, uint32_t nblk, unsigned long long *run_tot, unsigned long long *info,
                                                         const uint32_t *pool_used)
{
    typedef hipcub::BlockScan<uint32_t, 1024> Scan;
    typedef hipcub::BlockReduce<uint32_t, 1024> Reduce;
    __shared__ typename Scan::TempStorage tc, tf;
    __shared__ typename Reduce::TempStorage tr;
    const uint32_t worst = Reduce(tr).Reduce(threadIdx.x < PG_POOL_SHARDS ? pool_used[threadIdx.x * 16u] : 0u, hipcub::Max());
    // four consecutive block sums per thread: chunks of up to 4096 x 256 = 2^20 reads
    uint2 v[4];
    uint32_t tx = 0u, ty = 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t idx = threadIdx.x * 4u + (uint32_t)j;
        v[j] = idx < nblk ? blk[idx] : make_uint2(0u, 0u);
        tx += v[j].x;
        ty += v[j].y;
    }
    uint32_t ec, ef, sc, sf;
    Scan(tc).ExclusiveSum(tx, ec, sc);
    Scan(tf).ExclusiveSum(ty, ef, sf);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t idx = threadIdx.x * 4u + (uint32_t)j;
        if (idx < nblk) blk[idx] = make_uint2(ec, ef);
        ec += v[j].x;
        ef += v[j].y;
    }
    if (threadIdx.x == 0) {
        info[0] = run_tot[0];
        info[1] = run_tot[1];
        info[2] = sc;
        info[3] = sf;
        info[4] = worst;
        info[5] = 0ull;
        run_tot[0] += sc;
        run_tot[1] += sf;
    }
}

__global__ __launch_bounds__(256) void pg_deliver_gather(const PgOutRec *out, uint32_t cnt, const uint2 *local, const uint2 *blk,
                                                         const unsigned long long *info, const pg_run *pool, unsigned long long pool_runs,
                                                         pg_run *close_runs, pg_run *far_runs, unsigned long long cap,
                                                         unsigned long long *close_off, unsigned long long *far_off,
                                                         unsigned long long *overflow)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= cnt) return;
    const PgOutRec r = out[i];
    const uint2 l = local[i], b = blk[blockIdx.x];
    const unsigned long long oc = info[0] + b.x + l.x, of = info[1] + b.y + l.y;
    close_off[i] = oc;
    far_off[i] = of;
    // far_runs == null (a batch delivered in ONE chunk): the far runs follow the close runs in the same buffer of `cap`
    // runs, so that offsets, summaries and both lists reach the host in one copy
    if (!far_runs) {
        if (info[2] + info[3] > cap) {
            *overflow = 1ull;
            return;
        }
        far_runs = close_runs + info[2];
    } else if (oc + r.close_cnt > cap || of + r.far_cnt > cap) {
        *overflow = 1ull;
        return;
    }
    // a launch whose run pool overflowed leaves offsets past the pool: the host repeats the launch with a larger pool,
    // this copy must not run off the buffer meanwhile
    if ((unsigned long long)r.close_off + r.close_cnt > pool_runs || (unsigned long long)r.far_off + r.far_cnt > pool_runs) {
        *overflow = 1ull;
        return;
    }
    const u32 *sc = (const u32 *)(pool + r.close_off), *sf = (const u32 *)(pool + r.far_off);
    u32 *dc = (u32 *)(close_runs + oc), *df = (u32 *)(far_runs + of);
    for (uint32_t k = 0; k < 3u * r.close_cnt; k++) dc[k] = sc[k];
    for (uint32_t k = 0; k < 3u * r.far_cnt; k++) df[k] = sf[k];
}

extern "C" int pg_deliver_chunk(const PgOutRec *out, uint32_t cnt, uint8_t *rc_flag, uint32_t *close_last, uint16_t *close_max,
                                void *local, void *blk, unsigned long long *run_tot, unsigned long long *info,
                                const pg_run *pool, unsigned long long pool_runs, pg_run *close_runs, pg_run *far_runs,
                                unsigned long long cap, unsigned long long *close_off, unsigned long long *far_off,
                                const uint32_t *pool_used, void *stream)
{
    if (!cnt || cnt > PG_DELIVER_CHUNK) return cnt ? (int)hipErrorInvalidValue : 0;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nblk = (cnt + 255u) / 256u;
    pg_deliver_scan1<<<nblk, 256, 0, st>>>(out, cnt, rc_flag, close_last, close_max, (uint2 *)local, (uint2 *)blk);
    pg_deliver_scan2<<<1, 1024, 0, st>>>((uint2 *)blk, nblk, run_tot, info, pool_used);
    pg_deliver_gather<<<nblk, 256, 0, st>>>(out, cnt, (const uint2 *)local, (const uint2 *)blk, info, pool, pool_runs, close_runs,
                                           far_runs, cap, close_off, far_off, info + 5);
    return (int)hipGetLastError();
}

// pg_device.h -- structures shared by the host API (pg_api.cpp) and the HIP
// kernels (pg_kernels.hip).  Not part of the public C ABI.
#ifndef PG_DEVICE_H
#define PG_DEVICE_H

#include <stdint.h>
#include "pindel_pg.h"

// Reference in HBM: three 1-bit planes per chromosome ("2-bit planar" + N plane),
// 32 bases per 32-bit word, indexed by AbsLoc (spacer-padded coordinates).
//   lo bit = code & 1, hi bit = code >> 1, code: A=0 C=1 G=2 T=3; nn bit = base is N.
// Every chromosome is preceded and followed by PG_GUARD_WORDS words of N so that a
// staged window may overhang either end without bounds checks.
#define PG_GUARD_WORDS 64u

struct PgDevRef {
    const uint32_t *lo;
    const uint32_t *hi;
    const uint32_t *nn;
    const uint64_t *chr_word_off;  // [n_chr] index of the word holding AbsLoc 0
    const uint32_t *chr_size;      // [n_chr] getCompSize()
    int32_t n_chr;
};

#define PG_MM_BREAKS_N 32
struct PgDevParams {
    int32_t max_range_index;
    int32_t add_mm;          // ADDITIONAL_MISMATCH
    int32_t min_perfect;     // Min_Perfect_Match_Around_BP
    int32_t min_close;       // g_MinClose
    uint32_t spacer;
    // g_maxMismatch as breakpoints (the table is monotone): mm(L) = #{k : L >= mm_bp[k]}
    uint32_t mm_bp[PG_MM_BREAKS_N];
};

enum PgMode { PG_MODE_CLOSE = 1, PG_MODE_FAR = 2, PG_MODE_BOTH = 3 };

// Packed per-read records.  pg_pack_reads builds the input records on the device from the SoA arrays of the C ABI;
// pg_unpack_results scatters the output records back into SoA arrays for the CSR scan / download.
//
// The input record is 128 bytes = two lines of the scalar data cache (the second: the seed filter's symbol programs): the search kernel fetches it with scalar loads straight
// into SGPRs, and it holds everything about a read that is a function of (read, parameters) alone, worked out ONCE by the pack
// kernel -- a streaming kernel with every lane busy -- instead of by every wave's scalar unit at the start of every read (round 5:
// profiles/r05/everything_else_breakdown.txt; the CU's one scalar unit is the search kernel's tightest pipe):
//   * the close end's windows: GetCloseEndInner's (pindel.cpp:2250-2326) R = 1 window is [w1s, w1s + 3 isz), its R = 0 window
//     [w1s + isz, w1s + 2 isz) whichever the anchor's strand (w1s = anchor - isz for '+', anchor - 2 isz for '-'); whether the
//     R = 1 window fits one LDS chunk (all four attempts on one grid); the first window fill;
//   * the word offset and size of the anchor's chromosome (no table look-up, no fallback path);
//   * T = g_maxMismatch[len] + ADDITIONAL_MISMATCH + 1 mismatch levels, CheckMismatches' threshold, the seed filter's two
//     depths J (plain / wide windows) with their base masks bits [1, J) and relevance bounds min(T - 1, g_maxMismatch[J] + ADD);
//   * is the first consumed base of either orientation one of ACGT (farend_searcher.cpp:60-66; searcher.cpp:160).
#define PG_RF_CLOSE_OK 1u       // the close end is searched: len - 1 >= g_MinClose and MatchedD is '+' or '-'
#define PG_RF_PLUS 2u           // MatchedD == '+'
#define PG_RF_SHARED_GRID 4u    // 0 < 3 InsertSize <= PG_CHUNK
#define PG_RF_FIRST_OK_FWD 8u   // read[0] is one of ACGT (orientation 0: left to right)
#define PG_RF_FIRST_OK_REV 16u  // read[len - 1] is (orientation 1: from the last base)
#define PG_RF_EXACT 32u         // the read holds a character outside ACGTN: it is on the exact kernel's list (set by an atomic of the pack kernel)
// The seed filter in READ ORDER (round 6, pg_kernels.hip "seed_filter_ro"): the consumed bases 1 .. 3 G in groups of three, the one-hot
// plane of a base's symbol picked by VGPR index (s_set_gpr_idx): the symbols come as a PROGRAM, one dword per group -- byte k =
// 0x30 | symbol of base 3 g + 1 + k (A 0, C 1, G 2, T 3, N 4; the 0x3 nibble is the index mode's operand enables when a 16-bit field
// is moved into M0, and biases the index by 48), byte 3 = 0x30 (| symbol of base 0 in group 0: the seed).  prog[0] = the read left
// to right as it is, prog[1] = the read from its last base, COMPLEMENTED: the only two (orientation, complement) pairs any search
// consumes (kind F reads them as they are; kind B reads the complement, and its planes are laid out in complement order).
#define PG_RO_GROUPS_MAX 8
#define PG_RO_GROUPS_MIN 4
#define PG_RO_OK 0x80000000u     // PgInRec::ro: this read may take the read-order filter (<= 16 mismatch levels, >= 4 groups, ACGTN only
                                 // among the bases a program covers, g_MinClose >= 8)
struct PgInRec {
    // dwords 0 .. 11: fetched at the start of a read
    int32_t  w1s;           // AbsLoc where the close end's R = 1 window starts
    int32_t  isz;           // InsertSize
    int32_t  stage_s, stage_e;   // the first window fill covers positions [stage_s, stage_e) (+ overhangs); 0, 0: none
    uint32_t chr_wo_lo, chr_wo_hi;   // index of the word holding AbsLoc 0 of the anchor's chromosome in the reference planes
    uint32_t lenf;          // read length | PG_RF_* << 16
    uint32_t lvl;           // CheckMismatches' threshold (smallest n with (float)n >= (float)(len * u)) | g_maxMismatch[len] << 16 | T << 24
    uint32_t depth;         // J plain | J wide << 8 | bound plain << 16 | bound wide << 24
    uint32_t jmask0;        // bits [1, J plain)
    uint32_t ro;            // read-order filter: groups plain | groups wide << 4 | bound plain << 8 | bound wide << 16 | PG_RO_OK
                            // (bound = min(T - 1, g_maxMismatch[3 G + 1] + ADD): the depth the groups reach)
    int32_t  chr;           // chromosome of the anchor
    // dwords 12 .. 15: fetched at the start of the far end
    uint32_t chr_size;      // getCompSize() of that chromosome
    uint32_t bd_cnt;        // BreakDancer windows of this read ...
    uint32_t bd_off;        // ... starting at PgDevBatch::bd[bd_off]
    uint32_t jmask1;        // bits [1, J wide)
    // dwords 16 .. 31: fetched by a seed-filter run (one orientation)
    uint32_t prog[2][PG_RO_GROUPS_MAX];
};
#define PG_PACK_CLAIM 8u            // ... of a launch that packs in place (PgDevBatch::soa) ...
#define PG_PACK_CLAIM_LONG 16u      // ... and when a wave takes many of them (pg_launch_search)
#define PG_PACK_IN_PLACE_MIN 1u            // fewest reads of such a launch (measured 20 000 reads .. 10 M: never slower than a pack launch in front)
#define PG_CLAIM_DEFAULT 8u  // reads a workgroup of the persistent launch claims per atomic, at most
#define PG_IN_PAD 8u        // records allocated behind the last one (the kernel prefetches the next read's record)
struct PgOutRec {
    uint32_t close_off, close_cnt, far_off, far_cnt;   // runs in the pool
    uint32_t close_last;    // getLastAbsLocCloseEnd()
    uint16_t close_max;     // MaxLenCloseEnd(), 0 = no close end
    uint8_t  rc_flag;       // GetCloseEnd left the read reverse-complemented
    uint8_t  pad;
    uint32_t alg;           // algorithmic bytes of this read
    uint32_t reserved;      // diagnostics: candidates (survivors of the seed filter) folded for this read
};

struct PgSoaIn;
struct PgDevBatch {
    uint32_t n_reads;
    uint32_t first_read;           // offset into the batch arrays handled by this launch
    const PgInRec *in;
    PgOutRec *out;                 // close-end fields: written in CLOSE/BOTH mode, read in FAR mode
    const uint8_t *seq;
    // The reads as bit planes, built once by pg_pack_reads (every lane busy) instead of per read in the search kernel
    // (ten byte compares and eight ballots of a single wave): per read u64[2 orientations][4 planes][plane_blocks],
    // orientation 0 = left to right, 1 = reversed; planes = code bit 0, code bit 1, is-N, is-other; bit j of block b =
    // base 64 b + j.
    const uint64_t *planes;
    uint32_t plane_blocks;         // 64-base blocks per read in `planes` (pg_plane_blocks of the batch's longest read)
    const pg_window *bd;           // BreakDancer windows (nullable)
    // Run pool, split into PG_POOL_SHARDS equal regions with one bump cursor each (a single
    // device-wide cursor saturates at ~88 M atomics/s, MI355X_MICROARCH.md "dequeue").  Workgroup b
    // allocates from shard b % PG_POOL_SHARDS; cursors are 64 bytes apart.
    pg_run *pool;
    uint32_t pool_shard_cap;       // runs per shard
    uint32_t *pool_used;           // [PG_POOL_SHARDS * 16]; cursor > pool_shard_cap means overflow (retry bigger)
    uint32_t *work_ctr;            // [PG_WORK_CTRS * 16] reads claimed per XCD part (zeroed before every launch)
    uint32_t claim;                // reads per claim (set by pg_launch_search: a workgroup's share in the fewest equal claims <= PG_CLAIM)
    // the reads with a character outside ACGTN (pg_pack_kernel appends, pg_search_exact_kernel searches them again with the
    // reference's read-shortening semantics) and CheckMismatches' threshold per read length (the exact kernel recomputes a read's
    // length-dependent fields)
    const uint32_t *exact_list;
    const uint32_t *exact_count;
    const uint16_t *thr_tab;       // [512]
    // PACK IN PLACE (pg_launch_search, BOTH mode): non-null = the records and planes of the launch's reads are built from these SoA
    // arrays (a PgSoaIn in device memory) by the search kernel itself, claim by claim, instead of by a pg_pack_reads launch before it
    const struct PgSoaIn *soa;
};

// SoA views for the pack / unpack kernels
static inline uint32_t pg_plane_blocks(uint32_t max_len)
{
    return max_len <= 64u ? 1u : (max_len <= 128u ? 2u : (max_len <= 192u ? 3u : (max_len <= 256u ? 4u : 8u)));
}
struct PgSoaIn {
    const uint8_t *seq;            // the reads' bases ...
    uint64_t *planes;              // ... and where their bit planes go (PgDevBatch::planes)
    uint32_t plane_blocks;
    const uint64_t *seq_off;
    const uint8_t *strand;
    const int32_t *pos;
    const int16_t *isz;
    const int32_t *chr;
    const uint64_t *bd_off;        // nullable
    uint32_t *exact_list;          // nullable: indices of the reads with a character outside ACGTN ...
    uint32_t *exact_count;         // ... and how many (zeroed by the caller before the batch's first pack)
    const struct PgLenRec *len_tab; // [512] the fields of a read's record that follow from its length alone (pg_len_rec)
    const uint64_t *chr_word_off;  // [n_chr] as PgDevRef
    const uint32_t *chr_size;      // [n_chr]
    uint32_t spacer;
    int32_t add_mm;                // ADDITIONAL_MISMATCH
    int32_t min_close;             // g_MinClose
};
// consumed bases the seed filter inspects for a read of `len` bases with T mismatch levels (wide: the chunks of wide far-end
// windows, where a survivor costs a whole candidate pass for a handful of candidates: two more)
#ifndef PG_SEED_J
#define PG_SEED_J(T) (2 * (T) + 4)
#endif
#ifndef PG_SEED_J_WIDE
#define PG_SEED_J_WIDE 2
#endif
#ifdef __HIPCC__
__host__ __device__
#endif
static inline int pg_seed_depth(int len, int T, int wide)
{
    int J = len - 1 < 32 ? len - 1 : 32;
    const int jt = PG_SEED_J(T) + (wide ? PG_SEED_J_WIDE : 0);
    return J > jt ? jt : J;
}
// groups of three bases the read-order filter inspects (bases 1 .. 3 G): the depth above rounded to whole groups
#ifdef __HIPCC__
__host__ __device__
#endif
static inline int pg_ro_groups(int len, int T, int wide)
{
    const int J = pg_seed_depth(len, T, wide);
    int G = J / 3;                                    // (J - 1 bases: 15 -> 5 groups, 17 -> 6, 19 -> 6, 21 -> 7)
    if (G > PG_RO_GROUPS_MAX) G = PG_RO_GROUPS_MAX;
    while (G > 0 && 3 * G > len - 1) G--;
    return G;
}
// What PgInRec holds that is a function of the read's LENGTH and the search parameters alone (lvl, depth, jmask0, ro, jmask1): one
// table per context, built on the host when the context is created (the pack's per-read lane used to work these out -- two depths,
// two group counts, six table look-ups -- for every read; inside the search kernel that lane's instructions are paid for by a claim of
// eight reads).
struct PgLenRec { uint32_t lvl, depth, jmask0, ro, jmask1, pad[3]; };
static inline struct PgLenRec pg_len_rec(int len, const uint32_t *mm, const uint16_t *thr, int add_mm, int min_close)
{
    struct PgLenRec r;
    const uint32_t M = mm[len] & 0xffu, T = M + (uint32_t)add_mm + 1u;
    const int J0 = pg_seed_depth(len, (int)T, 0), J1 = pg_seed_depth(len, (int)T, 1);
    // relevance bound of a seed: min(T - 1, g_maxMismatch[J] + ADD)  (J >= 0; a one-base read has J = 0)
    uint32_t b0 = (mm[J0 < 0 ? 0 : J0] & 0xffu) + (uint32_t)add_mm, b1 = (mm[J1 < 0 ? 0 : J1] & 0xffu) + (uint32_t)add_mm;
    if (b0 > T - 1u) b0 = T - 1u;
    if (b1 > T - 1u) b1 = T - 1u;
    r.lvl = (uint32_t)thr[len] | (M << 16) | (T << 24);
    r.depth = (uint32_t)(J0 < 0 ? 0 : J0) | ((uint32_t)(J1 < 0 ? 0 : J1) << 8) | (b0 << 16) | (b1 << 24);
    r.jmask0 = J0 > 1 ? ((J0 >= 32 ? 0xffffffffu : (1u << J0) - 1u) & ~1u) : 0u;
    r.jmask1 = J1 > 1 ? ((J1 >= 32 ? 0xffffffffu : (1u << J1) - 1u) & ~1u) : 0u;
    // read-order filter (PgInRec::ro): whole groups of three bases, the relevance bound of the depth they reach
    const int G0 = pg_ro_groups(len, (int)T, 0), G1 = pg_ro_groups(len, (int)T, 1);
    uint32_t rb0 = (mm[3 * G0 + 1] & 0xffu) + (uint32_t)add_mm, rb1 = (mm[3 * G1 + 1] & 0xffu) + (uint32_t)add_mm;
    if (rb0 > T - 1u) rb0 = T - 1u;
    if (rb1 > T - 1u) rb1 = T - 1u;
    r.ro = (uint32_t)G0 | ((uint32_t)G1 << 4) | (rb0 << 8) | (rb1 << 16);
    if (T <= 16u && G0 >= PG_RO_GROUPS_MIN && G1 >= PG_RO_GROUPS_MIN && min_close >= 8) r.ro |= PG_RO_OK;     // (the close end's snapshot covers seven bases)
    r.pad[0] = r.pad[1] = r.pad[2] = 0u;
    return r;
}
struct PgSoaOut {
    uint8_t *rc_flag;
    uint32_t *close_last;
    uint16_t *close_max;
    uint32_t *close_off, *close_cnt, *far_off, *far_cnt, *alg;
    uint32_t *cand;                // nullable
};

// Candidate id = position relative to the search origin | kind (F/B) | window index of a BreakDancer cluster.
//   64-bit ids: rel(26) | kind(1) | region(7)   -- any window the ABI accepts
//   32-bit ids: rel(24) | kind(1) | region(7)   -- every search window of the launch has <= 2^24 positions
#define PG_REL_BITS 26
#define PG_REL_BITS_SMALL 24
#define PG_SMALL_MAX_WINDOW (1 << 24)
#define PG_SMALL_MAX_CLUSTER 127u   // windows per read the 32-bit ids can tell apart
#define PG_MAX_LEVELS 32
#define PG_MM_BREAKS 32
#define PG_POOL_SHARDS 1024u
#define PG_WORK_CTRS 8u           // per-XCD read counters of the persistent launch, 64 bytes apart
#define PG_DIAG_WORDS 64u         // behind the read counters: cycle accumulators of a -DPG_TIMING diagnostics build
// Run-pool slots a workgroup reserves per claimed read with ONE atomic per claim (instead of one or two dependent
// atomic round trips inside every read): the first PG_RES_CLOSE for the read's UP_Close runs, the rest for its UP_Far
// runs (1.04 / 1.03 runs on average); a list that does not fit takes an allocation of its own.
#define PG_RES_CLOSE 4u
#define PG_RES_FAR 4u
#define PG_RESERVE (PG_RES_CLOSE + PG_RES_FAR)

#define PG_CHUNK 2048u            // window positions staged per LDS fill (= 64 lanes x 32-base words)
#define PG_CHUNK_SHIFT 11
#define PG_EQ_ROWS 5u
// LDS window capacity in 32-base words: chunk(s) + overhang of nb 64-base blocks on both sides + slack.  Reads of
// up to 192 bases (nb <= 3) take the chunks of wide far-end windows two at a time; longer reads keep one chunk per
// fill, where the extra KB of LDS would cost a resident wave per SIMD.
#define PG_PAIR_CHUNKS(nb) ((nb) <= 3)
#define PG_WIN_WORDS(nb) (((PG_PAIR_CHUNKS(nb) ? 2u : 1u) * PG_CHUNK + 2u * (64u * (nb))) / 32u + 6u)
// ... of which this many are STATIC LDS.  For reads of 65..192 bases (nb = 2, 3) the second chunk's words are dynamic LDS that only
// launches with -x >= 3 ask for: the static part then fits 28 (nb = 2: 5120 B) or 24 (nb = 3: 6384 B) workgroups per CU -- seven or
// six waves per SIMD -- at the default -x 2, where nothing takes two chunks per fill (a cluster window of more than a chunk
// goes chunk by chunk).  The dynamic part starts where the static LDS ends (the window is the last member of the kernel's one
// static LDS object), so the window stays one array with compile-time addresses.
// (LDS is handed out in granules of 1280 bytes on gfx950 -- hence + 4 words of slack here, not + 6: a fill writes
// (2048 + 128 nb) / 32 + 2 words -- 74 / 78 -- and a candidate reads up to word 4 nb + 63 of a one-chunk window)
#define PG_WIN_STATIC_WORDS(nb) ((nb) == 3 || (nb) == 2 ? (PG_CHUNK + 2u * (64u * (nb))) / 32u + 4u : PG_WIN_WORDS(nb))
#define PG_WIN_DYN_BYTES(nb) ((PG_WIN_WORDS(nb) - PG_WIN_STATIC_WORDS(nb)) * 16u)

// Environment switches of tests and experiments, read once per process by pg_api.cpp (pg_debug_reload_env() re-reads them)
struct PgEnvSwitches {
    uint32_t host_chunk;        // PG_HOST_CHUNK: reads per chunk of the host path (0 = the default schedule)
    uint32_t lds_pad;           // PG_LDS_PAD: extra dynamic LDS per workgroup (lowers occupancy), bytes
    bool no_single_block;       // PG_NO_SINGLE_BLOCK: no one-copy delivery of one-chunk batches
    bool tiny_delivery;         // PG_TEST_TINY_DELIVERY: forces the delivery-overflow fallback
    bool tiny_pool;             // PG_TEST_TINY_POOL: forces the run-pool regrow path
    bool force_wide_cells;      // PG_FORCE_WIDE_CELLS: 64-bit candidate ids
    bool split_launch;          // PG_SPLIT_LAUNCH: close end and far end as two launches
    bool generic_kernels;       // PG_GENERIC_KERNELS: never the default-parameter kernels
    bool no_pack_in_place;      // PG_NO_PACK_IN_PLACE: pg_device_batch_pack_search = pack launch + search launch
    uint32_t pack_claim;        // PG_PACK_CLAIM: reads per claim of a launch that packs in place (0 = the default)
    uint32_t pack_in_place_min; // PG_PACK_IN_PLACE_MIN: fewest reads of such a launch (tests: the path on small batches)
};

#ifdef __cplusplus
extern "C" {
#endif
const struct PgEnvSwitches *pg_env_switches(void);
// TEST HOOK: not synchronised with readers -- call it only while no pg_* call is in flight on any thread.
void pg_debug_reload_env(void);
// number of kernel arguments whose value fetched from the kernarg segment at PgKArgs' offsets differs from the by-value one (0)
int pg_debug_kargs_check(const PgDevRef *ref, const PgDevParams *prm, const PgDevBatch *batch, uint32_t max_len, uint32_t levels,
                         uint32_t *scratch_dev, void *stream);
// Launches the search kernel for the reads of the batch on `stream`.  small_ids selects the
// 32-bit candidate ids (see above for when that is valid).
int pg_launch_search(const PgDevRef *ref, const PgDevParams *prm, const PgDevBatch *batch,
                     int mode, uint32_t max_len, uint32_t levels, int small_ids, void *stream);
// may that launch build the records of its reads itself (PgDevBatch::soa) ?
int pg_pack_in_place_ok(int mode, uint32_t max_len, int small_ids, uint32_t n_reads, uint32_t plane_blocks);
// ... then the reads on the batch's exact list that lie in the launch's range, with the reference's read-shortening semantics
int pg_launch_search_exact(const PgDevRef *ref, const PgDevParams *prm, const PgDevBatch *batch, int mode,
                           uint32_t max_len, uint32_t levels, void *stream);
// Device-side CSR of a result list: first the scan (gather = 0: csr[0..n] = exclusive sums of cnt, cnt has
// n + 1 readable entries), then the gather (gather = 1: out[csr[i] + k] = pool[off[i] + k]).
size_t pg_scan_tmp_bytes(uint32_t n);
// in[lo .. lo + cnt) from the SoA input arrays; cnt = 0 is allowed
int pg_pack_reads(const PgSoaIn *soa, PgInRec *in, uint32_t lo, uint32_t cnt, void *stream);
// close-end summary (rc flag, last AbsLoc, max length) of a host result into the output records
int pg_pack_close_summary(const PgSoaOut *soa, PgOutRec *out, uint32_t n, void *stream);
int pg_unpack_results(const PgOutRec *out, const PgSoaOut *soa, uint32_t n, void *stream);
// One chunk (cnt <= PG_DELIVER_CHUNK reads) of a searched batch to read-order CSR behind the earlier chunks: see the
// delivery kernels in pg_kernels.hip.  local / blk: scratch of cnt and 4096 uint2; run_tot: 2 running totals (zeroed
// per batch); info: 4 values for the host {close base, far base, close runs, far runs}.
#define PG_DELIVER_CHUNK (1u << 20)     // reads a delivery handles at most (scan2: 4096 blocks of 256)
#define PG_HOST_CHUNK (1u << 18)        // reads per chunk the host path starts from (and the largest batch that is ONE chunk)
int pg_deliver_chunk(const PgOutRec *out, uint32_t cnt, uint8_t *rc_flag, uint32_t *close_last, uint16_t *close_max,
                     void *local, void *blk, unsigned long long *run_tot, unsigned long long *info,
                     const pg_run *pool, unsigned long long pool_runs, pg_run *close_runs, pg_run *far_runs /* null: behind the close runs */,
                     unsigned long long cap, unsigned long long *close_off, unsigned long long *far_off, const uint32_t *pool_used,
                     void *stream);
int pg_compact_runs(const pg_run *pool, const uint32_t *off, const uint32_t *cnt, uint32_t *csr,
                    pg_run *out, uint32_t n, void *tmp, size_t tmp_bytes, int gather, void *stream);
#ifdef __cplusplus
}
#endif

#endif

/*
 * pindel_pg.h -- C ABI of the MI355X-native split-read pattern-growth engine.
 *
 * Drop-in boundary for Pindel 0.2.5b9's hot path.  The reference has no FFI
 * layer; these entry points replace its two internal batch seams and the
 * per-read primitives behind them:
 *
 *   pg_close_end_batch  <->  ReadBuffer::flush()           src/read_buffer.cpp:36-101
 *                            (GetCloseEnd per read,        src/pindel.cpp:2531-2605,
 *                             updateReadAfterCloseEndMapping src/reader.cpp:1531-1554)
 *                            and the two loops in ReadInRead src/reader.cpp:248-255,300-305
 *   pg_far_end_batch    <->  SearchFarEnds(chrSeq, reads, chr) src/pindel.cpp:1115-1138
 *                            (SearchFarEnd per read,        src/pindel.cpp:1001-1074,
 *                             SearchFarEndAtPos             src/farend_searcher.cpp:46-103)
 *   pg_search_batch     <->  both of the above back to back (no BreakDancer hints)
 *   pg_load_reference   <->  Genome::loadAll / Chromosome::getSeq()  src/pindel.cpp:236-312
 *   pg_params           <->  the globals the path reads: g_maxMismatch (pindel.cpp:794-819),
 *                            g_MinClose (:88), userSettings->{MaxRangeIndex, ADDITIONAL_MISMATCH,
 *                            Min_Perfect_Match_Around_BP, MaximumAllowedMismatchRate,
 *                            Seq_Error_Rate, sensitivity}, g_SpacerBeforeAfter (pindel.h:122)
 *
 * Conventions: plain pointers and sizes only; the caller owns every input
 * buffer; result objects are owned by the library until pg_result_free.
 * Every function returns 0 (PG_OK) or a negative pg_status; nothing calls
 * exit().  One pg_ctx drives one GPU (HIP device ordinal in pg_params); use
 * one process per GPU.  Calls on one ctx are synchronous and not re-entrant.
 *
 * Coordinates are the reference's AbsLoc: indices into the spacer-padded
 * chromosome string (biological position + spacer).
 */
#ifndef PINDEL_PG_H
#define PINDEL_PG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_ABI_VERSION 1
#define PG_MAX_READ_LEN 499      /* g_maxMismatch has 500 entries (pindel.cpp:801) */

typedef enum {
    PG_OK = 0,
    PG_E_INVALID = -1,       /* bad argument                                   */
    PG_E_NOMEM = -2,         /* host or device allocation failed               */
    PG_E_DEVICE = -3,        /* HIP runtime error (see pg_last_error)          */
    PG_E_NO_REFERENCE = -4,  /* pg_load_reference not called                   */
    PG_E_READ_TOO_LONG = -5, /* a read is longer than PG_MAX_READ_LEN          */
    PG_E_UNSUPPORTED = -6    /* parameter outside what the device path handles */
} pg_status;

typedef struct pg_ctx pg_ctx;
typedef struct pg_result pg_result;
typedef struct pg_device_batch pg_device_batch;

/* Pindel command-line flags the path reads (defaults of 0.2.5b9 in brackets). */
typedef struct {
    int32_t  abi_version;                 /* PG_ABI_VERSION                              */
    int32_t  device;                      /* HIP device ordinal                          */
    int32_t  max_range_index;             /* -x [2], capped at 9 (pindel.cpp:125,921)    */
    int32_t  additional_mismatch;         /* -a [1], raised to >=1 (pindel.cpp:927)      */
    int32_t  min_perfect_match_around_bp; /* -m [3]                                      */
    int32_t  min_close;                   /* -H [8]  g_MinClose                          */
    double   max_allowed_mismatch_rate;   /* -u [0.02]                                   */
    double   seq_error_rate;              /* -e [0.01]                                   */
    double   sensitivity;                 /* -E [0.95]                                   */
    uint32_t spacer;                      /* g_SpacerBeforeAfter [100000]                */
    uint32_t reserved;
} pg_params;

/* One batch of one-end-anchored reads: the SPLIT_READ fields the path reads
 * (src/pindel.h:265-383), structure-of-arrays. */
typedef struct {
    uint32_t        n_reads;
    const uint8_t  *seq;            /* concatenated UnmatchedSeq, ASCII, as after setUnmatchedSeq */
    const uint64_t *seq_off;        /* n_reads+1 offsets into seq                                  */
    const uint8_t  *anchor_strand;  /* MatchedD: '+' or '-'                                        */
    const int32_t  *anchor_pos;     /* MatchedRelPos (biological coordinate)                       */
    const int16_t  *insert_size;    /* InsertSize (short, pindel.h:318)                            */
    const int32_t  *chr_id;         /* index of FragName in the loaded reference                   */
} pg_read_batch;

/* A SearchWindow (src/pindel.h:665-716) in AbsLoc coordinates. */
typedef struct {
    int32_t chr_id;
    int32_t start;   /* may be < 0: then start = end-1 (farend_searcher.cpp:69-71) */
    int32_t end;
} pg_window;

/* Per-read BreakDancer clusters (result of BDData::getCorrespondingSearchWindowCluster,
 * src/bddata.cpp:949-971), CSR.  NULL / n = 0 means "no hints". */
typedef struct {
    const uint64_t  *offset;   /* n_reads+1 */
    const pg_window *windows;
} pg_windows;

/* Run-length-encoded UniquePoints (src/pindel.h:137-158).  A run stands for
 * the points  LengthStr = len_first..len_last  with
 * AbsLoc = abs_loc_first + (LengthStr-len_first)  for direction '+' (FORWARD)
 * AbsLoc = abs_loc_first - (LengthStr-len_first)  for direction '-' (BACKWARD),
 * all with the same Mismatches / Direction / Strand / chromosome. */
typedef struct {
    uint32_t abs_loc_first;
    uint16_t len_first;
    uint16_t len_last;
    uint8_t  mismatches;
    uint8_t  flags;          /* bit0: Direction BACKWARD, bit1: Strand ANTISENSE */
    int16_t  chr_id;
} pg_run;

#define PG_RUN_BACKWARD  0x1u
#define PG_RUN_ANTISENSE 0x2u

/* An expanded UniquePoint. */
typedef struct {
    uint32_t abs_loc;
    int16_t  length;       /* LengthStr  */
    int16_t  mismatches;
    int16_t  chr_id;
    char     direction;    /* '+' FORWARD / '-' BACKWARD */
    char     strand;       /* '+' SENSE   / '-' ANTISENSE */
} pg_point;

/* What a search leaves behind for each read (host memory, owned by the result). */
typedef struct {
    uint32_t        n_reads;
    const uint64_t *close_off;   /* n_reads+1: UP_Close runs of read i are close_runs[close_off[i]..close_off[i+1]) */
    const pg_run   *close_runs;  /* after CleanUniquePoints (pindel.cpp:2904-2941)                                  */
    const uint64_t *far_off;     /* n_reads+1 (all zero after pg_close_end_batch)                                   */
    const pg_run   *far_runs;    /* UP_Far                                                                          */
    const uint8_t  *rc_flag;     /* how GetCloseEnd left UnmatchedSeq (pindel.cpp:2545 "setUnmatchedSeq(ReverseComplement())"):
                                  * 0 as it came; 1 reverse-complemented once; 2 TWICE -- reported only for a read that holds a
                                  * character outside ACGTN, which two reverse complements do not restore: Convert2RC4N
                                  * (pindel.cpp:966-970) makes such characters NUL and setUnmatchedSeq (pindel.cpp:142-157) strips
                                  * the trailing ones, so the read is SHORTER by what it began with (after the first) and ended with
                                  * (after the second); every other read is its old self after two and keeps 0.  The search uses the
                                  * shortened read from that attempt on, exactly as the reference does; an adapter restores the
                                  * post-state by applying setUnmatchedSeq(ReverseComplement()) rc_flag times (pg_adapter.hpp).    */
} pg_result_view;

/* ---- lifetime ---------------------------------------------------------- */
void pg_default_params(pg_params *p);
int  pg_create(const pg_params *p, pg_ctx **out);
void pg_destroy(pg_ctx *ctx);
const char *pg_last_error(const pg_ctx *ctx);
/* g_maxMismatch as the library computed it (500 entries). */
int  pg_get_max_mismatch(const pg_ctx *ctx, uint32_t *table500);

/* ---- reference --------------------------------------------------------- */
/* seq_padded[c] is Chromosome::getSeq(): spacer N's + sequence + spacer N's, ASCII,
 * upper case ACGTN (anything else is taken as N).  Packs to 2-bit planes + N plane
 * and uploads to the ctx's GPU. */
int  pg_load_reference(pg_ctx *ctx, int32_t n_chr, const char *const *names,
                       const uint8_t *const *seq_padded, const uint64_t *len_padded);
/* FASTA loader with Genome::loadChromosome's semantics (pindel.cpp:272-312). */
int  pg_load_fasta(pg_ctx *ctx, const char *path);
/* Packed reference cache: writes / reads the loaded reference as the bit planes it
 * occupies in HBM, so that a genome is packed from FASTA once (Genome::loadChromosome semantics,
 * src/pindel.cpp:272-312, via pg_load_fasta) instead of being parsed character by character on every
 * run.  The file records the spacer it was packed with; loading it into a ctx with another spacer fails. */
int  pg_reference_save_packed(const pg_ctx *ctx, const char *path);
int  pg_reference_load_packed(pg_ctx *ctx, const char *path);
int  pg_reference_n_chr(const pg_ctx *ctx);
const char *pg_reference_name(const pg_ctx *ctx, int32_t chr_id);
uint64_t pg_reference_comp_size(const pg_ctx *ctx, int32_t chr_id);   /* getCompSize() */
/* Unpack [start, start+n) of a chromosome back to ASCII (tests, reporters). */
int  pg_reference_fetch(const pg_ctx *ctx, int32_t chr_id, uint64_t start, uint64_t n, uint8_t *out);

/* ---- the path, host buffers in / host results out ---------------------- */
int  pg_close_end_batch(pg_ctx *ctx, const pg_read_batch *reads, pg_result **out);
/* Both seams on ONE read vector: `close` must be the result of pg_close_end_batch on the same reads, given here in
 * their ORIGINAL orientation (the result carries UP_Close and the rc flags); it is extended in place with UP_Far.
 * The reference's own call site has a different vector by then -- see pg_far_end_batch_from_close. */
int  pg_far_end_batch(pg_ctx *ctx, const pg_read_batch *reads, pg_result *close,
                      const pg_windows *bd_hints /* nullable */);
/* Seam 2 exactly where the reference calls it: SearchFarEnds(chrSeq, state.Reads_SR, chr), src/pindel.cpp:1115-1138,
 * called at :1888 on the FILTERED UNION of many flushes -- ReadBuffer::flush keeps only the reads that got a close end
 * (src/read_buffer.cpp:55-64), 50 000 raw reads at a time (src/reader.cpp:55).  So this entry takes the reads as the
 * close end left them (UnmatchedSeq already reverse-complemented where GetCloseEnd did so, src/pindel.cpp:2545) and,
 * per read, the two things SearchFarEnd reads from UP_Close:
 *   close_last[i] = UP_Close.back().AbsLoc      getLastAbsLocCloseEnd(), src/pindel.cpp:475-478 (centre of the ranges)
 *   close_max[i]  = UP_Close.back().LengthStr   UP_Close.MaxLen(), src/pindel.cpp:480-483, 490-498 (goodFarEndFound);
 *                                               <= 0: the read has no close end and is not searched
 * No result of an earlier call is needed; anchor_strand / anchor_pos / insert_size of the batch are not read by the far
 * end (they must still be valid arrays).  The result holds UP_Far (far_off / far_runs); its close lists are empty and
 * its rc flags zero. */
int  pg_far_end_batch_from_close(pg_ctx *ctx, const pg_read_batch *reads, const uint32_t *close_last,
                                 const int16_t *close_max, const pg_windows *bd_hints /* nullable */, pg_result **out);
int  pg_search_batch(pg_ctx *ctx, const pg_read_batch *reads, pg_result **out);

/* Multi-GPU form of pg_search_batch: the reads are split into n_ctx contiguous ranges (reads are independent:
 * the loop of ReadBuffer::flush / SearchFarEnds, src/read_buffer.cpp:36-101, src/pindel.cpp:1115-1138), context k
 * (one GPU each, the same reference loaded in all of them) searches range k on its own host thread, and the
 * results are concatenated in order: identical to pg_search_batch on one context.  No collective between GPUs. */
int  pg_search_batch_multi(pg_ctx *const *ctxs, int32_t n_ctx, const pg_read_batch *reads, pg_result **out);

int  pg_result_view_get(const pg_result *r, pg_result_view *view);
void pg_result_free(pg_result *r);
/* Expand runs to UniquePoints; returns the number of points (call with out = NULL to count). */
uint64_t pg_expand_runs(const pg_run *runs, uint64_t n_runs, pg_point *out);

/* ---- the path, device-resident (what bench.py times) ------------------- */
int  pg_device_batch_upload(pg_ctx *ctx, const pg_read_batch *reads, pg_device_batch **out);
/* Attaches per-read BreakDancer/RP window clusters (the SearchWindowCluster of
 * g_bdData.getCorrespondingSearchWindowCluster(read), src/bddata.cpp:949-979, searched before the
 * ranges, src/pindel.cpp:1006-1018) to a device-resident batch; NULL detaches them.  Same limits as
 * pg_far_end_batch. */
int  pg_device_batch_set_windows(pg_ctx *ctx, pg_device_batch *b, const pg_windows *bd_hints);
/* Close end + far end for every read of the batch; results stay on the GPU.
 * Synchronous: returns when the kernels have finished. */
int  pg_device_batch_search(pg_ctx *ctx, pg_device_batch *b);
int  pg_device_batch_download(pg_ctx *ctx, pg_device_batch *b, pg_result **out);
/* Measurement: runs the pack stage of pg_device_batch_upload again on the resident batch -- ASCII bases (what
 * src/reader.cpp:852-856 hands over) -> the bit planes and packed records the search kernel reads; idempotent -- and
 * returns its HIP-event duration in ms (events on the ctx's own stream). */
int  pg_device_batch_repack(pg_ctx *ctx, pg_device_batch *b, double *pack_ms);
/* Pack + search of the resident batch as ONE step: the result of pg_device_batch_repack followed by pg_device_batch_search.
 * Wherever the batch's plane layout is that of its search kernels (32-bit candidate ids: always) the search kernel builds the planes
 * and records of its reads itself, claim by claim (the pack is HBM-bound, the search is not: no launch of its own).  Synchronous. */
int  pg_device_batch_pack_search(pg_ctx *ctx, pg_device_batch *b);
void pg_device_batch_free(pg_ctx *ctx, pg_device_batch *b);
/* HIP-event duration (ms) of the kernel of the last search on this ctx (events recorded on
 * the ctx's own stream around the launch) and the run-pool slots it took (reserved per read + allocated; the number
 * of runs is close_off[n] + far_off[n] of the downloaded result). */
int  pg_last_search_stats(const pg_ctx *ctx, double *kernel_ms, uint64_t *n_runs);
/* Algorithmic bytes of the last search of this batch (SURVEY.md 8d formula, accumulated per
 * read by the kernel; DESIGN.md "roofline accounting").  Copies n x 4 bytes back: call it
 * outside any timed region. */
int  pg_device_batch_algorithmic_bytes(pg_ctx *ctx, pg_device_batch *b, double *bytes);
/* Diagnostics: candidate positions (survivors of the kernel's seed filter) that went through the full
 * comparison in the last search of this batch -- what a repeat-rich reference drives up. */
int  pg_device_batch_candidates(pg_ctx *ctx, pg_device_batch *b, double *n_candidates);

#ifdef __cplusplus
}
#endif
#endif /* PINDEL_PG_H */

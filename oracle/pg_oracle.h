/*
 * pg_oracle.h -- CPU restatement of Pindel 0.2.5b9's split-read pattern growth.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (pindel_amd/, the
 * C-ABI library) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the CPU baseline, never as the thing shipped.
 *
 * Pinning status: see oracle/README.md (pinned through the reference's own
 * golden files devtools/gold_standard/simulated_test.out_{D,SI,TD,INV}; the
 * reference itself is unbuildable in this image because src/pindel.h:34-35
 * includes htslib headers that are absent).
 *
 * All coordinates are the reference's "AbsLoc": indices into the
 * spacer-padded chromosome string (biological position + 100 000,
 * src/pindel.h:122, src/pindel.cpp:297-309).
 */
#ifndef PG_ORACLE_H
#define PG_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_READ_LEN 500   /* g_maxMismatch has 500 entries (pindel.cpp:801) */

typedef struct {
    int32_t  max_range_index;      /* -x, userSettings->MaxRangeIndex            */
    int32_t  additional_mismatch;  /* -a, ADDITIONAL_MISMATCH (>=1, pindel.cpp:927) */
    int32_t  min_perfect_match;    /* -m, Min_Perfect_Match_Around_BP            */
    int32_t  min_close;            /* -H, g_MinClose (pindel.cpp:88)             */
    double   max_mismatch_rate;    /* -u, MaximumAllowedMismatchRate             */
    uint32_t spacer;               /* g_SpacerBeforeAfter = 100000               */
    uint32_t max_mismatch[ORC_MAX_READ_LEN]; /* g_maxMismatch (pindel.cpp:799-819) */
} orc_params;

/* One UniquePoint (src/pindel.h:137-158). */
typedef struct {
    uint32_t abs_loc;    /* AbsLoc                         */
    int16_t  length;     /* LengthStr                      */
    int16_t  mismatches; /* Mismatches                     */
    int16_t  chr_id;     /* index of chromosome_p          */
    char     direction;  /* '+' FORWARD / '-' BACKWARD     */
    char     strand;     /* '+' SENSE   / '-' ANTISENSE    */
} orc_point;

typedef struct {
    int32_t  chr_id;
    int32_t  start;      /* SearchWindow start (AbsLoc, may be <0: farend_searcher.cpp:69-71) */
    int32_t  end;
} orc_window;

/* createProbTable(seqErrorRate, sensitivity), pindel.cpp:781-819. */
void orc_make_max_mismatch(double seq_error_rate, double sensitivity, uint32_t *table500);

/* Fill defaults of Pindel 0.2.5b9 (-x 2 -a 1 -m 3 -H 8 -u 0.02 -e 0.01 -E 0.95). */
void orc_default_params(orc_params *p);

/*
 * GetCloseEnd (pindel.cpp:2531-2605) + CleanUniquePoints (pindel.cpp:2904-2941).
 * `seq` is the read's UnmatchedSeq (len bytes, already trimmed as
 * setUnmatchedSeq does); it is reverse-complemented IN PLACE exactly when the
 * reference leaves the read reverse-complemented -- through setUnmatchedSeq, which
 * strips the NULs that ReverseComplement puts where the read BEGAN with characters
 * outside ACGTN: *len_out (nullable) = ReadLength afterwards.  Returns the number of
 * points written to `out` (0 = no close end).  *rc_flag = 1: seq was left
 * reverse-complemented; 2: it went through two reverse complements and holds a
 * character outside ACGTN (so it is NOT the original again: those are NUL now, the
 * ones at either end are gone); 0: as it came.  If `clean` is non-zero
 * CleanUniquePoints is applied.
 */
int orc_close_end(const orc_params *p,
                  const char *chr_seq, uint64_t chr_len, int chr_id,
                  char *seq, int len,
                  char anchor_strand, int32_t anchor_pos, int16_t insert_size,
                  int clean,
                  orc_point *out, int cap, int *rc_flag, int *len_out);

/*
 * SearchFarEnd (pindel.cpp:1001-1074): BD-hint cluster first (may be empty),
 * then ranges 64*4^(r-1), r = 1..MaxRangeIndex+1.  `seq` is the read as left
 * by the close-end stage.  close_last_abs_loc / close_max_len describe
 * UP_Close (getLastAbsLocCloseEnd, MaxLenCloseEnd).  Returns number of
 * UP_Far points written.
 */
int orc_far_end(const orc_params *p,
                int n_chr, const char *const *chr_seq, const uint64_t *chr_len,
                int chr_id,
                const char *seq, int len,
                uint32_t close_last_abs_loc, int close_max_len,
                const orc_window *bd, int n_bd,
                orc_point *out, int cap);

/*
 * Whole hot path over a batch (ReadBuffer::flush + SearchFarEnds,
 * read_buffer.cpp:36-101, pindel.cpp:1115-1138), OpenMP over reads.
 * Reads are given as a flat SoA.  seq is modified in place for reads that end
 * up reverse-complemented.  Results are strided: read i's UP_Close points are
 * close_pts[i*stride .. i*stride+close_cnt[i]) (likewise far); stride must be
 * >= the longest read.  close_pts / far_pts may be NULL (counts only; used when timing).  bd/bd_off (per-read BD-hint clusters, CSR, n+1
 * offsets) may be NULL.  Returns 0 or a negative error.
 */
int orc_search_batch(const orc_params *p,
                     int n_chr, const char *const *chr_seq, const uint64_t *chr_len,
                     uint32_t n_reads,
                     char *seq, const uint64_t *seq_off,
                     const char *anchor_strand, const int32_t *anchor_pos,
                     const int16_t *insert_size, const int32_t *chr_id,
                     const orc_window *bd, const uint64_t *bd_off,
                     int do_far, uint32_t stride,
                     uint32_t *close_cnt, orc_point *close_pts,
                     uint32_t *far_cnt, orc_point *far_pts,
                     uint8_t *rc_flag, uint32_t *len_out /* nullable: ReadLength after the close end */, int n_threads);

#ifdef __cplusplus
}
#endif
#endif

/*
 * pg_oracle.c -- CPU restatement of Pindel 0.2.5b9's split-read pattern growth
 * (close end + far end), written from the reference's sources as a checker.
 *
 * TEST INFRASTRUCTURE ONLY (see pg_oracle.h).  It deliberately keeps the
 * reference's data flow -- one position list per mismatch level, rebuilt at
 * every extension step -- so that it is an independent statement of the
 * semantics the HIP kernels (which use a different, histogram-based
 * formulation) are checked against.
 *
 * Reference map (all under /root/reference/src):
 *   orc_make_max_mismatch  pindel.cpp:781-819   probOfReadWithTheseErrors / createProbTable
 *   matches                searcher.cpp:36-44   Matches, tables pindel.cpp:948-970
 *   categorize             searcher.cpp:48-63   CategorizePositions
 *   check_mismatches       searcher.cpp:331-388 CheckMismatches
 *   close_growth           searcher.cpp:153-197 CheckLeft_Close, :247-286 CheckRight_Close,
 *                          :65-97 ExtendMatchClose
 *   close_inner            pindel.cpp:2250-2326 GetCloseEndInner
 *   orc_close_end          pindel.cpp:2531-2575 GetCloseEnd, :2904-2941 CleanUniquePoints
 *   far_at_pos             farend_searcher.cpp:46-103 SearchFarEndAtPos
 *   far_growth             pindel.cpp:2823-2902 CheckBoth, :2673-2725 ExtendMatch
 *   orc_far_end            pindel.cpp:1001-1074 SearchFarEnd
 */
#include "pg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ tables */

/* pindel.cpp:781-792 */
static double prob_of_read_with_these_errors(unsigned length, unsigned n_err, double rate)
{
    double chance_correct = 1.0 - rate;
    unsigned n_correct = length - n_err;
    double matched = pow(chance_correct, (double)n_correct);
    double mismatched = 1.0;
    for (unsigned i = 0; i < n_err; i++)
        mismatched *= (((length - i) * rate) / (n_err - i));
    return matched * mismatched;
}

/* pindel.cpp:799-819 */
void orc_make_max_mismatch(double seq_error_rate, double sensitivity, uint32_t *t)
{
    for (unsigned length = 0; length < ORC_MAX_READ_LEN; length++) {
        double total = 0.0;
        t[length] = 0;
        for (unsigned n_err = 0; n_err <= length; n_err++) {
            total += prob_of_read_with_these_errors(length, n_err, seq_error_rate);
            if (total > sensitivity) {
                t[length] = n_err + 1;
                break;
            }
        }
    }
    t[0] = t[1] = t[2] = t[3] = 0;
}

void orc_default_params(orc_params *p)
{
    p->max_range_index = 2;
    p->additional_mismatch = 1;
    p->min_perfect_match = 3;
    p->min_close = 8;
    p->max_mismatch_rate = 0.02;
    p->spacer = 100000;
    /* pindel.cpp:856: createProbTable(0.001 + Seq_Error_Rate, sensitivity) */
    orc_make_max_mismatch(0.001 + 0.01, 0.95, p->max_mismatch);
}

/* Convert2RC4N, pindel.cpp:966-970 (all other entries are 0) */
static inline char rc4n(char c)
{
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    case 'N': return 'N';
    default:  return 0;
    }
}

/* ReverseComplement, pindel.cpp:2037-2048 */
static void reverse_complement(const char *in, int len, char *out)
{
    for (int j = 0; j < len; j++)
        out[j] = rc4n(in[len - j - 1]);
}

/* Matches, searcher.cpp:36-44; Match2N[ACGT]='N' (pindel.cpp:954-959) */
static inline int matches(char read_base, char ref_base)
{
    if (read_base != 'N')
        return ref_base == read_base;
    return ref_base == 'A' || ref_base == 'C' || ref_base == 'G' || ref_base == 'T';
}

/* ----------------------------------------------------------- small vectors */

typedef struct { uint32_t *v; uint32_t n, cap; } vec;

static inline void vec_push(vec *a, uint32_t x)
{
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 64;
        a->v = (uint32_t *)realloc(a->v, (size_t)a->cap * sizeof(uint32_t));
    }
    a->v[a->n++] = x;
}

#define ORC_MAX_LEVELS 64

typedef struct { vec lv[ORC_MAX_LEVELS]; } levels;

static void levels_free(levels *l)
{
    for (int i = 0; i < ORC_MAX_LEVELS; i++) free(l->lv[i].v);
    memset(l, 0, sizeof(*l));
}

static void levels_clear(levels *l, int total)
{
    for (int i = 0; i < total; i++) l->lv[i].n = 0;
}

/* CategorizePositions, searcher.cpp:48-63 */
static void categorize(char read_base, const char *chr, const levels *in, levels *out,
                       int level, int direction, int max_level)
{
    const vec *src = &in->lv[level];
    for (uint32_t j = 0; j < src->n; j++) {
        uint32_t pos = src->v[j] + (uint32_t)direction;
        if (matches(read_base, chr[pos]))
            vec_push(&out->lv[level], pos);
        else if (level < max_level)
            vec_push(&out->lv[level + 1], pos);
    }
}

/* ------------------------------------------------------- CheckMismatches */

/* searcher.cpp:331-388.  `seq` is the read's current UnmatchedSeq. */
static int check_mismatches(const orc_params *P, const char *chr, const char *seq, int len,
                            const orc_point *up)
{
    char cur[ORC_MAX_READ_LEN + 4];
    int m = P->min_perfect_match;
    if (up->strand == '+')
        memcpy(cur, seq, (size_t)len);
    else
        reverse_complement(seq, len, cur);
    uint32_t start = 0;
    if (up->direction == '+') {           /* FORWARD */
        start = up->abs_loc - (uint32_t)up->length + 1;
        if (up->length <= m) return 0;
        /* substr(LengthStr - m, m) vs ref.substr(AbsLoc - m + 1, m) */
        if (memcmp(cur + up->length - m, chr + up->abs_loc - m + 1, (size_t)m) != 0) return 0;
    } else {                              /* BACKWARD */
        start = up->abs_loc + (uint32_t)up->length - (uint32_t)len;
        if (len < up->length) return 0;
        /* substr(len - LengthStr, m) is clipped at the end of the read */
        int avail = up->length < m ? up->length : m;
        if (avail != m) return 0;          /* strings of different length differ */
        if (memcmp(cur + len - up->length, chr + up->abs_loc, (size_t)m) != 0) return 0;
    }
    float max_allowed = (float)((double)(size_t)len * P->max_mismatch_rate);
    short n_mis = 0;
    for (int i = 0; i < len; i++) {
        char r = chr[start + (uint32_t)i];
        if (cur[i] == 'N') {
            if (!(r == 'A' || r == 'C' || r == 'G' || r == 'T')) n_mis++;
        } else if (r != cur[i]) {
            n_mis++;
        }
    }
    return (float)n_mis >= max_allowed;
}

/* ------------------------------------------------------------- close end */

typedef struct {
    orc_point *out;
    int n, cap;
} point_sink;

static inline void sink_push(point_sink *s, const orc_point *p)
{
    if (s->n < s->cap) s->out[s->n] = *p;
    s->n++;
}

static uint32_t competing(const levels *pd, int max_index)
{
    uint32_t sum = 0;
    for (int j = 0; j <= max_index; j++) sum += pd->lv[j].n;
    return sum;
}

/*
 * CheckLeft_Close / CheckRight_Close + ExtendMatchClose as a loop
 * (the reference recurses once per base; there is no other state).
 * direction +1: left growth, points are (FORWARD, ANTISENSE);
 * direction -1: right growth, points are (BACKWARD, SENSE).
 */
static void close_growth(const orc_params *P, const char *chr, int chr_id,
                         const char *read_seq /* UnmatchedSeq */, const char *cur /* CurrentReadSeq */,
                         int len, int max_snp, int total_snp,
                         levels *a, levels *b, int direction, point_sink *up)
{
    const int bp_start = P->min_close, bp_end = len - 1;
    levels *in = a, *out = b;
    for (int L = 1;; L++) {
        if (L >= bp_start && L <= bp_end) {
            int lo = 0;                                   /* minimumNumberOfMismatches */
            for (; lo <= max_snp; lo++) if (in->lv[lo].n != 0) break;
            if ((uint32_t)lo > P->max_mismatch[L]) return;
            for (int i = 0; i <= max_snp; i++) {
                if (in->lv[i].n == 1 && L >= bp_start + i) {
                    uint32_t sum = competing(in, i + P->additional_mismatch);
                    if (sum == 1 && (uint32_t)i <= P->max_mismatch[L]) {
                        orc_point t;
                        t.abs_loc = in->lv[i].v[0];
                        t.length = (int16_t)L;
                        t.mismatches = (int16_t)i;
                        t.chr_id = (int16_t)chr_id;
                        t.direction = direction == 1 ? '+' : '-';
                        t.strand = direction == 1 ? '-' : '+';
                        if (check_mismatches(P, chr, read_seq, len, &t)) {
                            sink_push(up, &t);
                            break;
                        }
                    }
                }
            }
        }
        if (!(L < bp_end)) return;
        /* ExtendMatchClose */
        levels_clear(out, total_snp);
        char c = direction == 1 ? cur[L] : cur[len - 1 - L];
        for (int i = 0; i <= total_snp - 1; i++)
            categorize(c, chr, in, out, i, direction, total_snp - 1);
        if (competing(out, max_snp) == 0) return;
        levels *t = in; in = out; out = t;
    }
}

/* GetCloseEndInner, pindel.cpp:2250-2326.  Returns number of points. */
static int close_inner(const orc_params *P, const char *chr, int chr_id,
                       const char *seq, int len, char anchor_strand, int32_t anchor_pos,
                       int16_t insert_size, int range_index,
                       levels *a, levels *b, orc_point *out, int cap)
{
    const int max_snp = (int)P->max_mismatch[len];
    const int total_snp = max_snp + P->additional_mismatch + 1;
    char cur[ORC_MAX_READ_LEN + 4];
    point_sink up = { out, 0, cap };
    levels_clear(a, total_snp);
    int start, end;
    if (anchor_strand == '+') {
        reverse_complement(seq, len, cur);
        start = anchor_pos + (int)P->spacer - range_index * insert_size;
        end = start + (2 * range_index + 1) * insert_size;
        char left = cur[0];
        if (left != 'N')
            for (int pos = start; pos < end; pos++)
                if (chr[pos] == left) vec_push(&a->lv[0], (uint32_t)pos);
        close_growth(P, chr, chr_id, seq, cur, len, max_snp, total_snp, a, b, 1, &up);
    } else if (anchor_strand == '-') {
        memcpy(cur, seq, (size_t)len);
        end = anchor_pos + (int)P->spacer + range_index * insert_size;
        start = end - (2 * range_index + 1) * insert_size;
        char right = cur[len - 1];
        if (right != 'N')
            for (int pos = start; pos < end; pos++)
                if (chr[pos] == right) vec_push(&a->lv[0], (uint32_t)pos);
        close_growth(P, chr, chr_id, seq, cur, len, max_snp, total_snp, a, b, -1, &up);
    }
    return up.n;
}

/* CleanUniquePoints, pindel.cpp:2904-2941 (in place, returns new count) */
static int clean_unique_points(orc_point *pts, int n)
{
    if (n <= 0) return n;
    orc_point last = pts[n - 1];
    int m = 0;
    for (int i = 0; i < n; i++) {
        if (pts[i].chr_id != last.chr_id) continue;
        if (pts[i].direction != last.direction || pts[i].strand != last.strand) continue;
        if (last.direction == '+') {
            if (last.abs_loc - (uint32_t)last.length == pts[i].abs_loc - (uint32_t)pts[i].length)
                pts[m++] = pts[i];
        } else if (last.direction == '-') {
            if (last.abs_loc + (uint32_t)last.length == pts[i].abs_loc + (uint32_t)pts[i].length)
                pts[m++] = pts[i];
        }
    }
    return m;
}

/* setUnmatchedSeq, pindel.cpp:142-169: trailing characters that are not alphanumeric are stripped ("while
 * (!isalnum(UnmatchedSeq[lastCharIndex])) resize"); ReadLength, MAX_SNP_ERROR and TOTAL_SNP_ERROR_CHECKED follow the new
 * length (close_inner / far_end_ws take them from `len`).  (A string with NO alphanumeric character walks the reference's
 * unsigned index below zero -- undefined there; here the length becomes 0 and nothing is found.) */
static int strip_trailing_non_alnum(const char *seq, int len)
{
    while (len > 0) {
        unsigned char c = (unsigned char)seq[len - 1];
        if ((c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z')) break;
        len--;
    }
    return len;
}

/* "Temp_One_Read.setUnmatchedSeq( ReverseComplement( Temp_One_Read.getUnmatchedSeq() ) )", pindel.cpp:2545: Convert2RC4N maps
 * every character outside ACGTN to 0 (pindel.cpp:966-970: the table's other entries are zero-initialised), so the characters the
 * read BEGAN with, if they are not ACGTN, arrive at the end as NULs and are stripped: the read gets SHORTER, and stays short. */
static int rc_and_strip(char *seq, int len)
{
    char tmp[ORC_MAX_READ_LEN + 4];
    reverse_complement(seq, len, tmp);
    memcpy(seq, tmp, (size_t)len);
    return strip_trailing_non_alnum(seq, len);
}

static int close_end_ws(const orc_params *P, const char *chr, int chr_id, char *seq, int *len_io,
                        char anchor_strand, int32_t anchor_pos, int16_t insert_size, int clean,
                        levels *a, levels *b, orc_point *out, int cap, int *rc_flag)
{
    /* GetCloseEnd, pindel.cpp:2531-2575: MaxRange = 2; on failure the read is
     * reverse-complemented (setUnmatchedSeq, :2545) and STAYS so -- with whatever length setUnmatchedSeq left it. */
    int n = 0, n_rc = 0, len = *len_io, other = 0;
    for (int i = 0; i < len; i++) other |= rc4n(seq[i]) == 0;          /* a character two reverse complements do not restore */
    for (int range_index = 0; range_index < 2; range_index++) {
        n = len > 0 ? close_inner(P, chr, chr_id, seq, len, anchor_strand, anchor_pos, insert_size,
                                  range_index, a, b, out, cap) : 0;
        if (n == 0) {
            len = rc_and_strip(seq, len);
            n_rc++;
            n = len > 0 ? close_inner(P, chr, chr_id, seq, len, anchor_strand, anchor_pos, insert_size,
                                      range_index, a, b, out, cap) : 0;
        }
        if (n > 0) break;
    }
    /* 1: the read is left reverse-complemented; 2: it went through TWO reverse complements that did not give the original back
     * (characters outside ACGTN are NUL now, those at either end are gone); 0: it is as it came */
    if (rc_flag) *rc_flag = n_rc == 1 ? 1 : (n_rc == 2 && other ? 2 : 0);
    *len_io = len;
    if (n > cap) n = cap;
    if (clean && n > 0) n = clean_unique_points(out, n);
    return n;
}

int orc_close_end(const orc_params *P, const char *chr, uint64_t chr_len, int chr_id,
                  char *seq, int len, char anchor_strand, int32_t anchor_pos,
                  int16_t insert_size, int clean, orc_point *out, int cap, int *rc_flag, int *len_out)
{
    (void)chr_len;
    if (len <= 0 || len >= ORC_MAX_READ_LEN) return -1;
    levels a, b;
    memset(&a, 0, sizeof a);
    memset(&b, 0, sizeof b);
    int n = close_end_ws(P, chr, chr_id, seq, &len, anchor_strand, anchor_pos, insert_size, clean,
                         &a, &b, out, cap, rc_flag);
    if (len_out) *len_out = len;
    levels_free(&a);
    levels_free(&b);
    return n;
}

/* --------------------------------------------------------------- far end */

/* FarEndSearchPerRegion, farend_searcher.h:26-51 */
typedef struct {
    int chr_id;
    levels plus, minus;
} region;

typedef struct {
    region *r;
    int n, cap;
} region_set;

static region *region_set_add(region_set *s, int chr_id, int total)
{
    if (s->n == s->cap) {
        int ncap = s->cap ? s->cap * 2 : 4;
        s->r = (region *)realloc(s->r, (size_t)ncap * sizeof(region));
        memset(s->r + s->cap, 0, (size_t)(ncap - s->cap) * sizeof(region));
        s->cap = ncap;
    }
    region *g = &s->r[s->n++];
    g->chr_id = chr_id;
    levels_clear(&g->plus, total);
    levels_clear(&g->minus, total);
    return g;
}

static void region_set_free(region_set *s)
{
    for (int i = 0; i < s->cap; i++) {
        levels_free(&s->r[i].plus);
        levels_free(&s->r[i].minus);
    }
    free(s->r);
    memset(s, 0, sizeof *s);
}

/* CheckBoth + ExtendMatch as a loop, pindel.cpp:2823-2902 / :2673-2725 */
static void far_growth(const orc_params *P, const char *const *chr_seq,
                       const char *seq, int len, int max_snp, int total_snp,
                       region_set *a, region_set *b, point_sink *up)
{
    const int bp_start = 10, bp_end = len - 1;      /* farend_searcher.cpp:90-91 */
    region_set *in = a, *out = b;
    for (int L = 1;; L++) {
        if (L >= bp_start && L <= bp_end) {
            /* minimumNumberOfMismatches(regions), pindel.cpp:2727-2740 */
            uint32_t sum = 0;
            int lo = 0;
            for (; lo <= max_snp; lo++) {
                for (int r = 0; r < in->n; r++)
                    sum += in->r[r].plus.lv[lo].n + in->r[r].minus.lv[lo].n;
                if (sum != 0) break;
            }
            if ((uint32_t)lo > P->max_mismatch[L]) return;
            uint32_t less = 0;
            for (int nm = 0; nm <= max_snp; nm++) {
                if (less) break;
                int s = 0;
                for (int r = 0; r < in->n; r++)
                    s += (int)(in->r[r].plus.lv[nm].n + in->r[r].minus.lv[nm].n);
                less = (uint32_t)s;
                if (s == 1 && L >= bp_start + nm) {
                    s = 0;
                    if (P->additional_mismatch > 0) {
                        int region_with_match = 0;
                        for (int mc = 0; mc <= nm + P->additional_mismatch; mc++)
                            for (int r = 0; r < in->n; r++) {
                                uint32_t hits = in->r[r].plus.lv[mc].n + in->r[r].minus.lv[mc].n;
                                s += (int)hits;
                                if (hits > 0) region_with_match = r;
                            }
                        if (s == 1 && (uint32_t)nm <= P->max_mismatch[L]) {
                            const region *hit = &in->r[region_with_match];
                            orc_point t;
                            t.length = (int16_t)L;
                            t.mismatches = (int16_t)nm;
                            t.chr_id = (int16_t)hit->chr_id;
                            if (hit->plus.lv[nm].n == 1) {
                                t.abs_loc = hit->plus.lv[nm].v[0];
                                t.direction = '+'; t.strand = '+';   /* FORWARD, SENSE */
                            } else {
                                t.abs_loc = hit->minus.lv[nm].v[0];
                                t.direction = '-'; t.strand = '-';   /* BACKWARD, ANTISENSE */
                            }
                            if (check_mismatches(P, chr_seq[hit->chr_id], seq, len, &t)) {
                                sink_push(up, &t);
                                break;
                            }
                        }
                    }
                }
            }
        }
        if (!(L < bp_end)) return;
        /* ExtendMatch */
        char c = seq[L];
        char c_rc = rc4n(c);
        out->n = 0;
        for (int r = 0; r < in->n; r++) {
            const region *gi = &in->r[r];
            region *go = region_set_add(out, gi->chr_id, total_snp);
            const char *chr = chr_seq[gi->chr_id];
            for (int i = 0; i <= total_snp - 1; i++) {
                categorize(c, chr, &gi->plus, &go->plus, i, 1, total_snp - 1);
                categorize(c_rc, chr, &gi->minus, &go->minus, i, -1, total_snp - 1);
            }
            uint32_t cnt = 0;                        /* CountElements */
            for (int i = 0; i < total_snp; i++) cnt += go->plus.lv[i].n + go->minus.lv[i].n;
            if (cnt == 0) out->n--;                  /* region dropped */
        }
        if (out->n == 0) return;
        region_set *t = in; in = out; out = t;
    }
}

typedef struct {
    orc_point *pts;   /* current UP_Far */
    int n;
    orc_point *tmp;   /* scratch for one SearchFarEndAtPos */
    int cap;
} far_state;

/* SearchFarEndAtPos, farend_searcher.cpp:46-103 */
static void far_at_pos(const orc_params *P, const char *const *chr_seq, const uint64_t *chr_len,
                       const char *seq, int len, int close_max_len,
                       const orc_window *w, int nw, region_set *a, region_set *b, far_state *st)
{
    char base = seq[0];
    char base_rc = rc4n(base);
    if (base == 'N' || close_max_len == 0) return;
    const int max_snp = (int)P->max_mismatch[len];
    const int total_snp = max_snp + P->additional_mismatch + 1;
    a->n = 0;
    uint32_t hits = 0;
    for (int r = 0; r < nw; r++) {
        region *g = region_set_add(a, w[r].chr_id, total_snp);
        int start = w[r].start, end = w[r].end;
        if (start < 0) start = end - 1;
        const char *chr = chr_seq[w[r].chr_id];
        for (int pos = start; pos < end; pos++) {
            if (pos < 0 || (uint64_t)pos >= chr_len[w[r].chr_id]) continue; /* .at() would throw */
            if (chr[pos] == base) vec_push(&g->plus.lv[0], (uint32_t)pos);
            else if (chr[pos] == base_rc) vec_push(&g->minus.lv[0], (uint32_t)pos);
        }
        hits += g->plus.lv[0].n + g->minus.lv[0].n;
    }
    if (hits > 0) {
        point_sink up = { st->tmp, 0, st->cap };
        far_growth(P, chr_seq, seq, len, max_snp, total_snp, a, b, &up);
        int n_new = up.n > st->cap ? st->cap : up.n;
        /* NewUPFarIsBetter, farend_searcher.cpp:30-44: replace iff new.MaxLen >= old.MaxLen */
        int new_max = n_new ? st->tmp[n_new - 1].length : 0;
        int old_max = st->n ? st->pts[st->n - 1].length : 0;
        if (!(new_max < old_max)) {
            memcpy(st->pts, st->tmp, (size_t)n_new * sizeof(orc_point));
            st->n = n_new;
        }
    }
}

static int far_end_ws(const orc_params *P, int n_chr, const char *const *chr_seq,
                      const uint64_t *chr_len, int chr_id, const char *seq, int len,
                      uint32_t close_last_abs_loc, int close_max_len,
                      const orc_window *bd, int n_bd,
                      region_set *a, region_set *b, orc_point *out, orc_point *tmp, int cap)
{
    (void)n_chr;
    far_state st = { out, 0, tmp, cap };
    /* goodFarEndFound, pindel.cpp:480-483 */
#define GOOD_FAR_END() ((unsigned)((st.n ? st.pts[st.n - 1].length : 0) + close_max_len) >= (unsigned)len)
    if (n_bd != 0) {
        far_at_pos(P, chr_seq, chr_len, seq, len, close_max_len, bd, n_bd, a, b, &st);
        if (GOOD_FAR_END()) return st.n;
    }
    uint32_t span = 64;                              /* START_SEARCH_SPAN */
    uint32_t center = close_last_abs_loc;
    uint64_t size = chr_len[chr_id];
    for (int range_index = 1; range_index <= P->max_range_index + 1; range_index++) {
        uint32_t start, end;
        if (center > span + P->spacer) start = center - span;
        else start = P->spacer;
        if ((uint64_t)center + span + P->spacer < size) end = center + span;
        else end = (uint32_t)(size - P->spacer);
        orc_window w = { chr_id, (int32_t)start, (int32_t)end };
        far_at_pos(P, chr_seq, chr_len, seq, len, close_max_len, &w, 1, a, b, &st);
        if (GOOD_FAR_END()) return st.n;
        span *= 4;
    }
#undef GOOD_FAR_END
    return st.n;
}

int orc_far_end(const orc_params *P, int n_chr, const char *const *chr_seq,
                const uint64_t *chr_len, int chr_id, const char *seq, int len,
                uint32_t close_last_abs_loc, int close_max_len,
                const orc_window *bd, int n_bd, orc_point *out, int cap)
{
    if (len <= 0 || len >= ORC_MAX_READ_LEN) return -1;
    region_set a, b;
    memset(&a, 0, sizeof a);
    memset(&b, 0, sizeof b);
    orc_point *tmp = (orc_point *)malloc((size_t)cap * sizeof(orc_point));
    int n = far_end_ws(P, n_chr, chr_seq, chr_len, chr_id, seq, len, close_last_abs_loc,
                       close_max_len, bd, n_bd, &a, &b, out, tmp, cap);
    free(tmp);
    region_set_free(&a);
    region_set_free(&b);
    return n;
}

/* ----------------------------------------------------------------- batch */

int orc_search_batch(const orc_params *P, int n_chr, const char *const *chr_seq,
                     const uint64_t *chr_len, uint32_t n_reads, char *seq,
                     const uint64_t *seq_off, const char *anchor_strand,
                     const int32_t *anchor_pos, const int16_t *insert_size,
                     const int32_t *chr_id, const orc_window *bd, const uint64_t *bd_off,
                     int do_far, uint32_t stride, uint32_t *close_cnt, orc_point *close_pts,
                     uint32_t *far_cnt, orc_point *far_pts, uint8_t *rc_flag, uint32_t *len_out, int n_threads)
{
    int err = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
    #pragma omp parallel
    {
        levels a, b;
        region_set ra, rb;
        memset(&a, 0, sizeof a);
        memset(&b, 0, sizeof b);
        memset(&ra, 0, sizeof ra);
        memset(&rb, 0, sizeof rb);
        orc_point *tmp = (orc_point *)malloc((size_t)stride * sizeof(orc_point));
        /* close_pts / far_pts may be NULL (timing runs): points then go to per-thread scratch */
        orc_point *scratch_c = (orc_point *)malloc((size_t)stride * sizeof(orc_point));
        orc_point *scratch_f = (orc_point *)malloc((size_t)stride * sizeof(orc_point));
        #pragma omp for schedule(dynamic, 64)
        for (int64_t i = 0; i < (int64_t)n_reads; i++) {
            int len = (int)(seq_off[i + 1] - seq_off[i]);
            close_cnt[i] = 0;
            if (far_cnt) far_cnt[i] = 0;
            rc_flag[i] = 0;
            if (len <= 0 || len >= ORC_MAX_READ_LEN || (uint32_t)len > stride) { err = -1; continue; }
            char *s = seq + seq_off[i];
            int cid = chr_id[i];
            int flip = 0;
            orc_point *cp = close_pts ? close_pts + (size_t)i * stride : scratch_c;
            int nc = close_end_ws(P, chr_seq[cid], cid, s, &len, anchor_strand[i], anchor_pos[i],
                                  insert_size[i], 1, &a, &b, cp, (int)stride, &flip);
            close_cnt[i] = (uint32_t)nc;
            rc_flag[i] = (uint8_t)flip;
            if (len_out) len_out[i] = (uint32_t)len;           /* ReadLength as GetCloseEnd left it (the far end searches with it) */
            /* read_buffer.cpp:55: only reads with a close end go on */
            if (do_far && nc > 0) {
                const orc_window *w = NULL;
                int nw = 0;
                if (bd && bd_off) { w = bd + bd_off[i]; nw = (int)(bd_off[i + 1] - bd_off[i]); }
                int nf = far_end_ws(P, n_chr, chr_seq, chr_len, cid, s, len,
                                    cp[nc - 1].abs_loc, cp[nc - 1].length, w, nw, &ra, &rb,
                                    far_pts ? far_pts + (size_t)i * stride : scratch_f, tmp, (int)stride);
                far_cnt[i] = (uint32_t)nf;
            }
        }
        free(tmp);
        free(scratch_c);
        free(scratch_f);
        levels_free(&a);
        levels_free(&b);
        region_set_free(&ra);
        region_set_free(&rb);
    }
    return err;
}

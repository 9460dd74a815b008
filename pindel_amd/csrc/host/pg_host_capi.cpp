// pg_host_capi.cpp -- C entry points of the host library (libpindel_host.so): the steps
// before and after the hot path (loaders, classifiers, reporters), callable from tests and
// from the pindel_pg command line.  No search code here.
#include <cctype>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "pg_bam.hpp"
#include "pg_bdhints.hpp"
#include "pg_host.hpp"
#include "pg_host_priv.hpp"
#include "pg_pipeline.hpp"
#include "pg_rp.hpp"
#include "pindel_pg.h"

using namespace pgh;

static std::string g_err;

extern "C" {

const char *pgh_last_error(void) { return g_err.c_str(); }

struct pgh_settings {
    uint32_t spacer;
    uint32_t min_support;        /* -M */
    uint32_t balance_cutoff;     /* -B */
    double seq_error_rate;       /* -e */
    int32_t min_num_matched_bases; /* -d */
    int32_t min_inversion_size;  /* -v */
    int32_t analyze_td, analyze_inv;
    double window_mbp;           /* -w */
    uint32_t max_mismatch[500];  /* g_maxMismatch (pg_get_max_mismatch) */
};

/*
 * Pindel's post-search pipeline for a Pindel-text read file: attaches the given UP_Close /
 * UP_Far (CSR over ALL reads of the file, in file order; reads without close end have an
 * empty range) and the rc flags, then walks chromosomes and 5-Mbp bins like main()
 * (pindel.cpp:1778-1989) and appends <prefix>_D, _SI, _TD, _INV.
 */
int pgh_call_from_points(const char *fasta_path, const char *reads_path, const char *out_prefix,
                         const pgh_settings *st, uint32_t n_reads,
                         const uint64_t *close_off, const pg_point *close_pts,
                         const uint64_t *far_off, const pg_point *far_pts, const uint8_t *rc_flag)
{
    std::vector<Chromosome> genome;
    if (load_fasta(fasta_path, genome, st->spacer, g_err)) return -1;
    std::vector<SplitRead> all;
    if (load_pindel_text(reads_path, genome, all, g_err)) return -1;
    if (all.size() != n_reads) {
        g_err = "read count mismatch between the read file and the point arrays";
        return -1;
    }
    Settings S;
    S.spacer = st->spacer;
    S.NumRead2ReportCutOff = st->min_support;
    S.BalanceCutoff = st->balance_cutoff;
    S.Seq_Error_Rate = st->seq_error_rate;
    S.Min_Num_Matched_Bases = st->min_num_matched_bases;
    S.MIN_IndelSize_Inversion = st->min_inversion_size;
    S.Analyze_TD = st->analyze_td != 0;
    S.Analyze_INV = st->analyze_inv != 0;
    S.window_mbp = st->window_mbp;
    memcpy(S.max_mismatch, st->max_mismatch, sizeof S.max_mismatch);
    std::vector<unsigned> fai = read_fai(fasta_path, genome);
    auto to_up = [](const pg_point &p) {
        UniquePoint u;
        u.chr = p.chr_id;
        u.LengthStr = p.length;
        u.AbsLoc = p.abs_loc;
        u.Direction = p.direction;
        u.Strand = p.strand;
        u.Mismatches = p.mismatches;
        return u;
    };
    auto attach = [&](const Chromosome &, int, std::vector<SplitRead> &reads, const std::vector<uint32_t> &index) {
        for (size_t k = 0; k < reads.size(); k++) {
            const uint32_t i = index[k];
            SplitRead &r = reads[k];
            for (uint8_t k = 0; k < rc_flag[i] && k < 2; k++) {      // setUnmatchedSeq(ReverseComplement()), once or twice
                r.UnmatchedSeq = reverse_complement(r.UnmatchedSeq);
                while (!r.UnmatchedSeq.empty() && !std::isalnum((unsigned char)r.UnmatchedSeq.back())) r.UnmatchedSeq.pop_back();
            }
            for (uint64_t q = close_off[i]; q < close_off[i + 1]; q++) r.UP_Close.push_back(to_up(close_pts[q]));
            for (uint64_t q = far_off[i]; q < far_off[i + 1]; q++) r.UP_Far.push_back(to_up(far_pts[q]));
        }
        return 0;
    };
    return run_pipeline(genome, fai, all, S, out_prefix, attach, pgh::NoFarSearch(), g_err);
}

// BreakDancer hints (pg_bdhints.hpp) for one bin: clusters of the reads whose last close-end point is at
// q[i].  out_off has n_q + 1 entries, out_win 3 ints (chr id, start, end) per window; returns the status
// of load_file (0 / 1 = ignored), -1 = cannot open, -2 = unknown chromosome, -3 = out_win too small.
int pgh_bd_query(const char *path, uint32_t spacer, int32_t n_chr, const char *const *names, int32_t chr_id,
                 uint32_t start, uint32_t end, uint32_t n_q, const uint32_t *q, uint64_t *out_off,
                 int32_t *out_win, uint64_t cap, uint64_t *n_events)
{
    pgh::BDHints h;
    std::string note;
    const int rc = h.load_file(path, spacer, note);
    if (n_events) *n_events = h.n_events();
    g_err = note;
    if (rc < 0) return -1;
    std::vector<std::string> nm(names, names + n_chr);
    if (!h.load_region(nm, chr_id, start, end, g_err)) return -2;
    uint64_t k = 0;
    out_off[0] = 0;
    for (uint32_t i = 0; i < n_q; i++) {
        for (const pgh::BDWindow &w : h.cluster(q[i])) {
            if (k >= cap) return -3;
            out_win[3 * k] = w.chr_id;
            out_win[3 * k + 1] = (int32_t)w.start;
            out_win[3 * k + 2] = (int32_t)w.end;
            k++;
        }
        out_off[i + 1] = k;
    }
    return rc;
}

// BAM ingest (pg_bam.hpp): the split-read candidates of one window of one BAM file as the SoA batch of the C ABI.
// Returns a handle (pgh_bam_ingest_free) or null (pgh_last_error).
void *pgh_bam_ingest(const char *bam_path, const char *chr_name, int32_t chr_id, uint64_t chr_padded_size, int64_t win_start,
                     int64_t win_end, int32_t insert_size, const char *tag, uint32_t min_anchor_quality, uint32_t spacer,
                     int32_t use_index, uint64_t *n_reads, uint64_t *n_bases)
{
    pgh::BamFile bam;
    if (!bam.open(bam_path, g_err, use_index != 0)) return nullptr;
    pgh::BamIngestSettings st;
    st.min_anchor_quality = min_anchor_quality;
    st.spacer = spacer;
    pgh::BamIngest ing(st);
    pgh::IngestedReads *out = new pgh::IngestedReads();
    out->clear();
    if (!ing.read_window(bam, chr_name, chr_id, chr_padded_size, win_start, win_end, insert_size, tag ? tag : "", *out)) {
        g_err = ing.error;
        delete out;
        return nullptr;
    }
    if (n_reads) *n_reads = out->size();
    if (n_bases) *n_bases = out->batch.seq.size();
    return out;
}

int pgh_bam_ingest_view(void *h, const uint8_t **seq, const uint64_t **off, const uint8_t **strand, const int32_t **pos,
                        const int16_t **isz, const int32_t **chr, const int16_t **ms)
{
    if (!h) return -1;
    pgh::IngestedReads *r = (pgh::IngestedReads *)h;
    *seq = r->batch.seq.data();
    *off = r->batch.off.data();
    *strand = r->batch.strand.data();
    *pos = r->batch.pos.data();
    *isz = r->batch.isz.data();
    *chr = r->batch.chr.data();
    *ms = r->ms.data();
    return 0;
}

const char *pgh_bam_ingest_name(void *h, uint64_t i)
{
    pgh::IngestedReads *r = (pgh::IngestedReads *)h;
    return (r && i < r->names.size()) ? r->names[i].c_str() : nullptr;
}

// the reference-supporting reads of the window (isRefRead / build_record_RefRead): 3 values each (pos, length, tag index)
uint64_t pgh_bam_ingest_ref_reads(void *h, uint32_t *out, uint64_t cap)
{
    pgh::IngestedReads *r = (pgh::IngestedReads *)h;
    if (!r) return 0;
    for (size_t i = 0; i < r->ref_reads.size() && i < cap; i++) {
        out[3 * i] = r->ref_reads[i].pos;
        out[3 * i + 1] = r->ref_reads[i].length;
        out[3 * i + 2] = r->ref_reads[i].tag;
    }
    return r->ref_reads.size();
}

void pgh_bam_ingest_free(void *h) { delete (pgh::IngestedReads *)h; }

// What the BAI reader makes of an index file (test hook: the reference ships .bai files of its demo BAMs): per
// reference 4 values -- bins (without the pseudo-bin), chunks, 64-bit sum of all chunk begin/end offsets, linear-index
// entries; returns the number of references or -1.
int32_t pgh_bai_summary(const char *bai_path, uint64_t *out, uint32_t cap_refs)
{
    FILE *f = fopen(bai_path, "rb");
    if (!f) return -1;
    pgh::BamFile::BaiBins bins;
    std::vector<std::vector<uint64_t>> linear;
    const bool ok = pgh::BamFile::read_bai(f, bins, linear);
    fclose(f);
    if (!ok) return -1;
    for (size_t t = 0; t < bins.size() && t < cap_refs; t++) {
        uint64_t chunks = 0, sum = 0;
        for (const auto &kv : bins[t])
            for (const auto &c : kv.second) {
                chunks++;
                sum += c.first + c.second;
            }
        for (uint64_t v : linear[t]) sum += v;
        out[4 * t] = bins[t].size();
        out[4 * t + 1] = chunks;
        out[4 * t + 2] = sum;
        out[4 * t + 3] = linear[t].size();
    }
    return (int32_t)bins.size();
}

// Read-pair discovery (pg_rp.hpp): the BreakDancer-like events of one window of one BAM.  out receives 4 values per
// event (pos1, pos1b, pos2, pos2b, Pindel coordinates); rp_path (nullable): the lines of <prefix>_RP.
int64_t pgh_rp_events(const char *bam_path, const char *chr_name, int64_t win_start, int64_t win_end, int32_t insert_size,
                      const char *tag, uint32_t min_anchor_quality, uint32_t spacer, const char *rp_path, uint32_t *out, uint64_t cap)
{
    pgh::BamFile bam;
    if (!bam.open(bam_path, g_err)) return -1;
    std::vector<pgh::RpRead> rp;
    if (!pgh::rp_discover(bam, chr_name, win_start, win_end, insert_size, tag ? tag : "", min_anchor_quality, rp)) return -1;
    std::ofstream f;
    if (rp_path) f.open(rp_path, std::ios::trunc);
    const std::vector<pgh::RpEvent> ev = pgh::rp_events(rp, spacer, rp_path ? &f : nullptr);
    for (size_t i = 0; i < ev.size() && i < cap; i++) {
        out[4 * i] = ev[i].pos1;
        out[4 * i + 1] = ev[i].pos1b;
        out[4 * i + 2] = ev[i].pos2;
        out[4 * i + 3] = ev[i].pos2b;
    }
    return (int64_t)ev.size();
}

// The window hints of one bin exactly as `pindel_pg -i ... [-b file]` with -R hands them to the far end
// (run_bam_pipeline): events of the -b file (bd_path may be null / empty) + the read-pair events of this window of
// this BAM (UpdateBD; [win_start, win_end) = the window as clipped to the chromosome), loadRegion for the bin
// [win_start, region_end) (the unclipped bin, as main() hands it over), then the cluster of every query position
// (= last close-end AbsLoc).
// out_off: n_q + 1 entries, out_win: 3 ints (chr id, start, end) per window.  Returns the number of read-pair
// events, -1 on a file error, -2 unknown chromosome, -3 out_win too small.
int64_t pgh_window_hints(const char *bd_path, const char *bam_path, int32_t n_chr, const char *const *names, int32_t chr_id,
                         int64_t win_start, int64_t win_end, int64_t region_end, int32_t insert_size, const char *tag,
                         uint32_t min_anchor_quality, uint32_t spacer, uint32_t n_q, const uint32_t *q, uint64_t *out_off, int32_t *out_win, uint64_t cap)
{
    pgh::BDHints h;
    std::string note;
    if (bd_path && bd_path[0] && h.load_file(bd_path, spacer, note) < 0) {
        g_err = note;
        return -1;
    }
    std::vector<std::string> nm(names, names + n_chr);
    pgh::BamFile bam;
    if (!bam.open(bam_path, g_err)) return -1;
    std::vector<pgh::RpRead> rp;
    if (!pgh::rp_discover(bam, nm[chr_id], win_start, win_end, insert_size, tag ? tag : "", min_anchor_quality, rp)) return -1;
    const std::vector<pgh::RpEvent> ev = pgh::rp_events(rp, spacer, nullptr);
    std::vector<std::pair<pgh::BDHints::RpSide, pgh::BDHints::RpSide>> sides;
    for (const pgh::RpEvent &e : ev) {
        pgh::BDHints::RpSide a = { e.chr1, e.pos1, e.pos1b }, b = { e.chr2, e.pos2, e.pos2b };
        sides.push_back(std::make_pair(a, b));
    }
    h.update_with_rp(sides);
    if (!h.load_region(nm, chr_id, (unsigned)win_start + spacer, (unsigned)region_end + spacer, g_err)) return -2;
    uint64_t k = 0;
    out_off[0] = 0;
    for (uint32_t i = 0; i < n_q; i++) {
        for (const pgh::BDWindow &w : h.cluster(q[i])) {
            if (k >= cap) return -3;
            out_win[3 * k] = w.chr_id;
            out_win[3 * k + 1] = (int32_t)w.start;
            out_win[3 * k + 2] = (int32_t)w.end;
            k++;
        }
        out_off[i + 1] = k;
    }
    return (int64_t)ev.size();
}

// Test hook (tests/test_cpu_suite.py): sorts indices 0..n-1 by keys[] with the reference's O(n^2)
// exchange sort and with its fast equivalent; the two outputs must be identical.
void pgh_test_exchange_sort(const int32_t *keys, uint32_t n, uint32_t *out_reference, uint32_t *out_fast)
{
    std::vector<unsigned> a(n), b(n);
    for (uint32_t i = 0; i < n; i++) a[i] = b[i] = i;
    auto less = [&](unsigned x, unsigned y) { return keys[x] < keys[y]; };
    pgh::detail::exchange_sort_reference(a, less);
    pgh::detail::exchange_sort_fast(b, less);
    for (uint32_t i = 0; i < n; i++) {
        out_reference[i] = a[i];
        out_fast[i] = b[i];
    }
}

}  // extern "C"

// pg_pipeline.hpp -- main()'s chromosome / 5-Mbp-bin loop (src/pindel.cpp:1778-1989) around a
// pluggable search step.  Shared by the pindel_pg command line (search = GPU through the C ABI)
// and by pgh_call_from_points (search = attach externally computed points, used by the tests).
#ifndef PG_PIPELINE_HPP
#define PG_PIPELINE_HPP

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <future>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "pg_bam.hpp"
#include "pg_bdhints.hpp"
#include "pg_host.hpp"
#include "pg_rp.hpp"

namespace pgh {

// .fai sizes (init_g_ChrNameAndSizeAndIndex, pindel.cpp:1332-1348); 0 when absent
inline std::vector<unsigned> read_fai(const std::string &fasta_path, const std::vector<Chromosome> &genome)
{
    std::vector<unsigned> fai(genome.size(), 0);
    std::ifstream f((fasta_path + ".fai").c_str());
    std::string name, rest;
    unsigned size;
    while (f >> name >> size) {
        std::getline(f, rest);
        for (size_t c = 0; c < genome.size(); c++)
            if (genome[c].name == name) fai[c] = size;
    }
    return fai;
}

// Frees the strings and point lists of a window's reads on several threads (the destructors of ~10^7 reads are
// seconds of single-threaded work otherwise); the vector itself is left empty.
inline void release_reads(std::vector<SplitRead> &v)
{
    pg_adapter::parallel_ranges(v.size(), [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            SplitRead gone;
            std::swap(gone, v[i]);
        }
    });
    std::vector<SplitRead>().swap(v);
}

// The two seams, in the reference's order (src/pindel.cpp:1816-1888):
//   search(chrom, chr_id, reads, index_in_all)   on ALL reads of the window: must fill UP_Close (empty when there is no
//                                                close end) and leave UnmatchedSeq as GetCloseEnd would
//                                                (ReadBuffer::flush, src/read_buffer.cpp:36-101); may fill UP_Far too
//   far_search(chrom, chr_id, kept, ws, we)      on the reads that kept a close end (state.Reads_SR): fills UP_Far
//                                                (SearchFarEnds, src/pindel.cpp:1115-1138, called at :1888); [ws, we) =
//                                                the window being processed, biological coordinates (currentWindow,
//                                                src/pindel.cpp:1828: what g_bdData.loadRegion is given at :1853)
struct NoFarSearch {
    int operator()(const Chromosome &, int, std::vector<SplitRead> &, unsigned, unsigned) const { return 0; }
};

template <class Search, class FarSearch>
int run_pipeline(const std::vector<Chromosome> &genome, const std::vector<unsigned> &fai,
                 const std::vector<SplitRead> &all, const Settings &S, const std::string &prefix,
                 Search search, FarSearch far_search, std::string &err)
{
    Caller caller(S, &genome, prefix, true);
    const unsigned WINDOW = (unsigned)(S.window_mbp * 1000000);
    if (WINDOW == 0) {
        err = "window size (-w) must be at least 0.000001 Mbp";
        return -1;
    }
    // PGH_TIMING=1: wall-clock seconds per stage of this loop on stderr (diagnostics)
    const bool timing = getenv("PGH_TIMING") != nullptr;
    double t_copy = 0, t_search = 0, t_keep = 0, t_call = 0, t_free = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // The Pindel-text reader rescans the whole file for every window and raises g_maxPos for EVERY read it
    // passes (reader.cpp:224-226), so after the first window of a chromosome g_maxPos is the largest position
    // in the file; the windows then run until they pass it.  Here the reads are bucketed once: per chromosome
    // the indices of its reads by window, in input order (a stable counting sort), instead of one pass over
    // all reads per window.
    unsigned file_max_pos = 0;
    std::vector<std::vector<uint32_t>> by_chr(genome.size());
    for (uint32_t i = 0; i < all.size(); i++) {
        file_max_pos = std::max(file_max_pos, all[i].MatchedRelPos);
        if (all[i].chr_id >= 0 && all[i].chr_id < (int)genome.size()) by_chr[all[i].chr_id].push_back(i);
    }
    for (size_t c = 0; c < genome.size(); c++) {
        const Chromosome &chrom = genome[c];
        const unsigned biol = (unsigned)(chrom.seq.size() - 2 * S.spacer);
        const unsigned bed_start = 1, bed_end = fai[c] ? fai[c] : biol;   // "-c ALL": one BED record per chromosome
        const unsigned global_end = std::min(biol, bed_end + 10000u);      // AROUND_REGION_BUFFER
        // windows that will be visited: ws = 0, W, 2W, ... while !(ws >= g_maxPos || ws > global_end), at least one
        std::vector<unsigned> starts;
        {
            unsigned ws = 0;
            do {
                starts.push_back(ws);
                ws += WINDOW;
            } while (!(ws >= file_max_pos || ws > global_end));
        }
        std::vector<std::vector<uint32_t>> bins(starts.size());
        for (uint32_t i : by_chr[c]) {
            const size_t w = all[i].MatchedRelPos / WINDOW;
            // a read is picked up by window w iff ws <= pos < min(ws + W, global_end)
            if (w < starts.size() && all[i].MatchedRelPos < std::min(starts[w] + WINDOW, global_end)) bins[w].push_back(i);
        }
        for (size_t w = 0; w < starts.size(); w++) {
            if (bins[w].empty()) continue;
            const unsigned ws = starts[w], we = std::min(ws + WINDOW, global_end);
            double t0 = now();
            std::vector<SplitRead> reads(bins[w].size());
            pg_adapter::parallel_ranges(reads.size(), [&](size_t lo, size_t hi) {
                for (size_t k = lo; k < hi; k++) {
                    const uint32_t i = bins[w][k];
                    reads[k] = all[i];
                    if (reads[k].MatchedRelPos > biol) reads[k].MatchedRelPos = biol;   // reader.cpp:233-235
                    reads[k].MAX_SNP_ERROR = (short)S.max_mismatch[std::min<int>(all[i].ReadLength, 499)];
                }
            });
            t_copy += now() - t0; t0 = now();
            int rc = search(chrom, (int)c, reads, bins[w]);
            if (rc) {
                err = "search step failed";
                return rc;
            }
            t_search += now() - t0; t0 = now();
            caller.note_close_mapped_all(reads);                        // reader.cpp:258-291
            std::vector<SplitRead> kept;
            {
                size_t n_kept = 0;
                for (const SplitRead &r : reads) n_kept += r.UP_Close.empty() ? 0 : 1;
                kept.reserve(n_kept);
            }
            for (SplitRead &r : reads)
                if (!r.UP_Close.empty()) kept.push_back(std::move(r));      // `reads` is not used after this loop
            t_keep += now() - t0; t0 = now();
            if (!kept.empty() && (rc = far_search(chrom, (int)c, kept, ws, we))) {
                err = "far-end search step failed";
                return rc;
            }
            t_search += now() - t0; t0 = now();
            if (!kept.empty()) caller.process_window(chrom, kept, ws, we, bed_start, bed_end);
            t_call += now() - t0; t0 = now();
            release_reads(kept);
            release_reads(reads);
            t_free += now() - t0;
        }
    }
    if (timing)
        fprintf(stderr, "pgh timing: pipeline: copy reads %.3f s, search step %.3f s, keep %.3f s, classify + report %.3f s, free %.3f s\n",
                t_copy, t_search, t_keep, t_call, t_free);
    return 0;
}

// BAM input (`-i config`): main()'s loop with get_SR_Reads per window (src/pindel.cpp:1816-1982,
// src/reader.cpp:1427-1470).  Windows run from 0 in steps of the bin size while the start does not pass the
// end of the chromosome (LoopingSearchWindow::finished without the Pindel-text shortcut); every window reads
// its candidates from every BAM of the configuration through pg_bam.hpp, so only one window's reads are in
// memory at a time (a coordinate-sorted BAM delivers them bin by bin).
struct BamSource {
    std::string path, tag;
    int insert_size = 0;
};

// The close end of one window of the BAM path, straight on the ingested structure-of-arrays batch (no SplitRead per
// candidate): what comes back per contiguous part of the batch (one part per device).
struct ClosePart {
    size_t first = 0, n = 0;              // reads [first, first + n) of the batch
    const uint8_t *rc_flag = nullptr;     // per read of the part
    const uint64_t *close_off = nullptr;  // n + 1 offsets into close_runs
    const pg_run *close_runs = nullptr;
};
struct CloseView {
    std::vector<ClosePart> parts;
    std::function<void()> release;        // frees what the pointers point into
};

// bd != null && search_rp: before the reads of a window are taken, its discordant read pairs become BreakDancer-like
// events (get_RP_Reads_Discovery + BDData::UpdateBD, src/pindel.cpp:1838-1848; -R, default on) next to the events of
// a -b file; the search step then looks their windows up per read (loadRegion / getCorrespondingSearchWindowCluster).
//
//   close_soa(chrom, chr_id, batch, view)   ReadBuffer::flush on the window's candidates as ingested (SoA): the close
//                                           ends as run lists + rc flags; only the reads that have one become SplitReads
//                                           (UnmatchedSeq reverse-complemented where GetCloseEnd did, UP_Close filled)
//   far_search(chrom, chr_id, kept, ws, we) SearchFarEnds on those reads; [ws, we) = the window (for the hints' loadRegion)
//
// The windows are a three-stage pipeline on the host: while window k is searched and classified, a second thread
// already reads window k + 1 from the BAMs (read-pair discovery + ingest, the stage that dominates a BAM-fed run).
template <class CloseSoa, class FarSearch>
int run_bam_pipeline(const std::vector<Chromosome> &genome, const std::vector<unsigned> &fai,
                     const std::vector<BamSource> &bams, const BamIngestSettings &ingest, const Settings &S,
                     const std::string &prefix, CloseSoa close_soa, FarSearch far_search, std::string &err, size_t *n_reads_total = nullptr,
                     BDHints *bd = nullptr, bool search_rp = false, size_t *n_rp_events = nullptr)
{
    Caller caller(S, &genome, prefix, true);
    std::ofstream rp_out;
    if (bd && search_rp) rp_out.open((prefix + "_RP").c_str(), std::ios::trunc);
    const unsigned WINDOW = (unsigned)(S.window_mbp * 1000000);
    if (WINDOW == 0) {
        err = "window size (-w) must be at least 0.000001 Mbp";
        return -1;
    }
    std::vector<BamFile> files(bams.size());
    for (size_t k = 0; k < bams.size(); k++)
        if (!files[k].open(bams[k].path, err)) return -1;
    // PGH_TIMING=1: wall-clock seconds per stage of this loop on stderr (diagnostics)
    const bool timing = getenv("PGH_TIMING") != nullptr;
    double t_wait = 0, t_close = 0, t_keep = 0, t_far = 0, t_cov = 0, t_call = 0, t_free = 0;
    double t_rp = 0, t_ingest = 0;                         // (on the reader thread)
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();

    struct Win { size_t c; unsigned ws, we; };
    std::vector<Win> wins;
    for (size_t c = 0; c < genome.size(); c++) {
        const unsigned biol = (unsigned)(genome[c].seq.size() - 2 * S.spacer);
        const unsigned bed_end = fai[c] ? fai[c] : biol;
        const unsigned global_end = std::min(biol, bed_end + 10000u);
        for (unsigned ws = 0; !(ws > global_end); ws += WINDOW) wins.push_back({ c, ws, std::min(ws + WINDOW, global_end) });
    }
    // what the reader thread hands over for one window
    struct WinData {
        IngestedReads in;
        std::vector<std::pair<BDHints::RpSide, BDHints::RpSide>> sides;   // read-pair events of the window
        size_t n_events = 0;
        std::string error;
    };
    // (only this function touches `files` and `rp_out`, one window at a time, in window order)
    auto read_win = [&](size_t w) {
        std::unique_ptr<WinData> d(new WinData());
        const Win &win = wins[w];
        const Chromosome &chrom = genome[win.c];
        double t0 = now();
        if (bd && search_rp) {
            std::vector<RpRead> rp;
            for (size_t k = 0; k < bams.size(); k++)
                if (!rp_discover(files[k], chrom.name, win.ws, win.we, bams[k].insert_size, bams[k].tag, ingest.min_anchor_quality, rp)) {
                    d->error = bams[k].path + ": BAM read failed";
                    return d;
                }
            const std::vector<RpEvent> ev = rp_events(rp, S.spacer, &rp_out);
            for (const RpEvent &e : ev) {
                BDHints::RpSide a = { e.chr1, e.pos1, e.pos1b }, b = { e.chr2, e.pos2, e.pos2b };
                d->sides.push_back(std::make_pair(a, b));
            }
            d->n_events = ev.size();
        }
        t_rp += now() - t0; t0 = now();
        d->in.clear();
        BamIngest ing(ingest);
        for (size_t k = 0; k < bams.size(); k++)
            if (!ing.read_window(files[k], chrom.name, (int)win.c, chrom.seq.size(), win.ws, win.we, bams[k].insert_size, bams[k].tag, d->in)) {
                d->error = bams[k].path + ": " + ing.error;
                return d;
            }
        t_ingest += now() - t0;
        return d;
    };
    std::future<std::unique_ptr<WinData>> ahead;
    std::future<void> trash;
    if (!wins.empty()) ahead = std::async(std::launch::async, read_win, (size_t)0);
    int status = 0;
    for (size_t w = 0; w < wins.size(); w++) {
        double t0 = now();
        std::unique_ptr<WinData> d = ahead.get();
        if (w + 1 < wins.size() && d->error.empty()) ahead = std::async(std::launch::async, read_win, w + 1);
        t_wait += now() - t0; t0 = now();
        if (!d->error.empty()) {
            err = d->error;
            return -1;
        }
        const Win &win = wins[w];
        const size_t c = win.c;
        const Chromosome &chrom = genome[c];
        const unsigned biol = (unsigned)(chrom.seq.size() - 2 * S.spacer);
        const unsigned bed_start = 1, bed_end = fai[c] ? fai[c] : biol;
        if (bd && search_rp) {
            bd->update_with_rp(d->sides);
            if (n_rp_events) *n_rp_events += d->n_events;
        }
        IngestedReads &in = d->in;
        if (n_reads_total) *n_reads_total += in.size();
        if (in.size() == 0) continue;
        CloseView view;
        int rc = close_soa(chrom, (int)c, in.batch, view);
        if (rc) {
            err = "search step failed";
            status = rc;
            break;
        }
        t_close += now() - t0; t0 = now();
        // the reads that kept a close end (ReadBuffer::flush, src/read_buffer.cpp:55-64), in input order
        std::vector<uint32_t> kept_idx;
        std::vector<const ClosePart *> kept_part;
        for (const ClosePart &p : view.parts)
            for (size_t i = 0; i < p.n; i++)
                if (p.close_off[i + 1] > p.close_off[i]) {
                    kept_idx.push_back((uint32_t)(p.first + i));
                    kept_part.push_back(&p);
                }
        std::vector<SplitRead> kept(kept_idx.size());
        pg_adapter::parallel_ranges(kept.size(), [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; k++) {
                const size_t i = kept_idx[k];
                const ClosePart &p = *kept_part[k];
                const size_t j = i - p.first;
                SplitRead &r = kept[k];
                r.Name = in.names[i];
                r.UnmatchedSeq.assign((const char *)in.batch.seq.data() + in.batch.off[i], (size_t)(in.batch.off[i + 1] - in.batch.off[i]));
                pg_adapter::apply_rc_flag(r, p.rc_flag[j]);                        // setUnmatchedSeq(RC), pindel.cpp:2545 (once or twice)
                r.ReadLength = (short)r.UnmatchedSeq.size();
                r.MatchedD = (char)in.batch.strand[i];
                r.MatchedRelPos = (unsigned)in.batch.pos[i];
                r.MS = in.ms[i];
                r.InsertSize = in.batch.isz[i];
                r.Tag = in.tags[i];
                r.FragName = chrom.name;
                r.chr_id = (int)c;
                r.MAX_SNP_ERROR = (short)S.max_mismatch[std::min<int>(r.ReadLength, 499)];
                pg_adapter::fill_points(r.UP_Close, p.close_runs, p.close_off[j], p.close_off[j + 1], [](const pg_point &q) {
                    UniquePoint u;
                    u.chr = q.chr_id;
                    u.LengthStr = q.length;
                    u.AbsLoc = q.abs_loc;
                    u.Direction = q.direction;
                    u.Strand = q.strand;
                    u.Mismatches = q.mismatches;
                    return u;
                });
            }
        });
        if (view.release) view.release();
        caller.note_close_mapped_all(kept);
        t_keep += now() - t0; t0 = now();
        if (!kept.empty() && (rc = far_search(chrom, (int)c, kept, win.ws, win.we))) {
            err = "far-end search step failed";
            status = rc;
            break;
        }
        t_far += now() - t0; t0 = now();
        {   // UpdateRefReadCoverage, after the close ends (sample names) and before the classifiers
            std::vector<Caller::RefReadSpan> spans(in.ref_reads.size());
            for (size_t i = 0; i < spans.size(); i++) {
                spans[i].pos = in.ref_reads[i].pos;
                spans[i].length = in.ref_reads[i].length;
                spans[i].tag = in.ref_reads[i].tag;
            }
            caller.update_ref_coverage(spans, in.ref_tags, win.ws, win.we);
        }
        t_cov += now() - t0; t0 = now();
        if (!kept.empty()) caller.process_window(chrom, kept, win.ws, win.we, bed_start, bed_end);
        t_call += now() - t0; t0 = now();
        // the window's reads are freed behind the next window's work (one disposal in flight)
        if (trash.valid()) trash.wait();
        {
            std::shared_ptr<std::vector<SplitRead>> dead_reads(new std::vector<SplitRead>(std::move(kept)));
            std::shared_ptr<WinData> dead_win(d.release());
            trash = std::async(std::launch::async, [dead_reads, dead_win]() mutable {
                release_reads(*dead_reads);
                dead_reads.reset();
                dead_win.reset();
            });
        }
        t_free += now() - t0;
    }
    if (ahead.valid()) ahead.wait();                      // (an early exit must not leave the reader running on dead objects)
    if (trash.valid()) trash.wait();
    if (timing)
        fprintf(stderr, "pgh timing: BAM pipeline %.3f s wall: waiting for the reader %.3f s, close end %.3f s, keep + SplitReads %.3f s, "
                        "far end %.3f s, reference coverage %.3f s, classify + report %.3f s, free %.3f s | reader thread: read-pair "
                        "discovery %.3f s, ingest %.3f s (inflate + decode on threads %.3f, selection rules on threads %.3f, layout %.3f)\n",
                now() - t_begin, t_wait, t_close, t_keep, t_far, t_cov, t_call, t_free, t_rp, t_ingest,
                ingest_timing().inflate_decode, ingest_timing().select, ingest_timing().layout);
    return status;
}

}  // namespace pgh
#endif

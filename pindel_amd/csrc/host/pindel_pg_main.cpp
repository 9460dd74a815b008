// pindel_pg -- command line with Pindel's flags for the path this repository implements:
//   pindel_pg -f ref.fa -p reads.txt -o prefix [-x 2 -a 1 -m 3 -u 0.02 -e 0.01 -E 0.95 -H 8
//                                               -M 1 -B 100 -d 30 -v 50 -w 5 -G device]
// FASTA + Pindel-text reads -> close/far-end search on the MI355X (C ABI, libpindel_pg.so)
// -> SV classification and <prefix>_D/_SI/_TD/_INV reports (host code in this directory).
// Flags and their defaults follow src/fn_parameters.cpp; BAM input (-i) needs htslib and is
// not built here.
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fstream>
#include <algorithm>
#include <functional>
#include <iterator>
#include <string>
#include <thread>
#include <vector>

#include "pg_adapter.hpp"
#include "pg_bdhints.hpp"
#include "pg_host.hpp"
#include "pg_pipeline.hpp"
#include "pindel_pg.h"

using namespace pgh;

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const double t_start = now_s();
    double t_search = 0.0;
    std::string fasta, reads_path, prefix, bd_path, bam_config;
    unsigned min_anchor_quality = 0;
    int ref_read_nm = 2;                    // -n / --NM (isRefRead)
    bool search_rp = true;                 // -R: discordant read pairs as window hints (BAM input only; default true)
    bool use_bd = false;
    size_t flush_reads = 0;                // --flush-reads: reads per close-end call (0 = a whole bin)
    pg_params prm;
    pg_default_params(&prm);
    Settings S;
    // Flags as src/fn_parameters.cpp defines them: value flags need an argument that does not start with
    // '-'; unary switches take an optional true/false word (readParameters, fn_parameters.cpp:366-406).
    // Switches of report families this program does not write are accepted and ignored, like the reference
    // accepts them; anything else is an error, and so is a value that is not a number.
    struct Flag { const char *sh, *lg; char kind; };      // kind: i int, f float, s string, u unary
    static const Flag flags[] = {
        { "-f", "--fasta", 's' }, { "-p", "--pindel-file", 's' }, { "-i", "--config-file", 's' }, { "-o", "--output-prefix", 's' },
        { "-x", "--max_range_index", 'i' }, { "-a", "--additional_mismatch", 'i' },
        { "-m", "--min_perfect_match_around_BP", 'i' }, { "-u", "--maximum_allowed_mismatch_rate", 'f' },
        { "-e", "--sequencing_error_rate", 'f' }, { "-E", "--sensitivity", 'f' }, { "-H", "--min_close", 'i' },
        { "-M", "--minimum_support_for_event", 'i' }, { "-B", "--balance_cutoff", 'i' },
        { "-d", "--min_num_matched_bases", 'i' }, { "-v", "--min_inversion_size", 'i' },
        { "-w", "--window_size", 'f' }, { "-T", "--number_of_threads", 'i' }, { "-b", "--breakdancer", 's' },
        { "-G", "--gpus", 's' }, { "", "--bd-hints", 's' }, { "", "--flush-reads", 'i' }, { "-c", "--chromosome", 's' },
        { "-n", "--NM", 'i' }, { "", "--min_NT_size", 'i' }, { "-A", "--anchor_quality", 'i' }, { "-L", "--logfilename", 's' },
        { "-r", "--report_inversions", 'u' }, { "-t", "--report_duplications", 'u' },
        { "-l", "--report_long_insertions", 'u' }, { "-k", "--report_breakpoints", 'u' },
        { "-s", "--report_close_mapped_reads", 'u' }, { "-S", "--report_only_close_mapped_reads", 'u' },
        { "-I", "--report_interchromosomal_events", 'u' }, { "-C", "--IndelCorrection", 'u' },
        { "-N", "--NormalSamples", 'u' }, { "-R", "--RP", 'u' }, { "-q", "--detect_DD", 'u' },
    };
    std::string gpu_list;
    for (int i = 1; i < argc; i++) {
        const std::string f = argv[i];
        const Flag *fl = nullptr;
        for (const Flag &x : flags)
            if ((x.sh[0] && f == x.sh) || f == x.lg) fl = &x;
        if (!fl) {
            fprintf(stderr, "pindel_pg: unknown argument: %s\n", f.c_str());
            return 2;
        }
        const std::string key = fl->sh[0] ? fl->sh : fl->lg;
        if (fl->kind == 'u') {
            bool on = true;
            if (i + 1 < argc && argv[i + 1][0] != '-') {
                const char c0 = (char)tolower((unsigned char)argv[i + 1][0]);
                on = !(c0 == 'f' || c0 == '0');
                i++;
            }
            if (key == "-R") search_rp = on;
            else if (key == "-r") S.Analyze_INV = on;
            else if (key == "-t") S.Analyze_TD = on;
            // the other switches select reports (LI, BP, CloseEndMapped, INT ...) outside this program's scope
            continue;
        }
        if (i + 1 >= argc) {
            fprintf(stderr, "pindel_pg: argument of %s lacking.\n", f.c_str());
            return 2;
        }
        const char *v = argv[++i];
        if (v[0] == '-' && fl->kind != 's') {
            fprintf(stderr, "pindel_pg: argument of %s seems erroneous.\n", f.c_str());
            return 2;
        }
        long iv = 0;
        double fv = 0.0;
        if (fl->kind == 'i' || fl->kind == 'f') {
            char *endp = nullptr;
            if (fl->kind == 'i') iv = strtol(v, &endp, 10);
            else fv = strtod(v, &endp);
            if (endp == v || *endp != 0) {
                fprintf(stderr, "pindel_pg: argument of %s is not a number: %s\n", f.c_str(), v);
                return 2;
            }
        }
        if (key == "-f") fasta = v;
        else if (key == "-p") reads_path = v;
        else if (key == "-i") bam_config = v;
        else if (key == "-A") min_anchor_quality = (unsigned)iv;
        else if (key == "-n") ref_read_nm = (int)iv;       // "-n" is registered twice in the reference; --NM comes first
        else if (key == "-o") prefix = v;
        else if (key == "-x") prm.max_range_index = (int)iv;
        else if (key == "-a") prm.additional_mismatch = (int)iv;
        else if (key == "-m") prm.min_perfect_match_around_bp = (int)iv;
        else if (key == "-u") prm.max_allowed_mismatch_rate = fv;
        else if (key == "-e") prm.seq_error_rate = S.Seq_Error_Rate = fv;
        else if (key == "-E") prm.sensitivity = fv;
        else if (key == "-H") prm.min_close = (int)iv;
        else if (key == "-M") S.NumRead2ReportCutOff = (unsigned)iv;
        else if (key == "-B") S.BalanceCutoff = (unsigned)iv;
        else if (key == "-d") S.Min_Num_Matched_Bases = (int)iv;
        else if (key == "-v") S.MIN_IndelSize_Inversion = (int)iv;
        else if (key == "-w") {
            if ((unsigned)(fv * 1000000) == 0) {
                fprintf(stderr, "pindel_pg: -w must be at least 0.000001 (Mbp)\n");
                return 2;
            }
            S.window_mbp = fv;
        }
        else if (key == "--flush-reads") flush_reads = iv > 0 ? (size_t)iv : 0;
        else if (key == "-G") gpu_list = v;
        else if (key == "-b") bd_path = v;                                     // --breakdancer
        else if (key == "--bd-hints") use_bd = std::string(v) == "on";         // see below: off = what 0.2.5b9 does
        else if (key == "-c") {
            if (std::string(v) != "ALL") {
                fprintf(stderr, "pindel_pg: only -c ALL is supported\n");
                return 2;
            }
        }
        else if (key == "-T") {
            // host threads of the classifiers / reporters (the search itself runs on the GPU); PGH_THREADS wins
            if (iv >= 1) setenv("PGH_THREADS", std::to_string(iv).c_str(), 0);
        }
        // -L: accepted, no effect on this path
    }
    // -G 0,1,...: the reads of every bin are sharded over these devices in contiguous ranges (reads are
    // independent: the loop of SearchFarEnds / ReadBuffer::flush, src/pindel.cpp:1115-1138), reference
    // replicated per device, results concatenated in order -- identical reports for any device count
    std::vector<int> devices;
    if (gpu_list.empty()) devices.push_back(prm.device);
    else {
        const char *q = gpu_list.c_str();
        while (*q) {
            char *endp = nullptr;
            const long d = strtol(q, &endp, 10);
            if (endp == q || d < 0 || (*endp != 0 && *endp != ',')) {
                fprintf(stderr, "pindel_pg: bad device list %s\n", gpu_list.c_str());
                return 2;
            }
            devices.push_back((int)d);
            q = *endp ? endp + 1 : endp;
        }
        if (devices.empty()) {
            fprintf(stderr, "pindel_pg: bad device list %s\n", gpu_list.c_str());
            return 2;
        }
    }
    if (fasta.empty() || (reads_path.empty() == bam_config.empty()) || prefix.empty()) {
        fprintf(stderr, "usage: pindel_pg -f ref.fa (-p reads.txt | -i bam_config.txt) -o prefix [options]\n");
        return 2;
    }
    std::string err;
    std::vector<Chromosome> genome;
    if (load_fasta(fasta, genome, prm.spacer, err)) {
        fprintf(stderr, "pindel_pg: %s\n", err.c_str());
        return 1;
    }
    std::vector<SplitRead> all;
    if (!reads_path.empty() && load_pindel_text(reads_path, genome, all, err)) {
        fprintf(stderr, "pindel_pg: %s\n", err.c_str());
        return 1;
    }
    // -i: one line per BAM: file, insert size, sample tag (readBamConfigFile, src/pindel.cpp)
    std::vector<BamSource> bams;
    if (!bam_config.empty()) {
        std::ifstream cf(bam_config.c_str());
        if (!cf) {
            fprintf(stderr, "pindel_pg: cannot open %s\n", bam_config.c_str());
            return 1;
        }
        BamSource b;
        while (cf >> b.path >> b.insert_size >> b.tag) {
            if (b.path[0] != '/') {                       // relative to the configuration file
                const size_t sl = bam_config.rfind('/');
                if (sl != std::string::npos) b.path = bam_config.substr(0, sl + 1) + b.path;
            }
            bams.push_back(b);
        }
        if (bams.empty()) {
            fprintf(stderr, "pindel_pg: no BAM files in %s\n", bam_config.c_str());
            return 1;
        }
    }
    std::vector<pg_ctx *> ctxs;
    int rc = 0;
    for (int d : devices) {
        pg_params p = prm;
        p.device = d;
        pg_ctx *c = nullptr;
        rc = pg_create(&p, &c);
        if (rc) {
            fprintf(stderr, "pindel_pg: pg_create failed (%d) on device %d: no usable MI355X / HIP device\n", rc, d);
            return 1;
        }
        ctxs.push_back(c);
    }
    pg_ctx *ctx = ctxs[0];
    {
        std::vector<const char *> names;
        std::vector<const uint8_t *> seqs;
        std::vector<uint64_t> lens;
        for (const Chromosome &c : genome) {
            names.push_back(c.name.c_str());
            seqs.push_back((const uint8_t *)c.seq.data());
            lens.push_back(c.seq.size());
        }
        for (pg_ctx *c : ctxs) {
            rc = pg_load_reference(c, (int32_t)genome.size(), names.data(), seqs.data(), lens.data());
            if (rc) {
                fprintf(stderr, "pindel_pg: pg_load_reference: %s\n", pg_last_error(c));
                return 1;
            }
        }
    }
    S.spacer = prm.spacer;
    S.log_counts = true;
    pg_get_max_mismatch(ctx, S.max_mismatch);
    std::vector<unsigned> fai = read_fai(fasta, genome);
    const double t_loaded = now_s();
    auto chr_of = [](const SplitRead &r) { return r.chr_id; };
    auto make_point = [](const pg_point &p) {
        UniquePoint u;
        u.chr = p.chr_id;
        u.LengthStr = p.length;
        u.AbsLoc = p.abs_loc;
        u.Direction = p.direction;
        u.Strand = p.strand;
        u.Mismatches = p.mismatches;
        return u;
    };
    // -b: Pindel 0.2.5b9 loads the file but, for Pindel-text input, never hands its events to the far-end
    // search.  That is the default here too.  "--bd-hints on" searches the windows of
    // the file's events before the ranges, the way the BAM path of the reference does (pg_bdhints.hpp).
    BDHints bd;
    std::vector<std::string> chr_names;
    for (const Chromosome &c : genome) chr_names.push_back(c.name);
    if (!bd_path.empty()) {
        std::string note;
        const int brc = bd.load_file(bd_path, prm.spacer, note);
        if (brc < 0) {
            fprintf(stderr, "pindel_pg: %s\n", note.c_str());
            return 1;
        }
        if (brc > 0) printf("pindel_pg: %s\n", note.c_str());
        printf("pindel_pg: BD events: %zu%s\n", bd.n_events(), use_bd ? "" : " (not used for Pindel-text input; --bd-hints on to use them)");
    }
    size_t n_close = 0, n_far = 0;
    // fn(ctx, part) on contiguous shards of `reads`, one host thread and one ctx per device; the parts are moved out
    // and back, so the order is kept (reads are independent: identical results for any device count)
    auto on_devices = [&](std::vector<SplitRead> &reads, const std::function<int(pg_ctx *, std::vector<SplitRead> &)> &fn) {
        const size_t nd = ctxs.size(), n = reads.size();
        if (nd == 1 || n < 2 * nd) return fn(ctxs[0], reads);
        std::vector<std::vector<SplitRead>> parts(nd);
        std::vector<int> rcs(nd, 0);
        for (size_t d = 0; d < nd; d++) {
            const size_t lo = n * d / nd, hi = n * (d + 1) / nd;
            parts[d].assign(std::make_move_iterator(reads.begin() + lo), std::make_move_iterator(reads.begin() + hi));
        }
        std::vector<std::thread> th;
        for (size_t d = 0; d < nd; d++) th.emplace_back([&, d]() { rcs[d] = fn(ctxs[d], parts[d]); });
        for (std::thread &x : th) x.join();
        int r = 0;
        for (size_t d = 0; d < nd; d++) {
            if (rcs[d]) r = rcs[d];
            std::move(parts[d].begin(), parts[d].end(), reads.begin() + n * d / nd);
        }
        return r;
    };
    // Seam 1 (ReadBuffer::flush, src/read_buffer.cpp:36-101): the close end of ALL reads of the bin, `flush_reads` at a
    // time like the reference's 50 000-read buffer (src/reader.cpp:55; 0 = the whole bin in one call -- the results do
    // not depend on it).  The pipeline then keeps the reads with a close end, as flush() does (:55-64).
    auto close_search = [&](const Chromosome &, int, std::vector<SplitRead> &reads, const std::vector<uint32_t> &) {
        const double t0 = now_s();
        int r = on_devices(reads, [&](pg_ctx *c, std::vector<SplitRead> &part) {
            const size_t step = flush_reads ? flush_reads : std::max<size_t>(part.size(), 1);
            if (step >= part.size()) {
                pg_result *res = nullptr;
                const int rr = pg_adapter::CloseEndBatch(c, part, chr_of, make_point, &res);
                pg_result_free(res);
                return rr;
            }
            for (size_t lo = 0; lo < part.size(); lo += step) {
                const size_t hi = std::min(part.size(), lo + step);
                std::vector<SplitRead> buf(std::make_move_iterator(part.begin() + lo), std::make_move_iterator(part.begin() + hi));
                pg_result *res = nullptr;
                const int rr = pg_adapter::CloseEndBatch(c, buf, chr_of, make_point, &res);
                pg_result_free(res);
                std::move(buf.begin(), buf.end(), part.begin() + lo);
                if (rr) return rr;
            }
            return 0;
        });
        t_search += now_s() - t0;
        return r;
    };
    // Seam 2 (SearchFarEnds, src/pindel.cpp:1115-1138, called at :1888 on state.Reads_SR): the far end of the reads that
    // kept a close end -- the filtered union of the flushes -- through pg_far_end_batch_from_close.
    auto far_search = [&](const Chromosome &, int chr_id, std::vector<SplitRead> &kept, unsigned ws, unsigned we) {
        const double t0 = now_s();
        if (use_bd && bd.n_events() && !kept.empty()) {
            // the window main() is working on, as it hands it to g_bdData.loadRegion (currentWindow_cs, pindel.cpp:1828, 1853):
            // [ws, we) + spacer, we clipped to the end of the scanned region (LoopingSearchWindow::updateEndPositions).  NOT
            // derived from the reads: a BAM window also holds reads whose anchor lies before ws (reader.cpp has no position
            // filter on that path), and the bin of min(MatchedRelPos) would then be the previous window.
            std::string berr;
            if (!bd.load_region(chr_names, chr_id, ws + prm.spacer, we + prm.spacer, berr)) {
                fprintf(stderr, "pindel_pg: %s\n", berr.c_str());
                return (int)PG_E_INVALID;
            }
        }
        int r = on_devices(kept, [&](pg_ctx *c, std::vector<SplitRead> &part) {
            std::vector<uint64_t> hoff;
            std::vector<pg_window> hwin;
            pg_windows hints = { nullptr, nullptr };
            if (use_bd && bd.n_events() && !part.empty()) {
                hoff.push_back(0);
                for (const SplitRead &x : part) {
                    for (const BDWindow &w : bd.cluster(x.UP_Close.back().AbsLoc)) {
                        pg_window pw = { w.chr_id, (int32_t)w.start, (int32_t)w.end };
                        hwin.push_back(pw);
                    }
                    hoff.push_back(hwin.size());
                }
                hints.offset = hoff.data();
                hints.windows = hwin.empty() ? nullptr : hwin.data();
            }
            return pg_adapter::SearchFarEnds(c, part, chr_of, make_point, hints.offset ? &hints : nullptr);
        });
        size_t bin_far = 0;
        for (const SplitRead &x : kept) bin_far += !x.UP_Far.empty();
        n_close += kept.size();
        n_far += bin_far;
        // ReportCloseAndFarEndCounts (src/pindel.cpp:1094-1113), over the reads that kept a close end
        printf("Total: %zu;\tClose_end_found %zu;\tFar_end_found %zu;\tUsed\t0.\n\nFor LI and BP: %zu\n\n", kept.size(), kept.size(),
               bin_far, kept.size() - bin_far);
        t_search += now_s() - t0;
        return r;
    };
    size_t n_bam_reads = 0, n_rp_events = 0;
    if (!bams.empty()) {
        BamIngestSettings ing;
        ing.min_anchor_quality = min_anchor_quality;
        ing.spacer = prm.spacer;
        ing.nm = ref_read_nm;
        ing.max_mismatch_rate = prm.max_allowed_mismatch_rate;
        // BAM input: with -R (default) the window hints are live -- the events of a -b file plus the read-pair events of
        // every window; without -R the reference never hands any event to the search (UpdateBD is not called)
        use_bd = search_rp;
        // seam 1 on the ingested structure-of-arrays batch: one pg_close_end_batch per device on a contiguous part
        auto close_soa = [&](const Chromosome &, int, const pg_adapter::Batch &batch, CloseView &view) {
            const double t0 = now_s();
            const size_t nd = ctxs.size(), n = batch.strand.size();
            const size_t np = (nd == 1 || n < 2 * nd) ? 1 : nd;
            std::vector<pg_result *> res(np, nullptr);
            std::vector<int> rcs(np, 0);
            auto part = [&](size_t d) {
                const size_t lo = n * d / np, hi = n * (d + 1) / np;
                pg_read_batch v = batch.view();
                v.n_reads = (uint32_t)(hi - lo);
                v.seq_off += lo;
                v.anchor_strand += lo;
                v.anchor_pos += lo;
                v.insert_size += lo;
                v.chr_id += lo;
                rcs[d] = pg_close_end_batch(ctxs[d], &v, &res[d]);
            };
            if (np == 1) part(0);
            else {
                std::vector<std::thread> th;
                for (size_t d = 0; d < np; d++) th.emplace_back(part, d);
                for (std::thread &x : th) x.join();
            }
            int r = 0;
            for (size_t d = 0; d < np; d++)
                if (rcs[d]) r = rcs[d];
            view.release = [res]() { for (pg_result *x : res) pg_result_free(x); };
            if (r) {
                view.release();
                view.release = nullptr;
                return r;
            }
            for (size_t d = 0; d < np; d++) {
                pg_result_view rv;
                pg_result_view_get(res[d], &rv);
                ClosePart p;
                p.first = n * d / np;
                p.n = rv.n_reads;
                p.rc_flag = rv.rc_flag;
                p.close_off = rv.close_off;
                p.close_runs = rv.close_runs;
                view.parts.push_back(p);
            }
            t_search += now_s() - t0;
            return 0;
        };
        rc = run_bam_pipeline(genome, fai, bams, ing, S, prefix, close_soa, far_search, err, &n_bam_reads, &bd, search_rp, &n_rp_events);
        if (search_rp) printf("pindel_pg: read-pair events added as window hints: %zu\n", n_rp_events);
    } else
        rc = run_pipeline(genome, fai, all, S, prefix, close_search, far_search, err);
    if (rc) fprintf(stderr, "pindel_pg: %s (%s)\n", err.c_str(), pg_last_error(ctx));
    else {
        printf("pindel_pg: %zu reads, close end %zu, far end %zu\n", bams.empty() ? all.size() : n_bam_reads, n_close, n_far);
        // the phases the reference's Timer reports (pindel.cpp:1990-1996), wall-clock seconds
        printf("pindel_pg: loading %.2f s, split-read search (GPU, incl. adapters) %.2f s, classification + reports %.2f s\n",
               t_loaded - t_start, t_search, now_s() - t_loaded - t_search);
    }
    for (pg_ctx *c : ctxs) pg_destroy(c);
    return rc ? 1 : 0;
}

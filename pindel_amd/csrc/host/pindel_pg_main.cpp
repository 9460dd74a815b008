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
#include <string>
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
    std::string fasta, reads_path, prefix, bd_path;
    bool use_bd = false;
    pg_params prm;
    pg_default_params(&prm);
    Settings S;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string f = argv[i];
        const char *v = argv[i + 1];
        if (f == "-f") fasta = v;
        else if (f == "-p") reads_path = v;
        else if (f == "-o") prefix = v;
        else if (f == "-x") prm.max_range_index = atoi(v);
        else if (f == "-a") prm.additional_mismatch = atoi(v);
        else if (f == "-m") prm.min_perfect_match_around_bp = atoi(v);
        else if (f == "-u") prm.max_allowed_mismatch_rate = atof(v);
        else if (f == "-e") prm.seq_error_rate = S.Seq_Error_Rate = atof(v);
        else if (f == "-E") prm.sensitivity = atof(v);
        else if (f == "-H") prm.min_close = atoi(v);
        else if (f == "-M") S.NumRead2ReportCutOff = (unsigned)atoi(v);
        else if (f == "-B") S.BalanceCutoff = (unsigned)atoi(v);
        else if (f == "-d") S.Min_Num_Matched_Bases = atoi(v);
        else if (f == "-v") S.MIN_IndelSize_Inversion = atoi(v);
        else if (f == "-w") S.window_mbp = atof(v);
        else if (f == "-G") prm.device = atoi(v);
        else if (f == "-T") { /* thread count: the search runs on the GPU */ }
        else if (f == "-b") bd_path = v;                                     // --breakdancer
        else if (f == "--bd-hints") use_bd = std::string(v) == "on";         // see below: off = what 0.2.5b9 does
        else {
            fprintf(stderr, "pindel_pg: unknown flag %s\n", f.c_str());
            return 2;
        }
    }
    if (fasta.empty() || reads_path.empty() || prefix.empty()) {
        fprintf(stderr, "usage: pindel_pg -f ref.fa -p reads.txt -o prefix [options]\n");
        return 2;
    }
    std::string err;
    std::vector<Chromosome> genome;
    if (load_fasta(fasta, genome, prm.spacer, err)) {
        fprintf(stderr, "pindel_pg: %s\n", err.c_str());
        return 1;
    }
    std::vector<SplitRead> all;
    if (load_pindel_text(reads_path, genome, all, err)) {
        fprintf(stderr, "pindel_pg: %s\n", err.c_str());
        return 1;
    }
    pg_ctx *ctx = nullptr;
    int rc = pg_create(&prm, &ctx);
    if (rc) {
        fprintf(stderr, "pindel_pg: pg_create failed (%d): no usable MI355X / HIP device\n", rc);
        return 1;
    }
    {
        std::vector<const char *> names;
        std::vector<const uint8_t *> seqs;
        std::vector<uint64_t> lens;
        for (const Chromosome &c : genome) {
            names.push_back(c.name.c_str());
            seqs.push_back((const uint8_t *)c.seq.data());
            lens.push_back(c.seq.size());
        }
        rc = pg_load_reference(ctx, (int32_t)genome.size(), names.data(), seqs.data(), lens.data());
        if (rc) {
            fprintf(stderr, "pindel_pg: pg_load_reference: %s\n", pg_last_error(ctx));
            return 1;
        }
    }
    S.spacer = prm.spacer;
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
    auto search = [&](const Chromosome &, int, std::vector<SplitRead> &reads, const std::vector<uint32_t> &) {
        const double t0 = now_s();
        pg_result *res = nullptr;
        int r = pg_adapter::CloseEndBatch(ctx, reads, chr_of, make_point, &res);      // ReadBuffer::flush
        if (r) return r;
        std::vector<uint64_t> hoff;
        std::vector<pg_window> hwin;
        pg_windows hints = { nullptr, nullptr };
        if (use_bd && bd.n_events() && !reads.empty()) {
            // the bin of these reads, as main() hands it to g_bdData.loadRegion (pindel.cpp:1828, 1853)
            unsigned lo = reads[0].MatchedRelPos, hi = reads[0].MatchedRelPos;
            for (const SplitRead &x : reads) {
                lo = std::min(lo, x.MatchedRelPos);
                hi = std::max(hi, x.MatchedRelPos);
            }
            const unsigned W = (unsigned)(S.window_mbp * 1000000);
            const unsigned ws = lo / W * W, we = ws + W;
            (void)hi;
            std::string berr;
            if (!bd.load_region(chr_names, reads[0].chr_id, ws + prm.spacer, we + prm.spacer, berr)) {
                fprintf(stderr, "pindel_pg: %s\n", berr.c_str());
                pg_result_free(res);
                return (int)PG_E_INVALID;
            }
            hoff.push_back(0);
            for (const SplitRead &x : reads) {
                if (!x.UP_Close.empty())
                    for (const BDWindow &w : bd.cluster(x.UP_Close.back().AbsLoc)) {
                        pg_window pw = { w.chr_id, (int32_t)w.start, (int32_t)w.end };
                        hwin.push_back(pw);
                    }
                hoff.push_back(hwin.size());
            }
            hints.offset = hoff.data();
            hints.windows = hwin.empty() ? nullptr : hwin.data();
        }
        r = pg_adapter::SearchFarEnds(ctx, reads, chr_of, make_point, res, hints.offset ? &hints : nullptr);   // SearchFarEnds
        pg_result_free(res);
        for (const SplitRead &x : reads) {
            n_close += !x.UP_Close.empty();
            n_far += !x.UP_Far.empty();
        }
        t_search += now_s() - t0;
        return r;
    };
    rc = run_pipeline(genome, fai, all, S, prefix, search, err);
    if (rc) fprintf(stderr, "pindel_pg: %s (%s)\n", err.c_str(), pg_last_error(ctx));
    else {
        printf("pindel_pg: %zu reads, close end %zu, far end %zu\n", all.size(), n_close, n_far);
        // the phases the reference's Timer reports (pindel.cpp:1990-1996), wall-clock seconds
        printf("pindel_pg: loading %.2f s, split-read search (GPU, incl. adapters) %.2f s, classification + reports %.2f s\n",
               t_loaded - t_start, t_search, now_s() - t_loaded - t_search);
    }
    pg_destroy(ctx);
    return rc ? 1 : 0;
}

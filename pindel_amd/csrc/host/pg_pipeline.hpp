// pg_pipeline.hpp -- main()'s chromosome / 5-Mbp-bin loop (src/pindel.cpp:1778-1989) around a
// pluggable search step.  Shared by the pindel_pg command line (search = GPU through the C ABI)
// and by pgh_call_from_points (search = attach externally computed points, used by the tests).
#ifndef PG_PIPELINE_HPP
#define PG_PIPELINE_HPP

#include <algorithm>
#include <fstream>
#include <string>
#include <utility>
#include <vector>

#include "pg_host.hpp"

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

// search(chrom, chr_id, reads, index_in_all): must fill UP_Close / UP_Far of every read (leaving
// UP_Close empty when there is no close end) and leave UnmatchedSeq as GetCloseEnd would.
template <class Search>
int run_pipeline(const std::vector<Chromosome> &genome, const std::vector<unsigned> &fai,
                 const std::vector<SplitRead> &all, const Settings &S, const std::string &prefix,
                 Search search, std::string &err)
{
    Caller caller(S, &genome, prefix, true);
    const unsigned WINDOW = (unsigned)(S.window_mbp * 1000000);
    for (size_t c = 0; c < genome.size(); c++) {
        const Chromosome &chrom = genome[c];
        const unsigned biol = (unsigned)(chrom.seq.size() - 2 * S.spacer);
        const unsigned bed_start = 1, bed_end = fai[c] ? fai[c] : biol;   // "-c ALL": one BED record per chromosome
        const unsigned global_end = std::min(biol, bed_end + 10000u);      // AROUND_REGION_BUFFER
        unsigned g_max_pos = 0;                                            // reset per BED region, pindel.cpp:1798
        unsigned ws = 0;
        do {
            const unsigned we = std::min(ws + WINDOW, global_end);
            std::vector<SplitRead> reads;
            std::vector<uint32_t> index;
            for (uint32_t i = 0; i < all.size(); i++) {
                const SplitRead &src = all[i];
                if (src.MatchedRelPos > g_max_pos) g_max_pos = src.MatchedRelPos;          // reader.cpp:224-226
                if (src.chr_id != (int)c || !(src.MatchedRelPos >= ws && src.MatchedRelPos < we)) continue;
                reads.push_back(src);
                if (reads.back().MatchedRelPos > biol) reads.back().MatchedRelPos = biol;   // reader.cpp:233-235
                reads.back().MAX_SNP_ERROR = (short)S.max_mismatch[std::min<int>(src.ReadLength, 499)];
                index.push_back(i);
            }
            if (!reads.empty()) {
                int rc = search(chrom, (int)c, reads, index);
                if (rc) {
                    err = "search step failed";
                    return rc;
                }
                std::vector<SplitRead> kept;                                // reader.cpp:258-291
                for (SplitRead &r : reads)
                    if (!r.UP_Close.empty()) {
                        caller.note_close_mapped(r);
                        kept.push_back(std::move(r));      // `reads` is not used after this loop
                    }
                if (!kept.empty()) caller.process_window(chrom, kept, ws, we, bed_start, bed_end);
            }
            ws += WINDOW;
            // LoopingSearchWindow::finished, pindel.cpp:464-471 (Pindel-text input shortcut)
        } while (!(ws >= g_max_pos || ws > global_end));
    }
    return 0;
}

}  // namespace pgh
#endif

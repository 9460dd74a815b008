// pg_adapter.hpp -- the two reference seams re-expressed on top of the C ABI.
//
//   CloseEndBatch(ctx, chrom_ids, reads)   replaces the body of ReadBuffer::flush()
//                                          (src/read_buffer.cpp:36-101) and of the two OpenMP
//                                          loops in ReadInRead (src/reader.cpp:248-255, 300-305)
//   SearchFarEnds(ctx, reads, hints)       replaces SearchFarEnds(chrSeq, reads, chr)
//                                          (src/pindel.cpp:1115-1138)
//
// Both are templates over the read type so that they compile unchanged against the
// reference's SPLIT_READ (fields Name/UnmatchedSeq/MatchedD/MatchedRelPos/InsertSize/FragName/
// UP_Close/UP_Far, src/pindel.h:265-383) and against pgh::SplitRead used by this repository's
// own command line.  They restore exactly the post-state the reference leaves behind:
// UnmatchedSeq reverse-complemented when GetCloseEnd did so, UP_Close after CleanUniquePoints,
// UP_Far untouched by any pruning.
#ifndef PG_ADAPTER_HPP
#define PG_ADAPTER_HPP

#include <cstdint>
#include <string>
#include <vector>

#include "pindel_pg.h"

namespace pg_adapter {

struct Batch {
    std::vector<uint8_t> seq, strand;
    std::vector<uint64_t> off;
    std::vector<int32_t> pos, chr;
    std::vector<int16_t> isz;
    pg_read_batch view() const
    {
        pg_read_batch b;
        b.n_reads = (uint32_t)strand.size();
        b.seq = seq.data();
        b.seq_off = off.data();
        b.anchor_strand = strand.data();
        b.anchor_pos = pos.data();
        b.insert_size = isz.data();
        b.chr_id = chr.data();
        return b;
    }
};

// chr_of(read) -> index of read.FragName in the loaded reference
template <class Read, class ChrOf>
Batch make_batch(const std::vector<Read> &reads, ChrOf chr_of)
{
    Batch b;
    b.off.push_back(0);
    for (const Read &r : reads) {
        b.seq.insert(b.seq.end(), r.UnmatchedSeq.begin(), r.UnmatchedSeq.end());
        b.off.push_back(b.seq.size());
        b.strand.push_back((uint8_t)r.MatchedD);
        b.pos.push_back((int32_t)r.MatchedRelPos);
        b.isz.push_back((int16_t)r.InsertSize);
        b.chr.push_back((int32_t)chr_of(r));
    }
    return b;
}

// make_point(pg_point) -> the read type's UniquePoint
template <class Points, class MakePoint>
void fill_points(Points &dst, const pg_run *runs, uint64_t lo, uint64_t hi, MakePoint make_point)
{
    dst.clear();
    for (uint64_t k = lo; k < hi; k++) {
        pg_point p[512];
        uint64_t n = pg_expand_runs(runs + k, 1, p);
        for (uint64_t i = 0; i < n; i++) dst.push_back(make_point(p[i]));
    }
}

inline std::string rc(const std::string &s)
{
    std::string o(s.size(), 0);
    for (size_t j = 0; j < s.size(); j++) {
        char c = s[s.size() - 1 - j];
        o[j] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c == 'N' ? 'N' : 0;
    }
    return o;
}

// Close end for a whole batch.  Returns the pg status; `result` keeps UP_Close summaries for the
// far-end call and must be released with pg_result_free once SearchFarEnds is done.
template <class Read, class ChrOf, class MakePoint>
int CloseEndBatch(pg_ctx *ctx, std::vector<Read> &reads, ChrOf chr_of, MakePoint make_point,
                  pg_result **result)
{
    Batch b = make_batch(reads, chr_of);
    pg_read_batch v = b.view();
    int rc_ = pg_close_end_batch(ctx, &v, result);
    if (rc_) return rc_;
    pg_result_view rv;
    pg_result_view_get(*result, &rv);
    for (size_t i = 0; i < reads.size(); i++) {
        if (rv.rc_flag[i]) reads[i].UnmatchedSeq = rc(reads[i].UnmatchedSeq);   // setUnmatchedSeq(RC), pindel.cpp:2545
        fill_points(reads[i].UP_Close, rv.close_runs, rv.close_off[i], rv.close_off[i + 1], make_point);
    }
    return PG_OK;
}

// Far end for the reads of CloseEndBatch (same order, same count).  `hints` may be null.
template <class Read, class ChrOf, class MakePoint>
int SearchFarEnds(pg_ctx *ctx, std::vector<Read> &reads, ChrOf chr_of, MakePoint make_point,
                  pg_result *close_result, const pg_windows *hints)
{
    // pg_far_end_batch wants the sequences in their ORIGINAL orientation (the rc flags travel in
    // close_result); undo CloseEndBatch's flip for the upload only
    std::vector<Read> &rs = reads;
    pg_result_view rv0;
    pg_result_view_get(close_result, &rv0);
    const std::vector<uint8_t> was_rc(rv0.rc_flag, rv0.rc_flag + rs.size());
    Batch b;
    b.off.push_back(0);
    for (size_t i = 0; i < rs.size(); i++) {
        const std::string s = was_rc[i] ? rc(rs[i].UnmatchedSeq) : rs[i].UnmatchedSeq;
        b.seq.insert(b.seq.end(), s.begin(), s.end());
        b.off.push_back(b.seq.size());
        b.strand.push_back((uint8_t)rs[i].MatchedD);
        b.pos.push_back((int32_t)rs[i].MatchedRelPos);
        b.isz.push_back((int16_t)rs[i].InsertSize);
        b.chr.push_back((int32_t)chr_of(rs[i]));
    }
    pg_read_batch v = b.view();
    int rc_ = pg_far_end_batch(ctx, &v, close_result, hints);
    if (rc_) return rc_;
    pg_result_view rv;
    pg_result_view_get(close_result, &rv);
    for (size_t i = 0; i < rs.size(); i++)
        fill_points(rs[i].UP_Far, rv.far_runs, rv.far_off[i], rv.far_off[i + 1], make_point);
    return PG_OK;
}

}  // namespace pg_adapter
#endif

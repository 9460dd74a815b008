// pg_adapter.hpp -- the two reference seams re-expressed on top of the C ABI.
//
//   CloseEndBatch(ctx, chrom_ids, reads)   replaces the body of ReadBuffer::flush()
//                                          (src/read_buffer.cpp:36-101) and of the two OpenMP
//                                          loops in ReadInRead (src/reader.cpp:248-255, 300-305)
//   SearchFarEnds(ctx, reads, hints)       replaces SearchFarEnds(chrSeq, reads, chr)
//                                          (src/pindel.cpp:1115-1138, called at :1888 on state.Reads_SR =
//                                          the reads ReadBuffer::flush kept, src/read_buffer.cpp:55-64):
//                                          needs nothing but the reads themselves -- UnmatchedSeq as the
//                                          close end left it and UP_Close.back()
//
// Both are templates over the read type.  What they use of it is the public interface of the reference's SPLIT_READ /
// SortedUniquePoints / UniquePoint and nothing more (src/pindel.h:137-197, 265-383): the fields UnmatchedSeq /
// MatchedD / MatchedRelPos / InsertSize / UP_Close / UP_Far, setUnmatchedSeq() when the read type has one
// (src/pindel.cpp:142-169, what GetCloseEnd calls at :2545), and of the point lists only push_back / size / empty /
// operator[] / clear -- `reserve` is used where a container offers it (std::vector-backed lists such as
// pgh::SplitRead's) and skipped where it does not (the reference's SortedUniquePoints).
// tests/test_cpu_suite.py::test_adapter_compiles_against_reference_shapes instantiates all three entry points against
// tests/ref_shapes.hpp (those declarations restated), tests/test_gpu_parity.py::test_adapter_on_reference_shapes runs
// them on the GPU against the oracle; pindel_pg_main.cpp instantiates them with pgh::SplitRead.
// They restore the post-state the reference leaves behind: UnmatchedSeq reverse-complemented when GetCloseEnd did so,
// UP_Close after CleanUniquePoints, UP_Far untouched by any pruning.
//
// The per-read work either side of the GPU call (gathering the bases into one buffer, expanding the
// run-length-encoded lists into UniquePoints) is spread over host threads in contiguous read ranges: it
// is the part of the seam that scales with the number of points (~90 per read), not the search.
#ifndef PG_ADAPTER_HPP
#define PG_ADAPTER_HPP

#include <algorithm>
#include <cstdint>
#include <cctype>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "pindel_pg.h"

namespace pg_adapter {

// fn(lo, hi) over [0, n) in contiguous ranges on up to `max_threads` threads
template <class Fn>
void parallel_ranges(size_t n, Fn fn, unsigned max_threads = 16)
{
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    const unsigned nt = (unsigned)std::min<size_t>(std::min(hw, max_threads), (n + 4095) / 4096);
    if (nt <= 1) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(fn, n * t / nt, n * (t + 1) / nt);
    for (std::thread &x : th) x.join();
}

// Convert2RC4N (src/pindel.cpp:966-970): A<->T, C<->G, N->N, every other character -> 0
inline char rc_char(char c)
{
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c == 'N' ? 'N' : 0;
}

inline void rc_in_place(std::string &s)
{
    const size_t n = s.size();
    for (size_t i = 0; i < n / 2; i++) {
        const char a = rc_char(s[i]), b = rc_char(s[n - 1 - i]);
        s[i] = b;
        s[n - 1 - i] = a;
    }
    if (n & 1) s[n / 2] = rc_char(s[n / 2]);
}

inline std::string rc(const std::string &s)
{
    std::string o(s);
    rc_in_place(o);
    return o;
}

// setUnmatchedSeq's stripping (src/pindel.cpp:142-157): trailing characters that are not alphanumeric go -- after a reverse
// complement those are the NULs Convert2RC4N put where the read BEGAN with characters outside ACGTN
inline void strip_trailing_non_alnum(std::string &s)
{
    size_t n = s.size();
    while (n > 0 && !std::isalnum((unsigned char)s[n - 1])) n--;
    s.resize(n);
}

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

// chr_of(read) -> index of read.FragName in the loaded reference.  un_rc (nullable): reads whose flag is set
// are written reverse-complemented (pg_far_end_batch wants the ORIGINAL orientation; the rc flags travel in
// the close result).
template <class Read, class ChrOf>
Batch make_batch(const std::vector<Read> &reads, ChrOf chr_of, const uint8_t *un_rc = nullptr)
{
    Batch b;
    const size_t n = reads.size();
    b.off.resize(n + 1);
    b.off[0] = 0;
    for (size_t i = 0; i < n; i++) b.off[i + 1] = b.off[i] + reads[i].UnmatchedSeq.size();
    b.seq.resize(b.off[n]);
    b.strand.resize(n);
    b.pos.resize(n);
    b.isz.resize(n);
    b.chr.resize(n);
    parallel_ranges(n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const Read &r = reads[i];
            const std::string &s = r.UnmatchedSeq;
            uint8_t *dst = b.seq.data() + b.off[i];
            if (un_rc && (un_rc[i] & 1))        // (2: two reverse complements -- the read is in its original orientation)
                for (size_t j = 0; j < s.size(); j++) dst[j] = (uint8_t)rc_char(s[s.size() - 1 - j]);
            else
                std::copy(s.begin(), s.end(), dst);
            b.strand[i] = (uint8_t)r.MatchedD;
            b.pos[i] = (int32_t)r.MatchedRelPos;
            b.isz[i] = (int16_t)r.InsertSize;
            b.chr[i] = (int32_t)chr_of(r);
        }
    });
    return b;
}

// optional members of the caller's types (detection idiom): Points::reserve, Read::setUnmatchedSeq
template <class T, class = void> struct has_reserve : std::false_type {};
template <class T> struct has_reserve<T, decltype((void)std::declval<T &>().reserve((size_t)1))> : std::true_type {};
template <class T> inline void reserve_if_possible(T &c, size_t n, std::true_type) { c.reserve(n); }
template <class T> inline void reserve_if_possible(T &, size_t, std::false_type) {}
template <class T, class = void> struct has_set_seq : std::false_type {};
template <class T>
struct has_set_seq<T, decltype((void)std::declval<T &>().setUnmatchedSeq(std::declval<const std::string &>()))> : std::true_type {};
// "read.setUnmatchedSeq(ReverseComplement(read.getUnmatchedSeq()))" (src/pindel.cpp:2545)
template <class Read> inline void flip_read(Read &r, std::true_type) { r.setUnmatchedSeq(rc(r.UnmatchedSeq)); }
template <class Read> inline void flip_read(Read &r, std::false_type) { rc_in_place(r.UnmatchedSeq); strip_trailing_non_alnum(r.UnmatchedSeq); }
// pg_result_view::rc_flag -> the read as GetCloseEnd left it: 1 = one "setUnmatchedSeq(ReverseComplement())", 2 = two (a read with
// characters outside ACGTN is then NOT the original again: they are NUL, those at either end are gone; for every other read two
// reverse complements are the identity and the flag stays 0)
template <class Read> inline void apply_rc_flag(Read &r, uint8_t flag)
{
    for (uint8_t k = 0; k < flag && k < 2; k++) flip_read(r, has_set_seq<Read>());
}

// make_point(pg_point) -> the read type's UniquePoint; the runs [lo, hi) are expanded in place
template <class Points, class MakePoint>
void fill_points(Points &dst, const pg_run *runs, uint64_t lo, uint64_t hi, MakePoint make_point)
{
    dst.clear();
    if (has_reserve<Points>::value) {
        size_t total = 0;
        for (uint64_t k = lo; k < hi; k++) total += (size_t)(runs[k].len_last - runs[k].len_first) + 1;
        reserve_if_possible(dst, total, has_reserve<Points>());
    }
    for (uint64_t k = lo; k < hi; k++) {
        const pg_run &r = runs[k];
        const bool back = (r.flags & PG_RUN_BACKWARD) != 0;
        pg_point p;
        p.mismatches = r.mismatches;
        p.chr_id = r.chr_id;
        p.direction = back ? '-' : '+';
        p.strand = (r.flags & PG_RUN_ANTISENSE) ? '-' : '+';
        for (uint32_t L = r.len_first; L <= r.len_last; L++) {
            const uint32_t d = L - r.len_first;
            p.abs_loc = back ? r.abs_loc_first - d : r.abs_loc_first + d;
            p.length = (int16_t)L;
            dst.push_back(make_point(p));
        }
    }
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
    parallel_ranges(reads.size(), [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            apply_rc_flag(reads[i], rv.rc_flag[i]);
            fill_points(reads[i].UP_Close, rv.close_runs, rv.close_off[i], rv.close_off[i + 1], make_point);
        }
    });
    return PG_OK;
}

// Far end at the reference's own call site (src/pindel.cpp:1888): `reads` is whatever vector the caller holds by
// then -- the reads of many CloseEndBatch flushes that kept a close end, in any order -- each carrying UnmatchedSeq
// as GetCloseEnd left it and a non-empty UP_Close (a read with an empty UP_Close is passed through: no far end).
// `hints` may be null.
template <class Read, class ChrOf, class MakePoint>
int SearchFarEnds(pg_ctx *ctx, std::vector<Read> &reads, ChrOf chr_of, MakePoint make_point, const pg_windows *hints)
{
    Batch b = make_batch(reads, chr_of);
    const size_t n = reads.size();
    std::vector<uint32_t> close_last(n);
    std::vector<int16_t> close_max(n);
    parallel_ranges(n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const bool has = !reads[i].UP_Close.empty();
            close_last[i] = has ? (uint32_t)reads[i].UP_Close[reads[i].UP_Close.size() - 1].AbsLoc : 0u;   // getLastAbsLocCloseEnd
            close_max[i] = has ? (int16_t)reads[i].UP_Close[reads[i].UP_Close.size() - 1].LengthStr : (int16_t)0;   // MaxLen
        }
    });
    pg_read_batch v = b.view();
    pg_result *res = nullptr;
    int rc_ = pg_far_end_batch_from_close(ctx, &v, close_last.data(), close_max.data(), hints, &res);
    if (rc_) return rc_;
    pg_result_view rv;
    pg_result_view_get(res, &rv);
    parallel_ranges(n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++)
            fill_points(reads[i].UP_Far, rv.far_runs, rv.far_off[i], rv.far_off[i + 1], make_point);
    });
    pg_result_free(res);
    return PG_OK;
}

// Both seams on ONE read vector (same order, same count as CloseEndBatch; saves the close-summary upload when a
// caller does keep the flush's reads together).  `hints` may be null.
template <class Read, class ChrOf, class MakePoint>
int SearchFarEnds(pg_ctx *ctx, std::vector<Read> &reads, ChrOf chr_of, MakePoint make_point,
                  pg_result *close_result, const pg_windows *hints)
{
    pg_result_view rv0;
    pg_result_view_get(close_result, &rv0);
    // pg_far_end_batch wants the sequences in their ORIGINAL orientation: undo CloseEndBatch's flip for the upload only
    Batch b = make_batch(reads, chr_of, rv0.rc_flag);
    pg_read_batch v = b.view();
    int rc_ = pg_far_end_batch(ctx, &v, close_result, hints);
    if (rc_) return rc_;
    pg_result_view rv;
    pg_result_view_get(close_result, &rv);
    parallel_ranges(reads.size(), [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++)
            fill_points(reads[i].UP_Far, rv.far_runs, rv.far_off[i], rv.far_off[i + 1], make_point);
    });
    return PG_OK;
}

}  // namespace pg_adapter
#endif

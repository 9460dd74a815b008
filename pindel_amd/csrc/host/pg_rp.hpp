// pg_rp.hpp -- discordant read pairs of a window as BreakDancer-like events (second half):
// what `-R` (default on for BAM input) adds to the far-end search windows.
//
// Restated from the reference (no code shared):
//   fetch_func_RP_Discovery / build_record_RP_Discovery   src/reader.cpp:925-1097, 1197-1245   which pairs count
//   get_RP_Reads_Discovery                                 src/reader.cpp:1378-1410             all BAMs of the run
//   BDData::UpdateBD                                       src/bddata.cpp:646-812               events = external + RP
//   ModifyRP / InitializeA1B1 / RecipicalOverlap / ProcessSameChromosomeSameStrand / Summarize
//                                                          src/bddata.cpp:138-560
// The reference's arithmetic is kept as written, including `abs()` of unsigned differences (taken as int) and the
// never-true `shift_distance * 2 < shift_distance` of the second coordinate.  Its OpenMP loops race (ModifyRP updates
// reads that other iterations read; UpdateBD pushes events in completion order): this is the sequential order,
// and the final event list is sorted anyway.  Interchromosomal pairs (`-I`, default off) are not handled.
//
// PARITY STATUS: unpinned -- the reference's BAM path cannot be built here (htslib) and the one BAM it ships
// (demo/simulated_MEI) has no same-chromosome discordant cluster: checked against an independent restatement on
// synthetic pairs (tests/test_bam_ingest.py) and on that BAM (tests/test_mei_bam.py: both find no event).
#ifndef PG_RP_HPP
#define PG_RP_HPP

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "pg_bam.hpp"

namespace pgh {

struct RpRead {
    std::string ChrNameA, ChrNameB;
    char DA = 'k', DB = 'k';
    unsigned PosA = 0, PosB = 0, OriginalPosA = 0, OriginalPosB = 0, PosA1 = 0, PosB1 = 0;
    int InsertSize = 0;
    short ReadLength = 0;
    unsigned NumberOfIdentical = 0;
    bool Report = false, Visited = false;
    std::vector<std::string> Tags;
};

struct RpEvent {                       // one BreakDancerEvent (both directions are derived by the consumer)
    std::string chr1, chr2;
    unsigned pos1, pos1b, pos2, pos2b; // Pindel coordinates (spacer included): position, position2 of each side
};

// build_record_RP_Discovery for every record of the window, appended to `out` (same-chromosome pairs only)
inline bool rp_discover(BamFile &bam, const std::string &chr_name, int64_t win_start, int64_t win_end, int insert_size,
                        const std::string &tag, unsigned min_anchor_quality, std::vector<RpRead> &out)
{
    const BamHeader &hdr = bam.header();
    const int tid = hdr.id_of(chr_name);
    auto consider = [&](const BamRecord &r, std::vector<RpRead> &dst) {
        if (!(r.flag & BAM_FPAIRED)) return;
        if (r.mapq < min_anchor_quality) return;
        if ((r.flag & BAM_FUNMAP) || (r.flag & BAM_FMUNMAP)) return;                     // both mates mapped
        const bool rev = (r.flag & BAM_FREVERSE) != 0, mrev = (r.flag & BAM_FMREVERSE) != 0;
        if (!((r.tid != r.mtid) || std::abs(r.tlen) > 3 * insert_size + 1000 || rev == mrev)) return;
        if (r.tid != r.mtid) return;                                                      // -I: not handled
        if (r.mtid < 0 || (size_t)r.mtid >= hdr.names.size()) return;
        RpRead t;
        t.DA = rev ? '-' : '+';
        t.DB = mrev ? '-' : '+';
        t.PosA = t.OriginalPosA = (unsigned)r.pos;
        t.PosB = t.OriginalPosB = (unsigned)r.mpos;
        t.ChrNameA = hdr.names[(size_t)r.tid];
        t.ChrNameB = hdr.names[(size_t)r.mtid];
        t.InsertSize = insert_size;
        t.Tags.push_back(tag);
        t.ReadLength = (short)r.l_seq;
        if (!(t.PosA < t.PosB)) {                                                         // first coordinate = the smaller one
            std::swap(t.DA, t.DB);
            std::swap(t.PosA, t.PosB);
            std::swap(t.OriginalPosA, t.OriginalPosB);
            std::swap(t.ChrNameA, t.ChrNameB);
        }
        dst.push_back(t);
    };
    // with an index: sub-ranges of the window on several threads, taken in order afterwards (BamFile::query_split)
    const unsigned nt = bam.split_parts(tid, win_start, win_end);
    if (nt > 1) {
        std::vector<std::vector<RpRead>> parts(nt);
        if (!bam.query_split(tid, win_start, win_end, nt, [&](unsigned t, const BamRecord &r, const BamFile &) { consider(r, parts[t]); }))
            return false;
        for (const std::vector<RpRead> &part : parts) out.insert(out.end(), part.begin(), part.end());
        return true;
    }
    return bam.query(tid, win_start, win_end, [&](const BamRecord &r) { consider(r, out); });
}

namespace rp_detail {

inline int iabs_u(unsigned a, unsigned b) { return std::abs((int)(a - b)); }            // abs(unsigned - unsigned) as the reference compiles it

inline bool reciprocal_overlap(const RpRead &first, const RpRead &second)
{
    const int distance = 1000;
    if (iabs_u(first.PosA, first.PosA1) > distance || iabs_u(first.PosB, first.PosB1) > distance ||
        iabs_u(second.PosA, second.PosA1) > distance || iabs_u(second.PosB, second.PosB1) > distance)
        return false;
    const float cutoff = 0.9f;
    unsigned fa = (first.PosA + first.PosA1) / 2, fb = (first.PosB + first.PosB1) / 2;
    if (fa > fb) std::swap(fa, fb);
    unsigned sa = (second.PosA + second.PosA1) / 2, sb = (second.PosB + second.PosB1) / 2;
    if (sa > sb) std::swap(sa, sb);
    if (first.DA != second.DA || first.DB != second.DB) return false;
    if (fa > sb + 200 || fb + 200 < sa) return false;
    if (fa <= sa && sb <= fb && (double)(sb - sa) / (double)(fb - fa) >= cutoff) return true;
    if (sa <= fa && fb <= sb && (double)(fb - fa) / (double)(sb - sa) >= cutoff) return true;
    if (fa <= sa && sa <= fb && fb <= sb && (double)(fb - sa) / (double)(fb - fa) >= cutoff &&
        (double)(fb - sa) / (double)(sb - sa) >= cutoff)
        return true;
    if (sa <= fa && fa <= sb && sb <= fb && (double)(sb - fa) / (double)(fb - fa) >= cutoff &&
        (double)(sb - fa) / (double)(sb - sa) >= cutoff)
        return true;
    return false;
}

inline void initialize_a1b1(std::vector<RpRead> &v)
{
    for (RpRead &r : v) {
        const unsigned D = (unsigned)r.InsertSize, L = (unsigned)r.ReadLength;
        if (r.DA == '+') {
            r.PosA = r.PosA > L * 2 ? r.PosA - L * 2 : 1;
            r.PosA1 = r.PosA + D + L * 2;
        } else {
            r.PosA = r.PosA > D ? r.PosA - D : 1;
            r.PosA1 = r.PosA + D + L;
        }
        if (r.DB == '+') {
            r.PosB = r.PosB > L * 2 ? r.PosB - L * 2 : 1;
            r.PosB1 = r.PosB + D + L;
        } else {
            r.PosB = r.PosB > D ? r.PosB - D : 1;
            r.PosB1 = r.PosB + D + L;
        }
    }
}

inline void same_chr_same_strand(RpRead &f, const RpRead &s)
{
    if (s.PosA1 - s.PosA > 10000 || s.PosB1 - s.PosB > 10000) return;
    if ((f.DA == '+' && f.PosA < s.PosA && s.PosA < f.PosA1 && f.PosA1 < s.PosA1) ||
        (f.DA == '-' && f.PosA < s.PosA1 && s.PosA1 < f.PosA1 && s.PosA < f.PosA)) {
        f.PosA = s.PosA;
        f.PosA1 = s.PosA1;
    }
    if ((f.DB == '+' && f.PosB < s.PosB && s.PosB < f.PosB1 && f.PosB1 < s.PosB1) ||
        (f.DB == '-' && s.PosB < f.PosB && f.PosB < s.PosB1 && s.PosB1 < f.PosB1)) {
        f.PosB = s.PosB;
        f.PosB1 = s.PosB1;
    }
}

inline void modify_rp(std::vector<RpRead> &v)
{
    if (v.empty()) return;
    std::sort(v.begin(), v.end(), [](const RpRead &a, const RpRead &b) {          // Compare2RP: descending
        if (a.OriginalPosA > b.OriginalPosA) return true;
        if (a.OriginalPosA == b.OriginalPosA) return a.OriginalPosB > b.OriginalPosB;
        return false;
    });
    initialize_a1b1(v);
    for (size_t i = 0; i < v.size(); i++)
        for (size_t j = 0; j < v.size(); j++) {
            const RpRead &second = v[j];                                           // (i == j: itself; nothing changes)
            if (reciprocal_overlap(v[i], second) && v[i].DA == second.DA && v[i].DB == second.DB) same_chr_same_strand(v[i], second);
        }
    for (RpRead &r : v) {
        if (r.DA == '+') { r.PosA += (unsigned)r.ReadLength; r.PosA1 += (unsigned)r.ReadLength; }
        if (r.DB == '+') { r.PosB += (unsigned)r.ReadLength; r.PosB1 += (unsigned)r.ReadLength; }
        if (r.ChrNameA == r.ChrNameB && iabs_u(r.PosA, r.PosB) < 500) r.Visited = true;
    }
}

inline bool same_box(const RpRead &a, const RpRead &b)
{
    return a.PosA == b.PosA && a.PosB == b.PosB && a.PosA1 == b.PosA1 && a.PosB1 == b.PosB1 && a.DA == b.DA && a.DB == b.DB;
}

inline void summarize(std::vector<RpRead> &v)
{
    const unsigned Cutoff = 5;
    if (v.size() < 5) {
        for (RpRead &r : v) r.Report = false;
        return;
    }
    std::vector<size_t> good;
    for (size_t i = 0; i + 1 < v.size(); i++) {
        if (v[i].Visited) continue;
        v[i].NumberOfIdentical = 1;
        for (size_t j = i + 1; j < v.size(); j++) {
            if (v[j].Visited) continue;
            if (same_box(v[i], v[j])) {
                v[i].NumberOfIdentical++;
                v[j].Visited = true;
                v[i].Tags.insert(v[i].Tags.end(), v[j].Tags.begin(), v[j].Tags.end());
                v[j].Tags.clear();
            }
        }
        good.push_back(i);
    }
    if (good.empty()) return;
    if (good.size() == 1) {
        v[good[0]].Report = v[good[0]].NumberOfIdentical >= Cutoff;
        return;
    }
    for (size_t a = 0; a + 1 < good.size(); a++) {
        RpRead &ra = v[good[a]];
        if (ra.Visited) continue;
        for (size_t b = a + 1; b < good.size(); b++) {
            RpRead &rb = v[good[b]];
            if (rb.Visited) continue;
            if (same_box(ra, rb)) {
                ra.NumberOfIdentical += rb.NumberOfIdentical;
                rb.Visited = true;
                ra.Tags.insert(ra.Tags.end(), rb.Tags.begin(), rb.Tags.end());
                rb.Tags.clear();
            }
        }
        ra.Report = ra.NumberOfIdentical >= Cutoff;
    }
}

}  // namespace rp_detail

// BDData::UpdateBD for the read pairs of one window: the events they support (>= 5 identical pairs), in the
// reference's order of discovery; `rp_out` (nullable) receives the lines the reference appends to <prefix>_RP.
inline std::vector<RpEvent> rp_events(std::vector<RpRead> &reads, unsigned spacer, std::ofstream *rp_out)
{
    using namespace rp_detail;
    std::vector<RpEvent> events;
    std::sort(reads.begin(), reads.end(), [](const RpRead &a, const RpRead &b) {   // SortByFirstAndThenSecondCoordinate
        if (a.PosA != b.PosA) return a.PosA < b.PosA;
        if (a.PosB != b.PosB) return a.PosB < b.PosB;
        return false;
    });
    modify_rp(reads);
    summarize(reads);
    for (RpRead &r : reads) {
        if (!r.Report) continue;
        const unsigned shift = (unsigned)r.InsertSize;
        unsigned f1 = r.PosA + spacer, f2 = r.PosA1 + spacer;
        if (f1 > f2) std::swap(f1, f2);
        if (r.DA == '+' && f1 > shift) f1 -= shift;
        else if (shift * 2 < spacer) f2 += shift;
        unsigned s1 = r.PosB + spacer, s2 = r.PosB1 + spacer;
        if (s1 > s2) std::swap(s1, s2);
        if (r.DB == '+' && s1 > shift) s1 -= shift;
        // (the reference's "else if (shift_distance * 2 < shift_distance)" never holds: s2 stays)
        if (r.ChrNameA.empty() || r.ChrNameB.empty()) continue;
        RpEvent e = { r.ChrNameA, r.ChrNameB, f1, f2, s1, s2 };
        events.push_back(e);
        if (rp_out) {
            std::ofstream &o = *rp_out;
            o << r.ChrNameA << "\t" << (f1 > spacer ? f1 - spacer : 1) << "\t" << f2 - spacer << "\t" << r.DA << "\t" << f2 - f1 << "\t"
              << r.ChrNameB << "\t" << (s1 > spacer ? s1 - spacer : 1) << "\t" << s2 - spacer << "\t" << r.DB << "\t" << s2 - s1 << "\t"
              << std::abs((int)s1 - (int)f1) << "\tSupport: " << r.NumberOfIdentical << "\t";
            // DisplayBDSupportPerSample: tags sorted, "\t<tag> <count>" per tag
            std::sort(r.Tags.begin(), r.Tags.end());
            for (size_t i = 0; i < r.Tags.size();) {
                size_t j = i;
                while (j < r.Tags.size() && r.Tags[j] == r.Tags[i]) j++;
                o << "\t" << r.Tags[i] << " " << (j - i);
                i = j;
            }
            o << std::endl;
        }
    }
    reads.clear();
    return events;
}

}  // namespace pgh
#endif

// pg_host_sv.cpp -- tandem duplications and inversions: classifiers + reporters.
//   searchTandemDuplications(NT)   src/search_tandem_duplications.cpp:29-222, _nt.cpp:27-131
//   searchInversions(NT)           src/search_inversions.cpp:29-318, _nt.cpp:27-200
//   SortAndOutputTandemDuplications / OutputTDs   src/reporter.cpp:1157-1287, 157-269
//   DoSortAndOutputInversions / OutputInversions  src/output_sorter.cpp:69-257, reporter.cpp:446-628
//   OutputShortInversion                          src/reporter.cpp:1588-1696
#include <algorithm>
#include <fstream>
#include <sstream>

#include "pg_host_priv.hpp"

namespace pgh {

using namespace detail;

// Places a classified read into its box (the tail shared by all classifiers).
#define PGH_BOX_READ(r, ri, boxes, check_transgress)                                         \
    do {                                                                                     \
        if ((check_transgress) && transgresses((r), c.win_end)) {                            \
            (r).Used = true;                                                                 \
        } else if ((r).BPLeft + 1 >= c.region_start && (r).BPLeft + 1 <= c.region_end) {     \
            unsigned box_ = (unsigned)((int)(r).BPLeft / (int)BoxSize);                      \
            if (box_ < c.NumBoxes) {                                                         \
                (boxes)[box_].push_back(ri);                                                 \
                (r).Used = true;                                                             \
            }                                                                                \
        }                                                                                    \
    } while (0)

// LeftMostTD, src/search_tandem_duplications.cpp:194-222
static void left_most_td(const std::string &ref, unsigned spacer, SplitRead &r)
{
    unsigned pos = r.BPLeft + spacer, orig = pos, end = r.BPRight + spacer - 1;
    unsigned n = (unsigned)ref.size();
    if (pos >= n || end >= n) {
        r.BPLeft = 1;
        r.BPRight = 1;
        r.BP = 1;
        r.Used = true;
        return;
    }
    while (ref[pos] == ref[end]) {
        --pos;
        --end;
    }
    int diff = (int)(orig - pos);
    if (diff > 0) {
        if (diff >= r.BP) diff = r.BP - 1;
        r.BPLeft -= diff;
        r.BPRight -= diff;
        r.BP -= (short)diff;
    }
}

void Caller::search_tandem_dup(Ctx &c)
{
    std::vector<SplitRead> &reads = *c.reads;
    const std::string &ref = c.chrom->seq;
    std::vector<std::vector<unsigned>> boxes(c.NumBoxes);
    classify_reads(reads.size(), boxes, [&](unsigned lo_, unsigned hi_, BoxSink &sink) {
    for (unsigned ri = lo_; ri < hi_; ri++) {
        SplitRead &r = reads[ri];
        if (r.Used || r.UP_Far.empty() || r.FragName != r.FarFragName) continue;
        const bool plus = r.MatchedD == '+';
        if (!plus && r.MatchedD != '-') continue;
        const int nc = (int)r.UP_Close.size(), nf = (int)r.UP_Far.size();
        const FarByLength by_len(r);
        for (short budget = 0; budget <= r.MAX_SNP_ERROR; budget++) {
            for (int k = 0; k < nc; k++) {
                if (r.Used) break;
                const UniquePoint &cp = r.UP_Close[plus ? k : nc - 1 - k];
                if (cp.Mismatches > budget) continue;
                // only the far point of length ReadLength - LengthStr(close) can pass the tests below
                int fi_first, fi_last;
                by_len.range_desc(r.getReadLength(), cp.LengthStr, nf, fi_first, fi_last);
                int j_first = 0, j_last = nf - 1;
                if (by_len.usable) {
                    if (fi_first < 0) continue;
                    j_first = j_last = plus ? nf - 1 - fi_first : fi_first;
                }
                for (int j = j_first; j <= j_last; j++) {
                    if (r.Used) break;
                    const UniquePoint &fp = r.UP_Far[plus ? nf - 1 - j : j];
                    if (fp.Mismatches > budget) continue;
                    if (fp.Mismatches + cp.Mismatches > budget) continue;
                    if (plus) {
                        if (fp.Direction != '-') continue;
                        if (!(fp.LengthStr + cp.LengthStr == r.getReadLength() &&
                              fp.AbsLoc + fp.LengthStr < cp.AbsLoc && fp.AbsLoc + cp.LengthStr < cp.AbsLoc))
                            continue;
                        r.Right = (int)(cp.AbsLoc - cp.LengthStr + 1);
                        r.Left = (int)(fp.AbsLoc + fp.LengthStr - 1);
                        r.BP = (short)(cp.LengthStr - 1);
                        r.IndelSize = cp.AbsLoc - fp.AbsLoc + 1;
                        r.BPRight = cp.AbsLoc - S.spacer;
                        r.BPLeft = fp.AbsLoc - S.spacer;
                    } else {
                        if (fp.Direction != '+') continue;
                        if (!(cp.LengthStr + fp.LengthStr == r.getReadLength() &&
                              cp.AbsLoc + cp.LengthStr < fp.AbsLoc && cp.AbsLoc + fp.LengthStr < fp.AbsLoc))
                            continue;
                        r.Right = (int)(fp.AbsLoc - fp.LengthStr + 1);
                        r.Left = (int)(cp.AbsLoc + cp.LengthStr - 1);
                        r.BP = (short)(fp.LengthStr - 1);
                        r.IndelSize = fp.AbsLoc - cp.AbsLoc + 1;
                        r.BPRight = fp.AbsLoc - S.spacer;
                        r.BPLeft = cp.AbsLoc - S.spacer;
                    }
                    if (r.BPLeft == 0) continue;
                    left_most_td(ref, S.spacer, r);
                    PGH_BOX_READ(r, ri, sink, true);
                }
            }
        }
    }
    });
    sort_output_td(c, boxes, false);
}

void Caller::search_tandem_dup_nt(Ctx &c)
{
    std::vector<SplitRead> &reads = *c.reads;
    std::vector<std::vector<unsigned>> boxes(c.NumBoxes);
    classify_reads(reads.size(), boxes, [&](unsigned lo_, unsigned hi_, BoxSink &sink) {
    for (unsigned ri = lo_; ri < hi_; ri++) {
        SplitRead &r = reads[ri];
        if (r.Used || r.UP_Far.empty() || r.FragName != r.FarFragName) continue;
        const UniquePoint &cp = r.UP_Close.back();
        const UniquePoint &fp = r.UP_Far.back();
        if (fp.LengthStr + cp.LengthStr >= r.getReadLength()) continue;
        if (fp.Mismatches + cp.Mismatches > (short)(1 + S.Seq_Error_Rate * (fp.LengthStr + cp.LengthStr))) continue;
        if (r.MatchedD == '+') {
            if (fp.Direction != '-') continue;
            if (!(fp.AbsLoc + fp.LengthStr < cp.AbsLoc && fp.AbsLoc + cp.LengthStr < cp.AbsLoc &&
                  fp.LengthStr + cp.LengthStr > S.Min_Num_Matched_Bases))
                continue;
            r.Right = (int)(cp.AbsLoc - cp.LengthStr + 1);
            r.Left = (int)(fp.AbsLoc + fp.LengthStr - 1);
            r.BP = (short)(cp.LengthStr - 1);
            r.IndelSize = cp.AbsLoc - fp.AbsLoc + 1;
            r.NT_size = (unsigned short)(r.getReadLength() - cp.LengthStr - fp.LengthStr);
            r.NT_str = sub(reverse_complement(r.UnmatchedSeq), r.BP + 1, r.NT_size);
            r.BPRight = cp.AbsLoc - S.spacer;
            r.BPLeft = fp.AbsLoc - S.spacer;
        } else if (r.MatchedD == '-') {
            if (fp.Direction != '+') continue;
            if (!(cp.AbsLoc + cp.LengthStr < fp.AbsLoc && cp.AbsLoc + fp.LengthStr < fp.AbsLoc &&
                  fp.LengthStr + cp.LengthStr > S.Min_Num_Matched_Bases))
                continue;
            r.Right = (int)(fp.AbsLoc - fp.LengthStr + 1);
            r.Left = (int)(cp.AbsLoc + cp.LengthStr - 1);
            r.BP = (short)(fp.LengthStr - 1);
            r.IndelSize = fp.AbsLoc - cp.AbsLoc + 1;
            r.NT_size = (unsigned short)(r.getReadLength() - cp.LengthStr - fp.LengthStr);
            r.NT_str = sub(r.UnmatchedSeq, r.BP + 1, r.NT_size);
            r.BPRight = fp.AbsLoc - S.spacer;
            r.BPLeft = cp.AbsLoc - S.spacer;
        } else {
            continue;
        }
        PGH_BOX_READ(r, ri, sink, true);
    }
    });
    sort_output_td(c, boxes, true);
}

// OutputTDs, src/reporter.cpp:157-269
void Caller::output_td(Ctx &c, std::vector<SplitRead> &g, unsigned s, unsigned e, unsigned, unsigned)
{
    std::ostream &out = report(REP_TD);
    const std::string &ref = c.chrom->seq;
    const SplitRead &f = g[s];
    unsigned n_reads = 0;
    std::string sup = support_columns(g, s, e, f.BPLeft - 1, f.BPRight + 1, n_reads);
    out << HASHES << '\n';
    out << ev_no(EV_TD) << "\tTD " << f.IndelSize << "\tNT " << f.NT_size << " \"" << f.NT_str << "\"\tChrID "
        << f.FragName << "\tBP " << f.BPLeft << "\t" << f.BPRight + 2 << "\tBP_range " << f.BPLeft << "\t"
        << f.BPRight + 2 << sup << '\n';
    const long rl = g_reportLength;
    out << sub(ref, (long)f.BPRight + S.spacer - rl + 1, rl) << std::string(f.NT_size, ' ')
        << cap2low(sub(ref, (long)f.BPLeft + S.spacer, rl)) << '\n';
    for (unsigned i = s; i <= e; i++) {
        const SplitRead &r = g[i];
        short before = (short)(rl - r.BP - 1);
        out << std::string(before > 0 ? before : 0, ' ');
        out << (r.MatchedD == '-' ? r.UnmatchedSeq : reverse_complement(r.UnmatchedSeq)) << '\n';
        out << read_tail(r) << '\n';
    }
}

// SortAndOutputTandemDuplications, src/reporter.cpp:1157-1287
void Caller::sort_output_td(Ctx &c, std::vector<std::vector<unsigned>> &boxes, bool)
{
    std::vector<SplitRead> &reads = *c.reads;
    const std::string &ref = c.chrom->seq;
    struct Ev { unsigned s, e, bl, br, rs, re; };
    for_boxes(c.NumBoxes, [&](unsigned b) {
        std::vector<unsigned> &box = boxes[b];
        if (box.empty() || box.size() < S.NumRead2ReportCutOff) return;
        exchange_sort(reads, box);                       // bubblesortReads
        mark_duplicates(reads, box);                     // markDuplicates
        std::vector<SplitRead> good;
        for (unsigned i : box)
            if (reads[i].UniqueRead) good.push_back(reads[i]);
        if (good.empty()) return;
        std::vector<Ev> evs;
        Ev cur = { 0, 0, good[0].BPLeft, good[0].BPRight, 0, 0 };
        auto close_event = [&]() {
            cur.rs = cur.bl;
            cur.re = cur.br;
            real_start_deletion(ref, S.spacer, cur.rs, cur.re);
            evs.push_back(cur);
        };
        for (unsigned i = 1; i < good.size(); i++) {
            if (good[i].BPLeft == cur.bl && good[i].BPRight == cur.br) cur.e = i;
            else {
                close_event();
                cur.s = cur.e = i;
                cur.bl = good[i].BPLeft;
                cur.br = good[i].BPRight;
            }
        }
        close_event();
        for (const Ev &ev : evs) {
            if (ev.e - ev.s + 1 < S.NumRead2ReportCutOff) continue;
            // IsGoodTD (reporter.cpp:1093-1155) for Pindel-text / points input
            if (ev.re < ev.rs || ev.rs == 0) continue;
            if (good[ev.s].IndelSize < S.BalanceCutoff || report_event(good, ev.s, ev.e)) {
                output_td(c, good, ev.s, ev.e, ev.rs, ev.re);
            }
        }
    });
}

// ------------------------------------------------------------------------------ inversions
// LeftMostINV, src/search_inversions.cpp:285-318
static void left_most_inv(const std::string &ref, unsigned spacer, SplitRead &r)
{
    unsigned n = (unsigned)ref.size();
    unsigned pos = r.BPLeft + spacer + 1, orig = pos, end = r.BPRight + spacer - 1;
    if (n <= pos + spacer || n <= orig + spacer) {
        r.BPLeft = 1;
        r.BPRight = 1;
        r.BP = 1;
        r.Used = true;
        return;
    }
    while (ref[pos] == rc4n(ref[end])) {
        ++pos;
        --end;
    }
    short diff = (short)(pos - orig);
    if (diff > 0) {
        if (r.MatchedD == '+') {
            if (diff >= r.BP) diff = (short)(r.BP - 1);
        } else {
            if (diff + r.BP >= r.getReadLength()) diff = (short)(r.getReadLength() - r.BP - 1);
            r.BPLeft += diff;
            r.BPRight -= diff;
            r.BP += diff;
        }
    }
}

void Caller::search_inversions(Ctx &c)
{
    std::vector<SplitRead> &reads = *c.reads;
    const std::string &ref = c.chrom->seq;
    std::vector<std::vector<unsigned>> boxes(c.NumBoxes);
    const unsigned MIN = (unsigned)S.MIN_IndelSize_Inversion;
    classify_reads(reads.size(), boxes, [&](unsigned lo_, unsigned hi_, BoxSink &sink) {
    for (unsigned ri = lo_; ri < hi_; ri++) {
        SplitRead &r = reads[ri];
        if (r.Used || r.UP_Far.empty() || r.FragName != r.FarFragName) continue;
        if (!(r.UP_Close[0].Strand != r.UP_Far[0].Strand && r.UP_Close[0].Direction == r.UP_Far[0].Direction))
            continue;
        const int nc = (int)r.UP_Close.size(), nf = (int)r.UP_Far.size();
        const bool plus = r.MatchedD == '+';
        if (!plus && r.MatchedD != '-') continue;
        // which of the two geometries (search_inversions.cpp:53, 118, 186, 245)
        int geo;
        if (plus) {
            if (r.UP_Far[0].AbsLoc > r.UP_Close.back().AbsLoc + MIN) geo = 0;
            else if (r.UP_Far.back().AbsLoc + MIN < r.UP_Close[0].AbsLoc) geo = 1;
            else continue;
        } else {
            if (r.UP_Close.back().AbsLoc > r.UP_Far[0].AbsLoc + MIN) geo = 2;
            else if (r.UP_Close[0].AbsLoc + MIN < r.UP_Far.back().AbsLoc) geo = 3;
            else continue;
        }
        const bool close_desc = (geo == 0 || geo == 2);   // CloseIndex from the back, FarIndex ascending
        for (short budget = 0; budget <= r.MAX_SNP_ERROR; budget++) {
            for (int k = 0; k < nc; k++) {
                if (r.Used) break;
                const UniquePoint &cp = r.UP_Close[close_desc ? nc - 1 - k : k];
                if (cp.Mismatches > budget) continue;
                for (int j = 0; j < nf; j++) {
                    if (r.Used) break;
                    const UniquePoint &fp = r.UP_Far[close_desc ? j : nf - 1 - j];
                    if (fp.Mismatches > budget) continue;
                    if (fp.Mismatches + cp.Mismatches > budget) continue;
                    if (fp.Direction != (plus ? '+' : '-')) continue;
                    if (fp.LengthStr + cp.LengthStr != r.getReadLength()) continue;
                    if (geo == 0) {
                        if (!(fp.AbsLoc > cp.AbsLoc + MIN)) continue;
                        r.Left = (int)((cp.AbsLoc + 1) - cp.LengthStr);
                        r.Right = (int)(fp.AbsLoc - fp.LengthStr + r.getReadLength());
                        r.BP = (short)(cp.LengthStr - 1);
                        r.IndelSize = fp.AbsLoc - cp.AbsLoc;
                        r.BPLeft = cp.AbsLoc + 1 - S.spacer;
                        r.BPRight = fp.AbsLoc - S.spacer;
                    } else if (geo == 1) {
                        if (!(fp.AbsLoc + MIN < cp.AbsLoc)) continue;
                        r.Right = (int)(cp.AbsLoc - cp.LengthStr + r.getReadLength());
                        r.Left = (int)(fp.AbsLoc - fp.LengthStr + 1);
                        r.BP = (short)(fp.LengthStr - 1);
                        r.IndelSize = cp.AbsLoc - fp.AbsLoc;
                        r.BPRight = cp.AbsLoc - S.spacer;
                        r.BPLeft = (fp.AbsLoc + 1) - S.spacer;
                    } else if (geo == 2) {
                        if (!(cp.AbsLoc > fp.AbsLoc + MIN)) continue;
                        r.Left = (int)(fp.AbsLoc + fp.LengthStr - r.getReadLength());
                        r.Right = (int)(cp.AbsLoc + cp.LengthStr - 1);
                        r.BP = (short)(fp.LengthStr - 1);
                        r.IndelSize = cp.AbsLoc - fp.AbsLoc;
                        r.BPLeft = fp.AbsLoc - S.spacer;
                        r.BPRight = cp.AbsLoc - 1 - S.spacer;
                    } else {
                        if (!(cp.AbsLoc + MIN < fp.AbsLoc)) continue;
                        r.Right = (int)(fp.AbsLoc + fp.LengthStr - 1);
                        r.Left = (int)(cp.AbsLoc + cp.LengthStr - r.getReadLength());
                        r.BP = (short)(cp.LengthStr - 1);
                        r.IndelSize = fp.AbsLoc - cp.AbsLoc;
                        r.BPLeft = cp.AbsLoc - S.spacer;
                        r.BPRight = fp.AbsLoc - 1 - S.spacer;
                    }
                    r.NT_str = "";
                    r.NT_size = 0;
                    left_most_inv(ref, S.spacer, r);
                    PGH_BOX_READ(r, ri, sink, plus);   // the '-' branches skip the bin-border test
                }
            }
        }
    }
    });
    sort_output_inv(c, boxes, false);
}

void Caller::search_inversions_nt(Ctx &c)
{
    std::vector<SplitRead> &reads = *c.reads;
    std::vector<std::vector<unsigned>> boxes(c.NumBoxes);
    const unsigned MIN = (unsigned)S.MIN_IndelSize_Inversion;
    classify_reads(reads.size(), boxes, [&](unsigned lo_, unsigned hi_, BoxSink &sink) {
    for (unsigned ri = lo_; ri < hi_; ri++) {
        SplitRead &r = reads[ri];
        if (r.Used || r.UP_Far.empty() || r.FragName != r.FarFragName) continue;
        const UniquePoint &cp = r.UP_Close.back();
        const UniquePoint &fp = r.UP_Far.back();
        if (fp.Mismatches + cp.Mismatches > (short)(1 + S.Seq_Error_Rate * (fp.LengthStr + cp.LengthStr))) continue;
        if (!(r.UP_Close[0].Strand != r.UP_Far[0].Strand && r.UP_Close[0].Direction == r.UP_Far[0].Direction))
            continue;
        const bool short_enough = fp.LengthStr + cp.LengthStr < r.getReadLength() &&
                                  fp.LengthStr + cp.LengthStr >= S.Min_Num_Matched_Bases;
        if (!short_enough) continue;
        const unsigned short nt = (unsigned short)(r.getReadLength() - fp.LengthStr - cp.LengthStr);
        if (r.MatchedD == '+') {
            if (fp.Direction != '+') continue;
            if (fp.AbsLoc > cp.AbsLoc + MIN) {
                r.Left = (int)((cp.AbsLoc + 1) - cp.LengthStr);
                r.Right = (int)(fp.AbsLoc - fp.LengthStr + r.getReadLength());
                r.BP = (short)(cp.LengthStr - 1);
                r.IndelSize = fp.AbsLoc - cp.AbsLoc;
                r.NT_size = nt;
                r.NT_str = sub(reverse_complement(r.UnmatchedSeq), r.BP + 1, r.NT_size);
                r.BPLeft = cp.AbsLoc + 1 - S.spacer;
                r.BPRight = fp.AbsLoc - S.spacer;
                PGH_BOX_READ(r, ri, sink, true);
            }
            if (fp.AbsLoc + MIN < cp.AbsLoc) {
                r.Right = (int)(cp.AbsLoc - cp.LengthStr + r.getReadLength());
                r.Left = (int)(fp.AbsLoc - fp.LengthStr + 1);
                r.BP = (short)(fp.LengthStr - 1);
                r.IndelSize = cp.AbsLoc - fp.AbsLoc;
                r.NT_size = nt;
                r.NT_str = sub(r.UnmatchedSeq, r.BP + 1, r.NT_size);
                r.BPRight = cp.AbsLoc - S.spacer;
                r.BPLeft = (fp.AbsLoc + 1) - S.spacer;
                PGH_BOX_READ(r, ri, sink, true);
            }
        } else if (r.MatchedD == '-') {
            if (fp.Direction != '-') continue;
            if (cp.AbsLoc > fp.AbsLoc + MIN) {
                r.Left = (int)(fp.AbsLoc + fp.LengthStr - r.getReadLength());
                r.Right = (int)(cp.AbsLoc + cp.LengthStr - 1);
                r.BP = (short)(fp.LengthStr - 1);
                r.IndelSize = cp.AbsLoc - fp.AbsLoc;
                r.NT_size = nt;
                r.NT_str = sub(r.UnmatchedSeq, r.BP + 1, r.NT_size);
                r.BPLeft = fp.AbsLoc - S.spacer;
                r.BPRight = cp.AbsLoc - 1 - S.spacer;
                PGH_BOX_READ(r, ri, sink, true);
            }
            if (cp.AbsLoc + MIN < fp.AbsLoc) {
                r.Right = (int)(fp.AbsLoc + fp.LengthStr - 1);
                r.Left = (int)(cp.AbsLoc + cp.LengthStr - r.getReadLength());
                r.BP = (short)(cp.LengthStr - 1);
                r.IndelSize = fp.AbsLoc - cp.AbsLoc;
                r.NT_size = nt;
                r.NT_str = sub(reverse_complement(r.UnmatchedSeq), r.BP + 1, r.NT_size);
                r.BPLeft = cp.AbsLoc - S.spacer;
                r.BPRight = fp.AbsLoc - 1 - S.spacer;
                PGH_BOX_READ(r, ri, sink, true);
            }
        }
    }
    });
    sort_output_inv(c, boxes, true);
}

// OutputInversions, src/reporter.cpp:446-628
void Caller::output_inv(Ctx &c, std::vector<SplitRead> &g, unsigned s, unsigned e, unsigned, unsigned)
{
    std::ostream &out = report(REP_INV);
    const std::string &ref = c.chrom->seq;
    const SplitRead &f = g[s];
    short lnt = 0, rnt = 0;
    std::string lstr, rstr;
    for (unsigned i = s; i <= e; i++)
        if (g[i].MatchedD == '+') {
            lnt = (short)g[i].NT_size;
            lstr = g[i].NT_str;
            break;
        }
    for (unsigned i = s; i <= e; i++)
        if (g[i].MatchedD == '-') {
            rnt = (short)g[i].NT_size;
            rstr = g[i].NT_str;
            break;
        }
    unsigned n_reads = 0;
    std::string sup = support_columns(g, s, e, f.BPLeft - 1, f.BPRight + 1, n_reads);
    out << HASHES << '\n';
    out << ev_no(EV_INV) << "\tINV " << f.IndelSize << "\tNT " << lnt << ":" << rnt << " \"" << lstr << "\":\"" << rstr
        << "\"\tChrID " << f.FragName << "\tBP " << f.BPLeft + 1 - 1 << "\t" << f.BPRight + 1 + 1 << "\tBP_range "
        << f.BPLeft + 1 - 1 << "\t" << f.BPRight + 1 + 1 << sup << '\n';
    const long rl = g_reportLength;
    out << sub(ref, (long)f.BPLeft + S.spacer - rl, rl) << std::string(lnt > 0 ? lnt : 0, ' ')
        << cap2low(reverse_complement(sub(ref, (long)f.BPRight + 1 + S.spacer - rl, rl))) << '\n';
    for (unsigned i = s; i <= e; i++) {
        const SplitRead &r = g[i];
        if (r.MatchedD != '+') continue;
        short before = (short)(rl - r.BP - 1);
        out << std::string(before > 0 ? before : 0, ' ');
        if (r.UP_Close[0].AbsLoc < r.UP_Far[0].AbsLoc)
            out << reverse_complement(r.UnmatchedSeq) << std::string(r.BP > 0 ? r.BP : 0, ' ');
        else
            out << r.UnmatchedSeq;
        out << read_tail(r) << '\n';
    }
    out << DASHES << '\n';
    out << cap2low(reverse_complement(sub(ref, (long)f.BPLeft + S.spacer, rl))) << std::string(rnt > 0 ? rnt : 0, ' ')
        << sub(ref, (long)f.BPRight + 1 + S.spacer, rl) << '\n';
    for (unsigned i = s; i <= e; i++) {
        const SplitRead &r = g[i];
        if (r.MatchedD != '-') continue;
        short before = (short)(rl - r.BP - 1);
        out << std::string(before > 0 ? before : 0, ' ');
        if (r.UP_Close[0].AbsLoc > r.UP_Far[0].AbsLoc)
            out << r.UnmatchedSeq << std::string(r.BP > 0 ? r.BP : 0, ' ');
        else
            out << reverse_complement(r.UnmatchedSeq);
        out << read_tail(r) << '\n';
    }
}

// OutputShortInversion, src/reporter.cpp:1588-1696
void Caller::output_short_inv(Ctx &c, std::vector<SplitRead> &g, unsigned s, unsigned e)
{
    std::ostream &out = report(REP_INV);
    const std::string &ref = c.chrom->seq;
    const SplitRead &f = g[s];
    unsigned n_reads = 0;
    std::string sup = support_columns(g, s, e, f.BPLeft, f.BPRight, n_reads);
    out << HASHES << '\n';
    out << ev_no(EV_INV) << "\tINV " << f.IndelSize << "\tNT " << f.NT_size << " \"" << f.NT_str << "\"\tChrID "
        << f.FragName << "\tBP " << f.BPLeft + 1 << "\t" << f.BPRight + 1 << "\tBP_range " << f.BPLeft + 1 << "\t"
        << f.BPRight + 1 << sup << '\n';
    const long rl = g_reportLength;
    out << sub(ref, (long)f.Left - rl + f.BP + 1, rl)
        << cap2low(reverse_complement(sub(ref, (long)f.Left + f.BP + 1, f.NT_size)))
        << sub(ref, (long)f.Left + f.BP + 1 + f.IndelSize, rl) << '\n';
    for (unsigned i = s; i <= e; i++) {
        const SplitRead &r = g[i];
        short before = (short)(rl - r.BP - 1);
        out << std::string(before > 0 ? before : 0, ' ');
        out << (r.MatchedD == '-' ? r.UnmatchedSeq : reverse_complement(r.UnmatchedSeq)) << "\t";
        out << read_tail(r) << '\n';
    }
}

// OutputSorter::DoSortAndOutputInversions, src/output_sorter.cpp:69-257
void Caller::sort_output_inv(Ctx &c, std::vector<std::vector<unsigned>> &boxes, bool nt)
{
    std::vector<SplitRead> &reads = *c.reads;
    struct Ev { unsigned s, e, rs, re; };
    for_boxes(c.NumBoxes, [&](unsigned b) {
        std::vector<unsigned> &box = boxes[b];
        if (box.empty() || box.size() < S.NumRead2ReportCutOff) return;
        const size_t n = box.size();
        for (size_t a = 0; a + 1 < n; a++)
            for (size_t d = a + 1; d < n; d++) {
                const SplitRead &x = reads[box[a]], &y = reads[box[d]];
                bool swap = false;
                const unsigned sx = x.BPLeft + x.BPRight, sy = y.BPLeft + y.BPRight;
                if (sx < sy) continue;
                else if (sx > sy) swap = true;
                else if (x.IndelSize > y.IndelSize) continue;      // larger ones first
                else if (x.IndelSize < y.IndelSize) swap = true;
                else if (x.BPLeft < y.BPLeft) continue;
                else if (x.BPLeft > y.BPLeft) swap = true;
                else {
                    if (x.BPRight < y.BPRight) continue;
                    else if (x.BPRight > y.BPRight) swap = true;
                    else if (nt) {
                        if (x.NT_size < y.NT_size) continue;
                        else if (x.NT_size > y.NT_size) swap = true;
                        else if (x.BP > y.BP) swap = true;
                    } else if (x.BP > y.BP) swap = true;
                }
                if (swap) std::swap(box[a], box[d]);
            }
        for (size_t a = 0; a + 1 < n; a++)
            for (size_t d = a + 1; d < n; d++) {
                const SplitRead &x = reads[box[a]];
                SplitRead &y = reads[box[d]];
                if ((x.LeftMostPos == y.LeftMostPos ||
                     x.LeftMostPos + x.getReadLength() == y.LeftMostPos + y.getReadLength()) &&
                    x.MatchedD == y.MatchedD)
                    y.UniqueRead = false;
            }
        std::vector<SplitRead> good;                 // ALL reads of the box, unique or not
        for (unsigned i : box) good.push_back(reads[i]);
        if (good.empty()) return;
        std::vector<Ev> evs;
        unsigned cs = 0, ce = 0, cbl = good[0].BPLeft, cbr = good[0].BPRight;
        bool whether = true;                          // never re-armed inside a box (output_sorter.cpp:150)
        auto harmonise = [&]() {
            unsigned mx = 0;
            for (unsigned i = cs; i <= ce; i++) mx = std::max(mx, good[i].IndelSize);
            for (unsigned i = cs; i <= ce; i++) {
                SplitRead &r = good[i];
                if (r.IndelSize / (float)mx < 0.95 || mx + 30 > (unsigned)(r.getReadLength() + r.IndelSize)) {
                    whether = false;
                    break;
                }
                short diff = (short)((mx - r.IndelSize) / 2);
                r.IndelSize = mx;
                r.BPLeft = r.BPLeft - diff;
                r.BPRight = r.BPRight + diff;
                if (r.MatchedD == '+') {
                    if (r.BP > diff) r.BP = (short)(r.BP - diff);
                } else {
                    if (r.BP + diff < r.getReadLengthMinus()) r.BP = (short)(r.BP + diff);
                }
            }
        };
        for (unsigned i = 1; i < good.size(); i++) {
            if (good[i].BPLeft + good[i].BPRight == cbl + cbr) {
                ce = i;
            } else {
                harmonise();
                if (whether) evs.push_back({ cs, ce, good[cs].BPLeft, good[cs].BPRight });
                cs = ce = i;
                cbl = good[i].BPLeft;
                cbr = good[i].BPRight;
            }
        }
        harmonise();
        if (whether) evs.push_back({ cs, ce, cbl, cbr });
        for (const Ev &ev : evs) {
            if (ev.e - ev.s + 1 < S.NumRead2ReportCutOff) continue;
            if (ev.re < ev.rs || ev.rs == 0) continue;          // IsGoodINV, text/points input
            if (good[ev.s].IndelSize < S.BalanceCutoff || report_event(good, ev.s, ev.e))
                output_inv(c, good, ev.s, ev.e, ev.rs, ev.re);
        }
    });
}

}  // namespace pgh

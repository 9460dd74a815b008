// pg_host.cpp -- loaders, SV classifiers and text reporters (see pg_host.hpp).
#include "pg_host_priv.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <array>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <thread>

namespace pgh {

using namespace detail;

std::string reverse_complement(const std::string &s)   // src/pindel.cpp:2037-2048
{
    std::string o(s.size(), 'N');
    for (size_t j = 0; j < s.size(); j++) o[j] = rc4n(s[s.size() - 1 - j]);
    return o;
}

// ------------------------------------------------------------------ loaders
int load_fasta(const std::string &path, std::vector<Chromosome> &out, unsigned spacer, std::string &err)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) {
        err = "cannot open " + path;
        return -1;
    }
    std::string data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t i = 0, n = data.size();
    while (i < n && isspace((unsigned char)data[i])) i++;
    if (i >= n || data[i] != '>') {
        err = "fasta does not start with '>'";
        return -1;
    }
    const std::string pad(spacer, 'N');
    while (i < n) {
        i++;
        while (i < n && (data[i] == ' ' || data[i] == '\t')) i++;
        size_t j = i;
        while (j < n && !isspace((unsigned char)data[j])) j++;
        Chromosome c;
        c.name = data.substr(i, j - i);
        while (j < n && data[j] != '\n') j++;
        c.seq = pad;
        i = j;
        while (i < n && data[i] != '>') {
            unsigned char ch = (unsigned char)data[i++];
            if (isspace(ch)) continue;
            ch = (unsigned char)toupper(ch);
            if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') ch = 'N';
            c.seq.push_back((char)ch);
        }
        // the reference's extraction loop repeats the final base of the last record
        if (i >= n && c.seq.size() > pad.size()) c.seq.push_back(c.seq.back());
        c.seq += pad;
        out.push_back(c);
    }
    return 0;
}

int load_pindel_text(const std::string &path, const std::vector<Chromosome> &genome,
                     std::vector<SplitRead> &out, std::string &err)
{
    // Three lines per record, taken in order from the top whatever they contain (PindelReadReader: getline x 3); the
    // list ends at an empty name line or at an incomplete record.  The file is read in one piece, cut into lines,
    // and the records are parsed on several threads; the first malformed record (in file order) is the error.
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        err = "cannot open " + path;
        return -1;
    }
    std::string buf;
    {
        char tmp[1 << 16];
        size_t k;
        fseeko(f, 0, SEEK_END);
        const off_t sz = ftello(f);
        fseeko(f, 0, SEEK_SET);
        if (sz > 0) buf.reserve((size_t)sz);
        while ((k = fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, k);
        fclose(f);
    }
    std::vector<size_t> ls;                                   // line starts; line i = [ls[i], ls[i + 1] - 1)
    for (size_t at = 0; at < buf.size();) {
        ls.push_back(at);
        const void *nl = memchr(buf.data() + at, '\n', buf.size() - at);
        at = nl ? (size_t)((const char *)nl - buf.data()) + 1 : buf.size() + 1;
    }
    const size_t n_lines = ls.size();
    ls.push_back(buf.size() + 1);
    auto line = [&](size_t i) { return std::string(buf.data() + ls[i], std::min(ls[i + 1] - 1, buf.size()) - ls[i]); };
    size_t n_rec = n_lines / 3;
    for (size_t k = 0; k < n_rec; k++)
        if (std::min(ls[3 * k + 1] - 1, buf.size()) == ls[3 * k]) {                 // empty name line: end of the list
            n_rec = k;
            break;
        }
    const size_t base = out.size();
    out.resize(base + n_rec);
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(host_threads(), n_rec / 8192 + 1));
    std::vector<size_t> first_bad(nt, (size_t)-1);
    std::vector<std::string> bad_msg(nt);
    auto work = [&](unsigned t) {
        for (size_t k = n_rec * t / nt; k < n_rec * (t + 1) / nt; k++) {
            SplitRead &r = out[base + k];
            const std::string l1 = line(3 * k), l3 = line(3 * k + 2);
            std::string l2 = line(3 * k + 1);
            r.Name = l1;
            while (!l2.empty() && !isalnum((unsigned char)l2[l2.size() - 1])) l2.resize(l2.size() - 1);
            r.ReadLength = (short)l2.size();
            r.UnmatchedSeq.swap(l2);
            std::istringstream iss(l3);
            iss >> r.MatchedD >> r.FragName >> r.MatchedRelPos >> r.MS >> r.InsertSize >> r.Tag;
            if (l1[0] != '@') {
                first_bad[t] = k;
                bad_msg[t] = "Something wrong with the read name: " + l1;
                return;
            }
            if (r.MatchedD != '+' && r.MatchedD != '-') {
                first_bad[t] = k;
                bad_msg[t] = "+/- expected in read " + l1;
                return;
            }
            for (size_t c = 0; c < genome.size(); c++)
                if (genome[c].name == r.FragName) r.chr_id = (int)c;
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
        for (std::thread &x : th) x.join();
    }
    for (unsigned t = 0; t < nt; t++)
        if (first_bad[t] != (size_t)-1) {
            err = bad_msg[t];
            out.resize(base + first_bad[t]);
            return -1;
        }
    return 0;
}

// ------------------------------------------------------------------ caller
Caller::Caller(const Settings &s, const std::vector<Chromosome> *g, const std::string &out_prefix,
               bool truncate_outputs)
    : S(s), genome(g), prefix(out_prefix)
{
    if (truncate_outputs) {
        const char *suffixes[] = { "_D", "_SI", "_TD", "_INV" };
        for (const char *sf : suffixes) std::ofstream((prefix + sf).c_str(), std::ios::trunc);
    }
}

Caller::~Caller() { flush_reports(); }

void Caller::update_ref_coverage(const std::vector<RefReadSpan> &reads, const std::vector<std::string> &tags,
                                 unsigned start, unsigned end)
{
    std::vector<std::string> names(g_sampleNames.begin(), g_sampleNames.end());
    cov_start_ = start;
    ref_cov_.assign(names.size(), std::vector<int>());
    std::vector<int> sample_of(tags.size(), -1);
    for (size_t t = 0; t < tags.size(); t++)
        for (size_t i = 0; i < names.size(); i++)
            if (names[i] == tags[t]) sample_of[t] = (int)i;
    const size_t length = (size_t)end - start + 1;
    // every read adds one to the positions pos + 1 .. pos + length - 2: as +1 / -1 marks and one running sum per sample
    // (the same integers as incrementing position by position)
    for (const RefReadSpan &r : reads) {
        if (r.pos < start || (unsigned)(r.pos + r.length) > end) continue;
        const int s = r.tag < sample_of.size() ? sample_of[r.tag] : -1;
        if (s < 0) continue;           // a sample without any mapped split read yet (the reference dereferences end())
        std::vector<int> &cov = ref_cov_[(size_t)s];
        if (cov.empty()) cov.assign(length, 0);
        if (r.length < 3) continue;
        cov[r.pos - start + 1]++;
        const size_t stop = (size_t)(r.pos - start) + r.length - 1;       // first position not covered any more
        if (stop < length) cov[stop]--;
    }
    for (std::vector<int> &cov : ref_cov_) {
        int run = 0;
        for (int &v : cov) {
            run += v;
            v = run;
        }
    }
}

// the buffers of the worker that formats a box (for_boxes) and the places of the event numbers in them;
// null on any other thread
struct EvMark { size_t at; int kind; };
static thread_local std::ostringstream *tl_box_out = nullptr;
static thread_local std::vector<EvMark> *tl_box_marks = nullptr;

std::ostream &operator<<(std::ostream &out, Caller::EvNo e)
{
    // An event number is only known when the boxes are written out in order: inside for_boxes the place is recorded and
    // the number put in later.  Anywhere else there is nothing to put it in later -- a report written that way would
    // silently lack its event numbers, so that is a programming error, loudly.
    bool marked = false;
    if (tl_box_out)
        for (int k = 0; k < 4; k++)
            if (&out == static_cast<std::ostream *>(&tl_box_out[k])) {
                tl_box_marks[k].push_back(EvMark{ (size_t)out.tellp(), (int)e.kind });
                marked = true;
                break;
            }
    if (!marked) {
        fprintf(stderr, "pgh: event number written outside Caller::for_boxes (reporter called on the wrong stream)\n");
        abort();
    }
    return out;
}

unsigned Caller::take_event_number(int k)
{
    switch (k) {
    case EV_D: d_template++; return (unsigned)(d_template + d_nontemplate - 1);
    case EV_D_NT: d_nontemplate++; return (unsigned)(d_template + d_nontemplate - 1);
    case EV_SI: return n_si++;
    case EV_TD: return n_td++;
    default: return n_inv++;
    }
}

void Caller::for_boxes(unsigned n_boxes, const std::function<void(unsigned)> &body)
{
    struct BoxText {
        std::string text[REP_N];
        std::vector<EvMark> marks[REP_N];
    };
    std::vector<BoxText> texts(n_boxes);
    const unsigned nt = std::max(1u, std::min(host_threads(), n_boxes / 64u + 1u));
    std::atomic<unsigned> next(0);
    auto work = [&]() {
        std::ostringstream os[REP_N];
        std::vector<EvMark> marks[REP_N];
        tl_box_out = os;
        tl_box_marks = marks;
        for (unsigned b = next.fetch_add(1); b < n_boxes; b = next.fetch_add(1)) {
            body(b);
            for (int k = 0; k < REP_N; k++)
                if (os[k].tellp() > 0) {
                    texts[b].text[k] = os[k].str();
                    texts[b].marks[k].swap(marks[k]);
                    marks[k].clear();
                    os[k].str(std::string());
                    os[k].clear();
                }
        }
        tl_box_out = nullptr;
        tl_box_marks = nullptr;
    };
    if (nt == 1) work();
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(work);
        for (std::thread &x : th) x.join();
    }
    for (unsigned b = 0; b < n_boxes; b++)
        for (int k = 0; k < REP_N; k++) {
            const std::string &t = texts[b].text[k];
            if (t.empty()) continue;
            std::ostream &out = report(k);
            size_t from = 0;
            for (const EvMark &m : texts[b].marks[k]) {
                out.write(t.data() + from, (std::streamsize)(m.at - from));
                out << take_event_number(m.kind);
                from = m.at;
            }
            out.write(t.data() + from, (std::streamsize)(t.size() - from));
        }
}

std::ostream &Caller::report(int which)
{
    if (tl_box_out) return tl_box_out[which];
    static const char *suffixes[REP_N] = { "_D", "_SI", "_TD", "_INV" };
    std::ofstream &f = rep_[which];
    if (!f.is_open()) {
        rep_buf_[which].resize(4u << 20);
        f.rdbuf()->pubsetbuf(rep_buf_[which].data(), (std::streamsize)rep_buf_[which].size());
        f.open((prefix + suffixes[which]).c_str(), std::ios::app);
    }
    return f;
}

void Caller::flush_reports()
{
    for (int i = 0; i < REP_N; i++)
        if (rep_[i].is_open()) rep_[i].flush();
}

void Caller::note_close_mapped(SplitRead &r)
{
    // ReadInRead, src/reader.cpp:262-288
    if (g_reportLength < r.getReadLength()) g_reportLength = r.getReadLength();
    r.Used = false;
    r.UniqueRead = true;
    const UniquePoint &last = r.UP_Close.back();
    short close_len = last.LengthStr;
    if (r.MatchedD == '+') r.LeftMostPos = (int)(last.AbsLoc + 1 - close_len);
    else r.LeftMostPos = (int)(last.AbsLoc + close_len - r.getReadLength());
    r.SampleName2Number.insert(std::make_pair(r.Tag, 1u));
    g_sampleNames.insert(r.Tag);
}

void Caller::note_close_mapped_all(std::vector<SplitRead> &reads)
{
    const size_t n = reads.size();
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(host_threads(), n / 16384 + 1));
    std::vector<short> max_len(nt, 0);
    std::vector<std::set<std::string>> tags(nt);
    auto work = [&](unsigned t) {
        for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) {
            SplitRead &r = reads[i];
            if (r.UP_Close.empty()) continue;
            if (max_len[t] < r.getReadLength()) max_len[t] = r.getReadLength();
            r.Used = false;
            r.UniqueRead = true;
            const UniquePoint &last = r.UP_Close.back();
            const short close_len = last.LengthStr;
            if (r.MatchedD == '+') r.LeftMostPos = (int)(last.AbsLoc + 1 - close_len);
            else r.LeftMostPos = (int)(last.AbsLoc + close_len - r.getReadLength());
            r.SampleName2Number.insert(std::make_pair(r.Tag, 1u));
            if (tags[t].empty() || !tags[t].count(r.Tag)) tags[t].insert(r.Tag);
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
        for (std::thread &x : th) x.join();
    }
    for (unsigned t = 0; t < nt; t++) {
        if (g_reportLength < max_len[t]) g_reportLength = max_len[t];
        g_sampleNames.insert(tags[t].begin(), tags[t].end());
    }
}

// GetRealStart4Insertion, src/pindel.cpp:2134-2162
static void real_start_insertion(const std::string &chr, unsigned spacer, std::string &ins, unsigned &rs,
                                 unsigned &re)
{
    if (chr.size() < rs || chr.size() < re) return;
    unsigned after = re + spacer;
    while (chr[after] == ins[0] && chr[after] != 'N') {
        ins = ins.substr(1) + ins[0];
        after++;
    }
    re = after - spacer;
    unsigned before = after - 1;
    while (chr[before] == ins[ins.size() - 1] && chr[before] != 'N') {
        ins = ins[ins.size() - 1] + ins.substr(0, ins.size() - 1);
        before--;
    }
    rs = before - spacer;
}

// ---------------------------------------------------------------------------------
// The per-sample support block shared by every event header (reporter.cpp:284-316, 352-382).
std::string Caller::support_columns(const std::vector<SplitRead> &ev, unsigned s, unsigned e,
                                    unsigned bp_left, unsigned bp_right, unsigned &n_reads)
{
    struct Sup { int p = 0, m = 0, up = 0, um = 0; };
    std::vector<std::string> names(g_sampleNames.begin(), g_sampleNames.end());
    std::map<std::string, int> index;
    for (size_t i = 0; i < names.size(); i++) index[names[i]] = (int)i;
    std::vector<Sup> sup(names.size());
    for (unsigned i = s; i <= e; i++) {
        for (const auto &kv : ev[i].SampleName2Number) {
            Sup &t = sup[index[kv.first]];
            if (ev[i].MatchedD == '+') {
                t.p += kv.second;
                if (ev[i].UniqueRead) t.up += kv.second;
            } else {
                t.m += kv.second;
                if (ev[i].UniqueRead) t.um += kv.second;
            }
        }
    }
    unsigned LeftS = 0, LeftU = 0, RightS = 0, RightU = 0;
    short nsup = 0, nusup = 0;
    int n_u = 0;
    n_reads = 0;
    for (const Sup &t : sup) {
        LeftS += t.p; LeftU += t.up; RightS += t.m; RightU += t.um;
        if (t.p + t.m) nsup++;
        if (t.up + t.um) nusup++;
        n_reads += t.p + t.m;
        n_u += t.up + t.um;
    }
    unsigned easy = (LeftS + 1) * (RightS + 1);
    int sum_ms = 0;
    for (unsigned i = s; i <= e; i++) sum_ms += ev[i].MS;
    std::ostringstream o;
    o << "\tSupports " << n_reads << "\t" << n_u << "\t+ " << LeftS << "\t" << LeftU << "\t- " << RightS
      << "\t" << RightU << "\tS1 " << easy << "\tSUM_MS " << sum_ms << "\t" << names.size()
      << "\tNumSupSamples " << nsup << "\t" << nusup;
    // reference-coverage columns (update_ref_coverage; all zero for text input); -1 outside the current bin
    const bool in_s = bp_left + 2 >= g_RegionStart && bp_left + 2 < g_RegionEnd;
    const bool in_e = bp_right > g_RegionStart && bp_right < g_RegionEnd;
    auto cov_at = [&](size_t sample, unsigned pos) -> int {      // g_RefCoverageRegion[pos - g_RegionStart]
        if (sample >= ref_cov_.size() || ref_cov_[sample].empty()) return 0;
        const size_t k = (size_t)pos - cov_start_;
        return pos >= cov_start_ && k < ref_cov_[sample].size() ? ref_cov_[sample][k] : 0;
    };
    for (size_t i = 0; i < names.size(); i++)
        o << "\t" << names[i] << " " << (in_s ? cov_at(i, bp_left + 2) : -1) << " " << (in_e ? cov_at(i, bp_right) : -1)
          << " " << sup[i].p << " " << sup[i].up << " " << sup[i].m << " " << sup[i].um;
    return o.str();
}

// OutputDeletions, src/reporter.cpp:271-444
void Caller::output_deletion(Ctx &c, std::vector<SplitRead> &g, unsigned s, unsigned e, unsigned rs, unsigned re)
{
    std::ostream &out = report(REP_D);
    const std::string &ref = c.chrom->seq;
    const SplitRead &f = g[s];
    unsigned n_reads = 0;
    std::string sup = support_columns(g, s, e, f.BPLeft, f.BPRight, n_reads);
    short gap = f.IndelSize < 14 ? (short)f.IndelSize : (short)(13 + (int)log10((double)(f.IndelSize - 10)));
    out << HASHES << '\n';
    out << ev_no(EV_D) << "\tD " << f.IndelSize << "\tNT " << f.NT_size << " \"" << f.NT_str
        << "\"\tChrID " << f.FragName << "\tBP " << f.BPLeft + 1 << "\t" << f.BPRight + 1 << "\tBP_range "
        << rs + 1 << "\t" << re + 1 << sup << '\n';
    const long rl = g_reportLength;
    out << sub(ref, (long)f.Left - rl + f.BP + 1, rl);
    if (f.IndelSize >= 14) {
        out << cap2low(sub(ref, (long)f.Left + f.BP + 1, 5)) << "<" << f.IndelSize - 10 << ">"
            << cap2low(sub(ref, (long)f.Right - f.getReadLength() + f.BP - 3, 5));
    } else {
        out << cap2low(sub(ref, (long)f.Left + f.BP + 1, gap));
    }
    out << sub(ref, (long)f.Left + f.BP + 1 + f.IndelSize, rl - gap) << '\n';
    for (unsigned i = s; i <= e; i++) {
        const SplitRead &r = g[i];
        short before = (short)(rl - r.BP - 1);
        out << std::string(before > 0 ? before : 0, ' ');
        short after = (short)(rl + rl - before - r.getReadLength());
        const std::string seq = r.MatchedD == '-' ? r.UnmatchedSeq : reverse_complement(r.UnmatchedSeq);
        out << sub(seq, 0, r.BP + 1) << std::string(gap > 0 ? gap : 0, ' ')
            << sub(seq, r.BP + 1, r.getReadLength() - r.BP);
        out << std::string(after > 0 ? after : 0, ' ') << read_tail(r) << '\n';
    }
}

// OutputDI, src/reporter.cpp:757-872
void Caller::output_di(Ctx &c, std::vector<SplitRead> &g, unsigned s, unsigned e)
{
    std::ostream &out = report(REP_D);
    const std::string &ref = c.chrom->seq;
    const SplitRead &f = g[s];
    unsigned n_reads = 0;
    std::string sup = support_columns(g, s, e, f.BPLeft, f.BPRight, n_reads);
    out << HASHES << '\n';
    out << ev_no(EV_D_NT) << "\tD " << f.IndelSize << "\tNT " << f.NT_size << " \"" << f.NT_str
        << "\"\tChrID " << f.FragName << "\tBP " << f.BPLeft + 1 << "\t" << f.BPRight + 1 << "\tBP_range "
        << f.BPLeft + 1 << "\t" << f.BPRight + 1 << sup << '\n';
    const long rl = g_reportLength;
    out << sub(ref, (long)f.Left - rl + f.BP + 1, rl) << std::string(f.NT_size, ' ')
        << sub(ref, (long)f.Left + f.BP + 1 + f.IndelSize, rl) << '\n';
    for (unsigned i = s; i <= e; i++) {
        const SplitRead &r = g[i];
        short before = (short)(rl - r.BP - 1);
        out << std::string(before > 0 ? before : 0, ' ');
        out << (r.MatchedD == '-' ? r.UnmatchedSeq : reverse_complement(r.UnmatchedSeq)) << "\t";
        out << read_tail(r) << '\n';
    }
}

// GetConsensusInsertedStr, src/reporter.cpp:2360-2382
static std::string consensus_inserted(const std::vector<SplitRead> &g, unsigned s, unsigned e)
{
    std::map<std::string, int> cnt;
    for (unsigned i = s; i <= e; i++) cnt[g[i].NT_str]++;
    int best = 0;
    std::string o;
    for (const auto &kv : cnt)
        if (kv.second > best) {
            best = kv.second;
            o = kv.first;
        }
    return o;
}

// OutputSIs, src/reporter.cpp:630-755
void Caller::output_si(Ctx &c, std::vector<SplitRead> &g, unsigned s, unsigned e, unsigned rs, unsigned re)
{
    std::ostream &out = report(REP_SI);
    const std::string &ref = c.chrom->seq;
    const SplitRead &f = g[s];
    unsigned n_reads = 0;
    std::string sup = support_columns(g, s, e, f.BPLeft, f.BPRight, n_reads);
    out << HASHES << '\n';
    out << ev_no(EV_SI) << "\tI " << f.IndelSize << "\tNT " << f.IndelSize << " \"" << consensus_inserted(g, s, e)
        << "\"\tChrID " << f.FragName << "\tBP " << f.BPLeft + 1 << "\t" << f.BPRight + 1 << "\tBP_range "
        << rs + 1 << "\t" << re + 1 << sup << '\n';
    const long rl = g_reportLength;
    out << sub(ref, (long)f.Left - rl + f.BP + 1, rl) << std::string(f.IndelSize, ' ')
        << sub(ref, (long)f.Left + f.BP + 1, rl) << '\n';
    for (unsigned i = s; i <= e; i++) {
        const SplitRead &r = g[i];
        short before = (short)(rl - r.BP - 1);
        out << std::string(before > 0 ? before : 0, ' ');
        out << (r.MatchedD == '-' ? r.UnmatchedSeq : reverse_complement(r.UnmatchedSeq));
        short after = (short)(rl + rl - before - r.getReadLength());
        out << std::string(after > 0 ? after : 0, ' ') << read_tail(r) << '\n';
    }
}

// SortOutputD, src/reporter.cpp:1395-1570
void Caller::sort_output_d(Ctx &c, std::vector<std::vector<unsigned>> &boxes)
{
    std::vector<SplitRead> &reads = *c.reads;
    const std::string &ref = c.chrom->seq;
    struct Ev { unsigned s, e, bl, br, rs, re; };
    for_boxes(c.NumBoxes, [&](unsigned b) {
        std::vector<unsigned> &box = boxes[b];
        if (box.empty() || box.size() < S.NumRead2ReportCutOff) return;
        exchange_sort(reads, box);
        mark_duplicates(reads, box);
        std::vector<SplitRead> good;
        for (unsigned i : box)
            if (reads[i].UniqueRead) good.push_back(reads[i]);
        if (good.empty()) return;
        std::vector<Ev> evs;
        Ev cur = { 0, 0, good[0].BPLeft, good[0].BPRight, 0, 0 };
        std::string cur_chr = good[0].FragName;
        auto close_event = [&]() {
            cur.rs = cur.bl;
            cur.re = cur.br;
            real_start_deletion(ref, S.spacer, cur.rs, cur.re);
            evs.push_back(cur);
        };
        for (unsigned i = 1; i < good.size(); i++) {
            if (good[i].BPLeft == cur.bl && good[i].BPRight == cur.br && good[i].FragName == cur_chr &&
                good[i].FarFragName == cur_chr) {
                cur.e = i;
            } else {
                close_event();
                cur.s = cur.e = i;
                cur.bl = good[i].BPLeft;
                cur.br = good[i].BPRight;
                cur_chr = good[i].FragName;
            }
        }
        close_event();
        // only the first event of a box carries WhetherReport = true (it is never re-set
        // when OneIndelEvent is recycled, reporter.cpp:1448, 1478-1482) -- but the copy pushed
        // for later events inherits the flag of the first, so all are reported.
        for (const Ev &ev : evs) {
            unsigned support = ev.e - ev.s + 1;
            if (support < S.NumRead2ReportCutOff) continue;
            if (good[ev.s].IndelSize < S.BalanceCutoff || report_event(good, ev.s, ev.e)) {
                output_deletion(c, good, ev.s, ev.e, ev.rs, ev.re);
            }
        }
    });
}

// SortOutputDI, src/reporter.cpp:1709-1851
static bool is_inversion(const SplitRead &r, const std::string &ref, unsigned spacer)   // reporter.cpp:1698
{
    if (r.IndelSize == r.NT_size) {
        std::string replaced = sub(ref, (long)spacer + 1 + r.BPLeft, r.NT_size);
        if (reverse_complement(replaced) == r.NT_str) return true;
    }
    return false;
}

void Caller::sort_output_di(Ctx &c, std::vector<std::vector<unsigned>> &boxes)
{
    std::vector<SplitRead> &reads = *c.reads;
    const std::string &ref = c.chrom->seq;
    struct Ev { unsigned s, e, bl, isz; short nt; };
    for_boxes(c.NumBoxes, [&](unsigned b) {
        std::vector<unsigned> &box = boxes[b];
        if (box.empty() || box.size() < S.NumRead2ReportCutOff) return;
        const size_t n = box.size();
        // the reference's own exchange sort for DI (reporter.cpp:1732-1768)
        for (size_t a = 0; a + 1 < n; a++)
            for (size_t d = a + 1; d < n; d++) {
                const SplitRead &x = reads[box[a]], &y = reads[box[d]];
                bool swap = false;
                if (x.BPLeft < y.BPLeft) continue;
                else if (x.BPLeft > y.BPLeft) swap = true;
                else {
                    if (x.BPRight < y.BPRight) continue;
                    else if (x.BPRight > y.BPRight) swap = true;
                    else {
                        if (x.NT_size < y.NT_size) continue;
                        else if (x.NT_size > y.NT_size) swap = true;
                        else if (x.BP > y.BP) swap = true;
                    }
                }
                if (swap) std::swap(box[a], box[d]);
            }
        for (size_t a = 0; a + 1 < n; a++)
            for (size_t d = a + 1; d < n; d++) {
                const SplitRead &x = reads[box[a]];
                SplitRead &y = reads[box[d]];
                if (x.getReadLength() == y.getReadLength() &&
                    (x.LeftMostPos == y.LeftMostPos ||
                     x.LeftMostPos + x.getReadLength() == y.LeftMostPos + y.getReadLength()) &&
                    x.MatchedD == y.MatchedD)
                    y.UniqueRead = false;
            }
        std::vector<SplitRead> good;
        for (unsigned i : box)
            if (reads[i].UniqueRead) good.push_back(reads[i]);
        if (good.empty()) return;
        std::vector<Ev> evs;
        Ev cur = { 0, 0, good[0].BPLeft, good[0].IndelSize, (short)good[0].NT_size };
        for (unsigned i = 1; i < good.size(); i++) {
            if (good[i].BPLeft == cur.bl && good[i].IndelSize == cur.isz && (short)good[i].NT_size == cur.nt) {
                cur.e = i;
            } else {
                evs.push_back(cur);
                cur.s = cur.e = i;
                cur.bl = good[i].BPLeft;
                cur.isz = good[i].IndelSize;
                cur.nt = (short)good[i].NT_size;
            }
        }
        evs.push_back(cur);
        for (const Ev &ev : evs) {
            if (ev.e - ev.s + 1 < S.NumRead2ReportCutOff) continue;
            if (good[ev.s].IndelSize < S.BalanceCutoff || report_event(good, ev.s, ev.e)) {
                if (is_inversion(good[ev.s], ref, S.spacer)) {
                    output_short_inv(c, good, ev.s, ev.e);
                } else {
                    output_di(c, good, ev.s, ev.e);
                }
            }
        }
    });
}

// SortOutputSI, src/reporter.cpp:975-1091
void Caller::sort_output_si(Ctx &c, std::vector<std::vector<unsigned>> &boxes)
{
    std::vector<SplitRead> &reads = *c.reads;
    const std::string &ref = c.chrom->seq;
    struct Ev { unsigned s, e, bl, br, isz, rs, re; std::string str; };
    for_boxes(c.NumBoxes, [&](unsigned b) {
        std::vector<unsigned> &box = boxes[b];
        if (box.empty() || box.size() < S.NumRead2ReportCutOff) return;
        exchange_sort(reads, box);
        mark_duplicates(reads, box);
        std::vector<SplitRead> good;
        for (unsigned i : box)
            if (reads[i].UniqueRead) good.push_back(reads[i]);
        if (good.empty()) return;
        std::vector<Ev> evs;
        auto init = [&](unsigned i) {
            Ev e;
            e.s = e.e = i;
            e.bl = good[i].BPLeft;
            e.br = good[i].BPRight;
            e.isz = good[i].IndelSize;
            e.str = good[i].NT_str;
            e.rs = e.re = 0;
            return e;
        };
        Ev cur = init(0);
        auto complete = [&]() {
            cur.rs = cur.bl;
            cur.re = cur.br;
            real_start_insertion(ref, S.spacer, cur.str, cur.rs, cur.re);
            evs.push_back(cur);
        };
        for (unsigned i = 1; i < good.size(); i++) {
            if (good[i].BPLeft == cur.bl && good[i].IndelSize == cur.isz) cur.e = i;
            else {
                complete();
                cur = init(i);
            }
        }
        complete();
        for (const Ev &ev : evs) {
            unsigned short support = (unsigned short)(ev.e - ev.s + 1);
            if (support >= S.NumRead2ReportCutOff && ev.rs < ev.re) output_si(c, good, ev.s, ev.e, ev.rs, ev.re);
        }
    });
}

// ---------------------------------------------------------------------------------
// SearchVariant::Search, src/search_variant.cpp:48-266.  kind 0 = deletions
// (searchdeletions.cpp:38-64), kind 1 = short insertions (searchshortinsertions.cpp:38-65).
void Caller::search_variant(Ctx &c, int kind)
{
    std::vector<SplitRead> &reads = *c.reads;
    const std::string &ref = c.chrom->seq;
    std::vector<std::vector<unsigned>> boxes(c.NumBoxes);
    // "Checksum of far ends" (search_variant.cpp:53-71)
    unsigned bp_sum = 0;
    for (const SplitRead &r : reads) {
        if (!r.UP_Far.empty()) bp_sum += r.UP_Far.back().AbsLoc;
        if (bp_sum > 1000000000u) bp_sum -= 1000000000u;
    }
    far_end_checksum = bp_sum;
    if (S.log_counts) {
        // the reference's own cross-check lines (search_variant.cpp:67-69): a maintainer can diff them
        unsigned used = 0, far = 0;
        for (const SplitRead &r : reads) {
            used += r.Used ? 1u : 0u;
            far += r.UP_Far.empty() ? 0u : 1u;
        }
        printf("Reads already used: %u\nFar ends already mapped %u\nChecksum of far ends: %u\n", used, far, bp_sum);
    }

    classify_reads(reads.size(), boxes, [&](unsigned lo_, unsigned hi_, BoxSink &sink) {
    for (unsigned ri = lo_; ri < hi_; ri++) {
        SplitRead &r = reads[ri];
        if (r.FragName != r.FarFragName) continue;
        if (r.Used || r.UP_Far.empty()) continue;
        const bool plus = r.MatchedD == '+';
        if (!plus && r.MatchedD != '-') continue;
        const FarByLength by_len(r);
        const FarByLoc by_loc(r);
        for (short budget = 0; budget <= r.MAX_SNP_ERROR && !r.Used; budget++) {
            const int nc = (int)r.UP_Close.size();
            for (int k = 0; k < nc && !r.Used; k++) {
                const int ci = plus ? k : nc - 1 - k;
                const UniquePoint &cp = r.UP_Close[ci];
                if (cp.Mismatches > budget) continue;
                // candidates among the far points, visited from the highest index down like the full loop:
                // deletions: the one of length ReadLength - LengthStr(close); short insertions: the ones
                // at AbsLoc(close) + 1 ('+' anchor) / AbsLoc(close) - 1 ('-' anchor)
                int fi_first = (int)r.UP_Far.size() - 1, fi_last = 0;
                size_t lb = 0, le = 0;
                if (kind == 0) by_len.range_desc(r.getReadLength(), cp.LengthStr, (int)r.UP_Far.size(), fi_first, fi_last);
                else by_loc.range(plus ? cp.AbsLoc + 1 : cp.AbsLoc - 1, lb, le);
                for (size_t step_ = 0;; step_++) {
                    int fi;
                    if (kind == 0) {
                        fi = fi_first - (int)step_;
                        if (fi < fi_last) break;
                    } else {
                        if (lb + step_ >= le) break;
                        fi = -by_loc.v[lb + step_].second;
                    }
                    if (r.Used) break;
                    const UniquePoint &fp = r.UP_Far[fi];
                    if (fp.Mismatches > budget) continue;
                    if (fp.Mismatches + cp.Mismatches > budget) continue;
                    if (fp.Direction != (plus ? '-' : '+')) continue;
                    bool ok;
                    if (kind == 0) {
                        ok = plus ? (fp.LengthStr + cp.LengthStr == r.getReadLength() && fp.AbsLoc > cp.AbsLoc + 1)
                                  : (cp.LengthStr + fp.LengthStr == r.getReadLength() && cp.AbsLoc > fp.AbsLoc + 1);
                    } else {
                        ok = plus ? (fp.AbsLoc == cp.AbsLoc + 1 && cp.LengthStr + fp.LengthStr < r.getReadLength())
                                  : (cp.AbsLoc == fp.AbsLoc + 1 && fp.LengthStr + cp.LengthStr < r.getReadLength());
                    }
                    if (!ok) continue;
                    if (plus) {
                        r.Left = (int)(cp.AbsLoc - cp.LengthStr + 1);
                        r.Right = (int)(fp.AbsLoc + fp.LengthStr - 1);
                        r.BP = (short)(cp.LengthStr - 1);
                    } else {
                        r.Left = (int)(fp.AbsLoc - fp.LengthStr + 1);
                        r.Right = (int)(cp.AbsLoc + cp.LengthStr - 1);
                        r.BP = (short)(fp.LengthStr - 1);
                    }
                    if (kind == 0) {
                        r.IndelSize = (unsigned)((r.Right - r.Left) - r.getReadLengthMinus());
                        r.NT_str = "";
                    } else {
                        r.IndelSize = (unsigned)(r.getReadLengthMinus() - (r.Right - r.Left));
                        r.NT_str = plus ? sub(reverse_complement(r.UnmatchedSeq), r.BP + 1, r.IndelSize)
                                        : sub(r.UnmatchedSeq, r.BP + 1, r.IndelSize);
                    }
                    if (plus) {
                        r.BPLeft = cp.AbsLoc - S.spacer;
                        r.BPRight = fp.AbsLoc - S.spacer;
                    } else {
                        r.BPLeft = fp.AbsLoc - S.spacer;
                        r.BPRight = cp.AbsLoc - S.spacer;
                    }
                    unsigned real_l = r.BPLeft, real_r = r.BPRight;
                    if (ref.size() < real_l || ref.size() < real_r) {
                        r.Used = true;
                        break;
                    }
                    if (!r.NT_str.empty()) real_start_insertion(ref, S.spacer, r.NT_str, real_l, real_r);
                    else real_start_deletion(ref, S.spacer, real_l, real_r);
                    short diff = (short)(r.BPLeft - real_l);
                    diff = !((r.BP - 1) < diff) ? diff : (short)(r.BP - 1);
                    if (diff > 0) {
                        r.BP -= diff;
                        r.BPLeft -= diff;
                        r.BPRight -= diff;
                    }
                    if (transgresses(r, c.win_end)) {
                        r.Used = true;      // saveReadForNextCycle: the copy is cleared before reuse
                    } else if (r.BPLeft + 1 >= c.region_start && r.BPLeft + 1 <= c.region_end) {
                        unsigned box = (unsigned)((int)r.BPLeft / (int)BoxSize);
                        if (box < c.NumBoxes) {
                            sink[box].push_back(ri);
                            r.Used = true;
                        }
                    }
                }
            }
        }
    }
    });
    if (kind == 0) sort_output_d(c, boxes);
    else sort_output_si(c, boxes);
}

// searchIndels, src/search_deletions_nt.cpp:26-140
void Caller::search_indels(Ctx &c)
{
    std::vector<SplitRead> &reads = *c.reads;
    std::vector<std::vector<unsigned>> boxes(c.NumBoxes);
    classify_reads(reads.size(), boxes, [&](unsigned lo_, unsigned hi_, BoxSink &sink) {
    for (unsigned ri = lo_; ri < hi_; ri++) {
        SplitRead &r = reads[ri];
        if (r.Used || r.UP_Far.empty() || r.FragName != r.FarFragName) continue;
        const UniquePoint &cp = r.UP_Close.back();
        const UniquePoint &fp = r.UP_Far.back();
        if (fp.Mismatches + cp.Mismatches > (short)(1 + S.Seq_Error_Rate * (fp.LengthStr + cp.LengthStr))) continue;
        const bool plus = r.MatchedD == '+';
        if (!plus && r.MatchedD != '-') continue;
        if (fp.Direction != (plus ? '-' : '+')) continue;
        if (!(fp.LengthStr + cp.LengthStr < r.getReadLength() &&
              fp.LengthStr + cp.LengthStr >= S.Min_Num_Matched_Bases))
            continue;
        if (plus) {
            if (!(fp.AbsLoc > cp.AbsLoc + 1)) continue;
            r.Left = (int)(cp.AbsLoc - cp.LengthStr + 1);
            r.Right = (int)(fp.AbsLoc + fp.LengthStr - 1);
            r.BP = (short)(cp.LengthStr - 1);
            r.NT_size = (unsigned short)(r.getReadLength() - fp.LengthStr - cp.LengthStr);
            r.NT_str = sub(reverse_complement(r.UnmatchedSeq), r.BP + 1, r.NT_size);
            r.IndelSize = (unsigned)((r.Right - r.Left) + r.NT_size - r.getReadLengthMinus());
            r.BPLeft = cp.AbsLoc - S.spacer;
            r.BPRight = fp.AbsLoc - S.spacer;
        } else {
            if (!(cp.AbsLoc > fp.AbsLoc + 1)) continue;
            r.Left = (int)(fp.AbsLoc - fp.LengthStr + 1);
            r.Right = (int)(cp.AbsLoc + cp.LengthStr - 1);
            r.BP = (short)(fp.LengthStr - 1);
            r.NT_size = (unsigned short)(r.getReadLength() - cp.LengthStr - fp.LengthStr);
            r.NT_str = sub(r.UnmatchedSeq, r.BP + 1, r.NT_size);
            r.IndelSize = (unsigned)((r.Right - r.Left) - r.getReadLengthMinus() + r.NT_size);
            r.BPLeft = fp.AbsLoc - S.spacer;
            r.BPRight = cp.AbsLoc - S.spacer;
        }
        if (transgresses(r, c.win_end)) {
            r.Used = true;
        } else if (r.BPLeft + 1 >= c.region_start && r.BPLeft + 1 <= c.region_end) {
            unsigned box = (unsigned)((int)r.BPLeft / (int)BoxSize);
            if (box < c.NumBoxes) {
                sink[box].push_back(ri);
                r.Used = true;
            }
        }
    }
    });
    sort_output_di(c, boxes);
}

// main's per-bin sequence after the far-end search: UpdateFarFragName (pindel.cpp:1262-1270)
// and SearchSVs (pindel.cpp:1141-1175).
void Caller::process_window(const Chromosome &chrom, std::vector<SplitRead> &reads, unsigned win_start,
                            unsigned win_end, unsigned region_start, unsigned region_end)
{
    Ctx c;
    c.chrom = &chrom;
    c.reads = &reads;
    // BoxSize / NumBoxes, pindel.cpp:1806-1810
    BoxSize = (unsigned)(chrom.seq.size() / 30000);
    if (BoxSize == 0) BoxSize = 1;
    c.NumBoxes = (unsigned)(chrom.seq.size() * 2 / BoxSize) + 1;
    c.win_end = win_end;
    c.region_start = region_start;
    c.region_end = region_end;
    g_RegionStart = win_start;
    g_RegionEnd = win_end;
    for (SplitRead &r : reads) {
        if (!r.UP_Far.empty()) {
            int fc = r.UP_Far[0].chr;        // UpdateFarFragName, pindel.cpp:1262-1270
            r.FarFragName = (genome && fc >= 0 && fc < (int)genome->size()) ? (*genome)[fc].name : chrom.name;
            r.MatchedFarD = r.UP_Far[0].Strand;
        }
    }
    // PGH_TIMING=1: wall-clock seconds per classifier on stderr (diagnostics)
    static const bool timing = getenv("PGH_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t = now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const double t1 = now();
        fprintf(stderr, "pgh timing: %-18s %.3f s (%zu reads)\n", what, t1 - t, reads.size());
        t = t1;
    };
    search_variant(c, 0);
    lap("deletions");
    search_indels(c);
    lap("indels (DI)");
    if (S.Analyze_TD) {
        search_tandem_dup(c);
        lap("tandem dup");
        search_tandem_dup_nt(c);
        lap("tandem dup NT");
    }
    if (S.Analyze_INV) {
        search_inversions(c);
        lap("inversions");
        search_inversions_nt(c);
        lap("inversions NT");
    }
    search_variant(c, 1);
    lap("short insertions");
    flush_reports();
    lap("flush");
}

}  // namespace pgh

// pg_host.hpp -- C++ host side around the split-read search: the reference's data
// model (SPLIT_READ, UniquePoint, Chromosome), its text/FASTA loaders, the SV
// classifiers and the text reporters that consume UP_Close / UP_Far.
//
// This mirrors the reference's interface for the steps either side of the hot path
//  so that breakpoint calls can be compared byte for
// byte with the reference's golden files:
//   loaders      Genome::loadChromosome src/pindel.cpp:272-312, PindelReadReader
//                src/pindel_read_reader.cpp:53-66, ReadInRead src/reader.cpp:196-361
//   classifiers  SearchVariant::Search src/search_variant.cpp:48-266 (+ searchdeletions.cpp,
//                searchshortinsertions.cpp), searchIndels src/search_deletions_nt.cpp:26-140
//   reporters    SortOutputD / OutputDeletions, SortOutputDI / OutputDI, SortOutputSI /
//                OutputSIs  src/reporter.cpp
// The search itself is NOT here: UP_Close / UP_Far come from the GPU through the C ABI
// (include/pindel_pg.h).  No HIP dependency in this file; plain g++.
#ifndef PG_HOST_HPP
#define PG_HOST_HPP

#include <cstdint>
#include <map>
#include <set>
#include <fstream>
#include <functional>
#include <string>
#include <vector>

namespace pgh {

struct UniquePoint {              // src/pindel.h:137-158
    int chr = -1;
    short LengthStr = 0;
    unsigned AbsLoc = 0;
    char Direction = 'N';         // '+' FORWARD, '-' BACKWARD
    char Strand = 'N';            // '+' SENSE, '-' ANTISENSE
    short Mismatches = 0;
};

struct SplitRead {                // the SPLIT_READ fields used downstream, src/pindel.h:265-383
    std::string Name, UnmatchedSeq, FragName, FarFragName, Tag, NT_str;
    char MatchedD = 0, MatchedFarD = 0;
    unsigned MatchedRelPos = 0;
    short MS = 0, InsertSize = 0;
    short ReadLength = 0, MAX_SNP_ERROR = 0;
    std::vector<UniquePoint> UP_Close, UP_Far;
    short BP = 0;
    int Left = 0, Right = 0;
    unsigned BPLeft = 0, BPRight = 0, IndelSize = 0;
    unsigned short NT_size = 0;
    std::string NT_str_2;         // inversions: second non-template string
    unsigned short NT_size_2 = 0;
    bool UniqueRead = false, Used = false;
    int LeftMostPos = 0;
    int chr_id = -1;
    std::map<std::string, unsigned> SampleName2Number;
    short getReadLength() const { return ReadLength; }
    short getReadLengthMinus() const { return (short)(ReadLength - 1); }
};

struct Chromosome {
    std::string name;
    std::string seq;              // spacer + sequence + spacer
};

struct Settings {                 // the flags the downstream steps read (src/fn_parameters.cpp)
    unsigned spacer = 100000;
    unsigned NumRead2ReportCutOff = 1;   // -M
    unsigned BalanceCutoff = 100;        // -B
    double Seq_Error_Rate = 0.01;        // -e
    int Min_Num_Matched_Bases = 30;      // -d
    int MIN_IndelSize_Inversion = 50;    // -v
    bool Analyze_TD = true, Analyze_INV = true;   // -t, -r
    double window_mbp = 5.0;             // -w
    unsigned max_mismatch[500] = {0};    // g_maxMismatch
    bool log_counts = false;             // print the reference's cross-check lines (far-end counts and checksum)
};

int load_fasta(const std::string &path, std::vector<Chromosome> &out, unsigned spacer, std::string &err);

// Pindel-text reads (3 lines per read).  Trailing non-alphanumerics of SEQ are stripped
// (setUnmatchedSeq).  Reads on unknown chromosomes are kept with chr_id = -1.
int load_pindel_text(const std::string &path, const std::vector<Chromosome> &genome,
                     std::vector<SplitRead> &out, std::string &err);

std::string reverse_complement(const std::string &s);

// Everything that happens to the reads of ONE chromosome after the close-end stage, with the
// reference's global counters (SV indices, g_reportLength, g_sampleNames) kept across calls.
class Caller {
public:
    Caller(const Settings &s, const std::vector<Chromosome> *genome, const std::string &out_prefix,
           bool truncate_outputs);
    // reads: reads anchored on `chrom` in input order, each with UP_Close/UP_Far filled and
    // UnmatchedSeq in the orientation GetCloseEnd left it.  Reads without a close end must
    // already have been dropped (ReadInRead / ReadBuffer::flush do that).
    void process_window(const Chromosome &chrom, std::vector<SplitRead> &reads, unsigned win_start,
                        unsigned win_end, unsigned region_start, unsigned region_end);
    // post-close-end bookkeeping of ReadInRead (reader.cpp:258-291): CloseEndLength, LeftMostPos,
    // g_reportLength, sample names.  Call once per read that has a close end.
    void note_close_mapped(SplitRead &r);
    // ... for every read of `reads` that has a close end, on a few threads
    void note_close_mapped_all(std::vector<SplitRead> &reads);
    unsigned long far_end_checksum = 0;
    // UpdateRefReadCoverage (pindel.cpp:1272-1330), BAM input: per sample (in the order of the sample-name set as
    // it stands now) the number of reference-supporting reads over every position of the window [start, end];
    // a read counts from its second to its last-but-one base and only if it lies inside the window.  The two
    // coverage integers per sample of every report header come from here (0 0 without it, as for text input).
    struct RefReadSpan { uint32_t pos; uint16_t length; uint16_t tag; };
    void update_ref_coverage(const std::vector<RefReadSpan> &reads, const std::vector<std::string> &tags,
                             unsigned start, unsigned end);
    ~Caller();

private:
    // The four report files, opened once (append) with a large buffer and flushed at the end of every window.
    // (The reference re-opens the file for every event and flushes every line; the bytes are the same.)
    enum { REP_D = 0, REP_SI, REP_TD, REP_INV, REP_N };
    std::ofstream rep_[REP_N];
    std::vector<char> rep_buf_[REP_N];
    std::ostream &report(int which);        // the file -- or, inside for_boxes, the calling worker's buffer
    void flush_reports();
    // The event number at the head of a report entry.  Entries are formatted box by box in parallel (for_boxes),
    // so `out << ev_no(kind)` writes nothing and only marks the place; the number -- the running count of that kind
    // (D entries print template + non-template deletions so far) -- is put in when the boxes' texts are written in
    // box order.
    enum EvKind { EV_D = 0, EV_D_NT, EV_SI, EV_TD, EV_INV, EV_N };
    struct EvNo { EvKind kind; };
    static EvNo ev_no(EvKind k) { EvNo e = { k }; return e; }
    friend std::ostream &operator<<(std::ostream &out, EvNo e);
    unsigned take_event_number(int k);
    // body(b) for every box, on a few threads; what it wrote through report() lands in the files in box order
    void for_boxes(unsigned n_boxes, const std::function<void(unsigned)> &body);
    Settings S;
    const std::vector<Chromosome> *genome;
    std::string prefix;
    short g_reportLength = 1;
    std::set<std::string> g_sampleNames;
    int d_template = 0, d_nontemplate = 0;   // deletionFileData
    unsigned n_si = 0, n_td = 0, n_inv = 0;
    unsigned BoxSize = 1;
    unsigned g_RegionStart = 0, g_RegionEnd = 0;
    std::vector<std::vector<int>> ref_cov_;   // [sample][position - cov_start_]
    unsigned cov_start_ = 0;

    struct Ctx;
    void search_variant(Ctx &c, int kind);
    void search_indels(Ctx &c);
    void search_tandem_dup(Ctx &c);
    void search_tandem_dup_nt(Ctx &c);
    void search_inversions(Ctx &c);
    void search_inversions_nt(Ctx &c);
    void sort_output_d(Ctx &c, std::vector<std::vector<unsigned>> &boxes);
    void sort_output_di(Ctx &c, std::vector<std::vector<unsigned>> &boxes);
    void sort_output_si(Ctx &c, std::vector<std::vector<unsigned>> &boxes);
    void sort_output_td(Ctx &c, std::vector<std::vector<unsigned>> &boxes, bool nt);
    void sort_output_inv(Ctx &c, std::vector<std::vector<unsigned>> &boxes, bool nt);
    std::string support_columns(const std::vector<SplitRead> &ev, unsigned s, unsigned e,
                                unsigned bp_left, unsigned bp_right, unsigned &n_reads);
    void output_deletion(Ctx &c, std::vector<SplitRead> &g, unsigned s, unsigned e, unsigned rs, unsigned re);
    void output_di(Ctx &c, std::vector<SplitRead> &g, unsigned s, unsigned e);
    void output_si(Ctx &c, std::vector<SplitRead> &g, unsigned s, unsigned e, unsigned rs, unsigned re);
    void output_td(Ctx &c, std::vector<SplitRead> &g, unsigned s, unsigned e, unsigned rs, unsigned re);
    void output_inv(Ctx &c, std::vector<SplitRead> &g, unsigned s, unsigned e, unsigned rs, unsigned re);
    void output_short_inv(Ctx &c, std::vector<SplitRead> &g, unsigned s, unsigned e);
};

}  // namespace pgh
#endif

// pg_host_priv.hpp -- helpers shared by pg_host.cpp and pg_host_sv.cpp (not installed).
#ifndef PG_HOST_PRIV_HPP
#define PG_HOST_PRIV_HPP

#include <sstream>
#include <string>
#include <algorithm>
#include <functional>
#include <unordered_set>
#include <utility>
#include <cstdlib>
#include <thread>
#include <utility>
#include <vector>

#include "pg_host.hpp"

namespace pgh {

struct Caller::Ctx {
    const Chromosome *chrom;
    std::vector<SplitRead> *reads;
    unsigned NumBoxes;
    unsigned win_end;
    unsigned region_start, region_end;
};

namespace detail {

inline char rc4n(char c)   // Convert2RC4N, src/pindel.cpp:966-970 (unset entries are 0)
{
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    case 'N': return 'N';
    default: return 0;
    }
}

inline std::string cap2low(const std::string &s)        // Cap2LowArray, src/pindel.cpp:971-976
{
    std::string o(s.size(), 0);
    for (size_t i = 0; i < s.size(); i++) {
        switch (s[i]) {
        case 'A': o[i] = 'a'; break;
        case 'C': o[i] = 'c'; break;
        case 'G': o[i] = 'g'; break;
        case 'T': o[i] = 't'; break;
        case 'N': case '$': o[i] = 'n'; break;
        default: o[i] = 0; break;
        }
    }
    return o;
}

// std::string::substr that never throws (the reference only calls it in range)
inline std::string sub(const std::string &s, long pos, long n)
{
    if (pos < 0 || (size_t)pos > s.size() || n <= 0) return std::string();
    return s.substr((size_t)pos, (size_t)n);
}

// readTransgressesBinBoundaries, src/pindel.cpp:561-564
inline bool transgresses(const SplitRead &r, unsigned upper) { return r.BPRight > upper - 2 * r.InsertSize; }

static const char *const HASHES =
    "####################################################################################################";
static const char *const DASHES =
    "----------------------------------------------------------------------------------------------------";

inline std::string read_tail(const SplitRead &r)
{
    std::ostringstream o;
    o << "\t" << r.MatchedD << "\t" << r.MatchedRelPos << "\t" << r.MS << "\t" << r.Tag << "\t" << r.Name;
    return o.str();
}

// GetRealStart4Deletion, src/pindel.cpp:2095-2118
inline void real_start_deletion(const std::string &chr, unsigned spacer, unsigned &rs, unsigned &re)
{
    if (chr.size() < rs || chr.size() < re) return;
    unsigned pos = rs + spacer, start = pos + 1, end = re + spacer - 1;
    while (chr[pos] == chr[end] && chr[pos] != 'N') {
        --pos;
        --end;
    }
    rs = pos - spacer;
    pos = re + spacer;
    while (chr[pos] == chr[start] && chr[pos] != 'N') {
        ++pos;
        ++start;
    }
    re = pos - spacer;
}

// ReportEvent, src/pindel.cpp:2059-2093 (Min_Filter_Ratio = 0.5)
inline bool report_event(const std::vector<SplitRead> &g, unsigned s, unsigned e)
{
    bool lmin = false, lmax = false, rmin = false, rmax = false;
    for (unsigned i = s; i <= e; i++) {
        short rl = (short)(g[i].getReadLength() - g[i].NT_size);
        short mn = (short)((short)((rl * 0.5) + 0.5) - 1);
        short mx = (short)((short)(rl * (1 - 0.5) - 0.5) - 1);
        if (g[i].BP <= mn) lmin = true;
        if (g[i].getReadLength() - g[i].BP - g[i].NT_size <= mn) rmin = true;
        if (g[i].BP >= mx) lmax = true;
        if (g[i].getReadLength() - g[i].BP - g[i].NT_size >= mx) rmax = true;
    }
    return lmin && lmax && rmin && rmax;
}

// smaller(), src/reporter.cpp:908-929 (all reads of one window share FragName)
inline bool smaller(const SplitRead &a, const SplitRead &b)
{
    if (a.BPLeft != b.BPLeft) return a.BPLeft < b.BPLeft;
    if (a.BPRight != b.BPRight) return a.BPRight < b.BPRight;
    if (a.IndelSize != b.IndelSize) return a.IndelSize < b.IndelSize;
    if (a.NT_size != b.NT_size) return a.NT_size < b.NT_size;
    if (a.BP != b.BP) return a.BP < b.BP;
    return false;
}

// The classifiers pair every close-end point with every far-end point (budget x |UP_Close| x |UP_Far|
// iterations per read, search_variant.cpp:104-240 and friends) although most of them test
// "LengthStr(close) + LengthStr(far) == ReadLength" first.  UP_Far holds at most one point per length, in
// increasing length (it is one evaluation of the pattern growth), so the only far point that can pass
// that test is found by length; the loop body then runs for that index alone -- same first hit, same
// outcome.  If a caller hands in a list that is not strictly increasing the full loop is used.
struct FarByLength {
    short idx[512];
    bool usable;
    explicit FarByLength(const SplitRead &r) : usable(true)
    {
        for (int i = 0; i < 512; i++) idx[i] = -1;
        int prev = -1;
        const int nf = (int)r.UP_Far.size();
        for (int j = 0; j < nf; j++) {
            const int L = r.UP_Far[j].LengthStr;
            if (L <= prev || L < 0 || L >= 512 || nf > 32000) {
                usable = false;
                return;
            }
            idx[L] = (short)j;
            prev = L;
        }
    }
    // [first, last] = the far indices (descending walk first >= last) that can pair with a close point of
    // length close_len in a read of length read_len; empty when first < last
    void range_desc(int read_len, int close_len, int nf, int &first, int &last) const
    {
        if (!usable) {
            first = nf - 1;
            last = 0;
            return;
        }
        const int L = read_len - close_len;
        const int j = (L >= 0 && L < 512) ? idx[L] : -1;
        first = j;
        last = j < 0 ? 0 : j;
    }
};

// The same idea for the tests on positions (short insertions: "far AbsLoc == close AbsLoc + 1"): the far
// points sorted by (AbsLoc ascending, index descending); the entries of one AbsLoc are then exactly the
// points the descending loop over all far points would have tested successfully, in the same order.
struct FarByLoc {
    std::vector<std::pair<unsigned, int>> v;
    explicit FarByLoc(const SplitRead &r)
    {
        v.reserve(r.UP_Far.size());
        for (int j = 0; j < (int)r.UP_Far.size(); j++) v.push_back(std::make_pair(r.UP_Far[j].AbsLoc, -j));
        std::sort(v.begin(), v.end());
    }
    // [b, e): entries with the given AbsLoc; far index = -v[k].second, descending as k grows
    void range(unsigned loc, size_t &b, size_t &e) const
    {
        b = std::lower_bound(v.begin(), v.end(), std::make_pair(loc, -0x7fffffff)) - v.begin();
        e = b;
        while (e < v.size() && v[e].first == loc) e++;
    }
};

// bubblesortReads, src/reporter.cpp:932-942: an exchange sort that also swaps EQUAL elements,
//     for a < b: if (!smaller(x[a], x[b])) swap(x[a], x[b])
// Its (unstable-looking) order of equal reads is visible in the reports, so it has to be reproduced
// exactly -- but not in O(n^2): that loop leaves the elements sorted by key with equal elements in the
// REVERSE of their original order, i.e. it equals a stable sort of the reversed sequence.  (Each pass
// moves the last of the minimal elements to the front and shifts the others of its class one place
// down the chain; tests/test_cpu_suite.py checks the two against each other on random inputs.)
template <class Less>
inline void exchange_sort_reference(std::vector<unsigned> &idx, Less less)      // the O(n^2) original
{
    const size_t n = idx.size();
    for (size_t a = 0; a + 1 < n; a++)
        for (size_t b = a + 1; b < n; b++)
            if (!less(idx[a], idx[b])) std::swap(idx[a], idx[b]);
}
template <class Less>
inline void exchange_sort_fast(std::vector<unsigned> &idx, Less less)
{
    std::reverse(idx.begin(), idx.end());
    std::stable_sort(idx.begin(), idx.end(), less);
}
inline void exchange_sort(const std::vector<SplitRead> &reads, std::vector<unsigned> &idx)
{
    exchange_sort_fast(idx, [&](unsigned a, unsigned b) { return smaller(reads[a], reads[b]); });
}

// markDuplicates, src/reporter.cpp:946-972: walking the sorted box, a read that is still unique makes
// every LATER read with the same (Left, Right, Name) non-unique.  So within a (Left, Right, Name)
// class everything behind its first unique member is cleared: one hash lookup per read.
inline void mark_duplicates(std::vector<SplitRead> &reads, const std::vector<unsigned> &idx)
{
    struct Key {
        int left, right;
        const std::string *name;
        bool operator==(const Key &o) const { return left == o.left && right == o.right && *name == *o.name; }
    };
    struct Hash {
        size_t operator()(const Key &k) const
        {
            return std::hash<std::string>()(*k.name) ^ ((size_t)(unsigned)k.left * 0x9e3779b97f4a7c15ull) ^ ((size_t)(unsigned)k.right << 21);
        }
    };
    std::unordered_set<Key, Hash> seen_unique;
    seen_unique.reserve(idx.size() * 2);
    for (unsigned i : idx) {
        SplitRead &x = reads[i];
        const Key k = { x.Left, x.Right, &x.Name };
        if (seen_unique.count(k)) x.UniqueRead = false;
        else if (x.UniqueRead) seen_unique.insert(k);
    }
}

}  // namespace detail

// The classifiers' per-read loops on a few threads: body(lo, hi, sink) runs the loop over reads [lo, hi) and queues
// reads for boxes through sink[box].push_back(ri); afterwards the boxes receive them range by range, i.e. in ascending
// read index like the sequential loop would have pushed them.  (Every iteration touches its own read only.)
// threads for the classifiers and reporters: min(hardware, 16), or PGH_THREADS when set (1 = sequential)
inline unsigned host_threads()
{
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    unsigned v = std::min(hw, 16u);
    if (const char *e = getenv("PGH_THREADS")) {
        const int k = atoi(e);
        if (k >= 1) v = (unsigned)std::min(k, 64);
    }
    return v;
}

struct BoxSink {
    std::vector<std::pair<unsigned, unsigned>> v;
    struct Ref {
        BoxSink &s;
        unsigned b;
        void push_back(unsigned ri) { s.v.emplace_back(b, ri); }
    };
    Ref operator[](unsigned b) { return Ref{ *this, b }; }
};
template <class Body>
inline void classify_reads(size_t n, std::vector<std::vector<unsigned>> &boxes, Body body)
{
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(host_threads(), n / 8192 + 1));
    std::vector<BoxSink> sinks(nt);
    if (nt == 1) body(0u, (unsigned)n, sinks[0]);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() { body((unsigned)(n * t / nt), (unsigned)(n * (t + 1) / nt), sinks[t]); });
        for (std::thread &x : th) x.join();
    }
    for (BoxSink &s : sinks)
        for (const auto &p : s.v) boxes[p.first].push_back(p.second);
}

}  // namespace pgh
#endif

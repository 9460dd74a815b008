// pg_bdhints.hpp -- BreakDancer window hints for the far-end search: from a `-b`
// file to the per-read window clusters that pg_far_end_batch / pg_device_batch_set_windows take.
//
// What the reference does, restated (no code shared):
//   loadBDFile                              src/bddata.cpp:91-136   (+ format check :23-88)
//   BreakDancerCoordinate windows, ordering src/control_state.cpp:71-131  (span 200, control_state.h:46)
//   UpdateBD with no discordant read pairs  src/bddata.cpp:646-649, 809  (events = external events, sorted)
//   loadRegion / createRegionCluster        src/bddata.cpp:814-946
//   getCorrespondingSearchWindowCluster     src/bddata.cpp:949-979
//
// Scope and parity status: only the external (`-b` file) events are handled -- the read-pair events that
// UpdateBD adds need the BAM path (htslib).  In Pindel 0.2.5b9 a `-b` file has no
// effect on Pindel-text input at all (the events only reach loadRegion through UpdateBD, which main() calls
// for BAM input with -R), so there is NO reference output to pin this module against in this image:
// PARITY UNPINNED.  It is cross-checked against an independent restatement in tests/test_cpu_suite.py and
// is off by default in the pindel_pg command line (`-b file` alone reproduces the reference: ignored).
#ifndef PG_BDHINTS_HPP
#define PG_BDHINTS_HPP

#include <algorithm>
#include <cstdint>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace pgh {

struct BDWindow {
    int chr_id;          // index into the chromosome name list given to load_region
    unsigned start, end; // Pindel coordinates (spacer included), as SearchWindow(start, end)
};

class BDHints {
public:
    static const unsigned SPAN = 200;    // BREAKDANCER_WINDOWSPAN

    // 0 = loaded; 1 = the file failed the reference's format check and is ignored (as the reference
    // does, after printing a note); -1 = cannot open (the reference exits).
    int load_file(const std::string &path, unsigned spacer, std::string &note)
    {
        events_.clear();
        {
            std::ifstream probe(path.c_str());
            if (!probe.good()) {
                note = "cannot load breakdancer file '" + path + "'";
                return -1;
            }
        }
        if (!format_ok(path, note)) return 1;
        std::ifstream in(path.c_str());
        std::string line;
        while (std::getline(in, line)) {
            strip_cr(line);
            if (!line.empty() && line[0] == '#') continue;
            std::istringstream ls(line);
            std::string c1, c2, skip;
            unsigned p1 = 0, p2 = 0;
            ls >> c1 >> p1 >> skip >> c2 >> p2 >> skip;
            if (ls.fail()) continue;                     // short / empty line: nothing is pushed
            p1 += spacer;
            p2 += spacer;
            // "abs(firstPos - secondPos) < 500" on unsigned operands: the wrapped difference as int
            const int diff = (int)(p1 - p2);
            if (c1 == c2 && !c2.empty() && (diff < 0 ? -diff : diff) < 500) continue;
            if (c1.empty() || c2.empty()) continue;
            Event a = { { c1, p1, p1 }, { c2, p2, p2 } }, b = { { c2, p2, p2 }, { c1, p1, p1 } };
            events_.push_back(a);
            events_.push_back(b);
        }
        std::sort(events_.begin(), events_.end(), first_less);
        external_ = events_;
        return 0;
    }

    // BDData::UpdateBD (src/bddata.cpp:646-649, 809): the events of this window = the external ones (-b file) + the
    // read-pair events of the window (pg_rp.hpp), both directions, sorted on the first coordinate
    struct RpSide { std::string chr; unsigned pos, pos2; };
    void update_with_rp(const std::vector<std::pair<RpSide, RpSide>> &rp)
    {
        events_ = external_;
        for (const auto &e : rp) {
            Event a = { { e.first.chr, e.first.pos, e.first.pos2 }, { e.second.chr, e.second.pos, e.second.pos2 } };
            Event b = { { e.second.chr, e.second.pos, e.second.pos2 }, { e.first.chr, e.first.pos, e.first.pos2 } };
            events_.push_back(a);
            events_.push_back(b);
        }
        std::sort(events_.begin(), events_.end(), first_less);
    }
    size_t n_events_total() const { return events_.size() / 2; }

    size_t n_events() const { return events_.size() / 2; }

    // loadRegion for the bin [start, end] of chromosome chr_names[chr_id] (Pindel coordinates).
    // Returns false if an event names a chromosome that is not in chr_names (the reference exits).
    bool load_region(const std::vector<std::string> &chr_names, int chr_id, unsigned start, unsigned end,
                     std::string &err)
    {
        const unsigned INSERT_SIZE = 1000;
        win_start_ = start >= 3 * INSERT_SIZE ? start - 3 * INSERT_SIZE : 0;
        win_end_ = end + 3 * INSERT_SIZE;
        const std::string &chr = chr_names[chr_id];
        const Event lo_key = { { chr, win_start_, win_start_ }, { "", 0, 0 } };
        const Event hi_key = { { chr, win_end_, win_end_ }, { "", 0, 0 } };
        const size_t first = std::lower_bound(events_.begin(), events_.end(), lo_key, first_less) - events_.begin();
        const size_t last = std::upper_bound(events_.begin(), events_.end(), hi_key, first_less) - events_.begin();
        mask_.assign((size_t)(win_end_ - win_start_) + 1, 0u);     // getSize() entries; the last one is never written
        clusters_.clear();
        clusters_.push_back(std::vector<BDWindow>());
        size_t live_begin = first, live_end = first;              // startOfEventList, endOfEventList
        unsigned index = 0;
        for (unsigned position = win_start_; position < win_end_; position++) {
            bool changed = false;
            // drop events from the FRONT of the live list whose window has been passed
            for (size_t k = live_begin; k < live_end; k++) {
                if (position > events_[k].first.end_of_window()) {
                    live_begin++;
                    changed = true;
                } else {
                    break;
                }
            }
            // an event starting exactly here extends the live list by ONE (the next in order)
            for (size_t k = live_end; k < last; k++) {
                if (position < events_[k].first.start_of_window()) break;
                if (position == events_[k].first.start_of_window()) {
                    live_end++;
                    changed = true;
                }
            }
            if (live_begin == live_end) {
                mask_[position - win_start_] = 0;
            } else {
                if (changed) {
                    index++;
                    std::vector<BDWindow> cluster;
                    if (!make_cluster(chr_names, live_begin, live_end, cluster, err)) return false;
                    clusters_.push_back(cluster);
                }
                mask_[position - win_start_] = index;
            }
        }
        return true;
    }

    // getCorrespondingSearchWindowCluster for a read whose last UP_Close point is at last_close_absloc.
    const std::vector<BDWindow> &cluster(unsigned last_close_absloc) const
    {
        if (clusters_.empty()) return empty_;
        const unsigned rel = last_close_absloc - win_start_;        // unsigned like the reference
        const unsigned size = 1 + win_end_ - win_start_;
        if (rel > size) return clusters_[0];
        unsigned ci = 0;
        // the reference reads m_breakDancerMask[rel] for rel up to size, although only rel < size - 1 was
        // ever written (uninitialised / one past the end): those positions count as "no cluster" here
        if (last_close_absloc > win_start_ && rel < size - 1) ci = mask_[rel];
        return clusters_[ci];
    }

private:
    struct Coord {
        std::string chr;
        unsigned pos, pos2;
        unsigned start_of_window() const
        {
            unsigned t = pos;
            if (pos2 < pos && pos2 > 0) t = pos2;
            return t >= SPAN ? t - SPAN : 0;
        }
        unsigned end_of_window() const
        {
            unsigned t = pos;
            if (pos2 > pos && pos2 > 0) t = pos2;
            return t + SPAN;
        }
        bool differs(const Coord &o) const { return chr != o.chr || pos != o.pos; }
        bool less(const Coord &o) const
        {
            if (chr != o.chr) return chr < o.chr;
            if (pos != o.pos) return pos < o.pos;
            return false;
        }
    };
    struct Event {
        Coord first, second;
    };
    static bool first_less(const Event &a, const Event &b)       // sortOnFirstBDCoordinate
    {
        if (a.first.differs(b.first)) return a.first.less(b.first);
        if (a.second.differs(b.second)) return a.second.less(b.second);
        return false;
    }
    static bool second_less(const Event &a, const Event &b)      // sortOnSecondBDCoordinate
    {
        if (a.second.differs(b.second)) return a.second.less(b.second);
        if (a.first.differs(b.first)) return a.first.less(b.first);
        return false;
    }
    static void strip_cr(std::string &s)
    {
        if (!s.empty() && s[s.size() - 1] == '\r') s.erase(s.size() - 1);
    }
    static bool is_number(const std::string &s)
    {
        for (size_t i = 0; i < s.size(); i++)
            if (s[i] < '0' || s[i] > '9') return false;
        return true;
    }
    static bool at_least_6_fields(const std::string &s)
    {
        if (s.empty() || s[0] == '\t' || s[0] == ' ') return false;
        unsigned fields = 0;
        bool in_space = false;
        for (size_t i = 1; i < s.size(); i++) {
            if (s[i] == '\t' || s[i] == ' ') in_space = true;
            else if (in_space) {
                fields++;
                in_space = false;
            }
        }
        return fields >= 5;
    }
    // CheckBreakDancerFileFormat: every non-comment, non-empty line needs six fields with numeric 2nd and 5th
    static bool format_ok(const std::string &path, std::string &note)
    {
        std::ifstream in(path.c_str());
        std::string line;
        while (std::getline(in, line)) {
            strip_cr(line);
            if (!line.empty() && line[0] == '#') continue;
            if (at_least_6_fields(line)) {
                std::istringstream ls(line);
                std::string t, p1, p2;
                ls >> t >> p1 >> t >> t >> p2 >> t;
                if (!(is_number(p1) && is_number(p2))) {
                    note = "breakdancer file ignored, bad line: " + line;
                    return false;
                }
            } else if (!line.empty()) {
                note = "breakdancer file ignored, bad line: " + line;
                return false;
            }
        }
        return true;
    }
    // createRegionCluster: the live events ordered by their SECOND coordinate, overlapping windows merged
    bool make_cluster(const std::vector<std::string> &chr_names, size_t b, size_t e, std::vector<BDWindow> &out,
                      std::string &err) const
    {
        std::vector<Event> sub(events_.begin() + b, events_.begin() + e);
        std::sort(sub.begin(), sub.end(), second_less);
        for (size_t i = 0; i < sub.size(); i++) {
            int id = -1;
            for (size_t c = 0; c < chr_names.size(); c++)
                if (chr_names[c] == sub[i].second.chr) id = (int)c;
            if (id < 0) {
                err = "chromosome with name : " + sub[i].second.chr + " not yet loaded into memory";
                return false;
            }
            BDWindow w = { id, sub[i].second.start_of_window(), sub[i].second.end_of_window() };
            while (i + 1 < sub.size() && sub[i + 1].second.chr == sub[i].second.chr &&
                   sub[i + 1].second.start_of_window() <= sub[i].second.end_of_window() + 1) {
                i++;
                w.end = sub[i].second.end_of_window();
            }
            out.push_back(w);
        }
        return true;
    }

    std::vector<Event> events_, external_;
    std::vector<unsigned> mask_;
    std::vector<std::vector<BDWindow>> clusters_;
    std::vector<BDWindow> empty_;
    unsigned win_start_ = 0, win_end_ = 0;
};

}  // namespace pgh
#endif

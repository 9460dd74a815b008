// pg_bam.hpp -- BAM ingest for the split-read search without htslib: BGZF blocks are inflated
// with zlib, BAM records and the BAI index are decoded here, and the reference's read-selection rules are
// restated on top:
//
//   ReadInBamReads_SR     src/reader.cpp:483-559    region query of one window, pairing by read name, flush
//   fetch_func_SR         src/reader.cpp:1099-1151  which (anchor, read) combinations of a pair become records
//   isGoodAnchor          src/reader.cpp:561-615    isWeirdRead  src/reader.cpp:658-690
//   parse_flags_and_tags  src/reader.cpp:1258-1316  (mapped flag, NM)
//   build_record_SR       src/reader.cpp:799-898    name, N trimming, orientation, MatchedRelPos, clamps
//   bam_cigar2len / bam_cigar2mismatch              src/reader.cpp:1319-1346
//
// Output: the SoA read batch the C ABI takes (pg_adapter::Batch) plus names / mapping qualities / tags for the
// reporters, in the order the reference's single-threaded run emits the records (its OpenMP flush pushes the
// reads of a 50 000-read buffer in a race: the order here is the deterministic T = 1 order).
//
// Reference-supporting reads (isRefRead / build_record_RefRead, src/reader.cpp:620-656, 903-923) are collected by the
// same pass and become the per-sample coverage integers of the report headers (Caller::update_ref_coverage).
//
// PARITY STATUS: pinned on the one BAM the reference ships, demo/simulated_MEI/aln.sorted.bam (bwa + samtools,
// 24 000 records, with its .bai) and the 20 Pindel-text records `input` its authors derived from it: decoder == an
// independent decoding, index == scan, all 20 records reproduced field for field, every further record explained by a
// named rule of fetch_func_SR (tests/test_mei_bam.py).  Beyond that file the rules are checked against an independent
// restatement on a deliberately messy synthetic BAM and by the round trip "gold reads -> BAM (tests/bam_writer.py) ->
// this reader == the Pindel-text route, identical gold reports" (tests/test_bam_ingest.py).  The reference binary's
// BAM path itself cannot be built here (htslib).
#ifndef PG_BAM_HPP
#define PG_BAM_HPP

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "pg_adapter.hpp"

namespace pgh {

// ------------------------------------------------------------------------------------------ BGZF
class BgzfReader {
public:
    ~BgzfReader() { close(); }
    bool open(const std::string &path)
    {
        close();
        f_ = fopen(path.c_str(), "rb");
        block_addr_ = 0;
        block_len_ = 0;
        pos_ = 0;
        eof_ = false;
        bad_ = false;
        return f_ != nullptr;
    }
    void close()
    {
        if (f_) fclose(f_);
        f_ = nullptr;
    }
    // virtual offset = (file offset of the block << 16) | offset inside the inflated block
    uint64_t tell() const
    {
        if (pos_ == data_.size() && block_len_) return (block_addr_ + block_len_) << 16;
        return (block_addr_ << 16) | (uint64_t)pos_;
    }
    bool seek(uint64_t voff)
    {
        const uint64_t addr = voff >> 16;
        if (fseeko(f_, (off_t)addr, SEEK_SET) != 0) return false;
        block_addr_ = addr;
        block_len_ = 0;
        data_.clear();
        pos_ = 0;
        eof_ = false;
        bad_ = false;                                     // (a reader may be reused after a failed read)
        if (!load_block()) return !bad_ && (voff & 0xffff) == 0;
        pos_ = (size_t)(voff & 0xffff);
        return pos_ <= data_.size();
    }
    // reads exactly n bytes; false at end of file (or on a corrupt block).  got (nullable): the bytes delivered before the end
    bool read(void *dst, size_t n, size_t *got = nullptr)
    {
        uint8_t *d = (uint8_t *)dst;
        if (got) *got = 0;
        while (n) {
            if (pos_ == data_.size()) {
                block_addr_ += block_len_;
                block_len_ = 0;
                if (!load_block()) {
                    if (got) *got = (size_t)(d - (uint8_t *)dst);
                    return false;
                }
                continue;
            }
            const size_t k = std::min(n, data_.size() - pos_);
            memcpy(d, data_.data() + pos_, k);
            pos_ += k;
            d += k;
            n -= k;
        }
        return true;
    }
    bool ok() const { return f_ != nullptr; }
    // a corrupt or truncated block was met (as opposed to the clean end of the file)
    bool failed() const { return bad_; }
    void mark_failed() { bad_ = true; }

private:
    bool fail()
    {
        bad_ = true;
        return false;
    }
    bool load_block()
    {
        data_.clear();
        pos_ = 0;
        for (;;) {                                       // skip empty blocks (the end-of-file marker is one)
            uint8_t h[18];
            const size_t got = fread(h, 1, 18, f_);
            if (got != 18) {
                eof_ = true;
                if (got != 0) bad_ = true;               // a truncated block header is not a clean end of file
                return false;
            }
            if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return fail();
            const unsigned xlen = h[10] | (h[11] << 8);
            // the BC subfield is the first one in every BGZF writer; search the extra field to be safe
            std::vector<uint8_t> extra(xlen);
            memcpy(extra.data(), h + 12, std::min<size_t>(6, xlen));
            if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, f_) != xlen - 6) return fail();
            unsigned bsize = 0;
            for (size_t i = 0; i + 4 <= xlen;) {
                const unsigned slen = extra[i + 2] | (extra[i + 3] << 8);
                if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2 && i + 6 <= xlen) bsize = (extra[i + 4] | (extra[i + 5] << 8)) + 1u;
                i += 4 + slen;
            }
            if (bsize < 12 + xlen + 8) return fail();
            const size_t clen = bsize - 12 - xlen - 8;
            comp_.resize(clen + 8);
            if (fread(comp_.data(), 1, clen + 8, f_) != clen + 8) return fail();
            const uint32_t isize = comp_[clen + 4] | (comp_[clen + 5] << 8) | (comp_[clen + 6] << 16) | ((uint32_t)comp_[clen + 7] << 24);
            block_len_ = bsize;
            if (isize == 0) {
                block_addr_ += block_len_;
                block_len_ = 0;
                continue;
            }
            data_.resize(isize);
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) return fail();
            zs.next_in = comp_.data();
            zs.avail_in = (uInt)clen;
            zs.next_out = data_.data();
            zs.avail_out = isize;
            const int zr = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (zr != Z_STREAM_END || zs.total_out != isize) return fail();
            return true;
        }
    }
    FILE *f_ = nullptr;
    uint64_t block_addr_ = 0, block_len_ = 0;
    std::vector<uint8_t> data_, comp_;
    size_t pos_ = 0;
    bool eof_ = false, bad_ = false;
};

// ------------------------------------------------------------------------------------------ BAM records
enum {
    BAM_FPAIRED = 1, BAM_FPROPER_PAIR = 2, BAM_FUNMAP = 4, BAM_FMUNMAP = 8, BAM_FREVERSE = 16, BAM_FMREVERSE = 32,
    BAM_FREAD1 = 64, BAM_FREAD2 = 128, BAM_FSECONDARY = 256, BAM_FQCFAIL = 512, BAM_FDUP = 1024
};
enum { BAM_CMATCH = 0, BAM_CINS = 1, BAM_CDEL = 2, BAM_CREF_SKIP = 3, BAM_CSOFT_CLIP = 4, BAM_CHARD_CLIP = 5, BAM_CPAD = 6 };

struct BamRecord {
    int32_t tid = -1, pos = -1, mtid = -1, mpos = -1, tlen = 0, l_seq = 0;
    uint8_t mapq = 0;
    uint16_t flag = 0;
    std::string qname;
    std::vector<uint32_t> cigar;      // len << 4 | op
    std::vector<uint8_t> seq4;        // 4-bit packed bases
    std::vector<uint8_t> aux;

    // bam_endpos: an unmapped read or one without CIGAR covers one base
    int32_t end_pos() const
    {
        if ((flag & BAM_FUNMAP) || cigar.empty()) return pos + 1;
        int32_t l = 0;
        for (uint32_t c : cigar) {
            const int op = c & 15;
            if (op == BAM_CMATCH || op == BAM_CDEL || op == BAM_CREF_SKIP || op == 7 || op == 8) l += (int32_t)(c >> 4);
        }
        return pos + (l ? l : 1);
    }
    // integer value of an aux tag (types cCsSiI); false if absent or not an integer
    bool aux_int(const char *tag, int64_t &v) const
    {
        size_t i = 0;
        const size_t n = aux.size();
        while (i + 3 <= n) {
            const char t0 = (char)aux[i], t1 = (char)aux[i + 1], ty = (char)aux[i + 2];
            i += 3;
            size_t sz = 0;
            int64_t val = 0;
            bool is_int = true;
            switch (ty) {
            case 'A': sz = 1; is_int = false; break;
            case 'c': sz = 1; if (i < n) val = (int8_t)aux[i]; break;
            case 'C': sz = 1; if (i < n) val = aux[i]; break;
            case 's': sz = 2; if (i + 2 <= n) val = (int16_t)(aux[i] | (aux[i + 1] << 8)); break;
            case 'S': sz = 2; if (i + 2 <= n) val = (uint16_t)(aux[i] | (aux[i + 1] << 8)); break;
            case 'i': sz = 4; if (i + 4 <= n) val = (int32_t)(aux[i] | (aux[i + 1] << 8) | (aux[i + 2] << 16) | ((uint32_t)aux[i + 3] << 24)); break;
            case 'I': sz = 4; if (i + 4 <= n) val = (uint32_t)(aux[i] | (aux[i + 1] << 8) | (aux[i + 2] << 16) | ((uint32_t)aux[i + 3] << 24)); break;
            case 'f': sz = 4; is_int = false; break;
            case 'Z': case 'H':
                is_int = false;
                while (i + sz < n && aux[i + sz]) sz++;
                sz++;
                break;
            case 'B': {
                is_int = false;
                if (i + 5 > n) return false;
                const char sub = (char)aux[i];
                const uint32_t cnt = aux[i + 1] | (aux[i + 2] << 8) | (aux[i + 3] << 16) | ((uint32_t)aux[i + 4] << 24);
                const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                sz = 5 + es * cnt;
                break;
            }
            default: return false;
            }
            if (t0 == tag[0] && t1 == tag[1]) {
                if (!is_int) return false;
                v = val;
                return true;
            }
            i += sz;
        }
        return false;
    }
};

struct BamHeader {
    std::vector<std::string> names;
    std::vector<uint32_t> lengths;
    int id_of(const std::string &name) const
    {
        for (size_t i = 0; i < names.size(); i++)
            if (names[i] == name) return (int)i;
        return -1;
    }
};

class BamFile {
public:
    bool open(const std::string &path, std::string &err, bool use_index = true)
    {
        path_ = path;
        if (!z_.open(path)) {
            err = "cannot open " + path;
            return false;
        }
        char magic[4];
        int32_t l_text = 0, n_ref = 0;
        if (!z_.read(magic, 4) || memcmp(magic, "BAM\1", 4) != 0 || !z_.read(&l_text, 4) || l_text < 0) {
            err = path + ": not a BAM file";
            return false;
        }
        std::string text((size_t)l_text, ' ');
        if (l_text && !z_.read(&text[0], (size_t)l_text)) return bad(err);
        if (!z_.read(&n_ref, 4) || n_ref < 0) return bad(err);
        for (int i = 0; i < n_ref; i++) {
            int32_t l_name = 0;
            uint32_t l_ref = 0;
            if (!z_.read(&l_name, 4) || l_name <= 0) return bad(err);
            std::string nm((size_t)l_name, ' ');
            if (!z_.read(&nm[0], (size_t)l_name) || !z_.read(&l_ref, 4)) return bad(err);
            nm.resize((size_t)l_name - 1);
            hdr_.names.push_back(nm);
            hdr_.lengths.push_back(l_ref);
        }
        first_record_ = z_.tell();
        idx_.reset();
        if (use_index) load_index();
        return true;
    }
    // a second reader of the same file (own file handle; header and index shared): for reading sub-ranges of a
    // window on several threads
    bool open_like(const BamFile &o, std::string &err)
    {
        path_ = o.path_;
        if (!z_.open(path_)) {
            err = "cannot open " + path_;
            return false;
        }
        hdr_ = o.hdr_;
        first_record_ = o.first_record_;
        idx_ = o.idx_;
        return true;
    }
    const BamHeader &header() const { return hdr_; }
    bool has_index() const { return idx_ && !idx_->bins.empty(); }

    // next record of the stream; false at the end
    bool next(BamRecord &r)
    {
        int32_t block_size = 0;
        size_t got = 0;
        if (!z_.read(&block_size, 4, &got)) {             // the end of the file, or a bad block (z_.failed())
            if (got != 0) z_.mark_failed();               // ... or a file cut inside a record's length word: not a clean end
            return false;
        }
        if (block_size < 32) {
            z_.mark_failed();
            return false;
        }
        buf_.resize((size_t)block_size);
        if (!z_.read(buf_.data(), (size_t)block_size) || !decode(buf_.data(), (size_t)block_size, r)) {
            z_.mark_failed();                             // a record cut short or inconsistent: never a clean end
            return false;
        }
        return true;
    }
    // the bytes of the record `next` delivered last (block_size bytes, without the length word)
    const std::vector<uint8_t> &last_raw() const { return buf_; }
    // one record from its bytes
    static bool decode(const uint8_t *p, size_t block_size, BamRecord &r)
    {
        auto i32 = [&](size_t o) { return (int32_t)(p[o] | (p[o + 1] << 8) | (p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24)); };
        r.tid = i32(0);
        r.pos = i32(4);
        const unsigned l_read_name = p[8];
        r.mapq = p[9];
        const unsigned n_cigar = p[12] | (p[13] << 8);
        r.flag = (uint16_t)(p[14] | (p[15] << 8));
        r.l_seq = i32(16);
        r.mtid = i32(20);
        r.mpos = i32(24);
        r.tlen = i32(28);
        size_t o = 32;
        if (r.l_seq < 0 || o + l_read_name + 4u * n_cigar + (size_t)(r.l_seq + 1) / 2 + (size_t)r.l_seq > block_size) return false;
        r.qname.assign((const char *)p + o, l_read_name ? l_read_name - 1 : 0);
        o += l_read_name;
        r.cigar.resize(n_cigar);
        for (unsigned k = 0; k < n_cigar; k++) r.cigar[k] = (uint32_t)i32(o + 4 * k);
        o += 4u * n_cigar;
        r.seq4.assign(p + o, p + o + (size_t)(r.l_seq + 1) / 2);
        o += (size_t)(r.l_seq + 1) / 2 + (size_t)r.l_seq;        // + qualities
        r.aux.assign(p + o, p + block_size);
        return true;
    }

    // Records of reference `tid` overlapping [beg, end) in file order (sam_itr_queryi + sam_itr_next); with a
    // .bai next to the file only the chunks of the overlapping bins are read, without one the file is scanned.
    template <class Fn>
    bool query(int tid, int64_t beg, int64_t end, Fn fn)
    {
        BamRecord r;
        if (tid < 0) return true;
        if (beg < 0) beg = 0;
        if (!has_index()) {
            if (!z_.seek(first_record_)) return false;
            while (next(r))
                if (r.tid == tid && r.pos < end && r.end_pos() > beg) fn(r);
            return !z_.failed();                          // a damaged file must not pass for a short one
        }
        const BaiBins &bins_ = idx_->bins;
        const std::vector<std::vector<uint64_t>> &linear_ = idx_->linear;
        if ((size_t)tid >= bins_.size()) return true;
        // reg2bins (SAM specification section 5.3)
        std::vector<uint32_t> want;
        {
            const int64_t e = end - 1;
            want.push_back(0);
            for (int k = 1 + (int)(beg >> 26); k <= 1 + (int)(e >> 26); ++k) want.push_back((uint32_t)k);
            for (int k = 9 + (int)(beg >> 23); k <= 9 + (int)(e >> 23); ++k) want.push_back((uint32_t)k);
            for (int k = 73 + (int)(beg >> 20); k <= 73 + (int)(e >> 20); ++k) want.push_back((uint32_t)k);
            for (int k = 585 + (int)(beg >> 17); k <= 585 + (int)(e >> 17); ++k) want.push_back((uint32_t)k);
            for (int k = 4681 + (int)(beg >> 14); k <= 4681 + (int)(e >> 14); ++k) want.push_back((uint32_t)k);
        }
        const uint64_t min_off = linear_[tid].empty() ? 0 : linear_[tid][std::min<size_t>((size_t)(beg >> 14), linear_[tid].size() - 1)];
        std::vector<std::pair<uint64_t, uint64_t>> chunks;
        for (uint32_t b : want) {
            auto it = bins_[tid].find(b);
            if (it == bins_[tid].end()) continue;
            for (const auto &c : it->second)
                if (c.second > min_off) chunks.push_back(c);
        }
        std::sort(chunks.begin(), chunks.end());
        std::vector<std::pair<uint64_t, uint64_t>> merged;
        // chunks that touch, overlap or lie within 256 KB of file of each other are read in one go (the records in
        // between fail the overlap test below)
        for (const auto &c : chunks) {
            if (!merged.empty() && (c.first >> 16) <= (merged.back().second >> 16) + (256u << 10))
                merged.back().second = std::max(merged.back().second, c.second);
            else merged.push_back(c);
        }
        for (const auto &c : merged) {
            if (!z_.seek(c.first)) return false;
            while (z_.tell() < c.second && next(r))
                if (r.tid == tid && r.pos < end && r.end_pos() > beg) fn(r);
            if (z_.failed()) return false;
        }
        return true;
    }

    // The same query cut into `parts` sub-ranges by start position, each read by its own reader (own file handle, shared
    // header and index) on its own thread: fn(part, record, reader) is called on that thread for the records that START
    // in the sub-range (part 0: also those that only reach into [beg, end)), in file order within the part -- so the
    // parts taken one after the other are the records of query() in the same order, for a coordinate-sorted file
    // (which a file with an index is).  Threads: min(hardware, 16) or PGH_THREADS.
    static unsigned worker_threads()
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
    unsigned split_parts(int tid, int64_t beg, int64_t end) const
    {
        if (tid < 0 || !has_index() || end <= beg) return 1;
        return std::min<unsigned>(worker_threads(), (unsigned)((end - beg) >> 18) + 1u);
    }
    template <class Fn>
    bool query_split(int tid, int64_t beg, int64_t end, unsigned parts, Fn fn) const
    {
        std::vector<int> part_ok(parts, 1);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < parts; t++)
            th.emplace_back([&, t]() {
                BamFile mine;
                std::string err;
                if (!mine.open_like(*this, err)) {
                    part_ok[t] = 0;
                    return;
                }
                const int64_t lo = beg + (end - beg) * (int64_t)t / parts, hi = beg + (end - beg) * (int64_t)(t + 1) / parts;
                part_ok[t] = mine.query(tid, lo, hi, [&](const BamRecord &r) {
                    if (t == 0 || r.pos >= lo) fn(t, r, mine);
                }) ? 1 : 0;
            });
        for (std::thread &x : th) x.join();
        for (unsigned t = 0; t < parts; t++)
            if (!part_ok[t]) return false;
        return true;
    }

private:
    bool bad(std::string &err)
    {
        err = path_ + ": truncated BAM header";
        return false;
    }
    void load_index()
    {
        idx_.reset();
        FILE *f = fopen((path_ + ".bai").c_str(), "rb");
        if (!f) {
            std::string alt = path_;
            if (alt.size() > 4 && alt.substr(alt.size() - 4) == ".bam") alt = alt.substr(0, alt.size() - 4) + ".bai";
            f = fopen(alt.c_str(), "rb");
        }
        if (!f) return;
        std::shared_ptr<Index> ix(new Index());
        const bool ok = read_bai(f, ix->bins, ix->linear);
        fclose(f);
        if (ok) idx_ = ix;
    }

public:
    typedef std::vector<std::unordered_map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>>> BaiBins;
    // A .bai file: per reference the chunks of every bin (the metadata pseudo-bin 37450 is skipped) and the linear
    // index; whatever follows (n_no_coor) is not needed.
    static bool read_bai(FILE *f, BaiBins &bins, std::vector<std::vector<uint64_t>> &linear)
    {
        auto rd = [&](void *p, size_t n) { return fread(p, 1, n, f) == n; };
        char magic[4];
        int32_t n_ref = 0;
        bool ok = rd(magic, 4) && memcmp(magic, "BAI\1", 4) == 0 && rd(&n_ref, 4) && n_ref >= 0;
        bins.clear();
        linear.clear();
        if (ok) {
            bins.resize((size_t)n_ref);
            linear.resize((size_t)n_ref);
        }
        for (int t = 0; ok && t < n_ref; t++) {
            int32_t n_bin = 0;
            ok = rd(&n_bin, 4);
            for (int b = 0; ok && b < n_bin; b++) {
                uint32_t bin = 0;
                int32_t n_chunk = 0;
                ok = rd(&bin, 4) && rd(&n_chunk, 4) && n_chunk >= 0;
                std::vector<std::pair<uint64_t, uint64_t>> cs((size_t)(ok ? n_chunk : 0));
                for (auto &c : cs) ok = ok && rd(&c.first, 8) && rd(&c.second, 8);
                if (ok && bin != 37450) bins[t][bin] = cs;          // 37450: the metadata pseudo-bin
            }
            int32_t n_intv = 0;
            ok = ok && rd(&n_intv, 4) && n_intv >= 0;
            if (ok) {
                linear[t].resize((size_t)n_intv);
                ok = n_intv == 0 || rd(linear[t].data(), 8u * (size_t)n_intv);
            }
        }
        return ok;
    }

private:
    std::string path_;
    BgzfReader z_;
    BamHeader hdr_;
    uint64_t first_record_ = 0;
    std::vector<uint8_t> buf_;
    struct Index {
        BaiBins bins;
        std::vector<std::vector<uint64_t>> linear;
    };
    std::shared_ptr<const Index> idx_;
};

// ------------------------------------------------------------------------------------------ read selection
struct BamIngestSettings {
    unsigned min_anchor_quality = 0;   // -A  minimalAnchorQuality
    unsigned spacer = 100000;
    int nm = 2;                        // -n  NM (isRefRead)
    double max_mismatch_rate = 0.02;   // -u  MaximumAllowedMismatchRate (isRefRead)
};

// A read that supports the reference allele (REF_READ, pindel.h:199-212): what UpdateRefReadCoverage needs of it
struct RefRead {
    uint32_t pos;       // leftmost position of the read (BAM, 0-based)
    uint16_t length;    // l_qseq
    uint16_t tag;       // index into IngestedReads::ref_tags
};

// One window's split-read candidates of one or more BAM files: the SoA batch + what the reporters print.
struct IngestedReads {
    pg_adapter::Batch batch;           // seq = UnmatchedSeq as after setUnmatchedSeq, pos = MatchedRelPos
    std::vector<std::string> names;    // "@qname/1"
    std::vector<int16_t> ms;           // mapping quality of the anchor
    std::vector<std::string> tags;     // sample tag of the BAM
    std::vector<RefRead> ref_reads;    // RefSupportingReads of the window
    std::vector<std::string> ref_tags; // their sample tags (RefRead::tag indexes this)
    size_t size() const { return names.size(); }
    void clear()
    {
        batch = pg_adapter::Batch();
        batch.off.push_back(0);
        names.clear();
        ms.clear();
        tags.clear();
        ref_reads.clear();
        ref_tags.clear();
    }
};

// PGH_TIMING diagnostics: seconds spent in the three parts of the indexed ingest, summed over the windows of a run
struct IngestTiming {
    double inflate_decode = 0, select = 0, layout = 0;
};
inline IngestTiming &ingest_timing()
{
    static IngestTiming t;
    return t;
}

class BamIngest {
public:
    explicit BamIngest(const BamIngestSettings &s) : S(s) {}
    std::string error;

    // ReadInBamReads_SR for one BAM and one window [win_start, win_end) of chromosome `chr_name` (index chr_id in
    // the loaded reference, padded size chr_padded_size).  Appends to `out`.  false: error (see .error).
    bool read_window(BamFile &bam, const std::string &chr_name, int chr_id, uint64_t chr_padded_size, int64_t win_start,
                     int64_t win_end, int insert_size, const std::string &tag, IngestedReads &out)
    {
        if (out.batch.off.empty()) out.batch.off.push_back(0);
        const int tid = bam.header().id_of(chr_name);
        bool ok = true;
        // fetch_func_SR, the two cases: the first record of a name (a read that is "weird" is its own anchor), and
        // the second one together with the first (b2)
        auto first_of_name = [&](const BamRecord &b1) {
            if (is_weird(b1)) ok = ok && build_record(bam, b1, b1, chr_id, chr_padded_size, insert_size, tag, out);
        };
        auto pair_complete = [&](const BamRecord &b1, const BamRecord &b2) {
            if (is_weird(b2)) ok = ok && build_record(bam, b2, b2, chr_id, chr_padded_size, insert_size, tag, out);
            if (is_good_anchor(b1) && is_weird(b2)) ok = ok && build_record(bam, b1, b2, chr_id, chr_padded_size, insert_size, tag, out);
            if (is_good_anchor(b1) && is_ref_read(b2)) add_ref_read(b2, tag, out);
            if (is_good_anchor(b2) && is_weird(b1)) ok = ok && build_record(bam, b2, b1, chr_id, chr_padded_size, insert_size, tag, out);
            if (is_good_anchor(b2) && is_ref_read(b1)) add_ref_read(b1, tag, out);
        };
        std::unordered_map<std::string, BamRecord> waiting;         // read_to_map_qual: first mate seen, by name
        auto take = [&](const BamRecord &b1) {                      // the records in file order
            if (!ok) return;
            auto it = waiting.find(b1.qname);
            if (it == waiting.end()) {
                waiting.emplace(b1.qname, b1);
                first_of_name(b1);
                return;
            }
            const BamRecord b2 = it->second;
            waiting.erase(it);
            pair_complete(b1, b2);
        };
        // With an index (= a coordinate-sorted file) the window is cut into sub-ranges by start position; every
        // sub-range is inflated and decoded by its own reader on its own thread (BGZF inflation and record decoding
        // are most of the ingest time), and the records then pass through the selection above in file order: a
        // sub-range keeps the records that START in it (the first one also those that reach into the window).
        const unsigned nt = bam.split_parts(tid, win_start, win_end);
        if (nt > 1) {
            struct Part {
                std::vector<uint8_t> bytes;        // the records that start in the sub-range, back to back
                std::vector<uint64_t> ends;        // end offset of each in `bytes`
                std::vector<uint32_t> hash;        // hash of each record's read name
            };
            std::vector<Part> parts(nt);
            auto clock_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            double t_mark = clock_s();
            if (!bam.query_split(tid, win_start, win_end, nt, [&](unsigned t, const BamRecord &rec, const BamFile &reader) {
                    const std::vector<uint8_t> &raw = reader.last_raw();
                    parts[t].bytes.insert(parts[t].bytes.end(), raw.begin(), raw.end());
                    parts[t].ends.push_back(parts[t].bytes.size());
                    uint32_t h = 2166136261u;                          // FNV-1a
                    for (unsigned char ch : rec.qname) h = (h ^ ch) * 16777619u;
                    parts[t].hash.push_back(h);
                })) {
                error = "BAM read failed";
                return false;
            }
            // The selection runs on the records' bytes, which stay where they are (the name is a view into them, the
            // first mate is decoded again when the second one arrives), on several threads: a pair meets in the thread
            // that owns its name's hash, every thread walks the records in file order and notes which record (the
            // "trigger") produced each of its reads, and the reads are then laid out in trigger order -- the order
            // of the sequential pass.
            struct Raw { const uint8_t *p; uint32_t n, hash; };
            std::vector<Raw> recs;
            {
                size_t total = 0;
                for (const Part &part : parts) total += part.ends.size();
                recs.reserve(total);
                for (const Part &part : parts) {
                    uint64_t from = 0;
                    for (size_t k = 0; k < part.ends.size(); k++) {
                        recs.push_back(Raw{ part.bytes.data() + from, (uint32_t)(part.ends[k] - from), part.hash[k] });
                        from = part.ends[k];
                    }
                }
            }
            const size_t N = recs.size();
            ingest_timing().inflate_decode += clock_s() - t_mark;
            t_mark = clock_s();
            const unsigned T2 = (unsigned)std::max<size_t>(1, std::min<size_t>(BamFile::worker_threads(), N / 20000 + 1));
            struct Emit {
                IngestedReads out;
                std::vector<uint32_t> read_trig, ref_trig;       // trigger record of every read / reference read
                std::string error;
                bool ok = true;
            };
            std::vector<Emit> em(T2);
            auto select = [&](unsigned t) {
                Emit &e = em[t];
                e.out.clear();
                e.out.ref_tags = out.ref_tags;                     // (tag indices of the reference reads: same table in every thread)
                std::unordered_map<std::string_view, const Raw *> waiting_raw;
                BamRecord r, r2;
                BamIngest me(S);                                   // (build_record reports the fatal case through .error)
                for (size_t i = 0; i < N && e.ok; i++) {
                    const Raw &x = recs[i];
                    if (x.hash % T2 != t) continue;
                    if (!BamFile::decode(x.p, x.n, r)) {
                        e.ok = false;
                        e.error = "BAM read failed";
                        break;
                    }
                    const std::string_view name((const char *)x.p + 32, r.qname.size());
                    auto it = waiting_raw.find(name);
                    if (it == waiting_raw.end()) {
                        waiting_raw.emplace(name, &x);
                        if (is_weird(r)) e.ok = e.ok && me.build_record(bam, r, r, chr_id, chr_padded_size, insert_size, tag, e.out);
                    } else {
                        (void)BamFile::decode(it->second->p, it->second->n, r2);
                        waiting_raw.erase(it);
                        const BamRecord &b1 = r, &b2 = r2;
                        if (is_weird(b2)) e.ok = e.ok && me.build_record(bam, b2, b2, chr_id, chr_padded_size, insert_size, tag, e.out);
                        if (is_good_anchor(b1) && is_weird(b2)) e.ok = e.ok && me.build_record(bam, b1, b2, chr_id, chr_padded_size, insert_size, tag, e.out);
                        if (is_good_anchor(b1) && is_ref_read(b2)) add_ref_read(b2, tag, e.out);
                        if (is_good_anchor(b2) && is_weird(b1)) e.ok = e.ok && me.build_record(bam, b2, b1, chr_id, chr_padded_size, insert_size, tag, e.out);
                        if (is_good_anchor(b2) && is_ref_read(b1)) add_ref_read(b1, tag, e.out);
                    }
                    while (e.read_trig.size() < e.out.size()) e.read_trig.push_back((uint32_t)i);
                    while (e.ref_trig.size() < e.out.ref_reads.size()) e.ref_trig.push_back((uint32_t)i);
                }
                if (!e.ok && e.error.empty()) e.error = me.error;
            };
            {
                std::vector<std::thread> th;
                for (unsigned t = 1; t < T2; t++) th.emplace_back(select, t);
                select(0);
                for (std::thread &x : th) x.join();
            }
            for (const Emit &e : em)
                if (!e.ok) {
                    error = e.error;
                    return false;
                }
            ingest_timing().select += clock_s() - t_mark;
            t_mark = clock_s();
            // layout in trigger order: reads (and reference reads) per trigger, prefix sums, every thread moves its own
            std::vector<uint32_t> rstart(N + 1, 0), fstart(N + 1, 0);
            for (const Emit &e : em) {
                for (uint32_t i : e.read_trig) rstart[i + 1]++;
                for (uint32_t i : e.ref_trig) fstart[i + 1]++;
            }
            for (size_t i = 0; i < N; i++) {
                rstart[i + 1] += rstart[i];
                fstart[i + 1] += fstart[i];
            }
            const size_t base = out.size(), M = rstart[N], fbase = out.ref_reads.size(), FM = fstart[N];
            // (a reference read's tag index refers to out.ref_tags; a thread that met a new tag appended it to its copy)
            for (const Emit &e : em)
                for (size_t k = out.ref_tags.size(); k < e.out.ref_tags.size(); k++)
                    if (std::find(out.ref_tags.begin(), out.ref_tags.end(), e.out.ref_tags[k]) == out.ref_tags.end())
                        out.ref_tags.push_back(e.out.ref_tags[k]);
            out.names.resize(base + M);
            out.ms.resize(base + M);
            out.tags.resize(base + M);
            out.batch.strand.resize(base + M);
            out.batch.pos.resize(base + M);
            out.batch.isz.resize(base + M);
            out.batch.chr.resize(base + M);
            out.batch.off.resize(base + M + 1);
            out.ref_reads.resize(fbase + FM);
            std::vector<uint32_t> lens(M);
            auto place = [&](unsigned t, bool bases) {
                Emit &e = em[t];
                uint32_t prev = 0xffffffffu, sub = 0;
                for (size_t k = 0; k < e.read_trig.size(); k++) {
                    const uint32_t i = e.read_trig[k];
                    sub = i == prev ? sub + 1 : 0;
                    prev = i;
                    const size_t g = rstart[i] + sub;
                    if (!bases) {
                        out.names[base + g].swap(e.out.names[k]);
                        out.ms[base + g] = e.out.ms[k];
                        out.tags[base + g].swap(e.out.tags[k]);
                        out.batch.strand[base + g] = e.out.batch.strand[k];
                        out.batch.pos[base + g] = e.out.batch.pos[k];
                        out.batch.isz[base + g] = e.out.batch.isz[k];
                        out.batch.chr[base + g] = e.out.batch.chr[k];
                        lens[g] = (uint32_t)(e.out.batch.off[k + 1] - e.out.batch.off[k]);
                    } else {
                        std::copy(e.out.batch.seq.begin() + (long)e.out.batch.off[k], e.out.batch.seq.begin() + (long)e.out.batch.off[k + 1],
                                  out.batch.seq.begin() + (long)out.batch.off[base + g]);
                    }
                }
                if (bases) return;
                prev = 0xffffffffu;
                sub = 0;
                for (size_t k = 0; k < e.ref_trig.size(); k++) {
                    const uint32_t i = e.ref_trig[k];
                    sub = i == prev ? sub + 1 : 0;
                    prev = i;
                    RefRead rr = e.out.ref_reads[k];
                    const std::string &tg = e.out.ref_tags[rr.tag];
                    rr.tag = (uint16_t)(std::find(out.ref_tags.begin(), out.ref_tags.end(), tg) - out.ref_tags.begin());
                    out.ref_reads[fbase + fstart[i] + sub] = rr;
                }
            };
            auto on_threads = [&](bool bases) {
                std::vector<std::thread> th;
                for (unsigned t = 1; t < T2; t++) th.emplace_back(place, t, bases);
                place(0, bases);
                for (std::thread &x : th) x.join();
            };
            on_threads(false);
            for (size_t g = 0; g < M; g++) out.batch.off[base + g + 1] = out.batch.off[base + g] + lens[g];
            out.batch.seq.resize((size_t)out.batch.off[base + M]);
            on_threads(true);
            ingest_timing().layout += clock_s() - t_mark;
            return true;
        }
        const bool q = bam.query(tid, win_start, win_end, take);
        if (!q) error = "BAM read failed";
        return q && ok;
    }

private:
    BamIngestSettings S;

    static int nm_of(const BamRecord &b)                     // parse_flags_and_tags: edits = NM, 0 when absent
    {
        int64_t v = 0;
        return b.aux_int("NM", v) ? (int)v : 0;
    }
    static int cigar_mismatch(const BamRecord &b)            // bam_cigar2mismatch: bases of every non-M element
    {
        int n = 0;
        for (uint32_t c : b.cigar)
            if ((c & 15) != BAM_CMATCH) n += (int)(c >> 4);
        return n;
    }
    bool is_good_anchor(const BamRecord &b) const
    {
        if (b.flag & BAM_FUNMAP) return false;
        if (b.mapq < S.min_anchor_quality) return false;
        if (S.min_anchor_quality == 0) return true;
        return !(b.flag & (BAM_FSECONDARY | BAM_FQCFAIL | BAM_FDUP));
    }
    static bool is_weird(const BamRecord &b)
    {
        if (b.flag & BAM_FUNMAP) return true;
        for (uint32_t c : b.cigar) {
            const int op = c & 15;
            if (op == BAM_CINS || op == BAM_CDEL || op == BAM_CREF_SKIP || op == BAM_CSOFT_CLIP || op == BAM_CHARD_CLIP || op == BAM_CPAD)
                return true;
        }
        const int nm = nm_of(b);
        if (nm) return true;
        return nm + cigar_mismatch(b) > 0;
    }

    // isRefRead (reader.cpp:620-656): a primary, non-duplicate, QC-passing mapped read with few edits (the NM tag
    // against -n and against int(length * -u) + 1 when present; NM <= 2, <= 2 non-M CIGAR bases) and, when the
    // CIGAR has more than two elements, no I or D among them
    bool is_ref_read(const BamRecord &b) const
    {
        if (b.flag & (BAM_FSECONDARY | BAM_FQCFAIL | BAM_FDUP)) return false;
        int64_t nm = 0;
        if (b.aux_int("NM", nm)) {
            const int max_edits = (int)(b.l_seq * S.max_mismatch_rate) + 1;
            if (nm > S.nm || nm > max_edits) return false;
        }
        if (b.cigar.size() > 2)
            for (uint32_t c : b.cigar)
                if ((c & 15) == BAM_CINS || (c & 15) == BAM_CDEL) return false;
        return !(b.flag & BAM_FUNMAP) && nm_of(b) <= 2 && cigar_mismatch(b) <= 2;
    }
    // build_record_RefRead (reader.cpp:903-923); the anchor only decides whether this is called
    void add_ref_read(const BamRecord &ref, const std::string &tag, IngestedReads &out) const
    {
        if (ref.mapq < S.min_anchor_quality) return;
        size_t t = 0;
        while (t < out.ref_tags.size() && out.ref_tags[t] != tag) t++;
        if (t == out.ref_tags.size()) out.ref_tags.push_back(tag);
        RefRead r = { (uint32_t)ref.pos, (uint16_t)ref.l_seq, (uint16_t)t };
        out.ref_reads.push_back(r);
    }

    // build_record_SR(mapped_read, unmapped_read): false only for the fatal "insert size <= read length"
    bool build_record(const BamFile &bam, const BamRecord &mapped, const BamRecord &unmapped, int chr_id,
                      uint64_t chr_padded_size, int insert_size, const std::string &tag, IngestedReads &out)
    {
        (void)bam;
        if ((short)mapped.mapq < (short)S.min_anchor_quality) return true;
        std::string name = "@" + unmapped.qname;
        if (unmapped.flag & BAM_FREAD1) name += "/1";
        else if (unmapped.flag & BAM_FREAD2) name += "/2";
        static const char nt16[] = "=ACMGRSVTWYHKDBN";
        std::string seq((size_t)unmapped.l_seq, ' ');
        for (int i = 0; i < unmapped.l_seq; i++) seq[(size_t)i] = nt16[(unmapped.seq4[(size_t)i >> 1] >> ((~i & 1) << 2)) & 15];
        // "rudimentary n filter": leading / trailing N's go, more than 10 % N's or fewer than 22 bases: no record
        int length = unmapped.l_seq;
        size_t lead = 0;
        while (lead < seq.size() && seq[lead] == 'N') lead++;
        seq.erase(0, lead);
        length -= (int)lead;
        if (!seq.empty()) {
            // the reference indexes c_sequence[length - 1]: equal to the last character here
            while (length > 0 && seq[(size_t)length - 1] == 'N') {
                seq.erase((size_t)length - 1, 1);
                length--;
            }
        }
        int n_count = 0;
        for (char c : seq) n_count += c == 'N';
        const int max_ns = (int)(length * .10);
        if (n_count > max_ns || length < 22) return true;
        if (unmapped.flag & BAM_FREVERSE) pg_adapter::rc_in_place(seq);
        // setUnmatchedSeq: trailing non-alphanumerics go (a reverse-complemented IUPAC code becomes 0)
        while (!seq.empty() && !isalnum((unsigned char)seq[seq.size() - 1])) seq.resize(seq.size() - 1);
        unsigned rel_pos = (unsigned)mapped.pos;
        char strand = '+';
        if (mapped.flag & BAM_FREVERSE) {
            strand = '-';
            int rlen = 0;                                    // bam_cigar2len: M + I + S - D
            for (uint32_t c : mapped.cigar) {
                const int op = c & 15;
                if (op == BAM_CMATCH || op == BAM_CINS || op == BAM_CSOFT_CLIP) rlen += (int)(c >> 4);
                if (op == BAM_CDEL) rlen -= (int)(c >> 4);
            }
            rel_pos += (unsigned)rlen;
        }
        if (insert_size <= length) {
            error = "the insert size is only " + std::to_string(insert_size) + " while the read length is " + std::to_string(length);
            return false;
        }
        const unsigned biol = (unsigned)(chr_padded_size - 2ull * S.spacer);
        if (rel_pos > biol) rel_pos = biol;
        out.batch.seq.insert(out.batch.seq.end(), seq.begin(), seq.end());
        out.batch.off.push_back(out.batch.seq.size());
        out.batch.strand.push_back((uint8_t)strand);
        out.batch.pos.push_back((int32_t)rel_pos);
        out.batch.isz.push_back((int16_t)insert_size);
        out.batch.chr.push_back(chr_id);                     // FragName = the anchor's reference = the window's chromosome
        out.names.push_back(name);
        out.ms.push_back((int16_t)mapped.mapq);
        out.tags.push_back(tag);
        return true;
    }
};

}  // namespace pgh
#endif

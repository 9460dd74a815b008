// synth_bam -- load generator for the BAM-fed path (BASELINE configs[4] shape): a coordinate-sorted BAM + .bai + FASTA of
// a random reference with read pairs at a given spacing; a share of the mates are "weird" in Pindel's sense
// (src/reader.cpp:658-690): mapped with mismatches (NM > 0), soft-clipped, or unmapped split reads across a deletion /
// short insertion (the SV reads the search is for).  Not a parity fixture: tests/bam_writer.py + the reference's own
// demo BAM cover the decoder; this writes gigabytes quickly (BGZF blocks compressed on all host threads).
//   synth_bam <out_prefix> <ref_len> <n_pairs> [read_len=150] [seed=1]
// writes <prefix>.fa, <prefix>.fa.fai, <prefix>.bam, <prefix>.bam.bai, <prefix>.cfg ("<prefix>.bam 500 SYN")
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
    uint64_t next()
    {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        return s;
    }
    uint32_t below(uint32_t n) { return (uint32_t)((next() >> 16) % n); }
    double unit() { return (double)(next() >> 11) / 9007199254740992.0; }
};

static int reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (int)(beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (int)(beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (int)(beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (int)(beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (int)(beg >> 26);
    return 0;
}

struct Rec {                 // where a record sits in the uncompressed stream, for the index
    int64_t pos, end;
    uint64_t upos;           // offset in the uncompressed record stream
    uint32_t len;
};

static const char NT16[] = "=ACMGRSVTWYHKDBN";
static uint8_t nt16_of(char c)
{
    switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8; default: return 15; }
}
static char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }

static void put32(std::string &o, uint32_t v) { o.append((const char *)&v, 4); }
static void put16(std::string &o, uint16_t v) { o.append((const char *)&v, 2); }

// one BAM record appended to `o`
static void record(std::string &o, const std::string &name, int flag, int64_t pos, int mapq, const std::vector<uint32_t> &cigar,
                   const std::string &seq, int64_t mpos, int tlen, int nm, int64_t end)
{
    std::string b;
    put32(b, 0);                                   // refID
    put32(b, (uint32_t)pos);
    b.push_back((char)(name.size() + 1));
    b.push_back((char)mapq);
    put16(b, (uint16_t)reg2bin(std::max<int64_t>(pos, 0), std::max<int64_t>(end, 1)));
    put16(b, (uint16_t)cigar.size());
    put16(b, (uint16_t)flag);
    put32(b, (uint32_t)seq.size());
    put32(b, 0);                                   // mate refID
    put32(b, (uint32_t)mpos);
    put32(b, (uint32_t)tlen);
    b += name;
    b.push_back(0);
    for (uint32_t c : cigar) put32(b, c);
    for (size_t i = 0; i < seq.size(); i += 2)
        b.push_back((char)((nt16_of(seq[i]) << 4) | (i + 1 < seq.size() ? nt16_of(seq[i + 1]) : 0)));
    b.append(seq.size(), (char)0xff);
    if (nm >= 0) {
        b += "NMC";
        b.push_back((char)nm);
    }
    put32(o, (uint32_t)b.size());
    o += b;
}

static std::string bgzf(const char *data, size_t n)
{
    std::string out(n + 1024, 0);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = (Bytef *)data;
    zs.avail_in = (uInt)n;
    zs.next_out = (Bytef *)&out[18];
    zs.avail_out = (uInt)(out.size() - 18 - 8);
    deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    const uint8_t hdr[16] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0 };
    memcpy(&out[0], hdr, 16);
    const uint16_t bsize = (uint16_t)(18 + clen + 8 - 1);
    memcpy(&out[16], &bsize, 2);
    const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), (const Bytef *)data, (uInt)n), isize = (uint32_t)n;
    memcpy(&out[18 + clen], &crc, 4);
    memcpy(&out[18 + clen + 4], &isize, 4);
    out.resize(18 + clen + 8);
    return out;
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: synth_bam <out_prefix> <ref_len> <n_pairs> [read_len=150] [seed=1]\n");
        return 2;
    }
    const std::string prefix = argv[1];
    const int64_t ref_len = atoll(argv[2]);
    const int64_t n_pairs = atoll(argv[3]);
    const int L = argc > 4 ? atoi(argv[4]) : 150;
    Rng rng(argc > 5 ? (uint64_t)atoll(argv[5]) : 1);
    std::string ref((size_t)ref_len, 'A');
    for (int64_t i = 0; i < ref_len; i++) ref[(size_t)i] = "ACGT"[rng.next() >> 62];
    {
        FILE *f = fopen((prefix + ".fa").c_str(), "wb");
        fprintf(f, ">chrS\n");
        for (int64_t i = 0; i < ref_len; i += 60) {
            fwrite(ref.data() + i, 1, (size_t)std::min<int64_t>(60, ref_len - i), f);
            fputc('\n', f);
        }
        fclose(f);
        f = fopen((prefix + ".fa.fai").c_str(), "w");
        fprintf(f, "chrS\t%lld\t6\t60\t61\n", (long long)ref_len);
        fclose(f);
        f = fopen((prefix + ".cfg").c_str(), "w");
        fprintf(f, "%s.bam\t500\tSYN\n", prefix.substr(prefix.rfind('/') == std::string::npos ? 0 : prefix.rfind('/') + 1).c_str());
        fclose(f);
    }
    // pairs at a regular spacing with jitter below the spacing: first mates and second mates (+300) are two monotone
    // streams; the records come out coordinate-sorted by merging them
    const double spacing = (double)(ref_len - 2000 - 12000) / (double)n_pairs;
    std::string ustream;                           // uncompressed BAM (header + records), cut into blocks below
    {
        std::string text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chrS\tLN:" + std::to_string(ref_len) + "\n";
        ustream += "BAM\1";
        put32(ustream, (uint32_t)text.size());
        ustream += text;
        put32(ustream, 1);
        put32(ustream, 5);
        ustream += std::string("chrS") + '\0';
        put32(ustream, (uint32_t)ref_len);
    }
    const size_t header_bytes = ustream.size();
    struct Pending { int64_t pos; std::string bytes; int64_t end; };
    std::vector<Pending> second;                   // second mates waiting for their turn (sorted by construction)
    size_t second_head = 0;
    auto flush_second = [&](int64_t upto) {
        while (second_head < second.size() && second[second_head].pos <= upto) {
            Pending &p = second[second_head++];
            ustream += p.bytes;
            std::string().swap(p.bytes);
        }
        if (second_head > 4096 && second_head * 2 > second.size()) {
            second.erase(second.begin(), second.begin() + (long)second_head);
            second_head = 0;
        }
    };
    int64_t n_split = 0, n_weird = 0;
    for (int64_t k = 0; k < n_pairs; k++) {
        const int64_t p = 1000 + (int64_t)(k * spacing) + (int64_t)(rng.unit() * spacing * 0.9);
        flush_second(p);
        const std::string name = "q" + std::to_string(k);
        const double u = rng.unit();
        const std::vector<uint32_t> full = { (uint32_t)L << 4 };
        std::string a(ref, (size_t)p, (size_t)L), bytes;
        const int64_t mp = p + 300;
        if (u < 0.08) {
            // unmapped second mate: a split read across a deletion (70 %) or with a short insertion (30 %), on the
            // reverse strand as a mate of a forward anchor would be sequenced
            n_split++;
            const int cut = 30 + (int)rng.below((uint32_t)(L - 60));
            std::string s;
            if (rng.unit() < 0.7) {
                const int64_t del = (int64_t)std::min(1e4, std::max(1.0, std::exp(rng.unit() * 9.2)));
                s = ref.substr((size_t)mp, (size_t)cut) + ref.substr((size_t)(mp + cut + del), (size_t)(L - cut));
            } else {
                const int ins = 1 + (int)rng.below(20);
                std::string x((size_t)ins, 'A');
                for (char &c : x) c = "ACGT"[rng.below(4)];
                s = ref.substr((size_t)mp, (size_t)cut) + x + ref.substr((size_t)(mp + cut), (size_t)(L - cut - ins));
            }
            std::string rcs(s.rbegin(), s.rend());
            for (char &c : rcs) c = comp(c);
            record(ustream, name, 1 | 8 | 64, p, 60, full, a, p, 0, 0, p + L);             // anchor: paired, mate unmapped, read1
            record(ustream, name, 1 | 4 | 32 | 128, p, 0, {}, rcs, p, 0, -1, p + 1);        // unmapped mate placed at the anchor
            continue;
        }
        int nm_b = 0;
        std::string b(ref, (size_t)mp, (size_t)L);
        std::vector<uint32_t> cig_b = full;
        if (u < 0.65) {
            // a "weird" mapped mate: one or two mismatches (NM > 0), now and then a soft clip
            n_weird++;
            nm_b = 1 + (int)rng.below(2);
            for (int e = 0; e < nm_b; e++) {
                const size_t at = rng.below((uint32_t)L);
                b[at] = b[at] == 'A' ? 'C' : 'A';
            }
            if (rng.unit() < 0.1) {
                const uint32_t clip = 10 + rng.below(30);
                cig_b = { (clip << 4) | 4u, ((uint32_t)L - clip) << 4 };
                for (uint32_t i = 0; i < clip; i++) b[i] = "ACGT"[rng.below(4)];
            }
        }
        record(ustream, name, 1 | 2 | 32 | 64, p, 60, full, a, mp, 300 + L, 0, p + L);
        Pending pd;
        pd.pos = mp;
        pd.end = mp + L;
        record(pd.bytes, name, 1 | 2 | 16 | 128, mp, 60, cig_b, b, p, -(300 + L), nm_b, mp + L);
        second.push_back(std::move(pd));
    }
    flush_second(ref_len);
    // the index wants every record's place in the uncompressed stream: walk the stream once (simple and certain)
    std::vector<Rec> recs;
    recs.reserve((size_t)(2 * n_pairs));
    for (size_t at = header_bytes; at < ustream.size();) {
        uint32_t bs;
        memcpy(&bs, &ustream[at], 4);
        int32_t pos;
        memcpy(&pos, &ustream[at + 8], 4);
        uint16_t ncig, flag;
        memcpy(&ncig, &ustream[at + 16], 2);
        memcpy(&flag, &ustream[at + 18], 2);
        uint8_t lname = (uint8_t)ustream[at + 12];
        int64_t span = 0;
        for (int c = 0; c < ncig; c++) {
            uint32_t v;
            memcpy(&v, &ustream[at + 36 + lname + 4 * c], 4);
            const int op = v & 15;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += v >> 4;
        }
        recs.push_back({ pos, pos + ((flag & 4) || !span ? 1 : span), at, bs + 4 });
        at += bs + 4;
    }
    // BGZF: blocks of <= 0xff00 uncompressed bytes, compressed on all threads, written in order
    const size_t BLK = 0xff00;
    const size_t n_blocks = (ustream.size() + BLK - 1) / BLK;
    std::vector<std::string> comp_blocks(n_blocks);
    {
        const unsigned nt = std::max(1u, std::thread::hardware_concurrency());
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                for (size_t b = t; b < n_blocks; b += nt)
                    comp_blocks[b] = bgzf(ustream.data() + b * BLK, std::min(BLK, ustream.size() - b * BLK));
            });
        for (std::thread &x : th) x.join();
    }
    std::vector<uint64_t> coff(n_blocks + 1, 0);
    for (size_t b = 0; b < n_blocks; b++) coff[b + 1] = coff[b] + comp_blocks[b].size();
    {
        FILE *f = fopen((prefix + ".bam").c_str(), "wb");
        for (const std::string &c : comp_blocks) fwrite(c.data(), 1, c.size(), f);
        static const uint8_t eof[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        fwrite(eof, 1, 28, f);
        fclose(f);
    }
    auto voff = [&](uint64_t upos) { return (coff[upos / BLK] << 16) | (upos % BLK); };
    // .bai: bins -> chunks (consecutive records of a bin merge into one chunk), 16-kb linear index
    std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
    std::vector<uint64_t> linear;
    for (const Rec &r : recs) {
        const uint32_t bin = (uint32_t)reg2bin(std::max<int64_t>(r.pos, 0), std::max<int64_t>(r.end, 1));
        const uint64_t vb = voff(r.upos), ve = voff(r.upos + r.len);
        auto &ch = bins[bin];
        if (!ch.empty() && ch.back().second == vb) ch.back().second = ve;
        else ch.push_back({ vb, ve });
        for (int64_t w = std::max<int64_t>(r.pos, 0) >> 14; w <= (std::max<int64_t>(r.end, 1) - 1) >> 14; w++) {
            if ((size_t)w >= linear.size()) linear.resize((size_t)w + 1, 0);
            if (linear[(size_t)w] == 0) linear[(size_t)w] = vb;
        }
    }
    for (size_t w = 1; w < linear.size(); w++)
        if (linear[w] == 0) linear[w] = linear[w - 1];
    {
        FILE *f = fopen((prefix + ".bam.bai").c_str(), "wb");
        fwrite("BAI\1", 1, 4, f);
        const uint32_t one = 1, nb = (uint32_t)bins.size();
        fwrite(&one, 4, 1, f);
        fwrite(&nb, 4, 1, f);
        for (const auto &kv : bins) {
            const uint32_t nc = (uint32_t)kv.second.size();
            fwrite(&kv.first, 4, 1, f);
            fwrite(&nc, 4, 1, f);
            for (const auto &c : kv.second) {
                fwrite(&c.first, 8, 1, f);
                fwrite(&c.second, 8, 1, f);
            }
        }
        const uint32_t ni = (uint32_t)linear.size();
        fwrite(&ni, 4, 1, f);
        fwrite(linear.data(), 8, linear.size(), f);
        fclose(f);
    }
    printf("synth_bam: %lld pairs (%lld split-read mates, %lld weird mapped mates), %zu records, %.2f GB uncompressed, %.2f GB BAM\n",
           (long long)n_pairs, (long long)n_split, (long long)n_weird, recs.size(), ustream.size() / 1e9, coff[n_blocks] / 1e9);
    return 0;
}

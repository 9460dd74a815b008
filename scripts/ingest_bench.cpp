// Diagnostics: time BamIngest::read_window (pg_bam.hpp) on an indexed BAM and print a digest of what it produced.
//   g++ -O2 -std=c++17 -pthread -Ipindel_amd/csrc/host -Iinclude scripts/ingest_bench.cpp -lz -o /tmp/ingest_bench
//   /tmp/ingest_bench file.bam chr_name chr_padded_size win_start win_end insert_size
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include "pg_bam.hpp"

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    std::string err;
    pgh::BamFile bam;
    if (!bam.open(argv[1], err, true)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    pgh::BamIngestSettings st;
    for (int rep = 0; rep < 3; rep++) {
        pgh::ingest_timing() = pgh::IngestTiming();
        pgh::BamIngest ing(st);
        pgh::IngestedReads out;
        out.clear();
        const auto t0 = std::chrono::steady_clock::now();
        if (!ing.read_window(bam, argv[2], 0, strtoull(argv[3], 0, 10), atoll(argv[4]), atoll(argv[5]), atoi(argv[6]), "tag", out)) {
            fprintf(stderr, "%s\n", ing.error.c_str());
            return 1;
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        unsigned long long h = 1469598103934665603ull;
        auto mix = [&](const void *p, size_t n) { const unsigned char *c = (const unsigned char *)p; for (size_t i = 0; i < n; i++) h = (h ^ c[i]) * 1099511628211ull; };
        mix(out.batch.seq.data(), out.batch.seq.size());
        mix(out.batch.off.data(), out.batch.off.size() * 8);
        mix(out.batch.pos.data(), out.batch.pos.size() * 4);
        mix(out.batch.strand.data(), out.batch.strand.size());
        mix(out.batch.isz.data(), out.batch.isz.size() * 2);
        mix(out.ms.data(), out.ms.size() * sizeof(out.ms[0]));
        for (const std::string &n : out.names) mix(n.data(), n.size());
        mix(out.ref_reads.data(), out.ref_reads.size() * sizeof(out.ref_reads[0]));
        printf("%.3f s: %zu reads, %zu reference reads, digest %016llx | inflate + decode %.3f, selection %.3f, layout %.3f\n", s, out.size(),
               out.ref_reads.size(), h, pgh::ingest_timing().inflate_decode, pgh::ingest_timing().select, pgh::ingest_timing().layout);
    }
    return 0;
}

// adapter_ref_shapes.cpp -- TEST INFRASTRUCTURE.  The reference-side binding INTEGRATION.md sections 3 and 4 show,
// instantiated against tests/ref_shapes.hpp (the public interface of the reference's SPLIT_READ / SortedUniquePoints /
// UniquePoint, src/pindel.h:137-197, 265-383) instead of pgh::SplitRead:
//   * `g++ -fsyntax-only` of this file is the CPU test "the adapters compile against the reference's types"
//     (tests/test_cpu_suite.py::test_adapter_compiles_against_reference_shapes);
//   * built and linked with libpindel_pg.so it is the GPU test's driver (tests/test_gpu_parity.py::
//     test_adapter_on_reference_shapes): FASTA + a plain read table in, the flow of the reference out --
//     ReadBuffer::flush in `flush`-sized batches (seam 1), keep the reads with a close end, SearchFarEnds on their
//     union (seam 2, both overloads) -- one line per read with the sequence as it was left and every UniquePoint.
//
// read table: one read per line  "<name> <chr name> <strand> <MatchedRelPos> <InsertSize> <sequence>"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "pindel_pg.h"
#include "ref_shapes.hpp"
#include "pg_adapter.hpp"

static std::vector<Chromosome *> g_chr;              // stands in for g_genome
static pg_ctx *g_pg = NULL;

static UniquePoint pg_make_point(const pg_point &p)
{
   return UniquePoint(g_chr[(unsigned)p.chr_id], p.length, p.abs_loc, p.direction, p.strand, p.mismatches);
}
static int pg_chr_of(const SPLIT_READ &r)
{
   for (size_t i = 0; i < g_chr.size(); i++)
      if (g_chr[i]->getName() == r.FragName) return (int)i;
   return -1;
}

static void print_points(std::ostream &os, const SortedUniquePoints &pts)
{
   os << pts.size();
   for (unsigned i = 0; i < pts.size(); i++) {
      const UniquePoint &u = pts[i];
      os << ' ' << u.LengthStr << ',' << u.AbsLoc << ',' << u.Direction << ',' << u.Strand << ',' << u.Mismatches << ','
         << u.chromosome_p->getID();
   }
}

int main(int argc, char **argv)
{
   if (argc < 5) {
      std::fprintf(stderr, "usage: %s ref.fa reads.txt flush_size out.txt [both-overloads]\n", argv[0]);
      return 2;
   }
   const size_t flush = (size_t)std::atol(argv[3]);
   pg_params p;
   pg_default_params(&p);
   if (pg_create(&p, &g_pg) != PG_OK) { std::fprintf(stderr, "pg_create failed\n"); return 3; }
   if (pg_load_fasta(g_pg, argv[1]) != PG_OK) { std::fprintf(stderr, "%s\n", pg_last_error(g_pg)); return 3; }
   for (int c = 0; c < pg_reference_n_chr(g_pg); c++) g_chr.push_back(new Chromosome(pg_reference_name(g_pg, c), (unsigned)c));

   std::vector<SPLIT_READ> raw;
   {
      std::ifstream in(argv[2]);
      std::string line;
      while (std::getline(in, line)) {
         std::istringstream ls(line);
         SPLIT_READ r;
         std::string seq;
         int pos = 0, isz = 0;
         ls >> r.Name >> r.FragName >> r.MatchedD >> pos >> isz >> seq;
         r.MatchedRelPos = (unsigned)pos;
         r.InsertSize = (short)isz;
         r.setUnmatchedSeq(seq);
         raw.push_back(r);
      }
   }

   // seam 1 the way ReadBuffer does it: flush by flush, the reads that got a close end are kept
   std::vector<SPLIT_READ> kept;                    // = state.Reads_SR
   std::vector<SPLIT_READ> kept_first;              // the first flush alone, for the close_result overload
   pg_result *first_result = NULL;
   for (size_t lo = 0; lo < raw.size(); lo += flush) {
      std::vector<SPLIT_READ> m_rawreads(raw.begin() + (long)lo, raw.begin() + (long)std::min(raw.size(), lo + flush));
      pg_result *res = NULL;
      if (pg_adapter::CloseEndBatch(g_pg, m_rawreads, pg_chr_of, pg_make_point, &res) != PG_OK) {
         std::fprintf(stderr, "%s\n", pg_last_error(g_pg));
         return 4;
      }
      for (size_t i = 0; i < m_rawreads.size(); i++) {
         raw[lo + i] = m_rawreads[i];               // (the test also wants the reads that were dropped)
         if (m_rawreads[i].hasCloseEnd()) kept.push_back(m_rawreads[i]);
      }
      if (lo == 0 && argc > 5) {
         kept_first = m_rawreads;
         first_result = res;
      } else
         pg_result_free(res);
   }
   // seam 2 on the filtered union (src/pindel.cpp:1888)
   if (pg_adapter::SearchFarEnds(g_pg, kept, pg_chr_of, pg_make_point, (const pg_windows *)NULL) != PG_OK) {
      std::fprintf(stderr, "%s\n", pg_last_error(g_pg));
      return 5;
   }
   // ... and the overload that keeps one flush together
   if (first_result) {
      if (pg_adapter::SearchFarEnds(g_pg, kept_first, pg_chr_of, pg_make_point, first_result, (const pg_windows *)NULL) != PG_OK) {
         std::fprintf(stderr, "%s\n", pg_last_error(g_pg));
         return 6;
      }
      pg_result_free(first_result);
   }

   std::ofstream out(argv[4]);
   size_t k = 0;
   for (size_t i = 0; i < raw.size(); i++) {
      const SPLIT_READ &r = raw[i].hasCloseEnd() ? kept[k++] : raw[i];
      out << r.Name << ' ' << r.getUnmatchedSeq() << " C ";
      print_points(out, r.UP_Close);
      out << " F ";
      print_points(out, r.UP_Far);
      out << '\n';
   }
   for (size_t i = 0; i < kept_first.size(); i++) {
      out << "first " << kept_first[i].Name << " F ";
      print_points(out, kept_first[i].UP_Far);
      out << '\n';
   }
   pg_destroy(g_pg);
   return 0;
}

// ref_shapes.hpp -- TEST INFRASTRUCTURE.  The *public interface* of the three reference types the seam adapters touch,
// written from the declarations in src/pindel.h (UniquePoint :137-158, SortedUniquePoints :160-197, SPLIT_READ :265-383):
// the same member names, signatures and access levels, own bodies.  Deliberately NOT a superset: SortedUniquePoints has
// exactly push_back / size / MaxLen / NumMismatch / empty / operator[] / clear / swap (no reserve, no iterators, no
// data()) and keeps its storage private in a std::deque, so that an adapter which leans on std::vector's interface
// fails to compile here exactly as it would against the reference.  Fields the adapters never touch are left out.
#ifndef PG_TEST_REF_SHAPES_HPP
#define PG_TEST_REF_SHAPES_HPP

#include <cctype>
#include <deque>
#include <map>
#include <string>

class Chromosome {
public:
   Chromosome(const std::string &name, unsigned id) : m_name(name), m_id(id) {}
   const std::string &getName() const { return m_name; }
   unsigned getID() const { return m_id; }             // (test helper; the reference looks chromosomes up by name)
private:
   std::string m_name;
   unsigned m_id;
};

struct UniquePoint {
   const Chromosome *chromosome_p;
   short LengthStr;
   unsigned int AbsLoc;
   char Direction;
   char Strand;
   short Mismatches;
   UniquePoint(const Chromosome *chromosome_ptr, const short lengthStr, const unsigned int absLoc, const char direction,
               const char strand, const short mismatches)
      : chromosome_p(chromosome_ptr), LengthStr(lengthStr), AbsLoc(absLoc), Direction(direction), Strand(strand),
        Mismatches(mismatches) {}
   UniquePoint() : chromosome_p(NULL), LengthStr(0), AbsLoc(0), Direction('N'), Strand('N'), Mismatches(0) {}
};

class SortedUniquePoints {
public:
   void push_back(const UniquePoint &up) { m_positions.push_back(up); }
   unsigned int size() const { return (unsigned int)m_positions.size(); }
   unsigned int MaxLen() const { return m_positions.empty() ? 0u : (unsigned int)m_positions.back().LengthStr; }
   unsigned int NumMismatch() const { return m_positions.empty() ? 0u : (unsigned int)m_positions.back().Mismatches; }
   bool empty() const { return m_positions.empty(); }
   const UniquePoint &operator[](const unsigned int pos) const { return m_positions[pos]; }
   UniquePoint &operator[](const unsigned int pos) { return m_positions[pos]; }
   void clear() { m_positions.clear(); }
   void swap(SortedUniquePoints &otherPV) { m_positions.swap(otherPV.m_positions); }
private:
   std::deque<UniquePoint> m_positions;
};

struct SPLIT_READ {
   SPLIT_READ()
      : MapperSplit(false), MatchedD(0), MatchedFarD(0), MatchedRelPos(0), MS(0), InsertSize(0), Thickness(0), UniqueRead(false),
        Used(false), LeftMostPos(0), ReadLength(0), ReadLengthMinus(0) {}
   bool MapperSplit;
   std::string FragName;
   std::string FarFragName;
   std::string Name;
   // as the reference's: stores the sequence minus trailing non-alphanumeric characters and refreshes the lengths
   void setUnmatchedSeq(const std::string &unmatchedSeq)
   {
      UnmatchedSeq = unmatchedSeq;
      while (!UnmatchedSeq.empty() && !isalnum((unsigned char)UnmatchedSeq[UnmatchedSeq.size() - 1]))
         UnmatchedSeq.resize(UnmatchedSeq.size() - 1);
      ReadLength = (short)UnmatchedSeq.size();
      ReadLengthMinus = (short)(ReadLength - 1);
   }
   const std::string &getUnmatchedSeq() const { return UnmatchedSeq; }
   char MatchedD;
   char MatchedFarD;
   unsigned int MatchedRelPos;
   short MS;
   short InsertSize;
   std::string Tag;
   std::map<std::string, unsigned> SampleName2Number;
   unsigned Thickness;
   SortedUniquePoints UP_Close;
   SortedUniquePoints UP_Far;
   short getReadLength() const { return ReadLength; }
   short getReadLengthMinus() const { return ReadLengthMinus; }
   bool UniqueRead;
   bool Used;
   int LeftMostPos;
   unsigned int getLastAbsLocCloseEnd() const { return UP_Close[UP_Close.size() - 1].AbsLoc; }
   bool hasCloseEnd() const { return !UP_Close.empty(); }
   unsigned int MaxLenCloseEnd() const { return UP_Close.MaxLen(); }
   unsigned int MaxLenFarEnd() const { return UP_Far.MaxLen(); }
   bool goodFarEndFound() const { return MaxLenFarEnd() + MaxLenCloseEnd() >= UnmatchedSeq.size(); }
   std::string UnmatchedSeq;
private:
   short ReadLength;
   short ReadLengthMinus;
};

#endif

"""Reads that BEGIN or END with characters outside ACGTN, and what the reference does with them (round 6: the last known
divergence of oracle and kernels from Pindel 0.2.5b9 is closed).

Source (written out by hand from it):
  * GetCloseEnd, src/pindel.cpp:2531-2575: attempts (R0, seq), then "setUnmatchedSeq(ReverseComplement(seq))" and (R0, seq'),
    then (R1, seq'), then setUnmatchedSeq(ReverseComplement(seq')) and (R1, seq'').
  * ReverseComplement, :2037-2048 with Convert2RC4N, :966-970: every character that is not one of ACGTN becomes NUL.
  * setUnmatchedSeq, :142-169: trailing characters that are not alphanumeric are stripped; ReadLength, MAX_SNP_ERROR and
    TOTAL_SNP_ERROR_CHECKED are recomputed from the new length.
So for a read s = J + X + K (J / K = the leading / trailing runs of characters outside ACGTN, X clean at both ends):
  seq'  = RC(s) without its last |J| characters   = NUL^|K| + RC(X)          length n - |J|
  seq'' = RC(seq') without its last |K| characters = X (inner junk -> NUL)    length n - |J| - |K|
and a read the first attempt does not place is searched, from then on, exactly like the SHORTER read: attempts 1 and 2 see seq',
attempt 3 sees seq'', the far end sees whichever GetCloseEnd left.  The cases below are built so that this can be stated as an
equivalence with a CLEAN read whose result is pinned elsewhere (gold reports, the rest of the suite):

  lead   s = J + RC(c)        c a clean split read that keeps its close end at attempt 0 (rc_flag 0).  seq' = c: whenever attempt 0
                              on s finds nothing, s must give exactly c's UP_Close and UP_Far, rc_flag 1, ReadLength |c|.
  trail  s = c + K, anchor moved one insert size so that c's close end lies in the R = 1 window only: attempts 0-2 find nothing
                              (no seed: the first consumed character is junk / NUL), attempt 3 sees seq'' = c: s must give what the
                              clean read c gives with the same moved anchor when THAT is found at attempt 3, rc_flag 2, ReadLength |c|.
  126    lead with |c| = 125, |J| = 1: ReadLength drops from 126 to 125 and g_maxMismatch from 5 to 4 (one level fewer).
"""
import numpy as np

from pindel_amd import synth
from pindel_amd.synth import ReadBatch

_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b


def rc_ref(s: bytes) -> bytes:
    """ReverseComplement with Convert2RC4N: characters outside ACGTN become NUL."""
    return _COMP[np.frombuffer(s, dtype=np.uint8)][::-1].tobytes()


def batch_of(seqs, strand, pos, isz, chr_id=0):
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return ReadBatch(seq=np.frombuffer(b"".join(seqs), dtype=np.uint8).copy(), seq_off=off,
                     anchor_strand=np.asarray(strand, dtype=np.uint8), anchor_pos=np.asarray(pos, dtype=np.int32),
                     insert_size=np.asarray(isz, dtype=np.int16),
                     chr_id=np.full(len(seqs), chr_id, dtype=np.int32) if np.isscalar(chr_id) else np.asarray(chr_id, dtype=np.int32))


def seqs_of(batch):
    return [batch.seq[int(batch.seq_off[i]):int(batch.seq_off[i + 1])].tobytes() for i in range(batch.n)]


def clean_reads(ref, n, read_len, seed):
    """split reads (deletions, insertions, duplications) in the orientation attempt 0 tries"""
    return synth.make_reads(ref, n, seed=seed, read_len=read_len, mix=(0.6, 0.2, 0.2, 0.0, 0.0), rc_retry_frac=0.0, n_rate=0.0)


def lead_case(clean, junk: bytes):
    """s = junk + RC(c) for every clean read c, same anchors"""
    return batch_of([junk + rc_ref(c) for c in seqs_of(clean)], clean.anchor_strand, clean.anchor_pos, clean.insert_size)


def moved(clean):
    """the same reads with the anchor one insert size further from the close end: R = 0 misses it, R = 1 holds it"""
    plus = clean.anchor_strand == ord("+")
    pos = np.where(plus, clean.anchor_pos.astype(np.int64) + clean.insert_size, clean.anchor_pos.astype(np.int64) - clean.insert_size)
    return ReadBatch(seq=clean.seq, seq_off=clean.seq_off, anchor_strand=clean.anchor_strand, anchor_pos=pos.astype(np.int32),
                     insert_size=clean.insert_size, chr_id=clean.chr_id)


def trail_case(clean_moved, junk: bytes):
    return batch_of([c + junk for c in seqs_of(clean_moved)], clean_moved.anchor_strand, clean_moved.anchor_pos, clean_moved.insert_size)


def inner_case(clean, at: int, ch: bytes = b"R"):
    """one character outside ACGTN INSIDE the read: the length never changes, two reverse complements leave a NUL there"""
    return batch_of([c[:at] + ch + c[at + 1:] for c in seqs_of(clean)], clean.anchor_strand, clean.anchor_pos, clean.insert_size)


def same_points(a, i, b, j):
    for which in ("close", "far"):
        ca, cb = int(a[which + "_cnt"][i]), int(b[which + "_cnt"][j])
        if ca != cb or a[which + "_pts"][i][:ca].tobytes() != b[which + "_pts"][j][:cb].tobytes():
            return False
    return True

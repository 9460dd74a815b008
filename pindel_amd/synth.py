"""Seeded synthetic inputs: a chr20-shaped reference and
one-end-anchored reads carrying deletions, short insertions, tandem duplications,
inversions or nothing, with base errors and N's.  Vectorised numpy; used by the
tests (small sizes) and by bench.py (BASELINE.json configs).  There is no network,
so real WGS data cannot be fetched -- everything here is generated.
"""
from __future__ import annotations

import numpy as np

from .hostio import SPACER, ReadBatch

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def make_reference(length: int, seed: int = 20260927, n_gaps: int = 3, gap_len: int = 50000,
                   repeat_len: int = 2000, n_repeat_copies: int = 4, microsat_len: int = 600,
                   spacer: int = SPACER) -> bytes:
    """i.i.d. ACGT with a few N gaps, one repeat family and an AC microsatellite,
    returned spacer-padded like Chromosome::getSeq()."""
    rng = np.random.default_rng(seed)
    seq = _ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]
    if length > 20 * (gap_len + repeat_len + microsat_len):
        for g in range(n_gaps):
            s = int(length * (g + 1) / (n_gaps + 1.5))
            seq[s:s + gap_len] = ord("N")
        unit = seq[1000:1000 + repeat_len].copy()
        for k in range(n_repeat_copies):
            s = int(length * (0.07 + 0.11 * k))
            seq[s:s + repeat_len] = unit
        s = int(length * 0.61)
        seq[s:s + microsat_len] = np.resize(np.frombuffer(b"AC", dtype=np.uint8), microsat_len)
    pad = np.full(spacer, ord("N"), dtype=np.uint8)
    return np.concatenate([pad, seq, pad]).tobytes()


def make_reads(chr_padded: bytes, n_reads: int, read_len: int = 100, seed: int = 20260927,
               insert_size: int = 500, mix=(0.4, 0.2, 0.1, 0.1, 0.2), error_rate: float = 0.01,
               n_rate: float = 0.001, rc_retry_frac: float = 0.1, chr_id: int = 0,
               max_del: int = 10000, spacer: int = SPACER, read_lens=None,
               chunk: int = 1 << 20) -> ReadBatch:
    """mix = fractions of (D, SI, TD, INV, none).  Reads are emitted as Pindel would
    receive them: sequence in sequencing orientation, anchor strand/position of the
    mapped mate, so that the close end lies within one insert size of the anchor."""
    ref = np.frombuffer(chr_padded, dtype=np.uint8)
    biol = len(ref) - 2 * spacer
    rng = np.random.default_rng(seed)
    seqs, strands, poss = [], [], []
    lens_all = []
    done = 0
    mixc = np.cumsum(np.asarray(mix, dtype=np.float64) / np.sum(mix))
    while done < n_reads:
        n = min(chunk, n_reads - done)
        if read_lens is None:
            L = np.full(n, read_len, dtype=np.int64)
        else:
            L = rng.choice(np.asarray(read_lens, dtype=np.int64), size=n)
        Lmax = int(L.max())
        kind = np.searchsorted(mixc, rng.random(n), side="right").clip(0, 4)
        sp = (rng.integers(20, 81, size=n) * L // 100).clip(12, None)   # split point
        sp = np.minimum(sp, L - 12)
        margin = max_del + 4 * insert_size + 2 * Lmax + 64
        bp = rng.integers(margin, max(biol - margin, margin + 1), size=n) + spacer   # AbsLoc of the break
        # event geometry: read = left part (sp bases ending at bp) + [insert] + right part
        dsize = np.exp(rng.random(n) * np.log(max_del)).astype(np.int64).clip(1, max_del)
        isize = rng.integers(1, 21, size=n)
        j = np.arange(Lmax)[None, :]
        in_read = j < L[:, None]
        left = j < sp[:, None]
        idx = np.zeros((n, Lmax), dtype=np.int64)
        comp = np.zeros((n, Lmax), dtype=bool)
        junk = np.zeros((n, Lmax), dtype=bool)
        # left part is always ref[bp-sp .. bp)
        idx_left = bp[:, None] - sp[:, None] + j
        # right part by event type
        r = j - sp[:, None]                                  # offset inside the right part
        k = kind[:, None]
        right_D = bp[:, None] + dsize[:, None] + r           # deletion: skip dsize bases
        ins = (r < isize[:, None])                           # short insertion: isize random bases
        right_SI = bp[:, None] + (r - isize[:, None])
        right_TD = bp[:, None] - dsize[:, None].clip(sp[:, None] + 1, None) + r   # jump back: tandem dup
        q = bp[:, None] + dsize[:, None] + 200               # inversion [bp, q): right part is RC of the far side
        right_INV = q - 1 - r
        right_none = bp[:, None] + r                          # plain reference read
        idx_right = np.where(k == 0, right_D, np.where(k == 1, right_SI, np.where(
            k == 2, right_TD, np.where(k == 3, right_INV, right_none))))
        idx = np.where(left, idx_left, idx_right)
        comp = (~left) & (k == 3)
        junk = (~left) & (k == 1) & ins
        # half of the "none" reads are unmappable junk
        all_junk = (kind == 4) & (rng.random(n) < 0.5)
        junk |= all_junk[:, None]
        idx = idx.clip(0, len(ref) - 1)
        bases = ref[idx]
        bases = np.where(comp, _COMP[bases], bases)
        rnd = _ACGT[rng.integers(0, 4, size=(n, Lmax), dtype=np.uint8)]
        bases = np.where(junk, rnd, bases)
        err = rng.random((n, Lmax)) < error_rate
        bases = np.where(err, rnd, bases)
        isn = rng.random((n, Lmax)) < n_rate
        bases = np.where(isn, np.uint8(ord("N")), bases)
        # anchor: '+' anchors sit upstream (close end = left part, read given as RC of the fragment),
        # '-' anchors downstream (close end = right part, read given forward)
        plus = rng.random(n) < 0.5
        slack = rng.integers(0, max(insert_size - Lmax - 20, 1), size=n)
        left_start = bp - sp
        right_end = np.where(kind == 0, bp + dsize, bp) + (L - sp)
        right_end = np.where(kind == 3, bp + dsize + 200, right_end)
        pos_plus = left_start - slack - spacer
        pos_minus = right_end + slack - spacer
        # for '-' anchors of TD/INV reads the close end is the left part seen from the other side;
        # keep them simple: anchor after the left part instead
        pos_minus = np.where((kind == 2) | (kind == 3), bp + slack - spacer, pos_minus)
        pos = np.where(plus, pos_plus, pos_minus).clip(0, biol)
        # sequencing orientation: '+' anchor -> mate read is the reverse complement
        flip = plus ^ (rng.random(n) < rc_retry_frac)
        rev_idx = (L[:, None] - 1 - j).clip(0, Lmax - 1)
        rc = _COMP[np.take_along_axis(bases, rev_idx, axis=1)]
        bases = np.where(flip[:, None], rc, bases)
        flat = bases[in_read]
        seqs.append(flat)
        lens_all.append(L)
        strands.append(np.where(plus, ord("+"), ord("-")).astype(np.uint8))
        poss.append(pos.astype(np.int32))
        done += n
    lens = np.concatenate(lens_all)
    off = np.zeros(n_reads + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    return ReadBatch(seq=np.concatenate(seqs), seq_off=off,
                     anchor_strand=np.concatenate(strands), anchor_pos=np.concatenate(poss),
                     insert_size=np.full(n_reads, insert_size, dtype=np.int16),
                     chr_id=np.full(n_reads, chr_id, dtype=np.int32))

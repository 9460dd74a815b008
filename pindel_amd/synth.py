"""Seeded synthetic inputs: a chr20-shaped reference and
one-end-anchored reads carrying deletions, short insertions, tandem duplications,
inversions or nothing, with base errors and N's.  Written with torch tensor ops so
the same code generates small test inputs on the CPU and the 10 M-read bench inputs
on the GPU in seconds (torch is plumbing here: RNG + gathers).  There is no network,
so real WGS data cannot be fetched -- everything is generated.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .hostio import SPACER, ReadBatch

_COMP = torch.zeros(256, dtype=torch.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b
_ACGT = torch.tensor(list(b"ACGT"), dtype=torch.uint8)


def _gen(seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


# GRCh38 primary assembly, chr1..22, X, Y (lengths of the .fai; only the shape matters here)
GRCH38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
                  138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
                  83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
GRCH38_NAMES = [str(i) for i in range(1, 23)] + ["X", "Y"]


def _plant_repeats(seq, g, device, repeat_frac, n_families=6, divergence=(0.02, 0.15), tract_frac=0.03):
    """Repeat-rich variant.  Pindel's search is LOCAL (a few kb around the anchor), so what
    makes it harder is local repetitiveness: `repeat_frac` of the sequence is overwritten by ARRAYS of diverged
    copies of a few repeat families (consensus 150 bp - 3 kb; 2-6 copies within a few kb of each other, per-copy
    divergence drawn from `divergence`) -- segmental / tandem duplications -- plus low-complexity tracts
    (mono/di/tri-nucleotide runs of 30-300 bp, `tract_frac` of the sequence).  The i.i.d. default has a
    survivor rate of the seed filter that is best-case."""
    length = seq.numel()
    acgt = _ACGT.to(device)
    fam_len = [150, 300, 300, 800, 1500, 3000][:n_families]
    share = [0.15, 0.25, 0.15, 0.15, 0.15, 0.15][:n_families]
    for f in range(n_families):
        Lf = fam_len[f]
        cons = acgt[torch.randint(0, 4, (Lf,), generator=g, device=device)]
        k = max(1, int(length * repeat_frac * share[f] / Lf))        # copies
        n_arr = max(1, k // 4)                                         # arrays of ~4 copies
        done = 0
        while done < n_arr:
            na = min(n_arr - done, max(1, (1 << 22) // Lf))
            centre = torch.randint(8000, max(length - 8000 - Lf, 8001), (na,), generator=g, device=device)
            ncopy = torch.randint(2, 7, (na,), generator=g, device=device)
            for c in range(6):
                use = ncopy > c
                # copies of one array lie within +-(Lf + 2 kb) of its centre (tandem or interspersed)
                off = torch.randint(-(Lf + 2000), Lf + 2000, (na,), generator=g, device=device)
                pos = (centre + off)[use]
                kk = int(pos.numel())
                if kk == 0:
                    continue
                div = divergence[0] + (divergence[1] - divergence[0]) * torch.rand(kk, generator=g, device=device)
                idx = pos[:, None] + torch.arange(Lf, device=device)[None, :]
                mut = torch.rand(kk, Lf, generator=g, device=device) < div[:, None]
                rnd = acgt[torch.randint(0, 4, (kk, Lf), generator=g, device=device)]
                seq[idx.reshape(-1)] = torch.where(mut, rnd, cons[None, :].expand(kk, Lf)).reshape(-1)
            done += na
    n_tr = max(1, int(length * tract_frac / 165))
    pos = torch.randint(0, max(length - 400, 1), (n_tr,), generator=g, device=device)
    tl = torch.randint(30, 300, (n_tr,), generator=g, device=device)
    unit = torch.randint(1, 4, (n_tr,), generator=g, device=device)
    motif = acgt[torch.randint(0, 4, (n_tr, 3), generator=g, device=device)]
    j = torch.arange(300, device=device)[None, :]
    idx = pos[:, None] + j
    val = torch.gather(motif, 1, (j % unit[:, None]).expand(n_tr, 300))
    keep = (j < tl[:, None]).reshape(-1)
    seq[idx.reshape(-1)[keep]] = val.reshape(-1)[keep]


def make_reference(length: int, seed: int = 20260927, n_gaps: int = 3, gap_len: int = 50000,
                   repeat_len: int = 2000, n_repeat_copies: int = 4, microsat_len: int = 600,
                   spacer: int = SPACER, device="cpu", repeat_frac: float = 0.0) -> bytes:
    """i.i.d. ACGT with a few N gaps, one repeat family and an AC microsatellite,
    returned spacer-padded like Chromosome::getSeq() (src/pindel.cpp:297-309).
    repeat_frac > 0: additionally repeat-rich (see _plant_repeats)."""
    g = _gen(seed, device)
    code = torch.randint(0, 4, (length,), generator=g, device=device, dtype=torch.uint8)
    seq = _ACGT.to(device)[code.long()]
    del code
    if repeat_frac > 0:
        _plant_repeats(seq, g, device, repeat_frac)
    if length > 20 * (gap_len + repeat_len + microsat_len):
        for k in range(n_gaps):
            s = int(length * (k + 1) / (n_gaps + 1.5))
            seq[s:s + gap_len] = ord("N")
        unit = seq[1000:1000 + repeat_len].clone()
        for k in range(n_repeat_copies):
            s = int(length * (0.07 + 0.11 * k))
            seq[s:s + repeat_len] = unit
        s = int(length * 0.61)
        ac = torch.tensor([ord("A"), ord("C")], dtype=torch.uint8, device=device)
        seq[s:s + microsat_len] = ac.repeat(microsat_len // 2 + 1)[:microsat_len]
    pad = torch.full((spacer,), ord("N"), dtype=torch.uint8, device=device)
    return torch.cat([pad, seq, pad]).cpu().numpy().tobytes()


def make_genome(lengths, names=None, seed: int = 20260927, device="cpu", repeat_frac: float = 0.0, spacer: int = SPACER):
    """A multi-chromosome reference: [(name, padded_bytes)] with one make_reference per chromosome."""
    names = names or [f"chr{i + 1}" for i in range(len(lengths))]
    return [(nm, make_reference(int(L), seed=seed + 101 * i, device=device, repeat_frac=repeat_frac, spacer=spacer))
            for i, (nm, L) in enumerate(zip(names, lengths))]


def make_reads_genome(chroms, n_reads: int, seed: int = 20260927, device="cpu", interleave: bool = True, **kw):
    """Reads over every chromosome of `chroms`, proportional to its length.  interleave = False keeps the
    reads grouped by chromosome (the order a coordinate-sorted BAM delivers them)."""
    from .hostio import ReadBatch
    sizes = np.array([len(s) for _, s in chroms], dtype=np.float64)
    counts = np.floor(sizes / sizes.sum() * n_reads).astype(np.int64)
    counts[0] += n_reads - counts.sum()
    parts = []
    for c, (_, s) in enumerate(chroms):
        if counts[c] > 0:
            parts.append(make_reads(s, int(counts[c]), seed=seed + 7 * c, chr_id=c, device=device, **kw))
    off = np.concatenate([[0]] + [p.seq_off[1:].astype(np.uint64) + np.uint64(b) for p, b in
                                   zip(parts, np.cumsum([0] + [len(p.seq) for p in parts[:-1]]))]).astype(np.uint64)
    b = ReadBatch(seq=np.concatenate([p.seq for p in parts]), seq_off=off,
                  anchor_strand=np.concatenate([p.anchor_strand for p in parts]),
                  anchor_pos=np.concatenate([p.anchor_pos for p in parts]),
                  insert_size=np.concatenate([p.insert_size for p in parts]),
                  chr_id=np.concatenate([p.chr_id for p in parts]))
    if interleave and b.n:
        rng = np.random.default_rng(seed + 5)
        perm = rng.permutation(b.n)
        lens = b.lengths()
        if (lens == lens[0]).all():
            L = int(lens[0])
            seq = b.seq.reshape(b.n, L)[perm].reshape(-1)
            off = b.seq_off
        else:
            o = b.seq_off.astype(np.int64)
            seq = np.concatenate([b.seq[o[i]:o[i + 1]] for i in perm])
            off = np.concatenate([[0], np.cumsum(lens[perm])]).astype(np.uint64)
        b = ReadBatch(seq=seq, seq_off=off, anchor_strand=b.anchor_strand[perm], anchor_pos=b.anchor_pos[perm],
                      insert_size=b.insert_size[perm], chr_id=b.chr_id[perm])
    return b


def make_reads(chr_padded, n_reads: int, read_len: int = 100, seed: int = 20260927,
               insert_size: int = 500, mix=(0.4, 0.2, 0.1, 0.1, 0.2), error_rate: float = 0.01,
               n_rate: float = 0.001, rc_retry_frac: float = 0.1, chr_id: int = 0,
               max_del: int = 10000, spacer: int = SPACER, read_lens=None,
               chunk: int = 1 << 20, device="cpu") -> ReadBatch:
    """mix = fractions of (D, SI, TD, INV, none).  Reads are emitted the way Pindel receives
    them: sequence in sequencing orientation plus strand/position of the mapped mate, placed
    so that the close end lies within one insert size of the anchor."""
    if isinstance(chr_padded, (bytes, bytearray)):
        ref = torch.frombuffer(bytearray(chr_padded), dtype=torch.uint8).to(device)
    else:
        ref = chr_padded.to(device)
    biol = ref.numel() - 2 * spacer
    g = _gen(seed, device)
    comp_lut = _COMP.to(device)
    acgt = _ACGT.to(device)
    mixc = torch.cumsum(torch.tensor(mix, dtype=torch.float64), 0)
    mixc = (mixc / mixc[-1]).to(device)

    def rand(*shape):
        return torch.rand(*shape, generator=g, device=device)

    def randint(lo, hi, shape):
        return torch.randint(lo, hi, shape, generator=g, device=device, dtype=torch.int64)

    seqs, strands, poss, lens_all = [], [], [], []
    done = 0
    while done < n_reads:
        n = min(chunk, n_reads - done)
        if read_lens is None:
            L = torch.full((n,), read_len, dtype=torch.int64, device=device)
        else:
            choices = torch.tensor(list(read_lens), dtype=torch.int64, device=device)
            L = choices[randint(0, len(read_lens), (n,))]
        Lmax = int(L.max())
        kind = torch.searchsorted(mixc, rand(n).double(), right=True).clamp(0, 4)
        sp = (randint(20, 81, (n,)) * L // 100).clamp(min=12)            # split point in the read
        sp = torch.minimum(sp, L - 12)
        margin = max_del + 4 * insert_size + 2 * Lmax + 64
        bp = randint(margin, max(biol - margin, margin + 1), (n,)) + spacer   # AbsLoc of the break
        dsize = torch.exp(rand(n) * math.log(max_del)).long().clamp(1, max_del)
        isize = randint(1, 21, (n,))
        j = torch.arange(Lmax, device=device)[None, :]
        in_read = j < L[:, None]
        left = j < sp[:, None]
        r = j - sp[:, None]                                    # offset inside the right part
        k = kind[:, None]
        bpc, dsz = bp[:, None], dsize[:, None]
        idx_left = bpc - sp[:, None] + j                       # left part: ref[bp-sp .. bp)
        right_D = bpc + dsz + r                                # deletion of dsize bases
        right_SI = bpc + (r - isize[:, None])                  # short insertion of isize random bases
        right_TD = bpc - torch.maximum(dsz, sp[:, None] + 1) + r   # tandem duplication: jump back
        right_INV = (bpc + dsz + 200) - 1 - r                  # inversion: far side, reverse complement
        right_none = bpc + r                                   # plain reference read
        idx_right = torch.where(k == 0, right_D, torch.where(k == 1, right_SI, torch.where(
            k == 2, right_TD, torch.where(k == 3, right_INV, right_none))))
        idx = torch.where(left, idx_left, idx_right).clamp(0, ref.numel() - 1)
        comp = (~left) & (k == 3)
        junk = (~left) & (k == 1) & (r < isize[:, None])
        all_junk = (kind == 4) & (rand(n) < 0.5)               # half of "none" is unmappable junk
        junk = junk | all_junk[:, None]
        bases = ref[idx]
        bases = torch.where(comp, comp_lut[bases.long()], bases)
        rnd = acgt[randint(0, 4, (n, Lmax))]
        bases = torch.where(junk, rnd, bases)
        bases = torch.where(rand(n, Lmax) < error_rate, rnd, bases)
        bases = torch.where(rand(n, Lmax) < n_rate, torch.full_like(bases, ord("N")), bases)
        # '+' anchors sit upstream of the read (close end = left part), '-' anchors downstream
        plus = rand(n) < 0.5
        slack = randint(0, max(insert_size - Lmax - 20, 1), (n,))
        left_start = bp - sp
        right_end = torch.where(kind == 0, bp + dsize, bp) + (L - sp)
        pos_plus = left_start - slack - spacer
        pos_minus = right_end + slack - spacer
        pos_minus = torch.where((kind == 2) | (kind == 3), bp + slack - spacer, pos_minus)
        pos = torch.where(plus, pos_plus, pos_minus).clamp(0, biol)
        # sequencing orientation: the mate of a '+' anchor is read from the reverse strand
        flip = plus ^ (rand(n) < rc_retry_frac)
        rev_idx = (L[:, None] - 1 - j).clamp(0, Lmax - 1)
        rc = comp_lut[torch.gather(bases, 1, rev_idx).long()]
        bases = torch.where(flip[:, None], rc, bases)
        seqs.append(bases[in_read].cpu())
        lens_all.append(L.cpu())
        strands.append(torch.where(plus, ord("+"), ord("-")).to(torch.uint8).cpu())
        poss.append(pos.to(torch.int32).cpu())
        done += n
    lens = torch.cat(lens_all).numpy()
    off = np.zeros(n_reads + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    return ReadBatch(seq=torch.cat(seqs).numpy(), seq_off=off,
                     anchor_strand=torch.cat(strands).numpy(), anchor_pos=torch.cat(poss).numpy(),
                     insert_size=np.full(n_reads, insert_size, dtype=np.int16),
                     chr_id=np.full(n_reads, chr_id, dtype=np.int32))

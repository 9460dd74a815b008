"""BAM ingest without htslib (pindel_amd/csrc/host/pg_bam.hpp).

The snapshot holds no BAM files (and no htslib / samtools), so the parity of the read-selection rules against the
reference binary is UNPINNED; what is checked here:
  * the reference's gold reads (tests/golden/sim1chrVs2, Pindel text) written as read pairs into a BAM by the
    test-side writer (tests/bam_writer.py) come back from the BAM route as exactly the batch the text route
    loads, and (GPU) `pindel_pg -i` reproduces the reference's gold _D/_SI/_TD/_INV reports from that BAM;
  * index queries (.bai: binning + linear index) return what a sequential scan of the same sorted file returns,
    window by window, and records spanning BGZF block boundaries survive;
  * the selection rules (fetch_func_SR / isGoodAnchor / isWeirdRead / build_record_SR, src/reader.cpp:561-898,
    1099-1151) against an independent Python restatement on a deliberately messy file: unmapped mate first or
    second, clipped / indel / NM>0 reads that anchor themselves, reverse-strand mates, N trimming, IUPAC codes,
    secondary / duplicate / low-quality anchors with -A, pairs split over two windows.
"""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

from pindel_amd import hostio, hostlib
from tests import bam_writer as bw
from tests import golden_util as gu

F = bw.FLAG


def _lib():
    L = hostlib.lib()
    L.pgh_bam_ingest.restype = C.c_void_p
    L.pgh_bam_ingest.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_char_p,
                                 C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pgh_bam_ingest_view.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 7
    L.pgh_bam_ingest_name.restype = C.c_char_p
    L.pgh_bam_ingest_name.argtypes = [C.c_void_p, C.c_uint64]
    L.pgh_bam_ingest_free.argtypes = [C.c_void_p]
    return L


def ingest(path, chr_name, chr_id, padded, ws, we, isz, tag="S", min_q=0, use_index=True):
    """-> list of (name, seq, strand, pos, ms, isz, chr) in emission order"""
    L = _lib()
    n, nb = C.c_uint64(), C.c_uint64()
    h = L.pgh_bam_ingest(str(path).encode(), chr_name.encode(), chr_id, padded, ws, we, isz, tag.encode(), min_q, 100000,
                         1 if use_index else 0, C.byref(n), C.byref(nb))
    assert h, L.pgh_last_error()
    ptr = [C.c_void_p() for _ in range(7)]
    assert L.pgh_bam_ingest_view(h, *[C.byref(p) for p in ptr]) == 0
    n = n.value

    def arr(p, dt, cnt):
        if not cnt:
            return np.zeros(0, dtype=dt)
        return np.frombuffer((C.c_uint8 * (cnt * np.dtype(dt).itemsize)).from_address(p.value), dtype=dt).copy()
    seq = arr(ptr[0], np.uint8, nb.value)
    off = arr(ptr[1], np.uint64, n + 1)
    strand, pos, iszs, chrs, ms = (arr(ptr[2], np.uint8, n), arr(ptr[3], np.int32, n), arr(ptr[4], np.int16, n),
                                   arr(ptr[5], np.int32, n), arr(ptr[6], np.int16, n))
    out = [(L.pgh_bam_ingest_name(h, i).decode(), seq[int(off[i]):int(off[i + 1])].tobytes().decode("latin1"),
            chr(strand[i]), int(np.uint32(pos[i])), int(ms[i]), int(iszs[i]), int(chrs[i])) for i in range(n)]
    L.pgh_bam_ingest_free(h)
    return out


def _gold_text_records():
    with gzip.open(os.path.join(gu.GOLD, "simulated_test.out_CloseEndMapped.gz"), "rt") as fh:
        lines = fh.read().split("\n")
    recs = []
    for i in range(0, len(lines) - 2, 3):
        if not lines[i]:
            break
        d, chrom, pos, ms, isz, tag = lines[i + 2].split()
        recs.append((lines[i], lines[i + 1], d, int(pos), int(ms), int(isz), tag))
    return recs


def _pairs_for_text_records(recs, tid=0):
    """One read pair per Pindel-text record: a clean anchor (not 'weird': no NM, plain match) followed by its
    unmapped mate at the same position -> exactly build_record_SR(anchor, mate), nothing else."""
    out = []
    for name, seq, d, pos, ms, isz, tag in recs:
        qname, _, suffix = name[1:].rpartition("/")
        mate_flag = F["READ1"] if suffix == "1" else F["READ2"]
        anchor_flag = F["READ2"] if suffix == "1" else F["READ1"]
        alen = 40
        apos = pos if d == "+" else pos - alen          # '-' anchors: MatchedRelPos = pos + bam_cigar2len
        out.append(dict(qname=qname, flag=F["PAIRED"] | F["MUNMAP"] | anchor_flag | (F["REVERSE"] if d == "-" else 0),
                        tid=tid, pos=apos, mapq=ms, cigar=[(0, alen)], seq="ACGT" * (alen // 4), mtid=tid, mpos=apos))
        out.append(dict(qname=qname, flag=F["PAIRED"] | F["UNMAP"] | mate_flag | (F["MREVERSE"] if d == "-" else 0),
                        tid=tid, pos=apos, mapq=0, cigar=[], seq=seq, mtid=tid, mpos=apos))
    return out


def test_gold_reads_through_bam_equal_the_text_route(tmp_path):
    recs = _gold_text_records()
    assert len(recs) == 14862
    bam = tmp_path / "gold.bam"
    n_blocks = bw.write_bam(str(bam), [("1", 200000)], _pairs_for_text_records(recs), with_index=False, block_bytes=0x7000)
    assert n_blocks > 50                                  # records do span BGZF block boundaries
    got = ingest(bam, "1", 0, 200000 + 200001, 0, 5_000_000, 500, tag="SIM1CHRVS2")
    assert len(got) == len(recs)
    for g, r in zip(got, recs):
        assert g == (r[0], r[1], r[2], r[3], r[4], r[5], 0), (g, r)
    # ... and as the SoA batch of the C ABI: identical to what the text loader builds
    fa, reads_txt = gu.unpack(tmp_path)
    chroms = hostio.load_fasta(fa)
    text = hostio.read_pindel_text(reads_txt, [c[0] for c in chroms])
    assert text.n == len(got)
    assert b"".join(g[1].encode() for g in got) == text.seq.tobytes()
    assert [g[3] for g in got] == text.anchor_pos.tolist() and [ord(g[2]) for g in got] == text.anchor_strand.tolist()


def _restated(records, tid, ws, we, isz, min_q, biol):
    """Independent restatement of fetch_func_SR + build_record_SR over records in file order."""
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}

    def end_pos(r):
        if r["flag"] & F["UNMAP"] or not r["cigar"]:
            return r["pos"] + 1
        return r["pos"] + (sum(n for op, n in r["cigar"] if op in (0, 2, 3, 7, 8)) or 1)

    def weird(r):
        if r["flag"] & F["UNMAP"]:
            return True
        if any(op in (1, 2, 3, 4, 5, 6) for op, _ in r["cigar"]):
            return True
        nm = (r.get("tags") or {}).get("NM", 0)
        return nm != 0 or sum(n for op, n in r["cigar"] if op != 0) > 0

    def good_anchor(r):
        if r["flag"] & F["UNMAP"] or r.get("mapq", 0) < min_q:
            return False
        return min_q == 0 or not (r["flag"] & (F["SECONDARY"] | F["QCFAIL"] | F["DUP"]))

    out = []

    def build(m, u):
        if m.get("mapq", 0) < min_q:
            return
        name = "@" + u["qname"] + ("/1" if u["flag"] & F["READ1"] else "/2" if u["flag"] & F["READ2"] else "")
        s = u["seq"].lstrip("N").rstrip("N")
        if s.count("N") > int(len(s) * .10) or len(s) < 22:
            return
        if u["flag"] & F["REVERSE"]:
            s = "".join(comp.get(c, "\0") for c in reversed(s))
        while s and not s[-1].isalnum():
            s = s[:-1]
        pos, d = m["pos"] & 0xffffffff, "+"
        if m["flag"] & F["REVERSE"]:
            d = "-"
            pos = (pos + sum(n for op, n in m["cigar"] if op in (0, 1, 4)) - sum(n for op, n in m["cigar"] if op == 2)) & 0xffffffff
        out.append((name, s, d, min(pos, biol), m.get("mapq", 0), isz, 0))

    waiting = {}
    for b1 in records:
        if b1["tid"] != tid or not (b1["pos"] < we and end_pos(b1) > ws):
            continue
        b2 = waiting.pop(b1["qname"], None)
        if b2 is None:
            waiting[b1["qname"]] = b1
            if weird(b1):
                build(b1, b1)
            continue
        if weird(b2):
            build(b2, b2)
        if good_anchor(b1) and weird(b2):
            build(b1, b2)
        if good_anchor(b2) and weird(b1):
            build(b2, b1)
    return out


def _messy_records(rng, n_pairs, ref_len, tid=0):
    recs = []
    for k in range(n_pairs):
        pos = int(rng.integers(100, ref_len - 400))
        qn = f"q{k}"
        kind = int(rng.integers(0, 8))
        seq = "".join(rng.choice(list("ACGT"), 100))
        if rng.random() < 0.1:                                        # N's: ends (trimmed) and inside (10 % rule)
            nn = int(rng.integers(1, 16))
            seq = "N" * int(rng.integers(0, 4)) + seq[:60] + "N" * nn + seq[60 + nn:] + "N" * int(rng.integers(0, 3))
        if rng.random() < 0.05:
            seq = seq[:50] + "R" + seq[51:97] + "YK" + seq[99:]      # IUPAC codes, also at the end
        rev_a, rev_u = bool(rng.random() < 0.5), bool(rng.random() < 0.3)
        flags_a = F["PAIRED"] | F["READ1"] | (F["REVERSE"] if rev_a else 0)
        flags_u = F["PAIRED"] | F["READ2"] | (F["REVERSE"] if rev_u else 0)
        mapq = int(rng.choice([0, 10, 37, 60]))
        a = dict(qname=qn, flag=flags_a, tid=tid, pos=pos, mapq=mapq, cigar=[(0, 100)], seq="ACGT" * 25, mtid=tid, mpos=pos)
        u = dict(qname=qn, flag=flags_u | F["UNMAP"], tid=tid, pos=pos, mapq=0, cigar=[], seq=seq, mtid=tid, mpos=pos)
        if kind == 1:                                                  # mate mapped too, soft-clipped: anchors itself
            u.update(flag=flags_u, pos=pos + int(rng.integers(150, 350)), mapq=int(rng.choice([0, 29, 60])),
                     cigar=[(4, 30), (0, len(seq) - 30)])
        elif kind == 2:                                                # mate with a deletion and NM
            u.update(flag=flags_u, pos=pos + 200, mapq=50, cigar=[(0, 40), (2, 7), (0, len(seq) - 40)], tags={"NM": 7})
        elif kind == 3:                                                # both clean: nothing
            u.update(flag=flags_u, pos=pos + 250, mapq=60, cigar=[(0, len(seq))], tags={"NM": 0})
        elif kind == 4:                                                # anchor itself has mismatches (NM > 0)
            a["tags"] = {"NM": 3}
        elif kind == 5:                                                # duplicate / secondary anchors
            a["flag"] |= F["DUP"] if rng.random() < 0.5 else F["SECONDARY"]
        elif kind == 6:                                                # short read: dropped (< 22 bases)
            u["seq"] = seq[:int(rng.integers(5, 30))]
        first_unmapped = kind == 7 or rng.random() < 0.2              # unmapped mate BEFORE its anchor in the file
        recs.append((min(a["pos"], u["pos"]), k, [u, a] if first_unmapped and u["pos"] == a["pos"] else sorted([a, u], key=lambda r: r["pos"])))
    recs.sort(key=lambda t: (t[0], t[1]))
    flat = []
    for _, _, pair in recs:
        flat.extend(pair)
    flat.sort(key=lambda r: r["pos"])                                  # coordinate-sorted, stable
    return flat


@pytest.mark.parametrize("min_q", [0, 20])
def test_selection_rules_match_the_restatement_and_index_equals_scan(tmp_path, min_q):
    rng = np.random.default_rng(12 + min_q)
    ref_len = 2_300_000
    recs = _messy_records(rng, 6000, ref_len)
    bam = tmp_path / "messy.bam"
    bw.write_bam(str(bam), [("chrM", 16000), ("chrZ", ref_len)], [dict(r, tid=1, mtid=1) for r in recs], with_index=True,
                 block_bytes=0x9000)
    padded = ref_len + 200000
    total = 0
    for ws, we in ((0, 500_000), (500_000, 1_000_000), (1_000_000, 2_300_000), (123_456, 130_000), (2_299_000, 2_300_000)):
        want = _restated([dict(r, tid=1) for r in recs], 1, ws, we, 450, min_q, ref_len)
        by_index = ingest(bam, "chrZ", 1, padded, ws, we, 450, min_q=min_q, use_index=True)
        by_scan = ingest(bam, "chrZ", 1, padded, ws, we, 450, min_q=min_q, use_index=False)
        assert by_index == by_scan
        assert [(g[0], g[1], g[2], g[3], g[4], g[5]) for g in by_index] == [w[:6] for w in want]
        total += len(want)
    assert total > 2000
    assert ingest(bam, "chrM", 0, 16000 + 200000, 0, 16000, 450) == []
    assert ingest(bam, "nope", 5, 1_000_000, 0, 16000, 450) == []


def test_insert_size_not_above_read_length_is_an_error(tmp_path):
    bam = tmp_path / "short_insert.bam"
    recs = _pairs_for_text_records([("@x/1", "ACGT" * 25, "+", 5000, 37, 500, "S")])
    bw.write_bam(str(bam), [("1", 200000)], recs, with_index=True)
    L = _lib()
    n, nb = C.c_uint64(), C.c_uint64()
    h = L.pgh_bam_ingest(str(bam).encode(), b"1", 0, 400000, 0, 100000, 100, b"S", 0, 100000, 1, C.byref(n), C.byref(nb))
    assert not h and b"insert size" in L.pgh_last_error()
    assert len(ingest(bam, "1", 0, 400000, 0, 100000, 101)) == 1


@pytest.mark.gpu
def test_command_line_bam_input_reproduces_gold_reports(tmp_path):
    """pindel_pg -i config: gold reads as a BAM -> BGZF/BAM decode -> selection rules -> GPU search -> reports
    byte-identical to the reference's gold standard."""
    import subprocess
    from pindel_amd import binding
    fa, _ = gu.unpack(tmp_path)
    bam = tmp_path / "gold.bam"
    bw.write_bam(str(bam), [("1", 200000)], _pairs_for_text_records(_gold_text_records()), with_index=False)
    cfg = tmp_path / "config.txt"
    cfg.write_text("gold.bam\t500\tSIM1CHRVS2\n")
    exe = os.path.join(os.path.dirname(binding.LIB_PATH), "pindel_pg")
    prefix = str(tmp_path / "bam")
    out = subprocess.run([exe, "-f", fa, "-i", str(cfg), "-o", prefix, "-T", "2"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "14862 reads, close end 14862, far end 10968" in out.stdout
    gu.assert_reports_match_gold(prefix)

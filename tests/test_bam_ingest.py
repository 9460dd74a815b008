"""BAM ingest without htslib (pindel_amd/csrc/host/pg_bam.hpp).

The one BAM the reference ships (demo/simulated_MEI) pins the decoder and the selection rules in tests/test_mei_bam.py.
There is no htslib / samtools in the image, so further BAMs come from the test-side writer; what is checked here:
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
    L.pgh_bam_ingest_ref_reads.restype = C.c_uint64
    L.pgh_bam_ingest_ref_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.pgh_bai_summary.restype = C.c_int32
    L.pgh_bai_summary.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32]
    return L


def ingest(path, chr_name, chr_id, padded, ws, we, isz, tag="S", min_q=0, use_index=True, ref_reads=None):
    """-> list of (name, seq, strand, pos, ms, isz, chr) in emission order; ref_reads (a list): receives the
    (pos, length) of the reference-supporting reads of the window"""
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
    if ref_reads is not None:
        k = int(L.pgh_bam_ingest_ref_reads(h, None, 0))
        buf = np.zeros(3 * k, dtype=np.uint32)
        L.pgh_bam_ingest_ref_reads(h, buf.ctypes.data, k)
        ref_reads.extend((int(buf[3 * i]), int(buf[3 * i + 1])) for i in range(k))
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


def _ref_reads_restated(records, tid, ws, we, min_q, nm_max=2, rate=0.02):
    """Independent restatement of isRefRead + build_record_RefRead as fetch_func_SR calls them (src/reader.cpp:620-656,
    903-923, 1132-1147): (pos, length) of the reads that support the reference allele, in emission order."""
    def end_pos(r):
        if r["flag"] & F["UNMAP"] or not r["cigar"]:
            return r["pos"] + 1
        return r["pos"] + (sum(n for op, n in r["cigar"] if op in (0, 2, 3, 7, 8)) or 1)

    def good_anchor(r):
        if r["flag"] & F["UNMAP"] or r.get("mapq", 0) < min_q:
            return False
        return min_q == 0 or not (r["flag"] & (F["SECONDARY"] | F["QCFAIL"] | F["DUP"]))

    def is_ref(r):
        if r["flag"] & (F["SECONDARY"] | F["QCFAIL"] | F["DUP"]):
            return False
        tags = r.get("tags") or {}
        if "NM" in tags and (tags["NM"] > nm_max or tags["NM"] > int(len(r["seq"]) * rate) + 1):
            return False
        if len(r["cigar"]) > 2 and any(op in (1, 2) for op, _ in r["cigar"]):
            return False
        return (not r["flag"] & F["UNMAP"]) and tags.get("NM", 0) <= 2 and sum(n for op, n in r["cigar"] if op != 0) <= 2

    out, waiting = [], {}
    for b1 in records:
        if b1["tid"] != tid or not (b1["pos"] < we and end_pos(b1) > ws):
            continue
        b2 = waiting.pop(b1["qname"], None)
        if b2 is None:
            waiting[b1["qname"]] = b1
            continue
        for anchor, ref in ((b1, b2), (b2, b1)):
            if good_anchor(anchor) and is_ref(ref) and ref.get("mapq", 0) >= min_q:
                out.append((ref["pos"], len(ref["seq"])))
    return out


def _messy_records(rng, n_pairs, ref_len, tid=0):
    recs = []
    for k in range(n_pairs):
        pos = int(rng.integers(100, ref_len - 400))
        qn = f"q{k}"
        kind = int(rng.integers(0, 11))
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
        elif kind == 8:                                                # mapped mate with 0-4 edits: a reference read up to 2
            u.update(flag=flags_u, pos=pos + 180, mapq=int(rng.choice([5, 30, 60])), cigar=[(0, len(seq))],
                     tags={"NM": int(rng.integers(0, 5))} if rng.random() < 0.8 else None)
        elif kind == 9:                                                # one-base indel: three CIGAR elements = HasIndel
            u.update(flag=flags_u, pos=pos + 220, mapq=60, cigar=[(0, 50), (1, 1), (0, len(seq) - 51)], tags={"NM": 1})
        elif kind == 10:                                               # two CIGAR elements: not "HasIndel", still few edits
            u.update(flag=flags_u, pos=pos + 240, mapq=60, cigar=[(0, len(seq) - 1), (1, 1)], tags={"NM": 1})
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
    total = total_refs = 0
    for ws, we in ((0, 500_000), (500_000, 1_000_000), (1_000_000, 2_300_000), (123_456, 130_000), (2_299_000, 2_300_000)):
        want = _restated([dict(r, tid=1) for r in recs], 1, ws, we, 450, min_q, ref_len)
        refs_index, refs_scan = [], []
        by_index = ingest(bam, "chrZ", 1, padded, ws, we, 450, min_q=min_q, use_index=True, ref_reads=refs_index)
        by_scan = ingest(bam, "chrZ", 1, padded, ws, we, 450, min_q=min_q, use_index=False, ref_reads=refs_scan)
        assert by_index == by_scan and refs_index == refs_scan
        want_refs = _ref_reads_restated([dict(r, tid=1) for r in recs], 1, ws, we, min_q)
        assert refs_index == want_refs
        total_refs += len(want_refs)
        assert [(g[0], g[1], g[2], g[3], g[4], g[5]) for g in by_index] == [w[:6] for w in want]
        total += len(want)
    assert total > 2000 and total_refs > 500
    assert ingest(bam, "chrM", 0, 16000 + 200000, 0, 16000, 450) == []
    assert ingest(bam, "nope", 5, 1_000_000, 0, 16000, 450) == []


@pytest.mark.parametrize("name,n_ref", [("sim1chrVs2.bam.bai", 1), ("simulated_sample_1.bam.bai", 4)])
def test_reference_bai_files_parse(name, n_ref):
    """The index files the reference ships (written by samtools; THESE BAMs are not in the snapshot -- the demo/simulated_MEI one is, see tests/test_mei_bam.py): the BAI
    reader must take the metadata pseudo-bin and the trailing n_no_coor in its stride."""
    import struct
    path = os.path.join(os.path.dirname(gu.GOLD), "bai", name)
    d = open(path, "rb").read()
    assert d[:4] == b"BAI\1" and struct.unpack_from("<i", d, 4)[0] == n_ref
    o, want = 8, []
    for _ in range(n_ref):
        (n_bin,) = struct.unpack_from("<i", d, o)
        o += 4
        bins = chunks = total = 0
        for _ in range(n_bin):
            bid, nch = struct.unpack_from("<Ii", d, o)
            o += 8
            vals = struct.unpack_from("<%dQ" % (2 * nch), d, o)
            o += 16 * nch
            if bid != 37450:
                bins, chunks, total = bins + 1, chunks + nch, total + sum(vals)
        (n_intv,) = struct.unpack_from("<i", d, o)
        o += 4
        total += sum(struct.unpack_from("<%dQ" % n_intv, d, o))
        o += 8 * n_intv
        want.append((bins, chunks, total & (2 ** 64 - 1), n_intv))
    assert len(d) - o in (0, 8)
    out = np.zeros(4 * n_ref, dtype=np.uint64)
    assert _lib().pgh_bai_summary(path.encode(), out.ctypes.data, n_ref) == n_ref
    assert [tuple(int(x) for x in out[4 * t:4 * t + 4]) for t in range(n_ref)] == want
    assert all(w[0] >= 8 and w[3] == 13 for w in want)


def test_ingest_does_not_depend_on_host_threads(tmp_path, monkeypatch):
    """With an index a window is inflated and decoded sub-range by sub-range on several threads (PGH_THREADS) and the
    records then pass through the pairing in file order: same batch, same reference reads, for any thread count."""
    rng = np.random.default_rng(99)
    ref_len = 3_000_000
    recs = _messy_records(rng, 8000, ref_len)
    bam = tmp_path / "threads.bam"
    bw.write_bam(str(bam), [("chrZ", ref_len)], recs, with_index=True, block_bytes=0x4000)
    got = {}
    for threads in ("1", "3", "8", "16"):
        monkeypatch.setenv("PGH_THREADS", threads)
        refs = []
        got[threads] = (ingest(bam, "chrZ", 0, ref_len + 200000, 0, ref_len, 450, ref_reads=refs), refs)
    assert len(got["1"][0]) > 3000 and len(got["1"][1]) > 500
    assert got["1"] == got["3"] == got["8"] == got["16"]
    monkeypatch.delenv("PGH_THREADS")
    assert got["1"][0] == ingest(bam, "chrZ", 0, ref_len + 200000, 0, ref_len, 450, use_index=False)


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


@pytest.mark.gpu
def test_command_line_bam_reference_coverage_columns(tmp_path):
    """The two per-sample reference-coverage integers of the report headers (BAM input only: isRefRead /
    build_record_RefRead / UpdateRefReadCoverage, src/reader.cpp:620-656, 903-923, src/pindel.cpp:1272-1330;
    reporter.cpp:352-382): the gold reads as a BAM plus clean read pairs that support the reference.  Everything but
    the two integers must still equal gold; the integers must equal the coverage restated here."""
    import re
    import subprocess
    from pindel_amd import binding
    fa, _ = gu.unpack(tmp_path)
    recs = _pairs_for_text_records(_gold_text_records())
    rng = np.random.default_rng(77)
    cov = np.zeros(200_002, dtype=np.int64)
    extra = []
    for k in range(12000):
        pos = int(rng.integers(500, 198_000))
        mpos = pos + int(rng.integers(150, 400))
        kind = int(rng.integers(0, 6))
        tags_b = {"NM": 0} if kind == 0 else {"NM": 2} if kind == 1 else {"NM": 3} if kind == 2 else None
        dup = F["DUP"] if kind == 3 else 0
        a = dict(qname=f"ref{k}", flag=F["PAIRED"] | F["READ1"] | F["MREVERSE"], tid=0, pos=pos, mapq=60, cigar=[(0, 100)],
                 seq="ACGT" * 25, mtid=0, mpos=mpos)
        b = dict(qname=f"ref{k}", flag=F["PAIRED"] | F["READ2"] | F["REVERSE"] | dup, tid=0, pos=mpos, mapq=60,
                 cigar=[(0, 90)] if kind == 4 else [(0, 100)], mtid=0, mpos=pos, tags=tags_b,
                 # (reads with NM > 0 are split-read candidates too: all-N bases keep them out of the reports)
                 seq=("ACGT" * 25)[:90] if kind == 4 else "N" * 100 if kind in (1, 2) else "ACGT" * 25)
        extra += [a, b]
        cov[pos + 1:pos + 99] += 1                                    # a: clean, always a reference read
        if kind not in (2, 3):                                        # b: NM 3 > -n 2 and duplicates are not
            n = len(b["seq"])
            cov[mpos + 1:mpos + n - 1] += 1
    bam = tmp_path / "gold_plus_ref.bam"
    bw.write_bam(str(bam), [("1", 200000)], recs + extra, with_index=False)
    cfg = tmp_path / "config.txt"
    cfg.write_text("gold_plus_ref.bam\t500\tSIM1CHRVS2\n")
    exe = os.path.join(os.path.dirname(binding.LIB_PATH), "pindel_pg")
    prefix = str(tmp_path / "cov")
    out = subprocess.run([exe, "-f", fa, "-i", str(cfg), "-o", prefix, "-T", "2", "-R", "false"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "14862 reads, close end 14862, far end 10968" in out.stdout
    gu.assert_reports_match_gold(prefix)                               # (the comparison masks the two integers)
    checked = 0
    for suf, off_s, off_e in (("D", 1, -1), ("SI", 1, -1)):           # cov[BPLeft + 2], cov[BPRight]; BP printed 1-based
        for line in open(f"{prefix}_{suf}"):
            m = re.match(r"^\d+\t(?:D|I) .*\tBP (\d+)\t(\d+)\tBP_range .*\tSIM1CHRVS2 (-?\d+) (-?\d+) ", line)
            if not m:
                continue
            bp1, bp2, cs, ce = (int(x) for x in m.groups())
            assert (cs, ce) == (int(cov[bp1 + off_s]), int(cov[bp2 + off_e])), line[:200]
            checked += 1
    assert checked >= 10 and cov.max() > 5


# ------------------------------------------------------------------------------------------ read-pair discovery (-R)
def _rp_restated(records, tid, ws, we, isz, min_q, spacer):
    """Independent restatement of build_record_RP_Discovery + UpdateBD / ModifyRP / Summarize (src/reader.cpp:925-1097,
    src/bddata.cpp:138-812) for same-chromosome pairs.  Positions within a test are unique, so no sort has ties."""
    def i32(x):
        x &= 0xffffffff
        return x - (1 << 32) if x >= (1 << 31) else x

    def uabs(a, b):
        return abs(i32(a - b))

    rp = []
    for r in records:
        if r["tid"] != tid or not (r["pos"] < we and (r["pos"] + max(1, sum(n for op, n in r["cigar"] if op in (0, 2, 3)))) > ws):
            continue
        f = r["flag"]
        if not f & F["PAIRED"] or r.get("mapq", 0) < min_q or f & F["UNMAP"] or f & F["MUNMAP"]:
            continue
        rev, mrev = bool(f & F["REVERSE"]), bool(f & F["MREVERSE"])
        if not (abs(r.get("tlen", 0)) > 3 * isz + 1000 or rev == mrev) or r.get("mtid", tid) != tid:
            continue
        d = dict(DA="-" if rev else "+", DB="-" if mrev else "+", PosA=r["pos"], PosB=r["mpos"], L=len(r["seq"]), IS=isz, n=0,
                 visited=False, report=False)
        if not d["PosA"] < d["PosB"]:
            d["DA"], d["DB"], d["PosA"], d["PosB"] = d["DB"], d["DA"], d["PosB"], d["PosA"]
        d["OA"], d["OB"] = d["PosA"], d["PosB"]
        rp.append(d)
    rp.sort(key=lambda d: (d["PosA"], d["PosB"]))
    rp.sort(key=lambda d: (-d["OA"], -d["OB"]))
    for d in rp:
        D, L = d["IS"], d["L"]
        if d["DA"] == "+":
            d["PosA"] = d["PosA"] - 2 * L if d["PosA"] > 2 * L else 1
            d["A1"] = d["PosA"] + D + 2 * L
        else:
            d["PosA"] = d["PosA"] - D if d["PosA"] > D else 1
            d["A1"] = d["PosA"] + D + L
        if d["DB"] == "+":
            d["PosB"] = d["PosB"] - 2 * L if d["PosB"] > 2 * L else 1
        else:
            d["PosB"] = d["PosB"] - D if d["PosB"] > D else 1
        d["B1"] = d["PosB"] + D + L

    def overlap(a, b):
        if max(uabs(a["PosA"], a["A1"]), uabs(a["PosB"], a["B1"]), uabs(b["PosA"], b["A1"]), uabs(b["PosB"], b["B1"])) > 1000:
            return False
        fa, fb = sorted(((a["PosA"] + a["A1"]) // 2, (a["PosB"] + a["B1"]) // 2))
        sa, sb = sorted(((b["PosA"] + b["A1"]) // 2, (b["PosB"] + b["B1"]) // 2))
        if a["DA"] != b["DA"] or a["DB"] != b["DB"] or fa > sb + 200 or fb + 200 < sa:
            return False
        c = 0.9
        if fa <= sa and sb <= fb and (sb - sa) / (fb - fa) >= c:
            return True
        if sa <= fa and fb <= sb and (fb - fa) / (sb - sa) >= c:
            return True
        if fa <= sa <= fb <= sb and (fb - sa) / (fb - fa) >= c and (fb - sa) / (sb - sa) >= c:
            return True
        if sa <= fa <= sb <= fb and (sb - fa) / (fb - fa) >= c and (sb - fa) / (sb - sa) >= c:
            return True
        return False

    for a in rp:
        for b in rp:
            if a is b or not overlap(a, b):
                continue
            if b["A1"] - b["PosA"] > 10000 or b["B1"] - b["PosB"] > 10000:
                continue
            if (a["DA"] == "+" and a["PosA"] < b["PosA"] < a["A1"] < b["A1"]) or \
                    (a["DA"] == "-" and a["PosA"] < b["A1"] < a["A1"] and b["PosA"] < a["PosA"]):
                a["PosA"], a["A1"] = b["PosA"], b["A1"]
            if (a["DB"] == "+" and a["PosB"] < b["PosB"] < a["B1"] < b["B1"]) or \
                    (a["DB"] == "-" and b["PosB"] < a["PosB"] < b["B1"] < a["B1"]):
                a["PosB"], a["B1"] = b["PosB"], b["B1"]
    for d in rp:
        if d["DA"] == "+":
            d["PosA"] += d["L"]
            d["A1"] += d["L"]
        if d["DB"] == "+":
            d["PosB"] += d["L"]
            d["B1"] += d["L"]
        if uabs(d["PosA"], d["PosB"]) < 500:
            d["visited"] = True
    box = lambda d: (d["PosA"], d["PosB"], d["A1"], d["B1"], d["DA"], d["DB"])
    events = []
    if len(rp) >= 5:
        good = []
        for i in range(len(rp) - 1):
            if rp[i]["visited"]:
                continue
            rp[i]["n"] = 1
            for j in range(i + 1, len(rp)):
                if not rp[j]["visited"] and box(rp[i]) == box(rp[j]):
                    rp[i]["n"] += 1
                    rp[j]["visited"] = True
            good.append(i)
        if len(good) == 1:
            rp[good[0]]["report"] = rp[good[0]]["n"] >= 5
        else:
            for a in range(len(good) - 1):
                ra = rp[good[a]]
                if ra["visited"]:
                    continue
                for b in range(a + 1, len(good)):
                    rb = rp[good[b]]
                    if not rb["visited"] and box(ra) == box(rb):
                        ra["n"] += rb["n"]
                        rb["visited"] = True
                ra["report"] = ra["n"] >= 5
    for d in rp:
        if not d["report"]:
            continue
        sh = d["IS"]
        f1, f2 = sorted((d["PosA"] + spacer, d["A1"] + spacer))
        if d["DA"] == "+" and f1 > sh:
            f1 -= sh
        elif sh * 2 < spacer:
            f2 += sh
        s1, s2 = sorted((d["PosB"] + spacer, d["B1"] + spacer))
        if d["DB"] == "+" and s1 > sh:
            s1 -= sh
        events.append((f1, f2, s1, s2, d["n"]))
    return events


def test_read_pair_discovery_matches_the_restatement(tmp_path):
    rng = np.random.default_rng(77)
    ref_len, isz, L = 3_000_000, 400, 100
    recs, used = [], set()

    def uniq(p):
        while p in used:
            p += 1
        used.add(p)
        return p

    k = 0
    for c in range(40):                                               # clusters of discordant pairs: deletions, inversions
        a0 = int(rng.integers(20_000, ref_len - 200_000))
        span = int(rng.integers(3000, 60_000))
        kind = int(rng.integers(0, 3))
        for _ in range(int(rng.integers(2, 14))):
            pa = uniq(a0 + int(rng.integers(0, 250)))
            pb = uniq(a0 + span + int(rng.integers(0, 250)))
            ra, rb = (False, True) if kind == 0 else (False, False) if kind == 1 else (True, True)   # FR far apart / FF / RR
            common = dict(qname=f"p{k}", tid=0, mtid=0, mapq=int(rng.choice([20, 40, 60])), cigar=[(0, L)], seq="ACGT" * 25)
            recs.append(dict(common, flag=F["PAIRED"] | F["READ1"] | (F["REVERSE"] if ra else 0) | (F["MREVERSE"] if rb else 0),
                             pos=pa, mpos=pb, tlen=pb - pa + L))
            recs.append(dict(common, flag=F["PAIRED"] | F["READ2"] | (F["REVERSE"] if rb else 0) | (F["MREVERSE"] if ra else 0),
                             pos=pb, mpos=pa, tlen=-(pb - pa + L)))
            k += 1
    for _ in range(3000):                                             # concordant pairs: ignored
        pa = uniq(int(rng.integers(1000, ref_len - 2000)))
        pb = uniq(pa + int(rng.integers(150, 320)))
        common = dict(qname=f"p{k}", tid=0, mtid=0, mapq=60, cigar=[(0, L)], seq="ACGT" * 25)
        recs.append(dict(common, flag=F["PAIRED"] | F["PROPER"] | F["READ1"] | F["MREVERSE"], pos=pa, mpos=pb, tlen=pb - pa + L))
        recs.append(dict(common, flag=F["PAIRED"] | F["PROPER"] | F["READ2"] | F["REVERSE"], pos=pb, mpos=pa, tlen=-(pb - pa + L)))
        k += 1
    recs.sort(key=lambda r: r["pos"])
    bam = tmp_path / "pairs.bam"
    bw.write_bam(str(bam), [("chrP", ref_len)], recs, with_index=True)
    L_ = hostlib.lib()
    L_.pgh_rp_events.restype = C.c_int64
    L_.pgh_rp_events.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int32, C.c_char_p, C.c_uint32, C.c_uint32,
                                 C.c_char_p, C.c_void_p, C.c_uint64]
    total = 0
    for ws, we, min_q in ((0, 1_500_000, 0), (1_500_000, 3_000_000, 0), (0, 3_000_000, 30)):
        out = np.zeros(4 * 4096, dtype=np.uint32)
        rp_path = tmp_path / f"rp_{ws}_{min_q}.txt"
        n = L_.pgh_rp_events(str(bam).encode(), b"chrP", ws, we, isz, b"S1", min_q, 100000, str(rp_path).encode(), out.ctypes.data, 4096)
        want = _rp_restated(recs, 0, ws, we, isz, min_q, 100000)
        assert n == len(want), (n, len(want))
        got = sorted(tuple(int(v) for v in out[4 * i:4 * i + 4]) for i in range(n))
        assert got == sorted(w[:4] for w in want)
        lines = rp_path.read_text().splitlines()
        assert len(lines) == n and all(l.startswith("chrP\t") and "Support: " in l and l.endswith("S1 " + l.split("Support: ")[1].split("\t")[0]) for l in lines)
        total += n
    assert total >= 10


@pytest.mark.gpu
def test_command_line_read_pair_hints_find_large_deletions(tmp_path):
    """pindel_pg -i with -R (the reference's default for BAM input): discordant pairs around 8-30 kb deletions become
    window hints, so the split reads across those deletions get their far ends (out of reach of the ranges at -x 2)
    and the deletions are called; with `-R false` they are not.  Sharding over contexts does not change a byte."""
    import filecmp
    import subprocess
    from pindel_amd import binding, synth
    rng = np.random.default_rng(5)
    ref = synth.make_reference(600_000, seed=91)
    biol = np.frombuffer(ref, dtype=np.uint8)[100000:-100000]
    fa = tmp_path / "ref.fa"
    with open(fa, "wb") as fh:
        fh.write(b">chrD\n")
        for i in range(0, len(biol), 60):
            fh.write(biol[i:i + 60].tobytes() + b"\n")
    (tmp_path / "ref.fa.fai").write_text(f"chrD\t{len(biol)}\t6\t60\t61\n")
    L, isz, recs, dels, used, k = 100, 400, [], [], set(), 0

    def uniq(p):
        while p in used:
            p += 1
        used.add(p)
        return p

    for d in range(12):
        a = 30_000 + 45_000 * d + int(rng.integers(0, 2000))          # deletion of [a, b)
        b = a + int(rng.integers(8_000, 30_000))
        dels.append((a, b))
        for _ in range(8):                                            # discordant pairs: mates on both sides, far apart
            pa, pb = uniq(a - 350 + int(rng.integers(0, 200))), uniq(b + 50 + int(rng.integers(0, 200)))
            common = dict(qname=f"d{k}", tid=0, mtid=0, mapq=60, cigar=[(0, L)])
            recs.append(dict(common, flag=F["PAIRED"] | F["READ1"] | F["MREVERSE"], pos=pa, mpos=pb, tlen=pb - pa + L,
                             seq=biol[pa:pa + L].tobytes().decode()))
            recs.append(dict(common, flag=F["PAIRED"] | F["READ2"] | F["REVERSE"], pos=pb, mpos=pa, tlen=-(pb - pa + L),
                             seq=biol[pb:pb + L].tobytes().decode()))
            k += 1
        for _ in range(10):                                           # split reads: anchor upstream, mate across the deletion
            sp = int(rng.integers(30, 70))
            read = np.concatenate([biol[a - sp:a], biol[b:b + L - sp]])
            apos = uniq(a - sp - int(rng.integers(120, 250)))
            comp = np.zeros(256, np.uint8)
            for x, y in zip(b"ACGTN", b"TGCAN"):
                comp[x] = y
            recs.append(dict(qname=f"s{k}", flag=F["PAIRED"] | F["READ1"] | F["MUNMAP"], tid=0, mtid=0, pos=apos, mpos=apos, mapq=60,
                             cigar=[(0, L)], seq=biol[apos:apos + L].tobytes().decode(), tlen=0))
            recs.append(dict(qname=f"s{k}", flag=F["PAIRED"] | F["READ2"] | F["UNMAP"], tid=0, mtid=0, pos=apos, mpos=apos, mapq=0,
                             cigar=[], seq=comp[read[::-1]].tobytes().decode(), tlen=0))
            k += 1
    recs.sort(key=lambda r: r["pos"])
    bw.write_bam(str(tmp_path / "rp.bam"), [("chrD", len(biol))], recs, with_index=True)
    (tmp_path / "cfg.txt").write_text(f"rp.bam\t{isz}\tTUMOR\n")
    exe = os.path.join(os.path.dirname(binding.LIB_PATH), "pindel_pg")

    def run(prefix, *extra):
        out = subprocess.run([exe, "-f", str(fa), "-i", str(tmp_path / "cfg.txt"), "-o", str(tmp_path / prefix), *extra],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        return out.stdout

    so = run("hinted")
    # 12 clusters of 8 pairs, 11 events: Summarize never sets Report for the LAST group it looks at
    # (`index_a < GoodIndex.size() - 1`, src/bddata.cpp:514-551) -- after the descending sort that is the cluster
    # with the smallest coordinate
    assert "read-pair events added as window hints: 11" in so
    rp_lines = (tmp_path / "hinted_RP").read_text().splitlines()
    assert len(rp_lines) == 11 and all("TUMOR 16" in l for l in rp_lines)    # both mates of a pair lie in the window
    so_plain = run("plain", "-R", "false")

    def called(prefix):
        sizes = []
        for line in open(tmp_path / (prefix + "_D")):
            f = line.split("\t")
            if len(f) > 5 and f[1].startswith("D "):
                sizes.append(int(f[1].split()[1]))
        return sizes

    want = sorted(b - a for a, b in dels[1:])              # every deletion but the one whose event is never reported
    assert sorted(called("hinted")) == want
    assert called("plain") == []
    far = lambda s: int(s.split("far end ")[1].split()[0])
    assert far(so) == 110 and far(so_plain) <= 5          # (a couple of chance far ends inside the 2 kb ranges)
    run("sharded", "-G", "0,0")
    for sfx in ("_D", "_SI", "_TD", "_INV", "_RP"):
        assert filecmp.cmp(tmp_path / ("hinted" + sfx), tmp_path / ("sharded" + sfx), shallow=False), sfx


@pytest.mark.gpu
def test_command_line_hints_use_the_window_not_the_reads_bin(tmp_path):
    """Several windows (-w 0.1) over one BAM with a BreakDancer file: g_bdData.loadRegion gets the window main() is
    working on (src/pindel.cpp:1828, 1853), not a bin derived from the reads.  Every window from the second on holds a
    split read whose '+' anchor starts 50 bases BEFORE the window (the BAM query returns records reaching into it,
    reader.cpp has no position filter on this path); a bin taken from the smallest anchor position would be the previous
    window and every read further than 3 kb into the real one would lose its hints -- the 8-30 kb deletions in the middle
    of the windows, out of reach of the ranges at -x 2, would go uncalled."""
    import subprocess
    from pindel_amd import binding, synth
    rng = np.random.default_rng(11)
    ref = synth.make_reference(600_000, seed=93)
    biol = np.frombuffer(ref, dtype=np.uint8)[100000:-100000]
    fa = tmp_path / "ref.fa"
    with open(fa, "wb") as fh:
        fh.write(b">chrD\n")
        for i in range(0, len(biol), 60):
            fh.write(biol[i:i + 60].tobytes() + b"\n")
    (tmp_path / "ref.fa.fai").write_text(f"chrD\t{len(biol)}\t6\t60\t61\n")
    comp = np.zeros(256, np.uint8)
    for x, y in zip(b"ACGTN", b"TGCAN"):
        comp[x] = y
    L, isz, recs, dels, k = 100, 400, [], [], 0

    def split_pair(apos, a, b, sp):
        nonlocal k
        read = np.concatenate([biol[a - sp:a], biol[b:b + L - sp]])
        recs.append(dict(qname=f"s{k}", flag=F["PAIRED"] | F["READ1"] | F["MUNMAP"], tid=0, mtid=0, pos=apos, mpos=apos, mapq=60,
                         cigar=[(0, L)], seq=biol[apos:apos + L].tobytes().decode(), tlen=0))
        recs.append(dict(qname=f"s{k}", flag=F["PAIRED"] | F["READ2"] | F["UNMAP"], tid=0, mtid=0, pos=apos, mpos=apos, mapq=0,
                         cigar=[], seq=comp[read[::-1]].tobytes().decode(), tlen=0))
        k += 1

    W = 100_000
    for w in range(1, 5):
        # the straddler: anchor [w W - 50, w W + 50), its mate across a 60-base deletion 200 bases into the window
        split_pair(w * W - 50, w * W + 200, w * W + 260, 50)
        a = w * W + 40_000 + int(rng.integers(0, 3000))
        b = a + int(rng.integers(8_000, 30_000))
        dels.append((a, b))
        for j in range(10):
            sp = int(rng.integers(30, 70))
            split_pair(a - sp - 120 - 13 * j, a, b, sp)
    recs.sort(key=lambda r: r["pos"])
    bw.write_bam(str(tmp_path / "w.bam"), [("chrD", len(biol))], recs, with_index=True)
    (tmp_path / "cfg.txt").write_text(f"w.bam\t{isz}\tTUMOR\n")
    bd_lines = ["#Chr1\tPos1\tOri1\tChr2\tPos2\tOri2\tType\tSize\tScore\tReads"]
    bd_lines += [f"chrD\t{a}\t5+0-\tchrD\t{b}\t0+5-\tDEL\t{b - a}\t99\t5" for a, b in dels]
    (tmp_path / "bd.txt").write_text("\n".join(bd_lines) + "\n")
    exe = os.path.join(os.path.dirname(binding.LIB_PATH), "pindel_pg")

    def called(prefix, *extra):
        out = subprocess.run([exe, "-f", str(fa), "-i", str(tmp_path / "cfg.txt"), "-o", str(tmp_path / prefix),
                              "-b", str(tmp_path / "bd.txt"), *extra], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        sizes = set()
        for line in open(tmp_path / (prefix + "_D")):
            f = line.split("\t")
            if len(f) > 5 and f[1].startswith("D "):
                sizes.add(int(f[1].split()[1]))
        return sizes

    want = {b - a for a, b in dels}
    one = called("one_window")                               # the default 5-Mbp window: one bin, nothing to get wrong
    assert want <= one
    many = called("many_windows", "-w", "0.1")
    assert want <= many, sorted(want - many)
    assert {s for s in many if s >= 8000} == want


def test_damaged_bam_is_an_error_not_a_short_file(tmp_path):
    """A corrupt or truncated BGZF block must fail the window ("BAM read failed"), not end it quietly with a partial
    read set (the clean end-of-file marker is the only way a file may end)."""
    rng = np.random.default_rng(3)
    recs = _messy_records(rng, 3000, 1_000_000)
    bam = tmp_path / "ok.bam"
    bw.write_bam(str(bam), [("chrZ", 1_000_000)], recs, with_index=True, block_bytes=0x4000)
    good = ingest(bam, "chrZ", 0, 1_200_000, 0, 1_000_000, 450)
    assert len(good) > 500
    data = bytearray(open(bam, "rb").read())
    L = _lib()
    n, nb = C.c_uint64(), C.c_uint64()

    def try_ingest(path, use_index):
        return L.pgh_bam_ingest(str(path).encode(), b"chrZ", 0, 1_200_000, 0, 1_000_000, 450, b"S", 0, 100000, 1 if use_index else 0,
                                C.byref(n), C.byref(nb))
    # flipped bytes in the middle of the compressed stream
    bad = bytearray(data)
    mid = len(bad) // 2
    for k in range(64):
        bad[mid + k] ^= 0x5a
    p1 = tmp_path / "flipped.bam"
    open(p1, "wb").write(bad)
    shutil_copy = __import__("shutil").copy
    shutil_copy(str(bam) + ".bai", str(p1) + ".bai")
    for use_index in (True, False):
        assert not try_ingest(p1, use_index), "a corrupt block was read as the end of the file"
        assert b"BAM read failed" in L.pgh_last_error()
    # cut in the middle of a block (no end-of-file marker)
    p2 = tmp_path / "cut.bam"
    open(p2, "wb").write(data[:len(data) * 2 // 3])
    assert not try_ingest(p2, False)
    assert b"BAM read failed" in L.pgh_last_error()
    # cut INSIDE a record's 4-byte length word, exactly at a block boundary, followed by a clean end-of-file marker: every
    # block is intact and the file ends properly, yet the last record is missing its body -- not a clean end either
    import struct
    head = b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chrZ\0" + struct.pack("<i", 1_000_000)
    body = b"".join(bw.encode_record(r) for r in recs[:40])
    whole = head + body
    p3 = tmp_path / "lenword.bam"
    open(p3, "wb").write(bw._bgzf_block(whole + bw.encode_record(recs[40])[:2]) + bw._EOF)
    assert not try_ingest(p3, False), "a file cut inside a length word was read as a short file"
    assert b"BAM read failed" in L.pgh_last_error()
    p4 = tmp_path / "clean.bam"
    open(p4, "wb").write(bw._bgzf_block(whole) + bw._EOF)                 # the same records, ended properly: fine
    h = try_ingest(p4, False)
    assert h
    L.pgh_bam_ingest_free(h)

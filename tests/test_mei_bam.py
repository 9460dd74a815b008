"""BAM ingest pinned on the reference's OWN BAM: demo/simulated_MEI (committed as data under tests/golden/simulated_MEI:
aln.sorted.bam -- bwa 0.6.1 + samtools, 24 000 records on two chromosomes --, its .bai, reference.fa, config, bd.txt and
`input`, 20 Pindel-text records the reference's authors derived from that BAM).

  * the BGZF / BAM / BAI decoder of pg_bam.hpp reads a samtools-written file: index queries == sequential scan == an
    independent decoding in this test (python gzip + struct);
  * all 20 records of `input` come out of the ingest field for field (name, bases, strand, position, MAPQ, insert size);
    every further record is explained by a named rule of fetch_func_SR (src/reader.cpp:1099-1151);
  * selection rules, reference reads and read-pair discovery == the independent restatements on this real file;
  * (-m gpu) `pindel_pg -f reference.fa -i config -b bd.txt` (the demo's own runme line) == CPU oracle + reporters on the
    ingested batch with the same window hints.
"""
import ctypes as C
import gzip
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from pindel_amd import hostio, hostlib
from tests import golden_util as gu
from tests.test_bam_ingest import F, _ref_reads_restated, _restated, _rp_restated, ingest

MEI = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "simulated_MEI")
BAM = os.path.join(MEI, "aln.sorted.bam")
ISZ, TAG = 500, "MEI"                                # the demo's config line: "aln.sorted.bam 500 MEI"


def _unpack(tmp_path):
    d = str(tmp_path)
    for f in ("aln.sorted.bam", "aln.sorted.bam.bai", "config", "bd.txt", "input"):
        shutil.copy(os.path.join(MEI, f), os.path.join(d, f))
    with gzip.open(os.path.join(MEI, "reference.fa.gz"), "rb") as s, open(os.path.join(d, "reference.fa"), "wb") as o:
        shutil.copyfileobj(s, o)
    return d


def _chroms(tmp_path):
    d = _unpack(tmp_path)
    chroms = hostio.load_fasta(os.path.join(d, "reference.fa"))
    assert [(n, len(s)) for n, s in chroms] == [("chr1", 297021), ("chr2", 303980)]     # (+1: the reference's EOF quirk)
    return d, chroms


def _decode_bam(path):
    """Independent decoding: BGZF is a series of gzip members; BAM records with struct."""
    d = gzip.open(path, "rb").read()
    assert d[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<i", d, 4)
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", d, p)
    p += 4
    refs = []
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", d, p)
        refs.append((d[p + 4:p + 4 + l - 1].decode(), struct.unpack_from("<i", d, p + 4 + l)[0]))
        p += 8 + l
    recs = []
    while p < len(d):
        bs, = struct.unpack_from("<i", d, p)
        tid, pos, lrn, mapq, _bin, ncig, flag, lseq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", d, p + 4)
        q = p + 36
        name = d[q:q + lrn - 1].decode()
        q += lrn
        cigar = [(c & 15, c >> 4) for c in struct.unpack_from("<%dI" % ncig, d, q)]
        q += 4 * ncig
        sq = d[q:q + (lseq + 1) // 2]
        seq = "".join("=ACMGRSVTWYHKDBN"[(sq[i >> 1] >> (4 * (1 - (i & 1)))) & 15] for i in range(lseq))
        q += (lseq + 1) // 2 + lseq
        tags, end = {}, p + 4 + bs
        while q < end:
            tag, ty = d[q:q + 2].decode(), chr(d[q + 2])
            q += 3
            if ty in "cCA":
                val = d[q] if ty != "c" else struct.unpack_from("<b", d, q)[0]
                q += 1
            elif ty in "sS":
                val = struct.unpack_from("<h" if ty == "s" else "<H", d, q)[0]
                q += 2
            elif ty in "iIf":
                val = struct.unpack_from({"i": "<i", "I": "<I", "f": "<f"}[ty], d, q)[0]
                q += 4
            elif ty in "ZH":
                e = d.index(b"\0", q)
                val = d[q:e].decode()
                q = e + 1
            else:
                raise AssertionError("array tag in the demo BAM?")
            tags[tag] = val
        recs.append(dict(qname=name, flag=flag, tid=tid, pos=pos, mapq=mapq, cigar=cigar, seq=seq, mtid=mtid, mpos=mpos,
                         tlen=tlen, tags=tags))
        p += 4 + bs
    return refs, recs


def _input_records():
    lines = open(os.path.join(MEI, "input")).read().split("\n")
    out = []
    for i in range(0, len(lines) - 2, 3):
        if not lines[i]:
            break
        d, c, p, ms, isz, tag = lines[i + 2].split()
        out.append((lines[i], lines[i + 1], d, c, int(p), int(ms), int(isz), tag))
    return out


def test_reference_bam_reproduces_the_reference_s_input_records(tmp_path):
    d, chroms = _chroms(tmp_path)
    refs, recs = _decode_bam(BAM)
    assert refs == [("chr1", 97021), ("chr2", 103979)] and len(recs) == 24000
    got = []
    for cid, (name, s) in enumerate(chroms):
        by_index = ingest(BAM, name, cid, len(s), 0, 5_000_000, ISZ, tag=TAG, use_index=True)
        by_scan = ingest(BAM, name, cid, len(s), 0, 5_000_000, ISZ, tag=TAG, use_index=False)
        assert by_index == by_scan
        # sub-windows through the .bai's bins and linear index == scan, and together == the whole chromosome
        parts = []
        for ws, we in ((0, 5000), (5000, 5450), (5450, 20000), (20000, 20400), (20400, 110000)):
            a = ingest(BAM, name, cid, len(s), ws, we, ISZ, tag=TAG, use_index=True)
            assert a == ingest(BAM, name, cid, len(s), ws, we, ISZ, tag=TAG, use_index=False)
            parts.append(a)
        assert sum(len(x) for x in parts) >= len(by_index)        # (a pair split over two windows loses its partner, never gains)
        got += [(g[0], g[1], g[2], name, g[3], g[4], g[5], TAG) for g in by_index]
    assert len(got) == 42
    want = _input_records()
    assert len(want) == 20
    # every record of `input`, field for field, in the file's order within each chromosome
    it = iter(got)
    for w in sorted(want, key=lambda r: r[3]):                    # chr1 records, then chr2 (stable: order of `input` kept)
        assert any(g == w for g in it), f"input record {w[0]} not reproduced (in order)"
    # ... and every further record is one of three named cases of fetch_func_SR (src/reader.cpp:1099-1151)
    by_name = {}
    for r in recs:
        by_name.setdefault(r["qname"], []).append(r)
    extra = [g for g in got if g not in set(want)]
    assert len(extra) == 22
    kinds = {"self_anchored_unmapped_twice": 0, "same_rule_as_input_not_listed": 0, "self_anchored_indel_read": 0}
    for g in extra:
        qname = g[0][1:].rpartition("/")[0]
        pair = by_name[qname]
        first, second = pair
        if g[5] == 0:
            # (a) an UNMAPPED read that precedes its mapped mate in the file: isWeirdRead -> build_record_SR(b1, b1) when
            # it is first seen (:1117-1119) and build_record_SR(b2, b2) AGAIN when the mate arrives (:1129-1131).  The
            # anchor is the read's own record: MAPQ 0, its own (= the mate's) position.
            assert first["flag"] & F["UNMAP"] and not second["flag"] & F["UNMAP"] and first["pos"] == second["pos"]
            assert sum(1 for x in got if x[0] == g[0] and x[5] == 0) == 2
            kinds["self_anchored_unmapped_twice"] += 1
        elif first["flag"] & F["UNMAP"] or second["flag"] & F["UNMAP"]:
            # (b) mapped anchor first, unmapped mate second: isGoodAnchor(b2) && isWeirdRead(b1) (:1140-1142) -- the rule
            # that yields read_6988 ... read_6992, which ARE in `input`; `input` lists five of the six such pairs of chr1
            assert qname == "read_6993" and not first["flag"] & F["UNMAP"] and second["flag"] & F["UNMAP"]
            assert ("@read_6992/2", "+", "chr1") in {(w[0], w[2], w[3]) for w in want}
            kinds["same_rule_as_input_not_listed"] += 1
        else:
            # (c) a mapped read with indels (92M3D1M2I5M, NM 6): isWeirdRead, self-anchored when first seen (:1117-1119);
            # its mate lies on the other chromosome and never joins it in this chromosome's fetch
            assert qname == "read_7012" and first["tags"]["NM"] == 6 and len(first["cigar"]) == 5 and first["mtid"] != first["tid"]
            kinds["self_anchored_indel_read"] += 1
    assert kinds == {"self_anchored_unmapped_twice": 20, "same_rule_as_input_not_listed": 1, "self_anchored_indel_read": 1}


def test_reference_bam_selection_rules_ref_reads_and_read_pairs_match_the_restatements(tmp_path):
    d, chroms = _chroms(tmp_path)
    refs, recs = _decode_bam(BAM)
    L = hostlib.lib()
    L.pgh_rp_events.restype = C.c_int64
    L.pgh_rp_events.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int32, C.c_char_p, C.c_uint32, C.c_uint32,
                                C.c_char_p, C.c_void_p, C.c_uint64]
    n_ref_reads = n_rp = 0
    for cid, (name, s) in enumerate(chroms):
        biol = len(s) - 200000
        for min_q in (0, 30):
            ref_reads = []
            got = ingest(BAM, name, cid, len(s), 0, 5_000_000, ISZ, tag=TAG, min_q=min_q, ref_reads=ref_reads)
            want = _restated(recs, cid, 0, 5_000_000, ISZ, min_q, biol)
            assert [(g[0], g[1], g[2], g[3], g[4], g[5]) for g in got] == [w[:6] for w in want]
            assert ref_reads == _ref_reads_restated(recs, cid, 0, 5_000_000, min_q)
            n_ref_reads += len(ref_reads)
        out = np.zeros(4 * 1024, dtype=np.uint32)
        n = L.pgh_rp_events(BAM.encode(), name.encode(), 0, 5_000_000, ISZ, TAG.encode(), 0, 100000, None, out.ctypes.data, 1024)
        want = _rp_restated(recs, cid, 0, 5_000_000, ISZ, 0, 100000)
        assert n == len(want)
        assert sorted(tuple(int(v) for v in out[4 * i:4 * i + 4]) for i in range(n)) == sorted(w[:4] for w in want)
        n_rp += n
    assert n_ref_reads > 40000           # ~12 000 clean pairs, each mate a reference read, at two quality cut-offs
    print("read-pair events on the demo BAM:", n_rp)


def _write_pindel_text(path, records):
    with open(path, "w") as f:
        for name, seq, d, chrom, pos, ms, isz, tag in records:
            f.write(f"{name}\n{seq}\n{d}\t{chrom}\t{pos}\t{ms}\t{isz}\t{tag}\n")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["runme_with_bd", "no_hints"])
def test_command_line_on_the_reference_demo_equals_oracle_and_reporters(tmp_path, mode):
    """`pindel_pg -f reference.fa -i config -b bd.txt` (demo/simulated_MEI/runme) on the MI355X == the CPU oracle on the
    ingested batch, with the window hints of the same run (read-pair events + bd.txt), through the same reporters."""
    from oracle import pyoracle
    from pindel_amd import binding
    d, chroms = _chroms(tmp_path)
    exe = os.path.join(os.path.dirname(binding.LIB_PATH), "pindel_pg")
    prefix = os.path.join(d, "gpu")
    extra = ["-b", os.path.join(d, "bd.txt")] if mode == "runme_with_bd" else ["-R", "false"]
    out = subprocess.run([exe, "-f", os.path.join(d, "reference.fa"), "-i", os.path.join(d, "config"), "-o", prefix] + extra,
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    # expected: ingest -> oracle close end -> hints for the reads that kept one -> oracle far end -> reporters
    L = hostlib.lib()
    L.pgh_window_hints.restype = C.c_int64
    L.pgh_window_hints.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.c_int32, C.c_int64, C.c_int64,
                                   C.c_int64, C.c_int32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_uint64]
    names = (C.c_char_p * 2)(b"chr1", b"chr2")
    p = pyoracle.make_params()
    seqs = [s for _, s in chroms]
    text, res = [], []
    n_close = n_far = 0
    for cid, (name, s) in enumerate(chroms):
        got = ingest(BAM, name, cid, len(s), 0, 5_000_000, ISZ, tag=TAG)
        text += [(g[0], g[1], g[2], name, g[3], g[4], g[5], TAG) for g in got]
        b = hostio.batch_from_lists([g[1].encode() for g in got], [g[2].encode() for g in got], [g[3] for g in got],
                                    [g[5] for g in got], [cid] * len(got))
        args = (b.seq, b.seq_off, b.anchor_strand, b.anchor_pos, b.insert_size, b.chr_id)
        bd = bd_off = None
        if mode == "runme_with_bd":
            close = pyoracle.search_batch(p, seqs, *args, do_far=False)
            last = np.array([int(close["close_pts"][i][close["close_cnt"][i] - 1]["abs_loc"]) if close["close_cnt"][i] else 0
                             for i in range(b.n)], dtype=np.uint32)
            off = np.zeros(b.n + 1, dtype=np.uint64)
            win = np.zeros(3 * 4096, dtype=np.int32)
            biol = len(s) - 200000
            n_ev = L.pgh_window_hints(os.path.join(d, "bd.txt").encode(), BAM.encode(), 2, names, cid, 0, min(5_000_000, biol),
                                      5_000_000, ISZ, TAG.encode(), 0, 100000, b.n, last.ctypes.data, off.ctypes.data, win.ctypes.data, 4096)
            assert n_ev >= 0, L.pgh_last_error()
            cnt = np.diff(off.astype(np.int64))
            keep = np.repeat(close["close_cnt"] > 0, cnt)
            cnt[close["close_cnt"] == 0] = 0
            w3 = win[:3 * int(off[-1])].reshape(-1, 3)[keep]
            bd = np.zeros(len(w3), dtype=pyoracle.WINDOW_DTYPE)
            bd["chr_id"], bd["start"], bd["end"] = w3[:, 0], w3[:, 1], w3[:, 2]
            bd_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
            assert len(bd) > 0                                         # the bd.txt events sit right at the reads' close ends
        r = pyoracle.search_batch(p, seqs, *args, bd=bd, bd_off=bd_off)
        res.append(r)
        n_close += int((r["close_cnt"] > 0).sum())
        n_far += int((r["far_cnt"] > 0).sum())
    assert f"close end {n_close}, far end {n_far}" in out.stdout, out.stdout[-600:]
    assert n_close >= 20
    reads_txt = os.path.join(d, "ingested.txt")
    _write_pindel_text(reads_txt, text)
    cat = {k: np.concatenate([r[k] for r in res]) for k in ("close_cnt", "far_cnt", "rc_flag")}
    co, cp = gu.csr_from_strided(cat["close_cnt"], np.concatenate([r["close_pts"] for r in res]))
    fo, fp = gu.csr_from_strided(cat["far_cnt"], np.concatenate([r["far_pts"] for r in res]))
    st = hostlib.default_settings(pyoracle.max_mismatch_table())
    want_prefix = os.path.join(d, "oracle")
    hostlib.call_from_points(os.path.join(d, "reference.fa"), reads_txt, want_prefix, st, co, cp, fo, fp, cat["rc_flag"])
    events = 0
    for suf in gu.SUFFIXES:
        got_l = gu.normalise(open(f"{prefix}_{suf}", "rb").read())
        want_l = gu.normalise(open(f"{want_prefix}_{suf}", "rb").read())
        assert got_l == want_l, f"_{suf} differs"
        events += sum(1 for x in want_l if x.startswith(b"####"))
    print(mode, "close", n_close, "far", n_far, "events", events)

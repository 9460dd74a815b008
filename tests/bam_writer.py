"""Test-side BAM + BAI writer (pure Python: struct + zlib).  The snapshot has neither BAM fixtures nor
samtools/htslib, so the tests build their own files from the SAM/BAM specification: BGZF blocks, BAM records,
and the binning + linear index of the .bai.  Records are written in the order given (the tests control it)."""
import struct
import zlib

FLAG = dict(PAIRED=1, PROPER=2, UNMAP=4, MUNMAP=8, REVERSE=16, MREVERSE=32, READ1=64, READ2=128, SECONDARY=256,
            QCFAIL=512, DUP=1024)
_CIGAR_OPS = "MIDNSHP=X"
_NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def cigar_ops(text):
    """'50M2D48M' -> [(0, 50), (2, 2), (0, 48)]"""
    out, num = [], ""
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            out.append((_CIGAR_OPS.index(ch), int(num)))
            num = ""
    return out


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def _end_pos(rec):
    if rec["flag"] & FLAG["UNMAP"] or not rec["cigar"]:
        return rec["pos"] + 1
    span = sum(n for op, n in rec["cigar"] if op in (0, 2, 3, 7, 8))
    return rec["pos"] + (span or 1)


def encode_record(rec):
    qname = rec["qname"].encode() + b"\0"
    cigar = rec.get("cigar") or []
    seq = rec.get("seq", "")
    packed = bytearray((len(seq) + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= _NT16[ch] << (4 if i % 2 == 0 else 0)
    aux = b""
    for tag, val in (rec.get("tags") or {}).items():
        if isinstance(val, int):
            aux += tag.encode() + (b"C" + struct.pack("<B", val) if 0 <= val < 256 else b"i" + struct.pack("<i", val))
        else:
            aux += tag.encode() + b"Z" + str(val).encode() + b"\0"
    end = _end_pos(rec)
    body = struct.pack("<iiBBHHHiiii", rec["tid"], rec["pos"], len(qname), rec.get("mapq", 0),
                       reg2bin(max(rec["pos"], 0), max(end, 1)), len(cigar), rec["flag"], len(seq),
                       rec.get("mtid", -1), rec.get("mpos", -1), rec.get("tlen", 0))
    body += qname + b"".join(struct.pack("<I", (n << 4) | op) for op, n in cigar) + bytes(packed) + b"\xff" * len(seq) + aux
    return struct.pack("<i", len(body)) + body


def _bgzf_block(data, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = 12 + 6 + len(comp) + 8
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def write_bam(path, refs, records, with_index=True, block_bytes=0xff00, header_text="@HD\tVN:1.6\tSO:coordinate\n"):
    """refs: [(name, length)]; records: dicts with qname, flag, tid, pos, mapq, cigar [(op, len)], seq, mtid, mpos,
    tlen, tags.  Returns the number of BGZF blocks."""
    head = b"BAM\1" + struct.pack("<i", len(header_text)) + header_text.encode() + struct.pack("<i", len(refs))
    for name, length in refs:
        head += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", length)
    stream = bytearray(head)
    starts = []
    for rec in records:
        starts.append(len(stream))
        stream += encode_record(rec)
    starts.append(len(stream))
    # blocks of the uncompressed stream -> file offsets
    blocks, file_off, out = [], 0, bytearray()
    for u0 in range(0, len(stream), block_bytes):
        blk = _bgzf_block(bytes(stream[u0:u0 + block_bytes]))
        blocks.append((u0, file_off))
        out += blk
        file_off += len(blk)
    out += _EOF
    with open(path, "wb") as fh:
        fh.write(out)

    def voff(u):
        if u >= len(stream):
            return file_off << 16                      # the end-of-file marker block
        bi = u // block_bytes
        return (blocks[bi][1] << 16) | (u - blocks[bi][0])

    if with_index:
        bins = [dict() for _ in refs]
        linear = [dict() for _ in refs]
        for k, rec in enumerate(records):
            if rec["tid"] < 0:
                continue
            beg, end = max(rec["pos"], 0), max(_end_pos(rec), 1)
            v0, v1 = voff(starts[k]), voff(starts[k + 1])
            chunks = bins[rec["tid"]].setdefault(reg2bin(beg, end), [])
            if chunks and chunks[-1][1] == v0:
                chunks[-1][1] = v1
            else:
                chunks.append([v0, v1])
            for w in range(beg >> 14, ((end - 1) >> 14) + 1):
                lin = linear[rec["tid"]]
                lin[w] = min(lin.get(w, v0), v0)
        with open(path + ".bai", "wb") as fh:
            fh.write(b"BAI\1" + struct.pack("<i", len(refs)))
            for t in range(len(refs)):
                fh.write(struct.pack("<i", len(bins[t])))
                for b, chunks in bins[t].items():
                    fh.write(struct.pack("<Ii", b, len(chunks)))
                    for c in chunks:
                        fh.write(struct.pack("<QQ", c[0], c[1]))
                n_intv = (max(linear[t]) + 1) if linear[t] else 0
                fh.write(struct.pack("<i", n_intv))
                last = 0
                for w in range(n_intv):
                    last = linear[t].get(w, last)
                    fh.write(struct.pack("<Q", last))
    return len(blocks)

"""Golden data of the reference's regression test (tests/golden/sim1chrVs2) and the comparison
used to pin the search against it."""
import gzip
import os
import re
import shutil

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sim1chrVs2")
SUFFIXES = ("D", "SI", "TD", "INV")


def unpack(tmp_path):
    """Decompress reference + reads into tmp_path; returns (fasta_path, reads_path)."""
    fa = os.path.join(str(tmp_path), "sim1chrVs2.fa")
    reads = os.path.join(str(tmp_path), "reads.txt")
    with gzip.open(os.path.join(GOLD, "sim1chrVs2.fa.gz"), "rb") as s, open(fa, "wb") as d:
        shutil.copyfileobj(s, d)
    shutil.copy(os.path.join(GOLD, "sim1chrVs2.fa.fai"), fa + ".fai")
    with gzip.open(os.path.join(GOLD, "simulated_test.out_CloseEndMapped.gz"), "rb") as s, open(reads, "wb") as d:
        shutil.copyfileobj(s, d)
    return fa, reads


def gold_lines(suffix):
    with gzip.open(os.path.join(GOLD, f"simulated_test.out_{suffix}.gz"), "rb") as fh:
        return normalise(fh.read())


def normalise(data: bytes):
    """Split into lines and mask the two reference-coverage integers of each per-sample block of
    an event header (BAM-only information; `0 0` for Pindel-text input)."""
    out = []
    for line in data.split(b"\n"):
        if b"\tSupports " in line and b"NumSupSamples " in line:
            head, _, tail = line.partition(b"NumSupSamples ")
            parts = tail.split(b"\t")
            parts = parts[:2] + [re.sub(rb"^(\S+) -?\d+ -?\d+ ", rb"\1 X X ", x) for x in parts[2:]]
            line = head + b"NumSupSamples " + b"\t".join(parts)
        out.append(line)
    return out


def assert_reports_match_gold(prefix):
    for suf in SUFFIXES:
        with open(f"{prefix}_{suf}", "rb") as fh:
            got = normalise(fh.read())
        want = gold_lines(suf)
        assert len(got) == len(want), f"_{suf}: {len(got)} lines, gold has {len(want)}"
        for i, (a, b) in enumerate(zip(got, want)):
            assert a == b, f"_{suf} line {i + 1} differs:\n got  {a[:200]!r}\n gold {b[:200]!r}"


def csr_from_strided(cnt, pts):
    n = len(cnt)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(cnt)
    flat = np.concatenate([pts[i][:cnt[i]] for i in range(n)]) if n else pts.reshape(-1)[:0]
    return off, flat

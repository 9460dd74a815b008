"""Diagnostics (CPU only): BAM ingest rate of pg_bam.hpp by thread count on a synthetic coordinate-sorted BAM written by
the test-side writer (tests/bam_writer.py): 10 % unmapped mates, 10 % soft-clipped mates, 80 % clean pairs.
    python scripts/bam_ingest_rate.py [n_pairs]
"""
import ctypes as C
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests import bam_writer as bw

F = bw.FLAG
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
if len(sys.argv) > 2:                                       # child: time the ingest with the PGH_THREADS of the environment
    from tests.test_bam_ingest import _lib
    L = _lib()
    nr, nb = C.c_uint64(), C.c_uint64()
    best = 1e9
    for _ in range(3):
        t0, tot = time.time(), 0
        for ws in range(0, 50_000_000, 5_000_000):
            h = L.pgh_bam_ingest(b"/tmp/ingest_rate.bam", b"chrZ", 0, 50_200_000, ws, ws + 5_000_000, 500, b"S", 0, 100000, 1,
                                 C.byref(nr), C.byref(nb))
            assert h
            tot += nr.value
            L.pgh_bam_ingest_free(h)
        best = min(best, time.time() - t0)
    print(f"PGH_THREADS={os.environ.get('PGH_THREADS', 'default'):>7}: {tot} split-read candidates of {2 * n} records in {best:.3f} s = "
          f"{2 * n / best / 1e6:.2f} M records/s")
    sys.exit(0)
rng = np.random.default_rng(1)
ref_len = 50_000_000
bases = np.array(list("ACGT"))
pos = np.sort(rng.integers(1000, ref_len - 2000, n))
recs = []
for k in range(n):
    p = int(pos[k])
    seq = "".join(bases[rng.integers(0, 4, 100)])
    a = dict(qname=f"q{k}", flag=F["PAIRED"] | F["READ1"] | F["MREVERSE"], tid=0, pos=p, mapq=60, cigar=[(0, 100)], seq=seq,
             mtid=0, mpos=p + 300, tags={"NM": 0})
    if k % 10 == 0:
        b = dict(qname=f"q{k}", flag=F["PAIRED"] | F["READ2"] | F["UNMAP"], tid=0, pos=p, mapq=0, cigar=[], seq=seq[::-1], mtid=0, mpos=p)
    elif k % 10 == 1:
        b = dict(qname=f"q{k}", flag=F["PAIRED"] | F["READ2"] | F["REVERSE"], tid=0, pos=p + 300, mapq=60, cigar=[(4, 30), (0, 70)],
                 seq=seq, mtid=0, mpos=p, tags={"NM": 1})
    else:
        b = dict(qname=f"q{k}", flag=F["PAIRED"] | F["READ2"] | F["REVERSE"], tid=0, pos=p + 300, mapq=60, cigar=[(0, 100)], seq=seq,
                 mtid=0, mpos=p, tags={"NM": 0})
    recs += [a, b]
recs.sort(key=lambda r: r["pos"])
bw.write_bam("/tmp/ingest_rate.bam", [("chrZ", ref_len)], recs, with_index=True)
print("BAM:", os.path.getsize("/tmp/ingest_rate.bam") // 1000000, "MB,", 2 * n, "records")
for t in ("1", "2", "4", "8", "16"):
    subprocess.run([sys.executable, __file__, str(n), "child"], env=dict(os.environ, PGH_THREADS=t))

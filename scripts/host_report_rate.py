"""Diagnostics (CPU only): time the host classifiers + reporters on a synthetic read set whose UniquePoints come
from the CPU oracle.      PGH_TIMING=1 python scripts/host_report_rate.py [n_reads] [chr_len]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import pyoracle
from pindel_amd import hostlib, synth
from tests import golden_util as gu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 6_000_000
out = "/tmp/host_rate"
os.makedirs(out, exist_ok=True)
ref = synth.make_reference(L, seed=3)
biol = ref[100000:-100000]
with open(f"{out}/ref.fa", "wb") as f:
    f.write(b">chrS\n")
    for i in range(0, len(biol), 60):
        f.write(biol[i:i + 60] + b"\n")
b = synth.make_reads(ref, n, seed=4)
order = np.argsort(b.anchor_pos, kind="stable")
seq = np.asarray(b.seq).reshape(n, 100)
with open(f"{out}/reads.txt", "wb") as f:
    for k, i in enumerate(order):
        f.write(b"@r%d/1\n" % k + seq[i].tobytes() + b"\n" + bytes([b.anchor_strand[i]]) +
                b"\tchrS\t%d\t60\t500\tS1\n" % int(b.anchor_pos[i]))
p = pyoracle.make_params()
seq_sorted = seq[order].reshape(-1)
off = (np.arange(n + 1) * 100).astype(np.uint64)
t0 = time.time()
r = pyoracle.search_batch(p, [ref], seq_sorted, off, b.anchor_strand[order], b.anchor_pos[order],
                          b.insert_size[order], b.chr_id[order])
print("oracle", round(time.time() - t0, 1), "s")
co, cp = gu.csr_from_strided(r["close_cnt"], r["close_pts"])
fo, fp = gu.csr_from_strided(r["far_cnt"], r["far_pts"])
st = hostlib.default_settings(pyoracle.max_mismatch_table())
t0 = time.time()
hostlib.call_from_points(f"{out}/ref.fa", f"{out}/reads.txt", f"{out}/out", st, co, cp, fo, fp, r["rc_flag"])
dt = time.time() - t0
size = sum(os.path.getsize(f"{out}/out{s}") for s in ("_D", "_SI", "_TD", "_INV"))
print(f"load + classify + report: {dt:.2f} s for {n} reads, {size / 1e6:.0f} MB of reports")

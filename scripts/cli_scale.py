"""Diagnostics: time the pindel_pg command line on a synthetic FASTA + Pindel-text input.
    python scripts/cli_scale.py [n_reads] [chr_len]
"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pindel_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
out = "/tmp/cli_scale"
os.makedirs(out, exist_ok=True)
ref = synth.make_reference(L, seed=3)
biol = ref[100000:-100000]
with open(f"{out}/ref.fa", "wb") as f:
    f.write(b">chrS\n")
    for i in range(0, len(biol), 60):
        f.write(biol[i:i + 60] + b"\n")
with open(f"{out}/ref.fa.fai", "w") as f:
    f.write(f"chrS\t{len(biol)}\t6\t60\t61\n")
b = synth.make_reads(ref, n, seed=4)
order = np.argsort(b.anchor_pos, kind="stable")
seq = np.asarray(b.seq).reshape(n, 100)
t0 = time.time()
with open(f"{out}/reads.txt", "wb") as f:
    for k, i in enumerate(order):
        f.write(b"@r%d/1\n" % k + seq[i].tobytes() + b"\n" + bytes([b.anchor_strand[i]]) +
                b"\tchrS\t%d\t60\t500\tS1\n" % int(b.anchor_pos[i]))
print("wrote input in", round(time.time() - t0, 1), "s")
t0 = time.time()
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pindel_amd", "pindel_pg")
r = subprocess.run([exe, "-f", f"{out}/ref.fa", "-p", f"{out}/reads.txt", "-o", f"{out}/out"], capture_output=True, text=True)
dt = time.time() - t0
print(r.stdout[-2000:], r.stderr[-2000:])
print(f"pindel_pg: {dt:.1f} s for {n} reads = {n / dt / 1e6:.3f} M reads/s end to end")
for sfx in ("_D", "_SI", "_TD", "_INV"):
    p = f"{out}/out{sfx}"
    print(sfx, os.path.getsize(p) if os.path.exists(p) else None)

"""Diagnostics: what does a read that fails all four close-end attempts cost next to one that succeeds at attempt 0?
A -DPG_STOP=22 build (returns from a read when its close end is done: scripts/build_ladder.sh) runs the CLOSE END ONLY of two
batches on the same anchors: (a) split reads whose close end is found at attempt 0 (deletions, sequenced in the orientation attempt 0
tries), (b) the same records with random bases (no close end: (R0,seq) (R0,RC) (R1,RC) (R1,seq), pindel.cpp:2537-2575).
  python scripts/retry_cost.py <libpindel_pg_stop22.so> <libpindel_pg.so> [reads] [read_len]
The shipped library runs both batches too (whole search) and says how many reads of each kept a close end."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pindel_amd import binding, synth

stop_lib, full_lib = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
L = int(sys.argv[4]) if len(sys.argv) > 4 else 150
dev = torch.device("cuda", 0)
ref = synth.make_reference(62_435_964, seed=20260927, device=dev)
ok = synth.make_reads(ref, n, seed=20260930, device=dev, read_len=L, mix=(1.0, 0.0, 0.0, 0.0, 0.0), rc_retry_frac=0.0)
rng = np.random.default_rng(7)
junk = type(ok)(seq=np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, len(ok.seq))], seq_off=ok.seq_off,
                anchor_strand=ok.anchor_strand, anchor_pos=ok.anchor_pos, insert_size=ok.insert_size, chr_id=ok.chr_id)


def run(lib, batch):
    binding.use_library(lib)
    eng = binding.Engine()
    eng.load_reference([("20", ref)])
    db = eng.upload(batch)
    ms = []
    for _ in range(4):
        eng.search_device(db)
        ms.append(eng.last_stats()[0])
    res = eng.download(db)
    close = int((res.close_off[1:] > res.close_off[:-1]).sum())
    eng.free_device_batch(db)
    eng.close()
    return min(ms), close


full_ok, close_ok = run(full_lib, ok)
full_junk, close_junk = run(full_lib, junk)
stop_ok, _ = run(stop_lib, ok)
stop_junk, _ = run(stop_lib, junk)
print(f"{n} x {L} bp.  split reads: {close_ok} with a close end ({100.0 * close_ok / n:.1f} %), whole search {full_ok:.3f} ms; "
      f"random reads: {close_junk} with a close end ({100.0 * close_junk / n:.1f} %), whole search {full_junk:.3f} ms")
print(f"close end only (PG_STOP = 22): split reads {stop_ok:.3f} ms, random reads {stop_junk:.3f} ms -> a read without a close end costs "
      f"{stop_junk / stop_ok:.2f} x a read that succeeds at attempt 0")

#!/usr/bin/env python
"""Run bench.py with the given args and print a one-line summary (dev helper)."""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "bench.py"] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(out.stdout[-2000:], out.stderr[-2000:])
    sys.exit(1)
d = json.loads(line[-1])
print(f"cand/read {d['config'].get('candidates_per_read', 0):.1f}  Mreads/s {d['value'] / 1e6:.2f}  kernel_ms {d['roofline']['kernel_ms']:.2f}  ms/step {d['ms_per_step']:.2f}  "
      f"GB/s {d['roofline']['achieved']:.1f}")

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04e
python bench.py --no-cpu-baseline > gpurun_out/r04e/bench.json 2> gpurun_out/r04e/bench.err; tail -3 gpurun_out/r04e/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04e/bench.json') if l.startswith('{')][-1])
print(d['value'], d['config']['host_path_reads_per_s'], d['config']['host_path_pinned_inputs_reads_per_s'], d['config']['pack_ms_per_step'])
PY
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden_pin.py -m gpu -x -q 2>&1 | tail -3

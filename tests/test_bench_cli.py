"""-m gpu: the exact commands the driver's BENCH / SCALE tiers run, as tested code paths.

`bench.py --gpus N` re-executes itself under torch.distributed.run (self_spawn), every rank initialises
torch.distributed, shards the ONE seeded batch (--scaling strong), and rank 0 gathers the per-read digests in rank
order: config.result_sha256 must not depend on N.  A 1-GPU box cannot give two ranks a GPU each, so
PG_BENCH_SHARE_GPU=1 lets both ranks use device 0 (gloo for the rendezvous, the data path is unchanged: there is no
collective on it).  PG_BENCH_FORCE_DIST=1 runs the N = 1 line through the RCCL path the 8-GPU run takes:
init_process_group(backend="nccl", device_id=...), barrier, all_reduce(MAX) of the step time."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--scaling", "strong", "--reads", "400000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-path"]


def _run(cmd, **env):
    e = dict(os.environ)
    e.update(env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.pop("WORLD_SIZE", None), e.pop("RANK", None), e.pop("LOCAL_RANK", None)
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]            # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def one_gpu_line():
    return _run([sys.executable, "bench.py", "--gpus", "1"] + COMMON)


def test_two_ranks_self_spawned_give_the_one_rank_digest(one_gpu_line):
    two = _run([sys.executable, "bench.py", "--gpus", "2"] + COMMON, PG_BENCH_SHARE_GPU="1")
    assert two["n_gpus"] == 2 and one_gpu_line["n_gpus"] == 1
    assert two["scaling"] == "strong"
    assert two["config"]["reads_total"] == one_gpu_line["config"]["reads_total"] == 400000
    assert two["config"]["reads_per_gpu"] == 200000
    assert two["config"]["result_sha256"] == one_gpu_line["config"]["result_sha256"]
    assert two["value"] > 0 and two["ms_per_step"] > 0
    for k in ("roofline", "metric", "unit", "higher_is_better", "dtype", "data"):
        assert k in two


def test_the_drivers_launcher_command_line(one_gpu_line):
    # what the driver runs for N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + COMMON
    two = _run(cmd, PG_BENCH_SHARE_GPU="1")
    assert two["n_gpus"] == 2
    assert two["config"]["result_sha256"] == one_gpu_line["config"]["result_sha256"]
    # weak scaling (the default): every rank its own reads, the aggregate counts both
    weak = _run([sys.executable, "bench.py", "--gpus", "2", "--reads", "200000", "--steps", "2", "--warmup", "1",
                 "--no-cpu-baseline", "--no-host-path"], PG_BENCH_SHARE_GPU="1")
    assert weak["scaling"] == "weak" and weak["config"]["reads_total"] == 400000 and weak["n_gpus"] == 2


def test_rccl_path_on_one_gpu(one_gpu_line):
    # the NCCL (= RCCL) process group with device_id, the barrier and the MAX all-reduce execute on the GPU
    d = _run([sys.executable, "bench.py", "--gpus", "1"] + COMMON, PG_BENCH_FORCE_DIST="1", MASTER_PORT=str(_free_port()))
    assert d["n_gpus"] == 1
    assert d["config"]["result_sha256"] == one_gpu_line["config"]["result_sha256"]
    assert d["roofline"]["kernel_ms"] > 0 and d["ms_per_step"] >= d["roofline"]["kernel_ms"] * 0.98


def test_a_launcher_of_the_wrong_size_is_refused():
    e = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + COMMON, cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "launcher started 3" in (p.stdout + p.stderr)


def test_eight_ranks_on_one_device_strong_and_weak(one_gpu_line):
    # the SCALE tier's largest launch (--gpus 8) as far as a 1-GPU box can run it: eight processes, eight contexts on device 0,
    # the read shards of ONE batch (strong) / eight batches (weak), one JSON line, the N = 1 digest
    eight = _run([sys.executable, "bench.py", "--gpus", "8"] + COMMON, PG_BENCH_SHARE_GPU="1")
    assert eight["n_gpus"] == 8 and eight["scaling"] == "strong"
    assert eight["config"]["reads_total"] == 400000 and eight["config"]["reads_per_gpu"] == 50000
    assert eight["config"]["result_sha256"] == one_gpu_line["config"]["result_sha256"]
    weak = _run([sys.executable, "bench.py", "--gpus", "8", "--reads", "100000", "--steps", "2", "--warmup", "1",
                 "--no-cpu-baseline", "--no-host-path"], PG_BENCH_SHARE_GPU="1")
    assert weak["n_gpus"] == 8 and weak["scaling"] == "weak" and weak["config"]["reads_total"] == 800000
    assert weak["value"] > 0


def test_value_covers_the_pack_stage(one_gpu_line):
    # round 6: a step = pack + search (pg_device_batch_pack_search); the search-only figure of rounds 1-5 rides along, measured
    # outside the timed region
    c = one_gpu_line["config"]
    assert c["device_ms_per_step"] > 0 and c["pack_ms_standalone"] > 0 and c["search_ms_standalone"] > 0
    assert one_gpu_line["ms_per_step"] >= c["device_ms_per_step"] * 0.98
    assert c["device_ms_per_step"] >= c["search_ms_standalone"] * 0.98          # the pack is in the step
    assert c["value_search_only"] > one_gpu_line["value"]

#!/usr/bin/env python
"""bench.py -- one-end-anchored reads/s through the split-read search (close end + far end).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (pg_device_batch_search: close-end + far-end kernel)
over one batch of synthetic reads already resident in HBM.  Workload at N=1 = BASELINE.json
configs[2]: 10 M x 100 bp one-end-anchored reads on a chr20-shaped reference, all SV types,
Pindel defaults (-x 2 ...).  Reads shard across ranks (each rank generates its own 10 M reads,
reference replicated per GPU, no collective on the data path) -> weak scaling.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHR20_LEN = 62_435_964       # demo/hs_ref_chr20.fa.fai:1
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAFFIC_FILE = "v6_hbm_traffic.json"


def cpu_baseline(chroms, batch, params_kw, budget_s=15.0, bd=None, bd_off=None):
    """Time the CPU restatement (oracle, OpenMP over reads) on a bounded sample of the same reads."""
    from oracle import pyoracle
    cores = os.cpu_count() or 1
    p = pyoracle.make_params(**params_kw)
    seqs = [s for _, s in chroms]

    def run(n):
        b = batch.slice(0, n)
        w, woff = (bd[:int(bd_off[n])], bd_off[:n + 1]) if bd is not None else (None, None)
        t0 = time.perf_counter()
        pyoracle.search_batch(p, seqs, b.seq, b.seq_off, b.anchor_strand, b.anchor_pos,
                              b.insert_size, b.chr_id, bd=w, bd_off=woff, n_threads=cores, keep_points=False)
        return time.perf_counter() - t0

    n0 = min(batch.n, 20000)
    run(n0)                              # warms the pages and the OpenMP pool
    n1 = min(batch.n, 200000)
    t1 = run(n1)                         # calibrates the rate at a size where all threads are busy
    n2 = int(min(batch.n, max(n1, n1 / max(t1, 1e-6) * budget_s)))
    if n2 > n1:
        n1, t1 = n2, run(n2)
    return {"value": n1 / t1, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"first {n1} reads of the rank-0 batch, close+far end, OpenMP {cores} threads, {t1:.1f} s"}


def measured_traffic(args):
    """HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE with the gfx950
    x2 correction + WRITE_SIZE, MI355X_MICROARCH.md).  PMC counters cannot be read from inside this
    process, so the per-read figure measured with rocprofv3 on this same workload is scaled by the reads
    of one launch; null for any other workload."""
    path = os.path.join(ROOT, "profiles", "r01", TRAFFIC_FILE)
    if args.read_len != 100 or args.max_range_index != 2 or args.workload != "sv10m" or not os.path.exists(path):
        return None, None
    with open(path) as fh:
        t = json.load(fh)
    per_read = t["fetch_bytes_per_read"] + t["write_bytes_per_read_uncalibrated"]
    return per_read * args.reads, f"profiles/r01/{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, per read x reads per launch)"


def valu_issue(args, kernel_ms):
    """The kernel is VALU-issue bound, not HBM bound: VALU instructions per read from
    the committed PMC pass x 4 cycles per wave64 instruction over 1024 SIMDs at 2.4 GHz, against the kernel
    time measured in this run.  None for workloads the PMC pass was not taken on."""
    path = os.path.join(ROOT, "profiles", "r01", "v6_pmc_per_read.txt")
    if (args.read_len != 100 or args.max_range_index != 2 or args.workload != "sv10m" or not os.path.exists(path)
            or kernel_ms <= 0):
        return None
    valu = None
    with open(path) as fh:
        for line in fh:
            f = line.split()
            if f and f[0] == "SQ_INSTS_VALU":
                valu = float(f[1])
    if valu is None:
        return None
    issue_ms = valu * args.reads * 4.0 / (1024 * 2.4e9) * 1e3
    return {"valu_insts_per_read": valu, "valu_issue_ms": issue_ms, "valu_busy_frac": issue_ms / kernel_ms,
            "source": "profiles/r01/v6_pmc_per_read.txt (SQ_INSTS_VALU), 256 CUs x 4 SIMDs, 4 cycles per wave64 VALU instruction, 2.4 GHz"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--chr-len", type=int, default=CHR20_LEN)
    ap.add_argument("--max-range-index", type=int, default=2, help="Pindel -x")
    ap.add_argument("--seed", type=int, default=20260927)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["sv10m", "colo-bd"], default="sv10m",
                    help="sv10m = BASELINE configs[2] (default); colo-bd = configs[1]-shaped: deletions only, "
                         "1 M reads unless --reads is given, per-read BreakDancer window hints (synthetic)")
    args = ap.parse_args()

    import torch
    from pindel_amd import binding, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("PG_BENCH_FORCE_DIST"):     # the env knob exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    params_kw = dict(max_range_index=args.max_range_index)
    # ---- synthetic inputs: reference identical on every rank, reads sharded by rank
    ref = synth.make_reference(args.chr_len, seed=args.seed, device=dev)
    chroms = [("20", ref)]
    bd = bd_off = None
    if args.workload == "colo-bd":
        # configs[1]-shaped (SURVEY.md 8d cfg 2): deletions only + BreakDancer hints.  The COLO-829 inputs are
        # not in the image, so the hints are synthetic: 0-2 windows of 600 bases per read within 20 kb
        # downstream/upstream of the anchor (where a deletion's far end lies).
        import numpy as np
        if args.reads == 10_000_000:
            args.reads = 1_000_000
        batch = synth.make_reads(ref, args.reads, read_len=args.read_len, seed=args.seed + 1 + rank,
                                 device=dev, mix=(1.0, 0.0, 0.0, 0.0, 0.0))
        rng = np.random.default_rng(args.seed + 77 + rank)
        k = rng.integers(0, 3, batch.n)
        bd_off = np.concatenate([[0], np.cumsum(k)]).astype(np.uint64)
        owner = np.repeat(np.arange(batch.n), k)
        sign = np.where(batch.anchor_strand[owner] == ord("+"), 1, -1)
        centre = batch.anchor_pos[owner].astype(np.int64) + 100000 + sign * rng.integers(300, 20000, len(owner))
        centre = np.clip(centre, 100400, len(ref) - 100400)
        bd = np.zeros(len(owner), dtype=binding.WINDOW_DTYPE)
        bd["chr_id"], bd["start"], bd["end"] = 0, centre - 300, centre + 300
    else:
        batch = synth.make_reads(ref, args.reads, read_len=args.read_len, seed=args.seed + 1 + rank,
                                 device=dev)
    eng = binding.Engine(device=local_rank, **params_kw)
    eng.load_reference(chroms)
    dbatch = eng.upload(batch)           # inputs resident in HBM before the timed region
    if bd is not None:
        eng.set_windows(dbatch, bd, bd_off)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.search_device(dbatch)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        eng.search_device(dbatch)        # synchronous: returns when the kernel finished
        kernel_ms.append(eng.last_stats()[0])
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    alg_bytes = eng.algorithmic_bytes(dbatch)     # per launch (outside the timed region)
    n_runs = eng.last_stats()[1]
    res = eng.download(dbatch)
    n_close = int((res.close_off[1:] > res.close_off[:-1]).sum())
    n_far = int((res.far_off[1:] > res.far_off[:-1]).sum())

    if rank == 0:
        avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic(args)
        out = {
            "metric": "one-end-anchored reads/sec through split-read search (close end + far end)",
            "value": world * args.reads * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": ((f"BASELINE configs[2]: synthetic {args.reads} x {args.read_len} bp "
                              if args.workload == "sv10m" else
                              f"BASELINE configs[1]-shaped (deletions only, synthetic BreakDancer window hints): "
                              f"{args.reads} x {args.read_len} bp ") +
                             f"one-end-anchored reads per GPU on a chr20-shaped reference "
                             f"({args.chr_len} bp), " +
                             ("all SV types (D/SI/TD/INV/none)" if args.workload == "sv10m" else "deletions") +
                             f", Pindel defaults "
                             f"-x {args.max_range_index} -a 1 -m 3 -u 0.02 -e 0.01 -E 0.95 -H 8"),
                "reads_per_gpu": args.reads, "read_len": args.read_len, "insert_size": 500,
                "parallelism": f"reads sharded over {world} GPU(s), reference replicated, no collective",
                "reads_with_close_end": n_close, "reads_with_far_end": n_far, "runs_out": int(n_runs),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "pg_search_kernel", "kernel_ms": avg_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "actual_bound": "valu-issue", "valu": valu_issue(args, avg_ms),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(chroms, batch, params_kw, bd=bd, bd_off=bd_off)
        print(json.dumps(out), flush=True)
    eng.free_device_batch(dbatch)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

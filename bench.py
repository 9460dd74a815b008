#!/usr/bin/env python
"""bench.py -- one-end-anchored reads/s through the split-read search (close end + far end).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (pg_device_batch_search: close-end + far-end kernel) over one batch
of synthetic reads already resident in HBM.  Workload at N=1 = BASELINE.json configs[2]: 10 M x 100 bp
one-end-anchored reads on a chr20-shaped reference, all SV types, Pindel defaults (-x 2 ...).

Multi-GPU: reads shard by index, reference replicated per GPU, no collective on the data
path (torch.distributed only for the barrier and the max-over-ranks time).
  --scaling weak    (default) every rank generates its own --reads reads
  --scaling strong  ONE seeded batch of --reads reads, rank r searches shard.shard_bounds(...)[r]; rank 0
                    gathers the per-read digests in rank order: config.result_sha256 must equal the N=1 value
`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run with N ranks.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHR20_LEN = 62_435_964       # demo/hs_ref_chr20.fa.fai:1
# wgs-real: 5 % split reads (D / SI / TD / INV), 95 % "no event" -- of which one in five still finds a (chance or mismatch-rich) close
# end: three reads in four (76 %) walk all four attempts and leave without one
WGS_REAL_MIX = (0.02, 0.01, 0.01, 0.01, 0.95)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# the round's committed counter files (scripts/profile_round.sh); PMC counters cannot be read from inside this process
PROFILE_ROUND = next((r for r in ("r06", "r05", "r04") if os.path.exists(os.path.join(ROOT, "profiles", r, "hbm_traffic.json"))), "r05")
PROFILE_DIR = os.path.join(ROOT, "profiles", PROFILE_ROUND)
TRAFFIC_FILE = "hbm_traffic.json"
PMC_FILE = "pmc_sq.txt"
CALIB_FILE = "profiles/r04/issue_calibration.txt"      # (the in-situ cost per extra instruction was measured once, in round 4)
UBENCH_FILE = "profiles/r03/ubench_issue_rates.txt"


def cpu_baseline(chroms, batch, params_kw, budget_s=10.0, bd=None, bd_off=None, thread_points=True):
    """Time the CPU restatement (oracle, OpenMP over reads) on bounded samples of the same reads at several thread
    counts and report the best one (the restatement keeps the reference's list-per-level data flow; SURVEY.md
    section 6 has the reference binary's own numbers: about 6.5 k reads/s per core)."""
    from oracle import pyoracle
    cores = os.cpu_count() or 1
    p = pyoracle.make_params(**params_kw)
    seqs = [s for _, s in chroms]

    def run(n, threads):
        b = batch.slice(0, n)
        w, woff = (bd[:int(bd_off[n])], bd_off[:n + 1]) if bd is not None else (None, None)
        t0 = time.perf_counter()
        pyoracle.search_batch(p, seqs, b.seq, b.seq_off, b.anchor_strand, b.anchor_pos,
                              b.insert_size, b.chr_id, bd=w, bd_off=woff, n_threads=threads, keep_points=False)
        return time.perf_counter() - t0

    n0 = min(batch.n, 20000)
    run(n0, cores)                       # warms the pages and the OpenMP pool
    # The restatement allocates per read: it stops scaling (and then slows down) well before all hardware threads
    # of a 256-thread host are in use.  Measure a few thread counts on ~2 s samples and report the BEST as the baseline.
    rate = {}
    for th in sorted({1, 16, 32, 64, 128, cores}):
        if th > cores:
            continue
        guess = 25_000.0 * min(th, 32)                       # reads/s, first guess for the sample size
        n = int(min(batch.n, max(2000, guess * 1.0)))
        t = run(n, th)
        if thread_points and t < 1.0 and n < batch.n:        # too short to mean much: once more on a 2 s sample
            n = int(min(batch.n, max(n, n / max(t, 1e-6) * 2.0)))
            t = run(n, th)
        rate[th] = n / t
    quota = cores
    try:                                                     # a container may own fewer CPUs than it sees (cgroup v2 quota)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(round(int(q) / int(per))))
    except Exception:
        pass
    best = max(rate, key=lambda k: rate[k])
    n1 = int(min(batch.n, max(20000, rate[best] * budget_s)))
    t1 = run(n1, best)
    rate[best] = max(rate[best], n1 / t1)
    return {"value": n1 / t1, "unit": "reads/s", "cores": min(best, quota), "kind": "port", "threads": best,
            "host_threads_visible": cores, "cpu_quota_cores": quota,
            "sample": f"first {n1} reads of the rank-0 batch, close+far end, OpenMP {best} threads (the best of "
                      f"{sorted(rate)}; the host shows {cores} threads, the container's CPU quota is {quota} cores), {t1:.1f} s",
            "reads_per_s_by_threads": {str(k): rate[k] for k in sorted(rate)}}


def default_workload(args):
    return args.read_len == 100 and args.max_range_index == 2 and args.workload == "sv10m"


def measured_traffic(args, reads_per_launch):
    """HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE with the gfx950 x2
    correction + WRITE_SIZE, MI355X_MICROARCH.md).  PMC counters cannot be read from inside this process,
    so the per-read figure measured with rocprofv3 on this same workload (scripts/profile_round.sh) is
    scaled by the reads of one launch; null for any other workload."""
    path = os.path.join(PROFILE_DIR, TRAFFIC_FILE)
    if not default_workload(args) or not os.path.exists(path):
        return None, None
    with open(path) as fh:
        t = json.load(fh)
    per_read = t["fetch_bytes_per_read"] + t["write_bytes_per_read_uncalibrated"]
    return per_read * reads_per_launch, (f"profiles/{PROFILE_ROUND}/{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                          "separate passes, per read x reads per launch)")


def issue_model(args, kernel_ms, reads_per_launch):
    """The kernel is instruction-issue bound, not HBM bound.  Instructions per read and lane
    occupancy from the committed PMC passes of this round (profiles/r04/pmc_sq.txt); issue costs from the microbenchmark
    (profiles/r03/ubench_issue_rates.txt, MI355X): a SIMD issues a fast-rate wave64 VALU op (plain logic / add / v_bitop3
    on VGPR or constant operands) every 2.0 cycles and a slow-rate one (an SGPR operand, funnel shifts, compares, selects,
    DPP, lane reads: 44 % of this kernel's VALU instructions statically) every 3.2 cycles, the CU's one scalar unit
    1.0-1.35 ops per cycle.  Both VALU bounds are given; the truth lies between them.  `ns_per_instruction` = SIMD time
    per read / (VALU + SALU instructions per read); `in_situ_ns_per_extra_instruction` = what 128 more instructions
    per seed-filter run were measured to cost inside this kernel (CALIB_FILE)."""
    path = os.path.join(PROFILE_DIR, PMC_FILE)
    if not default_workload(args) or not os.path.exists(path) or kernel_ms <= 0:
        return None
    c = {}
    with open(path) as fh:
        for line in fh:
            f = line.split()
            if len(f) >= 2 and f[0].startswith(("SQ_", "GRBM_")):
                c[f[0]] = float(f[1])
    if "SQ_INSTS_VALU" not in c:
        return None
    valu, salu = c["SQ_INSTS_VALU"], c.get("SQ_INSTS_SALU", 0.0)
    per_simd = reads_per_launch / 1024.0                       # 256 CUs x 4 SIMDs
    cyc = kernel_ms * 1e-3 * 2.4e9
    out = {"valu_insts_per_read": valu, "salu_insts_per_read": salu,
           "valu_busy_frac_at_2p0_cycles": valu * per_simd * 2.0 / cyc,
           "valu_busy_frac_at_3p2_cycles": valu * per_simd * 3.2 / cyc,
           "salu_busy_frac_at_1_per_cu_cycle": salu * reads_per_launch / 256.0 / cyc,
           "ns_per_instruction": kernel_ms * 1e6 / per_simd / (valu + salu),
           "in_situ_ns_per_extra_instruction": {"salu": 1.65, "valu_fast_rate": 0.79, "valu_slow_rate": 1.06},
           "source": f"profiles/{PROFILE_ROUND}/{PMC_FILE} (instruction counts) + {UBENCH_FILE} (issue rates) + {CALIB_FILE} (in-situ costs), "
                     "256 CUs x 4 SIMDs, 2.4 GHz nominal"}
    if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
        # active lanes per executed VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU), of 64
        out["valu_active_lanes_per_inst"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
        out["valu_lane_utilisation"] = out["valu_active_lanes_per_inst"] / 64.0
    if c.get("SQ_WAIT_INST_ANY") and c.get("SQ_WAVE_CYCLES"):
        out["wave_cycles_waiting_frac"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    return out


def pack_bytes(batch):
    """Bytes the pack moves for a batch: offsets + SoA fields + bases in; bit planes + the 128-byte record out."""
    lens = batch.lengths()
    ml = int(lens.max()) if batch.n else 1
    blocks = 1 if ml <= 64 else 2 if ml <= 128 else 3 if ml <= 192 else 4 if ml <= 256 else 8
    return float(lens.sum()) + batch.n * (8 + 11 + 64 * blocks + 128)


def pack_roofline(batch, pack_ms):
    """Second roofline entry: the pack kernel (a streaming transpose, HBM-bound by nature)."""
    nbytes = pack_bytes(batch)
    achieved = nbytes / (pack_ms * 1e-3) / 1e9 if pack_ms > 0 else 0.0
    return {"bound": "hbm", "kernel": "pg_pack_kernel", "kernel_ms": pack_ms, "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes, "traffic": None,
            "note": "as a launch of its own, measured outside the timed region; inside the step "
                    "and more the same code runs in pg_search_kernel, claim by claim"}


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: run N ranks under torch.distributed.run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def build_workload(args, rank, world, dev):
    """-> chroms [(name, padded bytes)], batch (this rank's reads), bd, bd_off, description, total reads."""
    import numpy as np
    from pindel_amd import binding, shard, synth
    bd = bd_off = None
    strong = args.scaling == "strong"
    read_seed = args.seed + 1 + (0 if strong else rank)
    if args.workload == "wgs-bins":
        # BASELINE configs[4]-shaped.  30x WGS: ~400 M one-end-anchored reads over 3.1 Gbp = 0.65 M per 5-Mbp bin; Pindel
        # works bin by bin (src/pindel.cpp:1816-1982).  One rank's share here: --reads reads (default 13 M = 20 bins'
        # worth) on the first chromosomes of the GRCh38-shaped reference, in coordinate order.
        if args.read_len == 100:
            args.read_len = 150
        if args.reads == 10_000_000:
            args.reads = 13_000_000
        n_bins = max(1, int(round(args.reads / 650_000)))
        lens = [max(400_000, int(L * args.genome_scale)) for L in synth.GRCH38_LENGTHS]
        chroms = synth.make_genome(lens, synth.GRCH38_NAMES, seed=args.seed, device=dev)
        # the reads of n_bins consecutive bins: the first n_bins x 5 Mbp of the genome, chromosome by chromosome
        parts, left = [], n_bins * 5_000_000
        for c, (_, s) in enumerate(chroms):
            if left <= 0:
                break
            span = min(left, len(s) - 2 * synth.SPACER)
            n_c = int(round(args.reads * span / (n_bins * 5_000_000)))
            if n_c > 0 and span > 100_000:
                parts.append(synth.make_reads(s[:span + 2 * synth.SPACER], n_c, seed=read_seed + 7 * c, chr_id=c, device=dev,
                                              read_len=args.read_len))
            left -= span
        import numpy as _np
        off = _np.concatenate([[0]] + [p_.seq_off[1:].astype(_np.uint64) + _np.uint64(b_) for p_, b_ in
                                        zip(parts, _np.cumsum([0] + [len(p_.seq) for p_ in parts[:-1]]))]).astype(_np.uint64)
        batch = type(parts[0])(seq=_np.concatenate([p_.seq for p_ in parts]), seq_off=off,
                               anchor_strand=_np.concatenate([p_.anchor_strand for p_ in parts]),
                               anchor_pos=_np.concatenate([p_.anchor_pos for p_ in parts]),
                               insert_size=_np.concatenate([p_.insert_size for p_ in parts]),
                               chr_id=_np.concatenate([p_.chr_id for p_ in parts]))
        args.reads = batch.n
        # coordinate order inside every chromosome (a sorted BAM)
        order = np.lexsort((batch.anchor_pos, batch.chr_id))
        L = args.read_len
        batch = type(batch)(seq=batch.seq.reshape(batch.n, L)[order].reshape(-1), seq_off=batch.seq_off,
                            anchor_strand=batch.anchor_strand[order], anchor_pos=batch.anchor_pos[order],
                            insert_size=batch.insert_size[order], chr_id=batch.chr_id[order])
        desc = (f"BASELINE configs[4]-shaped: {args.reads} x {args.read_len} bp reads per GPU in coordinate order on "
                f"the first {n_bins} bins of the GRCh38-shaped reference (24 chromosomes, {sum(lens)} bp), searched 5-Mbp bin by bin "
                f"(one launch per bin), all SV types")
    elif args.workload == "grch38-150":
        # BASELINE configs[3]-shaped: 24 chromosomes, 3.1 Gbp, 150-bp reads; one rank's share of 100 M reads
        if args.read_len == 100:
            args.read_len = 150
        if args.reads == 10_000_000:
            args.reads = 12_500_000
        scale = args.genome_scale
        lens = [max(400_000, int(L * scale)) for L in synth.GRCH38_LENGTHS]
        chroms = synth.make_genome(lens, synth.GRCH38_NAMES, seed=args.seed, device=dev)
        batch = synth.make_reads_genome(chroms, args.reads, seed=read_seed, device=dev, read_len=args.read_len)
        desc = (f"BASELINE configs[3]-shaped: {args.reads} x {args.read_len} bp reads per GPU on a GRCh38-shaped "
                f"reference (24 chromosomes, {sum(lens)} bp, i.i.d. + repeats/gaps per chromosome), all SV types")
    else:
        rf = 0.45 if args.workload == "repeat-rich" else 0.0
        ref = synth.make_reference(args.chr_len, seed=args.seed, device=dev, repeat_frac=rf)
        chroms = [("20", ref)]
        if args.workload == "colo-bd":
            # configs[1]-shaped (SURVEY.md 8d cfg 2): deletions only + BreakDancer hints.  The COLO-829 BAM is not
            # in the image, so the hints are synthetic: 0-2 windows of 600 bases per read within 20 kb
            # downstream/upstream of the anchor (where a deletion's far end lies).
            if args.reads == 10_000_000:
                args.reads = 1_000_000
            batch = synth.make_reads(ref, args.reads, read_len=args.read_len, seed=read_seed,
                                     device=dev, mix=(1.0, 0.0, 0.0, 0.0, 0.0))
            rng = np.random.default_rng(args.seed + 77 + (0 if strong else rank))
            k = rng.integers(0, 3, batch.n)
            bd_off = np.concatenate([[0], np.cumsum(k)]).astype(np.uint64)
            owner = np.repeat(np.arange(batch.n), k)
            sign = np.where(batch.anchor_strand[owner] == ord("+"), 1, -1)
            centre = batch.anchor_pos[owner].astype(np.int64) + 100000 + sign * rng.integers(300, 20000, len(owner))
            centre = np.clip(centre, 100400, len(ref) - 100400)
            bd = np.zeros(len(owner), dtype=binding.WINDOW_DTYPE)
            bd["chr_id"], bd["start"], bd["end"] = 0, centre - 300, centre + 300
            desc = (f"BASELINE configs[1]-shaped (deletions only, synthetic BreakDancer window hints): {args.reads} x "
                    f"{args.read_len} bp reads on a chr20-shaped reference ({args.chr_len} bp)")
        elif args.workload == "wgs-real":
            # the retry path's workload (round-5 verdict, item 2): 95 % of the reads are "no event" -- unmappable junk or plain
            # reference reads, few of which keep a close end (GetCloseEnd walks (R0,seq) (R0,RC) (R1,RC) (R1,seq)) -- the
            # rest split reads of every type: three reads in four end without a close end; 150 bp, in coordinate order like a sorted BAM
            if args.read_len == 100:
                args.read_len = 150
            batch = synth.make_reads(ref, args.reads, read_len=args.read_len, seed=read_seed, device=dev,
                                     mix=WGS_REAL_MIX)
            order = np.argsort(batch.anchor_pos, kind="stable")
            L = args.read_len
            batch = type(batch)(seq=batch.seq.reshape(batch.n, L)[order].reshape(-1), seq_off=batch.seq_off,
                                anchor_strand=batch.anchor_strand[order], anchor_pos=batch.anchor_pos[order],
                                insert_size=batch.insert_size[order], chr_id=batch.chr_id[order])
            desc = (f"WGS-like read mix (75 % of the reads without a close end: all four attempts), {args.reads} x {args.read_len} bp "
                    f"reads in coordinate order on a chr20-shaped reference ({args.chr_len} bp)")
        else:
            batch = synth.make_reads(ref, args.reads, read_len=args.read_len, seed=read_seed, device=dev)
            kind = "BASELINE configs[2]" if args.workload == "sv10m" else "repeat-rich variant of configs[2] (45 % repeats)"
            desc = (f"{kind}: synthetic {args.reads} x {args.read_len} bp one-end-anchored reads on a chr20-shaped "
                    f"reference ({args.chr_len} bp), all SV types (D/SI/TD/INV/none)")
    total = batch.n
    if strong and world > 1:
        lo, hi = shard.shard_bounds(batch.n, world)[rank]
        if bd is not None:
            b0, b1 = int(bd_off[lo]), int(bd_off[hi])
            bd, bd_off = bd[b0:b1], (bd_off[lo:hi + 1] - bd_off[lo]).astype(np.uint64)
        batch = batch.slice(lo, hi)
    return chroms, batch, bd, bd_off, desc, total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (weak) / in total (strong)")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--chr-len", type=int, default=CHR20_LEN)
    ap.add_argument("--max-range-index", type=int, default=2, help="Pindel -x")
    ap.add_argument("--seed", type=int, default=20260927)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive pg_search_batch sample")
    ap.add_argument("--no-standalone", action="store_true",
                    help="skip the launches that time the pack and the search as kernels of their own after the timed region "
                         "(config.pack_ms_standalone / search_ms_standalone / value_search_only): a traced run then holds the steps' launches only")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--genome-scale", type=float, default=1.0, help="grch38-150: scale every chromosome (tests)")
    ap.add_argument("--workload", choices=["sv10m", "colo-bd", "grch38-150", "repeat-rich", "wgs-bins", "wgs-real"], default="sv10m",
                    help="sv10m = BASELINE configs[2] (default); colo-bd = configs[1]-shaped; grch38-150 = configs[3]-shaped "
                         "(one rank's 12.5 M x 150 bp on a 3.1 Gbp 24-chromosome reference); repeat-rich = configs[2] on a "
                         "reference that is 45 % diverged repeat copies; wgs-bins = configs[4]-shaped: 150-bp reads in "
                         "coordinate order on the GRCh38-shaped reference, searched 5-Mbp bin by bin (one launch per bin, "
                         "0.65 M reads per bin = 30x WGS with 400 M one-end-anchored reads), as main()'s loop does; wgs-real = the "
                         "read mix INTEGRATION.md expects of a real WGS run: three reads in four find no close end and walk all "
                         "four attempts (150 bp, coordinate order, chr20-shaped reference)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_spawn(args)

    import torch
    from pindel_amd import binding, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if world > torch.cuda.device_count() and not os.environ.get("PG_BENCH_SHARE_GPU"):
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1 or os.environ.get("PG_BENCH_FORCE_DIST"):     # the env knob exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if os.environ.get("PG_BENCH_SHARE_GPU"):     # tests on a 1-GPU box: ranks share device 0, gloo instead of RCCL
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)

    params_kw = dict(max_range_index=args.max_range_index)
    # (set-up, outside the timed region -- its wall time per stage goes on the record: config.setup_seconds.  The first 8-GPU
    # run must not die here: eight ranks build their inputs side by side on one host)
    t_setup = [time.perf_counter()]
    chroms, batch, bd, bd_off, desc, total_reads = build_workload(args, rank, world, dev)
    t_setup.append(time.perf_counter())
    eng = binding.Engine(device=local_dev, **params_kw)
    eng.load_reference(chroms)
    t_setup.append(time.perf_counter())
    bins = None
    if args.workload == "wgs-bins":
        import numpy as np
        key = batch.chr_id.astype(np.int64) * 100_000 + batch.anchor_pos // 5_000_000
        cuts = np.concatenate([[0], np.nonzero(np.diff(key))[0] + 1, [batch.n]])
        bins = [eng.upload(batch.slice(int(a), int(b))) for a, b in zip(cuts[:-1], cuts[1:])]
    dbatch = eng.upload(batch)           # inputs resident in HBM before the timed region
    if bd is not None:
        eng.set_windows(dbatch, bd, bd_off)
    t_setup.append(time.perf_counter())

    # One step = what the device does for one batch whose raw inputs (ASCII bases, offsets, anchor fields) are resident in HBM:
    # the PACK (bit planes + packed records, the form the search reads -- part of every call of the ABI, so part of `value`
    # since round 6) and the SEARCH, through pg_device_batch_pack_search: ONE launch
    # (pg_search_kernel packs the reads of each claim itself just before it searches them); pg_pack_kernel + pg_search_kernel
    # only where the batch's plane layout is not its kernels' (64-bit candidate ids).  Synchronous; the HIP-event time covers everything the step put on the device.
    in_place = []                        # per launch of the timed steps: did the search kernel pack its own reads?

    def one_step():
        if bins is None:
            eng.pack_search_device(dbatch)    # synchronous: returns when the kernel finished
            in_place.append(eng.last_step_in_place())
            return eng.last_stats()[0]
        ms = 0.0
        for h in bins:                   # one step per 5-Mbp bin
            eng.pack_search_device(h)
            in_place.append(eng.last_step_in_place())
            ms += eng.last_stats()[0]
        return ms

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        kernel_ms.append(one_step())
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if os.environ.get("PG_BENCH_SHARE_GPU") else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- outside the timed region: accounting, result digests, the host-buffer seam, the CPU baseline
    # The pack (ASCII bases as src/reader.cpp:852-856 hands them over -> the bit planes + packed records the search reads) is
    # part of every timed step.  Bytes per read: len + 8 (offset) + 11 (strand, position, insert size, chromosome) in, 64 x
    # blocks (planes) + 128 (record + symbol programs) out (roofline.pack).
    # The two stages on their own, OUTSIDE the timed region (three launches each, HIP events): pg_pack_kernel as a launch of its
    # own and pg_search_kernel on records already packed -- the latter is what rounds 1-5 reported as `value`
    # (config.value_search_only), kept for comparison.
    pack_ms = search_only_ms = None
    if not args.no_standalone:
        units_of_step = [dbatch] if bins is None else bins          # (wgs-bins: a launch per bin, as in the timed steps)
        pack_ms = min(sum(eng.repack(h) for h in units_of_step) for _ in range(3))
        search_only_ms = []
        for _ in range(3):
            ms = 0.0
            for h in units_of_step:
                eng.search_device(h)
                ms += eng.last_stats()[0]
            search_only_ms.append(ms)
        search_only_ms = min(search_only_ms)
    if bins is not None or not args.no_standalone:
        eng.pack_search_device(dbatch)            # the step once more on the whole batch, for the accounting below
    alg_bytes = eng.algorithmic_bytes(dbatch)     # per launch
    n_cand = eng.candidates(dbatch)
    res = eng.download(dbatch)
    n_runs = int(res.close_off[-1]) + int(res.far_off[-1])
    n_close = int((res.close_off[1:] > res.close_off[:-1]).sum())
    n_far = int((res.far_off[1:] > res.far_off[:-1]).sum())
    digests = shard.read_digests(res)
    if dist is not None and args.scaling == "strong":
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(digests, gathered, dst=0)
        if rank == 0:
            import numpy as np
            digests = np.concatenate(gathered)      # rank order == input order
    units = total_reads if args.scaling == "strong" else world * args.reads

    if rank == 0:
        avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic(args, batch.n)
        host_path = None
        if world == 1 and not args.no_host_path:
            # the seam a Pindel maintainer calls: host buffers in, host CSR out (PCIe both ways) -- never `value`
            import ctypes as C
            nb = min(batch.n, 4_000_000)
            st, keep = binding._batch_struct(batch.slice(0, nb))
            L = binding.lib()
            best_t = None
            for _ in range(5):                        # the first calls pin the result buffers / grow the device arena
                h = C.c_void_p()
                th = time.perf_counter()
                rc = L.pg_search_batch(eng._h, C.byref(st), C.byref(h))
                dt = time.perf_counter() - th
                if rc:
                    raise SystemExit(f"pg_search_batch failed: {rc}")
                L.pg_result_free(h)
                best_t = dt if best_t is None else min(best_t, dt)
            host_path = nb / best_t
        out = {
            "metric": "one-end-anchored reads/sec through split-read search (close end + far end)",
            "value": units * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": desc + f", Pindel defaults -x {args.max_range_index} -a 1 -m 3 -u 0.02 -e 0.01 -E 0.95 -H 8",
                "reads_per_gpu": batch.n, "reads_total": units, "read_len": args.read_len, "insert_size": 500,
                "parallelism": f"reads sharded over {world} GPU(s) ({args.scaling} scaling), reference replicated, no collective",
                "launches_per_step": 1 if bins is None else len(bins),
                "reads_with_close_end": n_close, "reads_with_far_end": n_far, "runs_out": int(n_runs),
                "candidates_per_read": n_cand / max(batch.n, 1),    # seed-filter survivors that went through the full comparison
                "result_sha256": shard.digest_hex(digests),
                "host_path_reads_per_s": host_path,
                # `value` covers pack + search; the search alone (rounds 1-5's `value`: inputs already packed) for comparison
                "step": ("pack + search on raw inputs resident in HBM (pg_device_batch_pack_search): " +
                         ("ONE launch, pg_search_kernel packs the reads of each claim before it searches them" if in_place and all(in_place)
                          else "pg_pack_kernel + pg_search_kernel" if not any(in_place)
                          else f"{sum(in_place)} of {len(in_place)} launches pack in place, the others are pg_pack_kernel + pg_search_kernel")),
                "setup_seconds": {"synthetic_inputs": t_setup[1] - t_setup[0], "reference_to_hbm": t_setup[2] - t_setup[1],
                                  "reads_to_hbm": t_setup[3] - t_setup[2]},
                "device_ms_per_step": avg_ms,
                # outside the timed region, three launches each: the stages as launches of their own
                "pack_ms_standalone": pack_ms,
                "search_ms_standalone": search_only_ms,
                "value_search_only": batch.n * world / (search_only_ms * 1e-3) if search_only_ms else None,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                # what the memory system actually moved per second (counter traffic / kernel time) and how much of it the
                # algorithm did not ask for; null when the workload is not the one the committed counter passes ran
                "hbm_achieved_gbs": traffic / (avg_ms * 1e-3) / 1e9 if traffic and avg_ms > 0 else None,
                "hbm_achieved_frac": traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic and avg_ms > 0 else None,
                "traffic_over_algorithmic": traffic / alg_bytes if traffic and alg_bytes else None,
                # the step's launch also packs: its inputs read + planes and records written, which
                # `achieved` does not count, beside the counter traffic
                "traffic_over_algorithmic_incl_pack": (traffic / (alg_bytes + pack_bytes(batch))
                                                       if traffic and alg_bytes and in_place and all(in_place) else None),
                # (the step's launch: with the pack inside; `achieved` counts the SEARCH's
                # algorithmic bytes only -- the planes and records the pack writes are intermediates, see roofline.pack)
                "kernel": "pg_search_kernel", "kernel_ms": avg_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "actual_bound": "instruction issue (VALU + scalar), see DESIGN.md section 4",
                "issue": issue_model(args, avg_ms, batch.n),
                "pack": pack_roofline(batch, pack_ms) if pack_ms else None,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(chroms, batch, params_kw, bd=bd, bd_off=bd_off)
        print(json.dumps(out), flush=True)
    for h in bins or []:
        eng.free_device_batch(h)
    eng.free_device_batch(dbatch)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the hot path: the hierarchical linear map-joining solve (all transforms + joins of the reference's
binary tree, the region the reference itself times: LinearSFMImp.cpp:1929 -> 2068) on the NC3500-like Stereo set
(BASELINE.json configs[2]: 3499 local maps; synthetic stand-in, the real dataset is a Google-Drive link only).

    python bench.py --gpus N --steps K --warmup W

A "step" = ONE full join tree over ONE resident set of local maps, whatever N is ("scaling": "strong"): with N > 1 the
tree is sharded twice (linearsfm_amd/distributed.py): rank r joins block r of 2^k consecutive local maps on its own (the
independent joins of the lower levels), then the sub-tree roots are cut by feature label, slice g of every root goes to rank g
(all-to-all of packed device buffers), and ALL ranks run the top log2(N) levels on their slice of the features with RCCL
all-reduces of the sums that cross features (`--top merge`: pairwise merge rounds instead).  `value` is the wall time of
the whole tree.  Inputs are uploaded once and stay in HBM; no level writes its input, so every step reads
them in place.  Every timed step does the symbolic work of every join itself (pattern of S, ordering, symbolic
factorisation), as the reference's timed region does; the repeat runs that reuse it are reported as `resolve_ms`.
At N = 1 the line also carries the CPU baseline (the oracle on the host cores, three ways), the stand-alone streaming
rate of the CG's SpMV kernel and one files-to-files run of the command line.  Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
# fp64 matrix peak: the guide lists FP32 matrix = FP32 vector = 157.3 TFLOP/s (64 FLOP/clk/SIMD); v_mfma_f64_16x16x4_f64 runs at half
# that rate (32 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz) = 78.6 TFLOP/s, AMD's dense FP64 matrix figure for MI355X
F64_MFMA_PEAK_TFLOPS = 78.6
METRIC_NAMES = {"nc3500": "NC3500-like stereo", "rs468": "RS468-like monocular", "rs90": "RS90-like monocular", "aerial": "AP_Vaihingen-like aerial monocular",
                "synth16k": "synthetic 16k monocular", "synth64k": "synthetic 64k stereo",
                "spmv-stream": "CG SpMV stand-alone on a 1.4 GB Schur-like matrix"}


ROUND = "r06"  # the round whose profiles/ files the line may quote: counters of an older build say nothing about this one's kernels


def pmc_traffic(config):
    """HBM bytes per launch / per tree from THIS round's rocprofv3 PMC passes for THIS configuration
    (profiles/<ROUND>_pmc_traffic_summary_<config>.json, written by tools/refresh_profiles.py from separate FETCH_SIZE and
    WRITE_SIZE runs with the gfx950 corrections of MI355X_MICROARCH.md).  None when the round has no such file -- an older
    round's counters are never quoted (the kernels have changed since)."""
    path = os.path.join(ROOT, "profiles", f"{ROUND}_pmc_traffic_summary_{config}.json")
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path))
    except Exception:
        return None


def pmc_mfma_busy(config):
    """The matrix cores' busy share of the SIMD cycles per kernel, from THIS round's counter pass of this command
    (profiles/<ROUND>_pmc_mfma_busy_<config>.json: SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)); None when the round has none."""
    path = os.path.join(ROOT, "profiles", f"{ROUND}_pmc_mfma_busy_{config}.json")
    try:
        k = json.load(open(path))["kernels"]
        return {"source": os.path.relpath(path, ROOT), "share_of_simd_cycles": {n: v["mfma_busy_share_of_simd_cycles"] for n, v in k.items() if n.startswith("k_schur_panel")}}
    except Exception:
        return None


def rocprof_kernel_avg(config, prefixes):
    """Per-launch average (us) and total of the kernels whose names start with one of `prefixes`, from THIS round's committed
    `rocprofv3 --kernel-trace --stats` summary of this command (profiles/<ROUND>_bench_<config>_kernel_stats.csv): the line carries it
    so that its roofline fraction can be recomputed from profiles/ alone.  None when the round has no such file."""
    import csv
    import re
    path = os.path.join(ROOT, "profiles", f"{ROUND}_bench_{config}_kernel_stats.csv")
    if not os.path.exists(path):
        return None
    rows = []
    try:
        for r in csv.DictReader(open(path)):
            name = r.get("Name") or r.get("KernelName") or ""
            mt = re.search(r"lsfm::(k_\w+(?:<[^>]*>)?)", name)
            short = mt.group(1) if mt else name
            if any(short.startswith(p) for p in prefixes):
                rows.append({"kernel": short[:60], "calls": int(r["Calls"]), "total_us": float(r["TotalDurationNs"]) / 1e3,
                             "avg_us": float(r["AverageNs"]) / 1e3})
    except Exception:
        return None
    if not rows:
        return None
    trees = 7 if config == "synth16k" else 14  # tools/measure.sh: --steps 2 --warmup 1 (synth16k) / --steps 5 --warmup 2: first run + warm-ups + both modes
    total = sum(r["total_us"] for r in rows)
    return {"source": os.path.relpath(path, ROOT), "kernels": rows, "total_us": total, "calls": sum(r["calls"] for r in rows),
            "trees_profiled": trees, "ms_per_tree": total / trees / 1e3}


def twin_floor(config):
    """How far the oracle (fp64) and its long-double twin differ on THIS set under the same two error definitions (tools/oracle_twin_floor.py
    -> profiles/r*_oracle_twin_floor_<config>.json; a CPU run of minutes, made once): what the device's distance from the oracle is to be
    read against -- a true relative error of a few 1e-5 on near-zero angles is the reference's own arithmetic, not the device's."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_oracle_twin_floor_{config}.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        return {"pose_param_max_rel_err_oracle_vs_long_double_twin": d["pose_param_max_rel_err_oracle_vs_twin"],
                "pose_param_max_true_rel_err_oracle_vs_long_double_twin": d["pose_param_max_true_rel_err_oracle_vs_twin"],
                "source": os.path.relpath(files[-1], ROOT)}
    except Exception:
        return None


def spmv_stream(args, ctx, rank, world, torch, dist, synth_mod):
    """--config spmv-stream: the CG's SpMV kernel (K10a, k_spmv) alone on a Schur-like matrix too large for the caches
    (pose chain with 12 neighbours + 12 dense hub rows, 262 144 poses, 1.38 GB of upper blocks).  On the tree configurations S
    is <= ~20 MB and the launch is latency-bound, so this line is where the kernel's HBM streaming rate is measured.
    A step = one launch; every rank runs its own replica (no exchange: "replicas only"), value = sum of the ranks' rates."""
    m, band, hubs = args.maps or 262144, 12, 12
    rp32, colidx, val = synth_mod.schur_like_matrix(m, band, hubs, seed=0)
    x = np.random.default_rng(1).normal(size=6 * m)
    if args.warmup:
        ctx.spmv_bench(rp32, colidx, val, x, reps=args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # lsfm_spmv_bench uploads (untimed), then brackets `reps` launches with HIP events on the library's stream
    y, ms, by = ctx.spmv_bench(rp32, colidx, val, x, reps=args.steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gbs = by / (ms * 1e-3) / 1e9
    tm = torch.tensor([gbs, ms], dtype=torch.float64, device="cuda")
    mx = tm.clone()
    if world > 1:
        dist.all_reduce(tm, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    if rank == 0:
        line = {"metric": "CG-SpMV HBM GB/s (6x6-block symmetric SpMV of the Schur system, stand-alone)", "value": float(tm[0].item()),
                "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(mx[1].item()),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"spmv-stream: {m} poses, band {band} + {hubs} hub rows, {len(colidx)} upper 6x6 blocks "
                                       f"({len(colidx) * 288 / 1e6:.0f} MB)", "replicas": world,
                           "value_definition": "algorithmic bytes of one launch (nnzb*(288+4) + 4(m+1) + 96 m, SURVEY 8d) / average "
                                               "launch time (HIP events around the timed launches), summed over the replicas",
                           "host_wall_s_including_upload": wall},
                "roofline": {"bound": "hbm", "kernel": "k_spmv (K10a)", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": ms, "algorithmic_bytes_per_launch": by}}
        print(json.dumps(line))


def cpu_baseline(dicts, mono, sub=1024):
    """Oracle (plain-C port of the reference path, single thread like the reference): the whole workload once -- the baseline
    figure -- and, on a bounded sample (the first `sub` local maps: a complete sub-tree), the same port matching common
    features the way the reference does and on many host threads, each next to the plain port on that same sample.
    Checker/baseline only -- never part of the measured product path."""
    from oracle import pyoracle as po
    po.build()
    out, timing, rc = po.divide_conquer(dicts, mono, match_hash=True)
    sample = dicts[:min(sub, len(dicts))]
    _, t_sort, _ = po.divide_conquer(sample, mono, match_hash=True)
    # as the reference matches common features: std::find of every End label in Cur's labels, O(n1 n2) (Imp.cpp:2581-2599)
    _, t_find, _ = po.divide_conquer(sample, mono, match_hash=False)
    # the "fair multi-core" figure: the independent joins of every level on many host threads (same result)
    threads = max(1, min(64, (os.cpu_count() or 1)))
    _, t_mt, _ = po.divide_conquer(sample, mono, match_hash=True, threads=threads)
    return out, timing, rc, dict(maps=len(sample), sort=t_sort, find=t_find, mt=t_mt[0], threads=threads)


def e2e_cli(maps, mono):
    """Files -> files through the reference's command line (linearsfm_amd/LinearSFM: the drop-in for the reference's console
    program): the set is written in the reference's localmap_k.txt format (setup, not timed), then ONE process reads it, uploads,
    joins, downloads and writes the pose / feature files.  Returns wall seconds of that process and its own phase times."""
    import re
    import shutil
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from linearsfm_amd import api
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="lsfm_e2e_", dir=base)
    try:
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max(1, min(32, os.cpu_count() or 1))) as ex:  # the C writer releases the GIL
            list(ex.map(lambda km: api.write_localmap(os.path.join(d, f"localmap_{km[0] + 1}.txt"), km[1].__dict__, mono), enumerate(maps)))
        write_s = time.perf_counter() - t0
        nbytes = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
        exe = os.path.join(ROOT, "linearsfm_amd", "LinearSFM")
        cmd = [exe, "-path", d, "-num", str(len(maps)), "-type", "Monocular" if mono else "Stereo", "-p", os.path.join(d, "Pose.txt"),
               "-f", os.path.join(d, "Feature.txt"), "-stats", "1"]
        t0 = time.perf_counter()
        p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        wall = time.perf_counter() - t0
        ph = re.search(r"lsfm_e2e: read ([\d.]+) s, context ([\d.]+) s, upload ([\d.]+) s, join tree ([\d.]+) s, download ([\d.]+) s, write ([\d.]+) s", p.stderr)
        ok = p.returncode == 0 and os.path.getsize(os.path.join(d, "Pose.txt")) > 0
        return {"e2e_cli_s": wall, "rc": p.returncode, "ok": ok, "input_MB": nbytes / 1e6, "setup_write_inputs_s": write_s,
                "phases_s": dict(zip(("read", "context", "upload", "join_tree", "download", "write"), map(float, ph.groups()))) if ph else None,
                "command": "linearsfm_amd/LinearSFM -path <dir> -num %d -type %s -p Pose.txt -f Feature.txt" % (len(maps), "Monocular" if mono else "Stereo")}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="nc3500", choices=sorted(METRIC_NAMES),
                    help="synthetic stand-in of a BASELINE.json configuration (linearsfm_amd/synth.py CONFIGS); the headline "
                         "metric is quoted on nc3500, the others are recorded in BASELINE.md")
    ap.add_argument("--maps", type=int, default=0, help="local maps (0 = the configuration's own count)")
    ap.add_argument("--new-per-frame", type=int, default=0)
    ap.add_argument("--vis", type=int, default=0)
    ap.add_argument("--cpu-baseline", type=int, default=1, help="1: time the oracle on the SAME set on the host cores (N=1 only); 0: skip")
    ap.add_argument("--cpu-max-maps", type=int, default=4096, help="sets larger than this time the oracle on their first maps only")
    ap.add_argument("--tol", type=float, default=1e-12, help="relative residual at which the refinement of a system stops (library default)")
    ap.add_argument("--mixed", action="store_true", help="Cholesky preconditioner kept and applied in fp32, fp64 residual correction (BASELINE configs[4])")
    ap.add_argument("--plans", action="store_true", help="time the repeat runs of the resident tree (structure analysed once) instead of "
                                                         "runs that analyse every join like the reference does; the default reports both")
    ap.add_argument("--no-plans", action="store_true", help="(the default since round 3; kept for old command lines)")
    ap.add_argument("--top", default="shard", choices=("shard", "merge"),
                    help="N > 1: how the levels above the ranks' blocks run -- shard: feature-sharded on ALL ranks, RCCL all-reduces of the "
                         "transform's pose rows, the camera system and the pose solution per level; merge: pairwise merge rounds of packed "
                         "sub-tree roots on half of the remaining ranks each")
    ap.add_argument("--extras", type=int, default=1, help="1: also the stand-alone SpMV streaming leg and the files-to-files CLI run "
                                                          "(N=1, default configuration sizes only); 0: skip")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process starts the N ranks itself (torch.distributed.run as a CHILD,
        # before anything here has touched the GPU -- never an exec from a process that has), relays what they print (rank 0's
        # JSON line) and ends with their status
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # the line's n_gpus must be what was asked for: a launcher that started another number of ranks is an error, not a silent 1-GPU run
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch one rank per GPU, or run `python bench.py --gpus N` alone: it starts them)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("LSFM_BENCH_ONE_GPU"):
            # development aid: all ranks on cuda:0 with gloo (two ranks of one RCCL communicator cannot share a device); checks the
            # multi-rank logic of this script on a one-GPU box -- the numbers it prints are not a scaling measurement
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0

    from linearsfm_amd import api, synth
    from linearsfm_amd.distributed import ShardedTree, shard_bounds

    if args.config == "spmv-stream":
        ctx = api.Context(local_rank)
        spmv_stream(args, ctx, rank, world, torch, dist, synth)
        ctx.close()
        if world > 1:
            dist.destroy_process_group()
        sys.exit(0)

    typ, cN, cnpf, cvis, cpath = synth.CONFIGS[args.config]
    mono = typ == "Monocular"
    n_maps = args.maps or cN
    npf = args.new_per_frame or cnpf
    vis = args.vis or cvis
    # every rank generates and uploads ITS block of the one set (same seed everywhere: a slice equals that part of the whole)
    _, bounds = shard_bounds(n_maps, world)
    lo, hi = bounds[rank]
    _, block = synth.make_config(args.config, n_maps, seed=0, new_per_frame=npf, vis=vis, only=(lo, hi))
    ctx = api.Context(local_rank)
    ctx.set_pcg(args.tol, 0)
    ctx.set_precision(args.mixed)
    t0 = time.perf_counter()
    tree = ShardedTree(ctx, block, lo, n_maps, mono, top=args.top)   # PCIe copy, outside the timed region: inputs are resident from here on
    upload_ms = 1e3 * (time.perf_counter() - t0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step_wall = []  # wall ms of every tree of the last timed() call: how far single steps scatter around `value` (their mean)

    def timed(steps):
        """`steps` whole trees between two barrier + synchronize points: (seconds, summed stats, last stats, worst rc)"""
        barrier()
        t0 = time.perf_counter()
        acc, last, worst = {}, None, 0
        del step_wall[:]
        for _ in range(steps):
            t1 = time.perf_counter()
            last, rc = tree.run()
            step_wall.append(1e3 * (time.perf_counter() - t1))  # (a run returns when its tree is done: no extra synchronisation)
            worst = max(worst, rc)
            for k, v in (last or {}).items():
                if isinstance(v, (int, float)):
                    acc[k] = acc.get(k, 0) + v
        barrier()
        return time.perf_counter() - t0, acc, last, worst

    # ---- the timed region.  A step = one whole join tree that, like the reference's timed region (Imp.cpp:1929 -> 2068), also
    # does the symbolic work of every join (pattern of S, ordering, symbolic factorisation: the reference's mask -> CRS and
    # cholmod_analyze_p): lsfm_tree_set_plans(tree, 0).  The repeat runs of the resident tree that reuse that analysis are
    # timed separately below (`resolve_ms`). ----
    analysing = not args.plans
    tree.set_plans(not analysing)
    first_s, _, _, _ = timed(1)     # the very first run (code objects, arenas, rocPRIM temporaries come up here)
    for _ in range(max(0, args.warmup - 1)):
        tree.run()
    elapsed, acc, stats, worst = timed(args.steps)
    sw = sorted(step_wall)
    step_spread = {"min": sw[0], "median": sw[len(sw) // 2], "max": sw[-1]} if sw else None
    rdev = "cpu" if os.environ.get("LSFM_BENCH_ONE_GPU") else "cuda"
    tm = torch.tensor([elapsed, first_s], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    elapsed, first_ms = float(tm[0].item()), 1e3 * float(tm[1].item())
    ms_per_step = 1e3 * elapsed / args.steps
    # ---- the other mode, same steps: plans on (one analysing run leaves them, then the repeats) / off ----
    tree.set_plans(analysing)
    tree.run()
    tree.run()
    other_s, acc2, _, _ = timed(args.steps)
    tm = torch.tensor([other_s], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    other_ms = 1e3 * float(tm[0].item()) / args.steps
    analyse_ms, resolve_ms = (ms_per_step, other_ms) if analysing else (other_ms, ms_per_step)
    # per-rank busy time of the timed steps (where a multi-GPU run loses its efficiency): device time of the trees the rank ran
    busy = torch.zeros(world, dtype=torch.float64, device=rdev)
    busy[rank] = acc.get("t_total_ms", 0.0) / args.steps
    # ... and the wall clock of its three phases (own block, exchange of the slices, feature-sharded top levels), every rank's
    phases = torch.zeros(world, 3, dtype=torch.float64, device=rdev)
    for k, name in enumerate(("phase_block_ms", "phase_exchange_ms", "phase_top_ms")):
        phases[rank, k] = acc.get(name, 0.0) / args.steps
    if world > 1:
        dist.all_reduce(busy, op=dist.ReduceOp.SUM)
        dist.all_reduce(phases, op=dist.ReduceOp.SUM)

    # (feature-sharded top: every rank holds a slice of the final map's features -- the download is collective)
    out = tree.download(full=False) if (world > 1 and args.top == "shard") else (tree.download() if rank == 0 else None)
    if rank == 0:
        # Per-kernel live measurements (HIP events on the library's stream around the launches, accumulated over the timed
        # steps; at N > 1: of the trees rank 0 ran, i.e. its block and the merges it took part in).  The roofline object
        # describes whichever of the two instrumented kernels took the most device time.
        kern = {}
        for key, name in (("schur", "k_schur_panel (K9: Schur assembly S -= W V^-1 W^T, E -= W V^-1 eb; fp64 MFMA panels)"),
                          ("trf", "k_tr_entries (K3/K4: information transform I' = J^T I J, one lane per W block)"),
                          ("spmv", "k_spmv (K10a: 6x6-block symmetric SpMV of the CG)")):
            n = max(1, acc.get(f"{key}_launches", 0))
            kern[key] = dict(name=name, total_ms=acc.get(f"{key}_ms", 0.0) / args.steps, launches_per_step=acc.get(f"{key}_launches", 0) / args.steps,
                             avg_ms=acc.get(f"{key}_ms", 0.0) / n, avg_bytes=acc.get(f"{key}_bytes", 0.0) / n)
            kern[key]["gbs"] = kern[key]["avg_bytes"] / (kern[key]["avg_ms"] * 1e-3) / 1e9 if kern[key]["avg_ms"] > 0 else 0.0
        kern["schur"]["avg_flops"] = acc.get("schur_flops", 0.0) / max(1, acc.get("schur_launches", 0))
        kern["schur"]["tflops"] = kern["schur"]["avg_flops"] / (kern["schur"]["avg_ms"] * 1e-3) / 1e12 if kern["schur"]["avg_ms"] > 0 else 0.0
        # (K9's event bracket and k_tr_entries' are within a few per cent of each other since round 5 -- 5.9 vs 5.9-6.0 ms per tree; by the
        # rocprof sum of its variants, which run beside each other, K9 is still the larger: 8.3 vs 6.0.  The object stays with K9 unless the
        # transform's kernel leads by more than a tenth, so that the line does not flip from run to run; both are under "kernels".)
        dom = "trf" if kern["trf"]["total_ms"] > 1.1 * kern["schur"]["total_ms"] else "schur"
        pmc = pmc_traffic(args.config) if (world == 1 and not args.maps) else None
        traffic = None
        traffic_note = None if pmc else f"no {ROUND} PMC passes for this configuration under profiles/ (counters of an older round are not quoted)"
        whole = None
        if pmc:
            traffic = pmc.get("per_launch", {}).get({"schur": "k_schur_panel", "trf": "k_tr_entries"}[dom])
            if pmc.get("bytes_per_tree"):
                whole = {"hbm_bytes_per_step": pmc["bytes_per_tree"], "GBps": pmc["bytes_per_tree"] / (ms_per_step * 1e-3) / 1e9,
                         "frac_of_hbm_peak": pmc["bytes_per_tree"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, "source": pmc.get("source")}
        line = {
            "metric": "hierarchical linear map-joining solve wall-clock, %s (all transforms + joins)" % METRIC_NAMES[args.config],
            "value": ms_per_step,
            "unit": "ms",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "step_ms_spread": step_spread,
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if not args.mixed else "f64 (S, right-hand side, iterate, residual) + f32 (Cholesky preconditioner)",
            "data": "synthetic",
            "config": {"workload": f"{args.config} stand-in ({typ}): {n_maps} local maps, {npf} new features/frame visible in {vis} "
                                   f"frames, camera path {cpath}, {out['m']} poses / {out['n']} features in the final map",
                       "maps": n_maps, "sharding": (f"{world} block(s) of {bounds[0][1] - bounds[0][0]} local maps, " +
                                    (f"{max(0, world.bit_length() - 1)} merge round(s)" if (world == 1 or args.top == "merge") else
                                     f"then the top {max(0, (world - 1).bit_length())} level(s) feature-sharded over all {world} ranks "
                                     "(features by label mod N; RCCL all-reduce of the transform's pose rows, of S and E, and of the pose solution, per level; "
                                     "the camera systems factored by block ownership with one int64 all-reduce of the inter-block separators)")),
                       "pcg_rel_tol": args.tol, "plans": not analysing,
                       "value_definition": "wall ms of ONE whole join tree over all GPUs (barrier + synchronize on both sides, max over ranks), inputs "
                                           "resident in HBM; " + ("every step analyses every join (pattern of S, ordering, symbolic factorisation) "
                                                                  "like the reference's timed region does" if analysing else
                                                                  "repeat runs of the resident tree: structure analysed once, before the timed steps")},
            "analysing_run_ms": analyse_ms,
            "resolve_ms": resolve_ms,
            "resolve_note": "resolve_ms: the same tree joined again with NEW work only (lsfm_tree_set_plans(tree, 1): container sizes, pattern of S, "
                            "ordering and symbolic factorisation of every level kept from an earlier run, a level enqueued without host round trips); "
                            "analysing_run_ms: every run does that analysis itself, as every run of the reference does (cholmod_analyze_p in every "
                            "join).  `value` is " + ("analysing_run_ms" if analysing else "resolve_ms") + ".",
            "first_run_ms": first_ms,
            "upload_ms": upload_ms,
            "upload_c_abi_ms": (stats or {}).get("upload_ms"),
            "upload_note": "upload_ms: building the lsfm_map views of this rank's block in Python + lsfm_tree_upload (pageable host memory -> "
                           "pinned ring -> HBM), before the timed region; upload_c_abi_ms: lsfm_tree_upload alone (lsfm_stats.upload_ms)",
            "per_rank_device_ms": [float(v) for v in busy.tolist()],
            "per_rank_phases_ms": ({"block": [float(v) for v in phases[:, 0].tolist()], "exchange": [float(v) for v in phases[:, 1].tolist()],
                                    "top": [float(v) for v in phases[:, 2].tolist()],
                                    "note": "host wall clock per step on every rank: its own block (+ cutting the root into slices), the all-to-all of "
                                            "the slices (+ unpacking), the feature-sharded top levels"} if "phase_top_ms" in acc else None),
            "distributed_solve": ({"levels_per_step": acc.get("dist_solves", 0) / args.steps,
                                   "replicated_share_of_factor_work": (acc.get("dist_work_shared", 0.0) / acc["dist_work_total"]) if acc.get("dist_work_total") else None,
                                   "amdahl_bound_of_the_factorisation": (1.0 / ((acc["dist_work_shared"] / acc["dist_work_total"]) +
                                                                                (1.0 - acc["dist_work_shared"] / acc["dist_work_total"]) / world))
                                   if acc.get("dist_work_total") else None,
                                   "note": "camera systems of the feature-sharded levels factored by block ownership (lsfm_tree_set_comm_blocks): rank r the columns "
                                           "interior to block r, one exact int64 all-reduce of the inter-block separators' accumulators, the separators by every rank; "
                                           "share = block products (6x6x6) of the shared columns / all, over rank 0's top-tree levels"} if world > 1 else None),
            "device_breakdown_ms": {k: acc.get(k, 0.0) / args.steps for k in
                                    ("t_total_ms", "t_transform_ms", "t_join_ms", "t_schur_ms", "t_pcg_ms", "t_backsub_ms", "t_small_ms")},
            "dense_path": {"levels_per_step": acc.get("small_levels", 0) / args.steps, "ms_per_step": acc.get("t_small_ms", 0.0) / args.steps,
                           "note": "tree levels whose camera systems (at most 5 poses) are assembled, factored and solved by one launch, one "
                                   "work-group per join (lsfm_small.hip, lsfm_set_small_solve); K9's roofline counts the levels K9 runs on"},
            "pcg_iterations_per_step": acc.get("pcg_iterations", 0) / args.steps,
            "max_rel_residual": (stats or {}).get("max_rel_residual"),
            "not_converged": (stats or {}).get("not_converged"),
            "roofline": ({"bound": "hbm", "kernel": kern[dom]["name"], "achieved": kern[dom]["gbs"],
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kern[dom]["gbs"] / HBM_PEAK_GBS, "traffic": traffic,
                          "avg_launch_ms": kern[dom]["avg_ms"], "algorithmic_bytes_per_launch": kern[dom]["avg_bytes"],
                          "launches_per_step": kern[dom]["launches_per_step"], "traffic_note": traffic_note,
                          "rocprof": rocprof_kernel_avg(args.config, ("k_tr_entries",)),
                          "note": "one launch per tree level (12 levels + final re-anchoring); average over all of them, small "
                                  "low-level launches included; algorithmic bytes = every input and output moved once "
                                  "(DESIGN.md); traffic = HBM bytes per launch from the rocprofv3 PMC passes under profiles/ for this "
                                  "configuration, null when none is kept"} if dom == "trf" else
                         {"bound": "mfma", "kernel": kern[dom]["name"], "achieved": kern[dom]["tflops"], "peak": F64_MFMA_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "frac": kern[dom]["tflops"] / F64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                          "avg_launch_ms": kern[dom]["avg_ms"], "algorithmic_flops_per_launch": kern[dom]["avg_flops"],
                          "hbm_view": {"algorithmic_bytes_per_launch": kern[dom]["avg_bytes"], "achieved_GBps": kern[dom]["gbs"],
                                       "frac_of_hbm_peak": kern[dom]["gbs"] / HBM_PEAK_GBS},
                          "launches_per_step": kern[dom]["launches_per_step"], "traffic_note": traffic_note,
                          "rocprof": rocprof_kernel_avg(args.config, ("k_schur_panel", "k_schur_w", "k_schur_slots", "k_schur_lists")),
                          "mfma_busy": pmc_mfma_busy(args.config),
                          "rocprof_note": "per-variant averages of the committed rocprofv3 --kernel-trace --stats summary of this command; the variants of a "
                                          "level run beside each other on two streams, so their SUM (total_us / trees profiled) is above the event "
                                          "bracket avg_launch_ms x launches_per_step that `achieved` is computed from",
                          "note": "K9 is the one real contraction of the path: a tile's contribution to S is P P^T on v_mfma_f64_16x16x4_f64. "
                                  "ALGORITHMIC flops per feature with k W blocks: k (108 + 36) + k (k + 1) / 2 * 216 (Imp.cpp:2260-2328) -- "
                                  "the zero blocks the dense panel also multiplies are not counted; ~17 flop per HBM byte at the top "
                                  "levels, where the fp64 matrix rate binds, latency at the low ones; one launch per tree level, average "
                                  "over all of them; traffic = HBM bytes per launch from the rocprofv3 PMC passes under profiles/"}),
            "whole_step_hbm": whole,
            "cg_spmv": {"algorithmic_GBps": kern["spmv"]["gbs"], "frac_of_hbm_peak": kern["spmv"]["gbs"] / HBM_PEAK_GBS,
                        "avg_launch_ms": kern["spmv"]["avg_ms"], "matrix_MB_top_level": (stats or {}).get("spmv_nnzb_upper_last", 0) * 288 / 1e6,
                        "note": "cache-resident on this configuration: the Schur matrix of a level is <= ~20 MB (L2 / Infinity Cache), the "
                                "launch is latency-bound and -- with the exact factor as preconditioner -- runs ~5 times per level; the "
                                "kernel's HBM-streaming rate is the cg_spmv_stream object of this line"},
            "kernels": {k: {"avg_launch_ms": v["avg_ms"], "algorithmic_GBps": v["gbs"], "frac_of_hbm_peak": v["gbs"] / HBM_PEAK_GBS,
                            "ms_per_step": v["total_ms"], "launches_per_step": v["launches_per_step"],
                            **({"algorithmic_TFLOPs": v["tflops"], "frac_of_f64_mfma_peak": v["tflops"] / F64_MFMA_PEAK_TFLOPS} if k == "schur" else {})}
                        for k, v in kern.items()},
        }
        extras = args.extras and world == 1
        if extras:
            # the CG's SpMV kernel where it streams: a Schur-like matrix far beyond the caches (same kernel, same entry point as
            # `--config spmv-stream`, a little shorter: 196 608 poses, 1.0 GB of upper blocks)
            m_s, band, hubs = 196608, 12, 12
            rp32, colidx, val = synth.schur_like_matrix(m_s, band, hubs, seed=0)
            xs = np.random.default_rng(1).normal(size=6 * m_s)
            ctx.spmv_bench(rp32, colidx, val, xs, reps=3)
            _, sms, sby = ctx.spmv_bench(rp32, colidx, val, xs, reps=20)
            line["cg_spmv_stream"] = {"GBps": sby / (sms * 1e-3) / 1e9, "frac_of_hbm_peak": sby / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "avg_launch_ms": sms, "algorithmic_bytes_per_launch": sby, "launches": 20,
                                      "matrix": f"{m_s} poses, band {band} + {hubs} hub rows, {len(colidx)} upper 6x6 blocks ({len(colidx) * 288 / 1e6:.0f} MB)",
                                      "note": "k_spmv (K10a) stand-alone through lsfm_spmv_bench, HIP events around the timed launches; "
                                              "algorithmic bytes nnzb*(288+4) + 4(m+1) + 96 m (SURVEY 8d)"}
            del rp32, colidx, val
        if args.cpu_baseline and world == 1:  # the CPU leg runs at N=1 only, on the same set
            from oracle import pyoracle as po
            S = min(n_maps, args.cpu_max_maps)
            dicts = [po.localmap_to_dict(m) for m in block[:S]]
            o_out, timing, orc, smp = cpu_baseline(dicts, mono)
            if S == n_maps:
                g_out, g_analyse, g_resolve = out, analyse_ms, resolve_ms
            else:  # same prefix on the device, for a like-for-like ratio and a parity check of this very run
                tp = ctx.tree_upload(dicts, mono)
                ctx.tree_set_plans(tp, False)
                ctx.tree_run(tp)
                st2, _ = ctx.tree_run(tp)
                g_analyse = st2["t_total_ms"]
                g_out = ctx.tree_download(tp)
                ctx.tree_set_plans(tp, True)
                ctx.tree_run(tp)
                st3, _ = ctx.tree_run(tp)
                g_resolve = st3["t_total_ms"]
                ctx.tree_free(tp)
            mask = o_out["stno"] <= 0
            perr = float(np.max(np.abs(g_out["stVal"][mask] - o_out["stVal"][mask]) / np.maximum(1.0, np.abs(o_out["stVal"][mask]))))
            ferr = float(np.max(np.abs(g_out["stVal"][~mask] - o_out["stVal"][~mask]) / np.maximum(1.0, np.abs(o_out["stVal"][~mask]))))
            # ... and without the unit floor: |a - b| / |b| per pose scalar (scalars below 1e-3 of the largest of their kind -- translations,
            # angles -- held against that 1e-3: the relative error of a number that happens to be ~0 says nothing)
            ga, oa = g_out["stVal"][mask].reshape(-1, 6), o_out["stVal"][mask].reshape(-1, 6)
            perr_true = 0.0
            for cols in (slice(0, 3), slice(3, 6)):
                fl = 1e-3 * max(1e-300, float(np.max(np.abs(oa[:, cols]))))
                perr_true = max(perr_true, float(np.max(np.abs(ga[:, cols] - oa[:, cols]) / np.maximum(fl, np.abs(oa[:, cols])))))
            line["cpu_baseline"] = {"value": 1e3 * timing[0], "unit": "ms", "cores": 1, "kind": "port",
                                    "sample": (f"the whole set: all {n_maps} local maps" if S == n_maps else f"first {S} of the {n_maps} local maps")
                                              + f" (same generator/seed), whole join tree, oracle/lsfm_oracle.c single thread, sort-based "
                                                f"feature matching; host has {os.cpu_count()} cores",
                                    "oracle_breakdown_ms": {"transform": 1e3 * timing[1], "join_assembly": 1e3 * timing[2],
                                                            "schur_cholesky_backsub": 1e3 * timing[3]},
                                    "sample_legs": {
                                        "sample": f"first {smp['maps']} local maps of the set (a complete sub-tree), same port",
                                        "sort_matching_ms": 1e3 * smp["sort"][0],
                                        "as_reference_match_ms": 1e3 * smp["find"][0],
                                        "as_reference_match_join_assembly_ms": 1e3 * smp["find"][2],
                                        "multicore_ms": 1e3 * smp["mt"], "multicore_threads": smp["threads"],
                                        "note": "as_reference_match: common features matched the way the reference does -- std::find of every End "
                                                "label over Cur's labels, O(n1 n2) (Imp.cpp:2581-2599); `value` above uses a sort instead, so that "
                                                "the device is not credited for an algorithmic fix (on the whole NC3500-like set the std::find port "
                                                "took 48.8-51.5 s against 28.0-28.8 s: "
                                                "profiles/r03_bench_default_wholeset_cpu_legs.json).  multicore: the independent joins of a level on OpenMP threads; the top levels hold one "
                                                "join each, so this saturates at a few x.  These legs run on a sample so that the CPU part of the "
                                                "default run stays bounded"},
                                    "gpu_same_sample_ms": g_analyse,
                                    "gpu_same_sample_resolve_ms": g_resolve,
                                    "gpu_same_sample_note": "gpu_same_sample_ms: device runs that analyse every join, like the CPU figure does -- the like-for-like "
                                                            "pair; gpu_same_sample_resolve_ms: repeat runs of the resident tree",
                                    "pose_param_max_rel_err_vs_oracle": perr,
                                    "pose_param_max_true_rel_err_vs_oracle": perr_true,
                                    "reference_arithmetic_floor": twin_floor(args.config),
                                    "pose_param_err_definition": "max_rel_err: |a - b| / max(1, |b|) (translations of a monocular set are scale-normalised to O(1), "
                                                                 "angles are <= 0.3 rad: an absolute error below 1); max_true_rel_err: |a - b| / max(|b|, 1e-3 x the "
                                                                 "largest scalar of its kind)",
                                    "feature_param_max_rel_err_vs_oracle": ferr,
                                    "parity_tolerance": 1e-6,
                                    "parity_note": "fixed tolerance; how far two fp64 evaluations of the reference path differ on this set, and both "
                                                   "against the long-double evaluation of the solves: profiles/r02_full_parity_*.json (tools/full_parity.py)"}
        if extras and not args.maps:
            tree.close()
            tree = None
            line["e2e_cli"] = e2e_cli(block, mono)
        print(json.dumps(line))
    if tree is not None:
        tree.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    sys.exit(0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its 1-GPU configuration (configs[1]): Leduc Hold'em external-sampling
MCCFR with the full regret/strategy tables resident in HBM, infoset-updates/sec.

One "step" = Solver::step (crates/mccfr/src/solver/solver.rs:96-105): a batch of `--batch` sampled trees per GPU
traversed against the current profile, then every Decisions applied in tree-id order, then epoch += 1.
`value` = infoset-updates (the reference's `infos` counter, solver.rs:273) processed by ALL ranks / wall time,
inputs resident in HBM before the timed region.  N > 1 (launched by torch.distributed.run, one rank per GPU):
trees are sharded by rank and the per-cell composed maps are exchanged with one RCCL all-gather per step
(weak scaling: per-GPU batch fixed).

The same JSON line carries `roofline` (dominant kernel vs its roofline, algorithmic bytes from SURVEY §8d),
`cpu_baseline` (the CPU oracle timed on this host, rank 0, N=1 only) and, LAST, `kmeans`: the second hot path at
BASELINE configs[2]'s full size in the reference's own arithmetic (glibc expf / logf, the reference's k-means++ draw — the pass whose
buckets are the reference's bit for bit), with the f32 contract arithmetic on the same draw as `kmeans.contract_arithmetic` and how
far apart the two partitions are.  The line printed by default is a compact view (compact_line) that fits the driver's stdout window;
the whole detail object is written to gpurun_out/bench_detail.json and is the line itself under --verbose.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("RP_ORACLE_NATIVE", "1")  # the cpu_baseline legs time the oracle built -O3 -march=native on this machine

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1 << 23,
                    help="trees per step (Solver::batch_size): per GPU under weak scaling, in total under strong scaling")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="strong (default for N > 1): --batch trees per step in TOTAL, split across the GPUs (north_star's "
                         "strong-scaling figure); weak (N = 1, or on request): --batch trees per step on EVERY GPU.  With "
                         "N > 1 the line carries the other mode's measurement as well (`other_scaling`)")
    ap.add_argument("--workload", default="leduc", choices=["leduc", "nlhe-synth", "nlhe"],
                    help="leduc: BASELINE configs[1] (default, the quoted metric); nlhe: the blueprint trainer's own step — "
                         "external-sampling MCCFR over heads-up NLHE generated on the device (rp_nlhe_*, BASELINE configs[3] on "
                         "one GPU); nlhe-synth: synthetic NLHE-scale infoset batches through the sparse profile (SURVEY §8d config 4)")
    ap.add_argument("--nlhe-batch", type=int, default=262144,
                    help="nlhe: trees per step per GPU (the reference's batch_size is 128: timed beside it; up to 2 048 trees take the "
                         "one-tree-per-workgroup kernel, larger batches the level-synchronous ones: 37 GB of node arrays at the default)")
    ap.add_argument("--nlhe-cap", type=int, default=27, help="nlhe: log2 of the infoset table's rows")
    ap.add_argument("--rows", type=int, default=1 << 27, help="nlhe-synth: table rows (infoset slots)")
    ap.add_argument("--decisions", type=int, default=128 * 1500, help="nlhe-synth: Decisions per step per GPU")
    ap.add_argument("--game", default="leduc", choices=["leduc", "kuhn", "rps", "leduc_wide"])
    ap.add_argument("--regret", default="floored")
    ap.add_argument("--weight", default="linear")
    ap.add_argument("--sampling", default="external", choices=["external", "prunable", "pluribus"],
                    help="the SamplingScheme at walker nodes; nlhe: pluribus = the Flagship type (nlhe/src/lib.rs:86-90)")
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--update", default="composed", choices=["composed", "ordered"],
                    help="composed: blocked per-cell map composition (tables within 1e-4/step of the reference's "
                         "application order, bit-exact vs the oracle's model of it; the only mode that shards); "
                         "ordered: the reference's sequential tree-id order exactly (N=1 only)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-oracle baseline sample length (0 = skip)")
    ap.add_argument("--no-kmeans", action="store_true", help="skip the secondary k-means measurement")
    ap.add_argument("--verbose", action="store_true",
                    help="print the whole detail object (per-iteration tables, notes, every leg) as the JSON line instead of the compact one; "
                         "the detail is always written to gpurun_out/bench_detail.json as well")
    ap.add_argument("--kmeans-counter-leg", action="store_true",
                    help="also time the flop layer in the contract arithmetic with the counter-hash draw (rounds 1-5's default leg)")
    ap.add_argument("--no-kmeans-reference", action="store_true",
                    help="skip the flop layer in the reference's own arithmetic and draw (two more full-size passes, ~70 s)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed MCCFR loop (no other-mode rate, convergence run, k-means or CPU baselines): the "
                         "command profiled by scripts/profile_round.sh, so that rocprof's per-kernel averages are those of "
                         "the timed region")
    ap.add_argument("--kmeans", default="flop", choices=["slice", "flop", "turn"],
                    help="k-means measurement in the `kmeans` object: the FULL flop-street configuration (default: BASELINE "
                         "configs[2], 1 286 792 x 32 Elkan iterations with k-means++, init_bounds and lookup, ~1.5 min), one "
                         "GPU's share of configs[4] (turn), or a bounded flop-layer slice (~10 s)")
    ap.add_argument("--kmeans-libm", default="contract", choices=["contract", "glibc"],
                    help="glibc: additionally time the k-means slice in the lm_glibc pass of the kernels (rp_kmeans_set_libm: the "
                         "reference's own libm arithmetic, unpruned) and report it beside the contract pass; opt-in")
    ap.add_argument("--window", type=int, default=None,
                    help="N > 1: local steps per exchange (the composed maps of `window` consecutive steps are folded "
                         "locally and all-gathered once; 1 = exchange every step).  Default: 4 on several GPUs (the periodic "
                         "exchange of north_star), 1 otherwise")
    ap.add_argument("--comm", default="native", choices=["native", "torch"],
                    help="N > 1: the library's own RCCL communicator driven from C (rp_mccfr_step_comm), or torch.distributed "
                         "collectives driven from Python")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the rank bookkeeping (unique-id broadcast, max-over-ranks time); nccl = RCCL "
                         "on the GPUs; gloo only for tests/test_emul_bench.py, where the ranks have no GPU")
    ap.add_argument("--prune-warmup", type=int, default=0, help="nlhe: warm-up epochs of the pruned-regime leg")
    ap.add_argument("--prune-threshold", type=float, default=-100.0, help="nlhe: regret threshold of the pruned-regime leg")
    ap.add_argument("--prune-explore", type=float, default=0.05, help="nlhe: exploration probability of the pruned-regime leg")
    ap.add_argument("--projection", action="store_true",
                    help="add an 8-GPU strong-scaling PROJECTION (one rank's share timed by a child run, an assumed wire time) "
                         "under the key `unmeasured`; never part of the default line")
    ap.add_argument("--force-sharded", action="store_true",
                    help="exercise the RCCL all-gather path even with one rank (plumbing check)")
    return ap.parse_args()


def cpu_baseline(args):
    """The CPU oracle (plain C restatement, 1 thread) on a bounded sample of the same workload."""
    import oracle
    from robopoker_amd import Game

    g = Game(args.game)
    B = 4096
    s = oracle.OracleSolver(g, args.regret, args.weight, args.sampling, batch=B, seed=args.seed)
    s.step()  # warm
    _, i0 = s.counters()
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < args.cpu_seconds:
        s.step()
        steps += 1
    dt = time.perf_counter() - t0
    _, i1 = s.counters()
    return {
        "value": (i1 - i0) / dt,
        "unit": "infoset-updates/s",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle/rp_oracle_mccfr.c, {args.game} {args.regret}/{args.weight}/{args.sampling}, "
                  f"batch {B}, {steps * B} trees in {dt:.1f} s on 1 host thread",
        "build": oracle.ORACLE_BUILD,
    }


def host_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU boxes show 256
    logical CPUs under a 16-core quota; more threads than that only get throttled)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline_all_cores(args, seconds=4.0):
    """The CPU port in the reference's parallel structure: ONE process, tree-parallel batch() on every
    core the process may use (rayon -> OpenMP: ora_mccfr_step_mt; host_cores), then the sequential update on one thread; median of 3 runs."""
    import oracle
    from robopoker_amd import Game

    threads = host_cores()
    if threads < 2:
        return None
    g = Game(args.game)
    best = None
    for B in (1 << 12, 1 << 14, 1 << 16):  # the batch that suits the host best (small: fork/join bound; large: the
        s = oracle.OracleSolver(g, args.regret, args.weight, args.sampling, batch=B, seed=args.seed)  # Decisions leave the caches)
        s.step_mt(threads)  # warm (thread pool, page faults)
        rates = []
        for _ in range(3):
            _, i0 = s.counters()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < seconds / 9.0:
                s.step_mt(threads)
            dt = time.perf_counter() - t0
            _, i1 = s.counters()
            rates.append((i1 - i0) / dt)
        if best is None or float(np.median(rates)) > best[0]:
            best = (float(np.median(rates)), B, rates)
    return {"value": best[0], "unit": "infoset-updates/s", "cores": threads, "kind": "port", "runs": best[2], "batch": best[1],
            "sample": f"oracle/rp_oracle_mccfr.c ora_mccfr_step_mt: one process, {threads} OpenMP threads traverse a {best[1]}-tree batch "
                      f"(Solver::batch, rayon in the reference), one thread applies the Decisions in tree order; best of batch 2^12 / "
                      f"2^14 / 2^16, median of 3 runs of {seconds / 9.0:.1f} s each"}


KERNEL_GROUPS = {
    "traverse": ("k_prepare_infos", "k_traverse"),
    "compact": ("k_count", "k_scan", "k_compact"),
    "update": ("k_chain", "k_block_maps", "k_combine", "k_fold"),
}


def profiled_traffic(group, batch):
    """HBM bytes per launch of a kernel group from the newest committed PMC reduction (scripts/profile_round.sh),
    valid only for the batch it was collected at; None otherwise."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mccfr_hbm_traffic.json")))
    if not files:
        return None, None
    doc = json.load(open(files[-1]))
    if doc.get("batch") != batch or doc.get("update", "ordered") != group[1]:
        return None, os.path.basename(files[-1])
    # a step launches ONE instantiation of a templated kernel (the walker alternates): average over the instances of a
    # kernel name, add up the different kernels of the group
    per_name = {}
    for name, v in doc["kernels"].items():
        if any(k in name for k in KERNEL_GROUPS[group[0]]):
            base = name.split("(")[0].split("<")[0].replace("void ", "")
            per_name.setdefault(base, []).append(v["hbm_bytes_per_launch"])
    total = sum(sum(v) / len(v) for v in per_name.values())
    return (total if per_name else None), os.path.basename(files[-1])


VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 2.0  # 4 SIMD-32 per CU, a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md)


def profiled_valu(group):
    """Wave-level VALU instructions per launch of a kernel group from the newest committed SQ counter reduction
    (scripts/round_start.sh step 9: rocprofv3 --pmc SQ_INSTS_VALU ... of this same command), or None."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mccfr_sq_counters.json")))
    if not files:
        return None, None
    doc = json.load(open(files[-1]))
    per_name = {}
    for name, v in doc["kernels"].items():
        if any(k in name for k in KERNEL_GROUPS[group]) and v.get("dispatches"):
            base = name.split("(")[0].split("<")[0].replace("void ", "")
            per_name.setdefault(base, []).append(v["counters"].get("SQ_INSTS_VALU", 0.0) / v["dispatches"])
    total = sum(sum(v) / len(v) for v in per_name.values())
    return (total if per_name else None), os.path.basename(files[-1])


def side_rate(args, g, local_rank, mode, steps=10):
    """The other update mode's rate on the same workload, reported beside `value` for transparency."""
    from robopoker_amd.mccfr import Solver

    s = Solver(g, args.regret, args.weight, args.sampling, batch=args.batch, seed=args.seed, device=local_rank)
    s.set_update_mode(mode)
    s.step_async(3)
    s.sync()
    _, i0 = s.counters()
    t0 = time.perf_counter()
    s.step_async(steps)
    s.sync()
    dt = time.perf_counter() - t0
    _, i1 = s.counters()
    s.close()
    return (i1 - i0) / dt


def strong_scaling_projection(args, local_rank, ms_single, ranks=8, window=4, timeout=120):
    """A PROJECTION, labelled as such: what the timed step would take spread over `ranks` GPUs.  Measured, on this one GPU: a rank's
    share of the step (batch / ranks trees) through the sharded code path — exchange windows, the library's own communicator and a
    real all-gather, with a world of one (`bench.py --force-sharded` in a child process under a timeout: a collective that hung
    must not take the contract line with it).  Assumed: the wire.  An exchange moves ranks x the per-rank summary (8.6 KB for
    Leduc) once per window; at that size an xGMI all-gather is latency, taken as 20 us per exchange."""
    import subprocess

    share = max(64, args.batch // ranks)
    cmd = [sys.executable, os.path.abspath(__file__), "--force-sharded", "--no-extras", "--batch", str(share), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--window", str(window), "--game", args.game, "--regret", args.regret, "--weight", args.weight,
           "--sampling", args.sampling, "--update", "composed", "--seed", str(args.seed), "--dist-backend", args.dist_backend]
    import socket

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, LOCAL_RANK=str(local_rank), RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"child exited {r.returncode}: {r.stderr[-300:]}")
    child = json.loads(lines[-1])
    share_ms = child["ms_per_step"]
    wire_ms = 0.020 / window
    return {"kind": "projection (one GPU measured, the wire assumed)", "ranks": ranks, "scaling": "strong",
            "batch_per_rank": share, "measured_ms_per_step_of_a_rank_share": share_ms, "exchange_window": window,
            "assumed_exchange_ms_per_step": wire_ms, "single_gpu_ms_per_step": ms_single,
            "projected_speedup": ms_single / (share_ms + wire_ms)}


RCCL_NRANKS = None  # the communicator's own count of ranks (an all-reduce of ones), set by init_rccl


def init_rccl(rank, world, backend="nccl"):
    """init_process_group + communicator creation with fd 1 pointed at stderr: RCCL prints a version banner to stdout
    when the communicator is created, and stdout must carry exactly one JSON line."""
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group(backend, rank=rank, world_size=world)
        warm = torch.ones(1, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(warm)  # sums one per rank: what the communicator itself says its size is
        if backend == "nccl":
            torch.cuda.synchronize()
        global RCCL_NRANKS
        got = RCCL_NRANKS = int(warm.item())
        if got != world or dist.get_world_size() != world:
            raise SystemExit(f"bench.py --gpus {world}: the communicator reports {got} ranks (get_world_size {dist.get_world_size()})")
    finally:
        import ctypes

        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)  # the banner sits in the C library's stdout buffer: flush it to stderr now
        os.dup2(saved, 1)
        os.close(saved)


def kmeans_sharded(rank, world, local_rank, n_points=16384, K=256, bins=256, iters=2):
    """The k-means exchange on real GPUs: every rank owns `n_points` flop-like histograms (weak scaling), k-means++ draws
    through the exact integer prefix over ranks, one all-reduce(sum) of the integer centroid sums per Elkan iteration
    (RCCL).  Returns whole-job points/s (all ranks call this; rank 0 reports)."""
    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from robopoker_amd.fixtures import flop_like_points, smooth_metric
    from robopoker_amd import lloyd
    from robopoker_amd.parallel import ShardedLayer

    seed = 0xF10F
    pts = flop_like_points(n_points, bins=bins, mass=47, seed=seed + rank)
    layer = lloyd.Layer(K, pts, "sinkhorn", smooth_metric(bins, 1), seed=seed, device=local_rank)
    sh = ShardedLayer(layer, K, bins, seed, device="cuda")
    t0 = time.perf_counter()
    sh.init_centroids()
    t_kpp = time.perf_counter() - t0
    sh.init_bounds()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        sh.step()
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    layer.close()
    return {"metric": "kmeans_points_per_sec", "value": n_points * world * iters / float(dt.item()), "unit": "points/s",
            "n_gpus": world, "scaling": "weak",
            "workload": f"flop-layer slice: {n_points} points per GPU, K={K}, bins={bins}, Sinkhorn EMD, {iters} Elkan iterations, "
                        "point-sharded with one integer all-reduce per iteration",
            "kmeanspp_s": t_kpp}


def convergence(args, g, local_rank, batch=1 << 16, epochs=512):
    """That the measured configuration actually solves the game: exploitability of the average strategy after a short
    run in the benchmarked update mode (the reference's Leduc test asserts < 0.08, crates/leduc/src/solver.rs:105-123)."""
    from robopoker_amd.mccfr import Solver

    s = Solver(g, args.regret, args.weight, args.sampling, batch=batch, seed=args.seed, device=local_rank)
    s.set_update_mode(args.update)
    t0 = time.perf_counter()
    s.solve(batch * epochs)
    s.sync()
    dt = time.perf_counter() - t0
    out = {"exploitability": s.exploitability(), "epochs": epochs, "batch": batch, "trees": batch * epochs, "seconds": dt,
           "update": args.update, "reference_threshold": 0.08}
    s.close()
    return out


def partition_agreement(a, b, K):
    """How alike two clusterings of the same points are, whatever their label order: the adjusted Rand index (Hubert & Arabie) and the
    fraction of points whose labels agree under the best one-to-one matching of clusters (Hungarian assignment on the K x K table)."""
    a, b = np.asarray(a, dtype=np.int64), np.asarray(b, dtype=np.int64)
    n = a.size
    table = np.bincount(a * K + b, minlength=K * K).reshape(K, K).astype(np.float64)
    comb = lambda x: x * (x - 1.0) / 2.0  # noqa: E731
    s_ij, s_a, s_b = comb(table).sum(), comb(table.sum(1)).sum(), comb(table.sum(0)).sum()
    expected = s_a * s_b / comb(float(n))
    ari = (s_ij - expected) / (0.5 * (s_a + s_b) - expected) if 0.5 * (s_a + s_b) != expected else 1.0
    try:
        from scipy.optimize import linear_sum_assignment

        r, c = linear_sum_assignment(-table)
        matched = float(table[r, c].sum() / n)
    except ImportError:
        matched = None
    return {"adjusted_rand_index": float(ari), "matched_label_fraction": matched}


def kmeans_secondary(args):
    """The second hot path.  At configs[2] (flop) the layer of record is the REFERENCE'S OWN arithmetic — glibc's expf / logf and
    Layer::init_centroids' SmallRng + WeightedIndex<f32> draw — because that is the pass whose buckets are the reference's bit for bit
    (north_star); the f32 contract arithmetic on the same draw rides beside it as `contract_arithmetic`."""
    try:
        from robopoker_amd import lloyd
    except ImportError:
        return None
    import oracle

    if args.kmeans == "slice":
        out = lloyd.bench_slice()
        centroids = out.pop("_centroids", None)
    elif args.kmeans == "turn":
        out = lloyd.bench_full("turn")
        centroids = out.pop("_centroids", None)
    else:
        from robopoker_amd.fixtures import flop_like_points

        pts = flop_like_points(1286792, bins=256, mass=47, seed=0xF10F)  # drawn once, shared by the flop passes below
        keep, same = {}, {}
        libm, rng = ("contract", "counter") if args.no_kmeans_reference else ("glibc", "reference")
        out = lloyd.bench_full("flop", pts=pts, libm=libm, rng=rng, keep=keep)
        centroids = out.pop("_centroids", None)
        out["arithmetic_note"] = ("rp_kmeans_set_libm(RP_LIBM_GLIBC) + rp_kmeans_set_rng(RP_RNG_REFERENCE, Flop): what a Linux build of the reference "
                                  "computes for this layer, bit for bit (tests/test_gpu_z_glibc_mode.py, tests/test_reference_seed.py)"
                                  if libm == "glibc" else "the f32 contract arithmetic with the counter-hash draw (--no-kmeans-reference)")
        if not args.no_kmeans_reference:
            # the f32 contract arithmetic (include/rp_math.h) on the same draw, and how far apart the two clusterings are
            try:
                con = lloyd.bench_full("flop", pts=pts, libm="contract", rng="reference", keep=same)
                con.pop("_centroids", None)
                slim = {k: con[k] for k in ("libm", "rng", "create_s", "kmeanspp_s", "init_bounds_s", "elkan_total_s", "lookup_s", "end_to_end_s",
                                            "points_per_s", "distances_total", "rms", "kernels_ms", "roofline_sinkhorn", "roofline_mfma")}
                slim["picks_differing"] = int((keep["picks"] != same["picks"]).sum())
                slim["buckets_differing"] = int((keep["buckets"] != same["buckets"]).sum())
                slim.update(partition_agreement(keep["buckets"], same["buckets"], 256))
                slim["note"] = ("labels permute once a pick differs, so `buckets_differing` says 'another clustering', not how different: the "
                                "adjusted Rand index and the matched-label fraction compare the two partitions themselves")
                out["contract_arithmetic"] = slim
            except Exception as exc:  # noqa: BLE001  (a reported extra, never fatal)
                out["contract_arithmetic"] = {"error": f"{type(exc).__name__}: {exc}"}
        if args.kmeans_counter_leg:
            try:
                cnt = lloyd.bench_full("flop", pts=pts)
                cnt.pop("_centroids", None)
                out["contract_counter_draw"] = cnt
            except Exception as exc:  # noqa: BLE001
                out["contract_counter_draw"] = {"error": f"{type(exc).__name__}: {exc}"}
        # configs[4]'s share of one GPU rides along (under a second of device time)
        try:
            turn = lloyd.bench_full("turn")
            turn.pop("_centroids", None)
            out["kmeans_turn"] = turn
        except Exception as exc:  # noqa: BLE001  (a reported extra, never fatal)
            out["kmeans_turn"] = {"error": f"{type(exc).__name__}: {exc}"}
    if args.kmeans_libm == "glibc":
        try:
            small = lloyd.bench_slice(n_points=4096, iters=1)
            gl = lloyd.bench_slice(n_points=4096, iters=1, libm="glibc")
            out["glibc_pass"] = {"points_per_sec": gl["value"], "contract_points_per_sec": small["value"], "unit": "points/s",
                                 "init_bounds_points_per_sec": gl["init_bounds_points_per_sec"],
                                 "contract_init_bounds_points_per_sec": small["init_bounds_points_per_sec"],
                                 "workload": "flop-layer slice N=4096, K=256, bins=256: one Elkan iteration after init_bounds, the contract pass "
                                             "beside the lm_glibc pass"}
        except Exception as exc:  # noqa: BLE001  (a reported extra, never fatal)
            out["glibc_pass"] = {"error": f"{type(exc).__name__}: {exc}"}
    if args.cpu_seconds > 0:
        out["cpu_baseline"] = lloyd.cpu_baseline_slice(oracle, seconds=min(args.cpu_seconds, 8.0))
        if centroids is not None:
            try:
                out["cpu_baseline_all_cores"] = lloyd.cpu_baseline_full(oracle, out, centroids, threads=host_cores())
            except Exception as exc:  # noqa: BLE001  (a reported extra, never fatal)
                out["cpu_baseline_all_cores"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _short(x, n=96):
    return x if not isinstance(x, str) or len(x) <= n else x[: n - 1] + "…"


def compact_line(line):
    """The default stdout line: the contract's keys whole, every extra reduced to its numbers, the k-means object LAST (the driver keeps
    the end of stdout).  Nothing is measured here: it is a view of `line`, whose full text goes to gpurun_out/bench_detail.json (and is
    the line itself under --verbose)."""
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in line}
    cfg = dict(line.get("config", {}))
    cfg["update_tolerance"] = _short(cfg.get("update_tolerance"), 60)
    out["config"] = cfg
    rf = line.get("roofline", {})
    out["roofline"] = {**_pick(rf, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "valu_source",
                               "hbm_frac_measured", "updates_per_launch", "avg_launch_ms", "traversal_kernel"),
                       **({"hbm_algorithmic": _pick(rf["hbm_algorithmic"], "achieved", "frac", "bytes_per_update")} if "hbm_algorithmic" in rf else {})}
    cb = line.get("cpu_baseline")
    out["cpu_baseline"] = {**_pick(cb, "value", "unit", "cores", "kind"), "sample": _short(cb.get("sample"), 120)} if isinstance(cb, dict) else cb
    if isinstance(line.get("cpu_baseline_all_cores"), dict):
        out["cpu_baseline_all_cores"] = _pick(line["cpu_baseline_all_cores"], "value", "unit", "cores", "kind", "error")
    for k in ("other_update_mode", "other_scaling", "rccl_nranks"):
        if k in line:
            out[k] = line[k]
    if isinstance(line.get("convergence"), dict):
        out["convergence"] = _pick(line["convergence"], "exploitability", "epochs", "trees", "seconds", "error")
    nl = line.get("nlhe")
    if isinstance(nl, dict):
        out["nlhe"] = {**_pick(nl, "value", "unit", "ms_per_step", "trees_per_step", "error"),
                       "roofline": _pick(nl.get("roofline", {}), "bound", "kernel", "achieved", "peak", "frac", "traffic", "traffic_source"),
                       "reference_batch_128": _pick(nl.get("reference_batch_128", {}), "value", "ms_per_step"),
                       "cpu_baseline": _pick(nl.get("cpu_baseline", {}), "value", "cores"),
                       "cpu_baseline_all_cores": _pick(nl.get("cpu_baseline_all_cores", {}), "value", "cores")}
    ab = line.get("abstraction_inputs")
    if isinstance(ab, dict):
        out["abstraction_inputs"] = {"river_equity": _pick(ab.get("river_equity", {}), "n", "device_ms", "showdowns_per_s"),
                                     "project_turn": _pick(ab.get("project_turn", {}), "n", "device_ms"),
                                     "cpu_baseline": _pick(ab.get("cpu_baseline", {}), "value", "unit", "cores"), **_pick(ab, "error")}
    km = line.get("kmeans")
    if isinstance(km, dict):
        def leg(d):  # one full-size layer: phase times, rates, the rooflines' numbers
            r = _pick(d, "libm", "rng", "N", "K", "bins", "iterations", "create_s", "kmeanspp_s", "init_bounds_s", "elkan_total_s", "lookup_s",
                      "end_to_end_s", "value", "points_per_s", "rms", "distances_total", "picks_differing", "buckets_differing",
                      "adjusted_rand_index", "matched_label_fraction", "error")
            if "kernels_ms" in d:
                r["kernels_ms"] = {k: round(v["total_ms"]) for k, v in d["kernels_ms"].items()}
            for name in ("roofline_sinkhorn", "roofline_mfma", "roofline_bounds", "roofline_variation"):
                if name in d:
                    r[name] = _pick(d[name], "bound", "kernel", "achieved", "peak", "unit", "frac", "useful_frac")
                    if "issue" in d[name]:
                        r[name]["issue_frac"] = d[name]["issue"].get("frac")
            return r
        k = {"metric": km.get("metric"), "unit": km.get("unit"), "workload": _short(km.get("workload"), 150)}
        if isinstance(km.get("kmeans_turn"), dict):
            k["kmeans_turn"] = _pick(leg(km["kmeans_turn"]), "N", "bins", "end_to_end_s", "value", "rms", "roofline_variation", "error")
        if isinstance(km.get("cpu_baseline"), dict):
            k["cpu_baseline"] = {**_pick(km["cpu_baseline"], "value", "unit", "cores", "kind"), "sample": _short(km["cpu_baseline"].get("sample"), 100)}
        if isinstance(km.get("cpu_baseline_all_cores"), dict):
            k["cpu_baseline_all_cores"] = _pick(km["cpu_baseline_all_cores"], "value", "unit", "cores", "estimate", "iteration_s_at_full_size", "error")
        if isinstance(km.get("mfma_bound"), dict):
            k["prune"] = _pick(km["mfma_bound"], "survivors", "candidates", "sampled_points", "sample_mismatches", "kpp_bound_pairs", "kpp_bound_kept")
        if isinstance(km.get("reference_seed_draw"), dict):
            k["reference_seed_draw"] = _pick(km["reference_seed_draw"], "chunks", "walked_term_by_term")
        if isinstance(km.get("contract_arithmetic"), dict):
            k["contract_arithmetic"] = leg(km["contract_arithmetic"])
        k.update(leg(km))  # the layer of record last: kmeans.value / end_to_end_s are the last numbers of the line
        k["end_to_end_s_by_arithmetic"] = {str(km.get("libm")): km.get("end_to_end_s"),
                                           **({"contract": km["contract_arithmetic"].get("end_to_end_s")} if isinstance(km.get("contract_arithmetic"), dict) else {})}
        out["kmeans"] = k
    return out


def abstraction_inputs(args, local_rank):
    """SURVEY §8f row f2 at full size on this GPU, with the oracle's river equity timed on one host core beside it."""
    # in a child process: robopoker_amd.deuce keeps its bulk arrays in torch tensors, and torch must be imported before
    # a process makes its first HIP call (one HIP runtime per process) — this process already runs the solver
    import subprocess

    code = ("import json, torch; from robopoker_amd import deuce; "
            f"print('ABSTRACTION ' + json.dumps(deuce.bench_inputs({int(local_rank)})))")
    res = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("ABSTRACTION ")]
    if res.returncode != 0 or not lines:
        raise RuntimeError(f"child exited {res.returncode}: {res.stderr.strip()[-300:]}")
    out = json.loads(lines[-1][len("ABSTRACTION "):])
    if args.cpu_seconds > 0:
        import random

        import oracle_deuce as od

        rng = random.Random(args.seed)
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < min(args.cpu_seconds, 3.0):
            cards = rng.sample(range(52), 7)
            od.river_equity(sum(1 << c for c in cards[:2]), sum(1 << c for c in cards[2:]))
            n += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n * 990 / dt, "unit": "showdowns/s", "cores": 1, "kind": "port",
                               "sample": f"{n} random river observations x 990 opposing holes (oracle/rp_oracle_deuce.c)"}
    return out


def nlhe_synth(args, rank, world, local_rank):
    """2^27-row table (A <= 9), 128 x 1500 Zipf(1.1)-popular Decisions per step per GPU,
    LinearRegret / LinearWeight, exchanged every step.  One step = one Solver::step worth of updates."""
    import torch
    import torch.distributed as dist

    from robopoker_amd.sparse import DeviceBatch, SparseProfile, synthetic_batch

    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        init_rccl(rank, world, args.dist_backend)
    A = 9
    prof = SparseProfile(args.rows, A, "linear", "linear", max_batch=args.decisions * (world if sharded else 1), device=local_rank)
    host = [synthetic_batch(args.decisions, args.rows, A, seed=args.seed + 17 * rank + k) for k in range(4)]
    batches = [DeviceBatch(*b, device=local_rank) for b in host]
    algo_bytes = sum(int((24 + 32 * b[1].astype(np.int64)).sum()) for b in host) / len(host)
    if sharded:
        from robopoker_amd.parallel import ShardedProfile

        sh = ShardedProfile(prof, max_batch=args.decisions, device="cuda")
        step = lambda k: sh.step(batches[k % 4])  # noqa: E731
    else:
        step = lambda k: prof.apply(batches[k % 4], args.update)  # noqa: E731

    def fence():
        torch.cuda.synchronize()
        prof.sync()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    fence()
    prof.profile(True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    fence()
    dt = time.perf_counter() - t0
    sort_ms, sort_n = prof.kernel_time("sort")
    app_ms, app_n = prof.kernel_time("apply")
    prof.profile(False)
    if sharded:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        app_avg = app_ms / max(app_n, 1)
        achieved = algo_bytes / (app_avg * 1e-3) / 1e9 if app_avg > 0 else 0.0
        line = {
            "metric": "mccfr_infoset_updates_per_sec", "value": args.decisions * world * args.steps / dt,
            "unit": "infoset-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "NLHE-scale synthetic infoset batches through the sparse profile (SURVEY §8d config 4; "
                                   "BASELINE configs[3] without the game engine)",
                       "rows": args.rows, "max_actions": A, "decisions_per_gpu": args.decisions, "row_popularity": "zipf(1.1)",
                       "regret": "linear", "weight": "linear",
                       "update": ("composed+allgather" if sharded else args.update), "parallelism": f"batch-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": "apply", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": algo_bytes,
                         "avg_launch_ms": app_avg, "kernels_ms": {"sort": sort_ms / max(sort_n, 1), "apply": app_avg},
                         "note": "24 + 32*|choices| bytes per update (SURVEY §8d); the hottest row of a Zipf(1.1) batch "
                                 "receives ~12 % of the touches and is one sequential chain in the ordered mode"},
        }
        if world == 1 and args.cpu_seconds > 0:
            import oracle

            rows_cpu = min(args.rows, 1 << 22)  # the CPU port allocates the table densely
            o = oracle.OracleProfile(rows_cpu, A, "linear", "linear")
            b = tuple([host[0][0] % rows_cpu] + list(host[0][1:]))
            nact = (2 + (b[0].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) >> np.uint64(40)) % np.uint64(A - 1)).astype(np.uint8)
            b = (b[0], nact, ((1 << nact.astype(np.uint32)) - 1).astype(np.uint16)) + b[3:]
            t0 = time.perf_counter()
            done = 0
            while time.perf_counter() - t0 < args.cpu_seconds:
                o.apply(b)
                done += len(b[0])
            dtc = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": done / dtc, "unit": "infoset-updates/s", "cores": 1, "kind": "port",
                                    "sample": f"oracle ora_profile_apply, the same batch on a {rows_cpu}-row table, {done} updates "
                                              f"in {dtc:.1f} s on 1 host thread"}
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    prof.close()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves (one process per
    GPU under torch.distributed.run on 127.0.0.1) and pass rank 0's line through."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def timed_mccfr(args, g, batch, rank, world, local_rank, sharded_mode, torch, dist):
    """W warm-up steps, then exactly K timed steps between barrier + synchronize fences; max over ranks."""
    from robopoker_amd.mccfr import Solver

    solver = Solver(g, args.regret, args.weight, args.sampling, batch=batch, seed=args.seed, device=local_rank)
    solver.set_update_mode(args.update)
    comm = None
    if sharded_mode and args.comm == "native":
        from robopoker_amd.parallel import Comm

        try:
            Comm.unique_id()  # opens librccl: fails on every rank alike, before any collective, if the library is not there
        except Exception as exc:  # noqa: BLE001
            if rank == 0:
                print(f"bench: native RCCL communicator unavailable ({exc}); using --comm torch", file=sys.stderr, flush=True)
            args.comm = "torch"
    if sharded_mode and args.comm == "native":

        # tree ids [rank*B, (rank+1)*B); the library's own RCCL communicator (csrc/comm.cpp): the kernels of a window and its
        # ncclAllGather of the per-cell composed maps are queued on the solver's stream by ONE C call, no host work between
        comm = Comm.from_process_group(local_rank)  # collective; raises on every rank alike if librccl cannot be opened
        solver.set_shard(rank, world)
        pending = [0]

        def step():
            pending[0] += 1
            if pending[0] == max(args.window, 1) * 8:  # a few windows per call: the queue stays ahead of the GPU
                solver.step_comm(comm, pending[0], args.window)
                pending[0] = 0

        def fence():
            if pending[0]:
                solver.step_comm(comm, pending[0], args.window)
                pending[0] = 0
            solver.sync()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
    elif sharded_mode:
        from robopoker_amd.parallel import ShardedSolver

        # the same exchange driven from Python over torch.distributed (one all_gather per window)
        sharded = ShardedSolver(solver, device="cuda", window=args.window)

        def step():
            sharded.step()

        def fence():
            sharded.flush()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
    else:
        def step():
            solver.step_async(1)

        def fence():
            solver.sync()

    for _ in range(args.warmup):
        step()
    fence()
    _, infos0 = solver.counters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    _, infos1 = solver.counters()
    out = {"infos_local": infos1 - infos0, "infos": infos1 - infos0, "dt": dt, "batch": batch,
           "variant": solver.kernel_variant() + ("+maps fused" if args.update == "composed" and solver.kernel_variant() == "static" else "")}
    # per-kernel durations for the roofline: the same K steps again with a HIP event pair around every kernel group on the
    # launch stream (the event packets cost a few microseconds per launch, so they stay out of the timed region above)
    solver.profile(True)
    for _ in range(args.steps):
        step()
    fence()
    for name in ("traverse", "update", "compact"):
        ms, n = solver.kernel_time(name)
        out[name + "_ms"] = ms / max(n, 1)
    solver.profile(False)
    solver.close()
    if comm is not None:
        comm.close()
    if sharded_mode:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        out["dt"] = float(tt.item())
        ti = torch.tensor([out["infos"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(ti, op=dist.ReduceOp.SUM)
        out["infos"] = int(ti.item())
    return out


def time_to_exploitability(make, trees_per_check, max_trees, thresholds=(0.08, 0.01, 0.003)):
    """Wall time of solve() until the average strategy's exploitability first drops below each threshold (the
    reference's Leduc test asserts < 0.08 after 2^18 trees, crates/leduc/src/solver.rs:105-123).  The exploitability
    evaluations themselves (validation, host side in the reference too) are outside the clock."""
    s = make()
    reached, spent, trees = {}, 0.0, 0
    while trees < max_trees and len(reached) < len(thresholds):
        t0 = time.perf_counter()
        s.solve(trees_per_check)
        if hasattr(s, "sync"):
            s.sync()
        spent += time.perf_counter() - t0
        trees += trees_per_check
        e = s.exploitability()
        for th in thresholds:
            if e < th and str(th) not in reached:
                reached[str(th)] = {"seconds": spent, "trees": trees, "exploitability": e}
    if hasattr(s, "close"):
        s.close()
    return reached


def convergence_times(args, g, local_rank):
    """time-to-exploitability for the GPU (benchmarked update mode, a few batch sizes: small batches update the table
    more often, large ones amortise the launches) and for the CPU oracle at the reference's batch_size = 1."""
    import oracle
    from robopoker_amd.mccfr import Solver

    out = {"thresholds": [0.08, 0.01, 0.003], "gpu": {}, "cpu": None}
    for batch in (1 << 10, 1 << 13, 1 << 16):
        def make(batch=batch):
            s = Solver(g, args.regret, args.weight, args.sampling, batch=batch, seed=args.seed, device=local_rank)
            s.set_update_mode(args.update)
            return s

        out["gpu"][str(batch)] = time_to_exploitability(make, batch * 8, batch * 2048)
    if args.cpu_seconds > 0:
        out["cpu"] = {"batch": 1, "cores": 1,
                      "reached": time_to_exploitability(
                          lambda: oracle.OracleSolver(g, args.regret, args.weight, args.sampling, batch=1, seed=args.seed),
                          1 << 15, 1 << 23)}
    return out


def nlhe_profiled_traffic(kernel, batch):
    """HBM bytes per launch of an NLHE kernel from the newest committed PMC reduction taken at THIS batch (profiles/r*_nlhe_hbm_traffic*.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH prescribes); (None, file) otherwise."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_nlhe_hbm_traffic*.json")))
    for f in reversed(files):
        doc = json.load(open(f))
        if doc.get("batch") != batch:
            continue
        vals = [v["hbm_bytes_per_launch"] for name, v in doc["kernels"].items() if kernel in name]
        if vals:
            return sum(vals) / len(vals), os.path.basename(f)
    return None, (os.path.basename(files[-1]) + " (another batch)" if files else None)


def nlhe_extra(args, local_rank):
    """BASELINE configs[3] on this GPU, as an extra of the default line: the Flagship solver type's step (level-synchronous traversal
    + composed table update) at a GPU-sized batch and at the reference's 128, with the dominant kernel's roofline; CPU oracle beside."""
    from robopoker_amd.nlhe import NlheSolver

    def run(batch, steps, warmup, profile):
        s = NlheSolver(cap_log2=args.nlhe_cap, regret="linear", weight="linear", batch=batch, seed=args.seed, device=local_rank,
                       sampling="pluribus")
        for _ in range(warmup):
            s.step("composed")
        n0, i0, _ = s.counters()
        t0 = time.perf_counter()
        for _ in range(steps):
            s.step("composed")
        n1, i1, keys = s.counters()  # synchronises
        dt = time.perf_counter() - t0
        out = {"value": (i1 - i0) / dt, "unit": "infoset-updates/s", "ms_per_step": dt / steps * 1e3, "trees_per_step": batch,
               "nodes_per_s": (n1 - n0) / dt, "infosets_in_table": keys}
        if profile:
            s.profile(True)
            for _ in range(steps):
                s.step("composed")
            g, c = s.kernel_times(), s.census()
            s.profile(False)
            n_all = c["terminal"] + c["chance"] + c["walker"] + c["opponent"]
            alg = (4 * n_all + c["walker"] * (48 + 32 + 36 + 24 + 4) + 8 * c["walker_children"] + c["opponent"] * (48 + 32 + 72 + 20 + 8)
                   + c["chance"] * 20)
            ms = g["expand"][0]
            ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else None
            out["kernel_ms_per_step"] = {k: v[0] / steps for k, v in g.items()}
            traffic, traffic_src = nlhe_profiled_traffic("k_nl_expand", batch)
            out["roofline"] = {"bound": "hbm", "kernel": "k_nl_expand", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBPS if ach else None, "traffic": traffic, "traffic_source": traffic_src,
                               "hbm_frac_measured": (traffic / (ms * 1e-3 / max(g["expand"][1], 1)) / 1e9 / HBM_PEAK_GBPS) if traffic and ms > 0 else None,
                               "algorithmic_bytes_per_launch": alg / max(g["expand"][1], 1),
                               "avg_launch_us": ms * 1e3 / max(g["expand"][1], 1),
                               "note": "one launch per tree level; DESIGN 3c: bytes = 4 N + 144 walker + 8 walker-children + 180 opponent + 20 chance"}
        s.close()
        return out

    big = run(args.nlhe_batch, 6, 3, True)
    big["workload"] = ("heads-up NLHE blueprint MCCFR, Nlhe<LinearRegret, LinearWeight, PluribusSampling> (BASELINE configs[3] on one GPU): "
                       "trees generated on the device, hash encoder, 2^%d-row table" % args.nlhe_cap)
    ref = run(128, 100, 10, False)  # 0.36 ms a step: a hundred of them, behind ten that fill the table's first infosets
    big["reference_batch_128"] = {"value": ref["value"], "unit": "infoset-updates/s", "ms_per_step": ref["ms_per_step"], "steps": 100, "warmup": 10}
    if args.cpu_seconds > 0:
        import oracle_nlmc

        o = oracle_nlmc.OracleNlhe(cap_log2=20, regret="linear", weight="linear", batch=128, seed=args.seed, sampling="pluribus")
        o.step()
        _, i0, _ = o.counters()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < min(5.0, args.cpu_seconds):
            o.step()
            n += 1
        dtc = time.perf_counter() - t0
        _, i1, _ = o.counters()
        big["cpu_baseline"] = {"value": (i1 - i0) / dtc, "unit": "infoset-updates/s", "cores": 1, "kind": "port",
                               "sample": f"oracle/rp_oracle_nlmc.c, batch 128, {n * 128} trees in {dtc:.1f} s on 1 host thread"}
        try:
            big["cpu_baseline_all_cores"] = nlhe_cpu_all_cores(args.seed, min(4.0, args.cpu_seconds), sampling="pluribus")
        except Exception as exc:  # noqa: BLE001  (a reported extra, never fatal)
            big["cpu_baseline_all_cores"] = {"error": f"{type(exc).__name__}: {exc}"}
    return big


def nlhe_cpu_all_cores(seed, seconds, sampling="external", cap_log2=20):
    """The NLHE CPU baseline on every host core: one oracle solver per thread (private table, batch 128, ctypes releases the GIL),
    all stepping for `seconds`; the sum of their infoset-updates.  Independent instances share no table, so this is an UPPER
    bound on what the reference's rayon batch over one shared profile can reach on these cores."""
    import threading

    import oracle_nlmc

    T = host_cores()
    sols = [oracle_nlmc.OracleNlhe(cap_log2=cap_log2, regret="linear", weight="linear", batch=128, seed=seed + 7919 * t, sampling=sampling)
            for t in range(T)]
    for o in sols:
        o.step()
    base = [o.counters()[1] for o in sols]
    steps = [0] * T
    t0 = time.perf_counter()

    def work(t):
        while time.perf_counter() - t0 < seconds:
            sols[t].step()
            steps[t] += 1

    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    done = sum(o.counters()[1] - b for o, b in zip(sols, base))
    return {"value": done / dt, "unit": "infoset-updates/s", "cores": T, "kind": "port",
            "sample": f"oracle/rp_oracle_nlmc.c, {T} independent solvers (one per host thread, private tables, batch 128), "
                      f"{sum(steps) * 128} trees in {dt:.1f} s; an upper bound for one shared-profile solver on these cores"}


def nlhe_real(args, rank, world, local_rank):
    """BASELINE configs[3]: Solver::step of the NLHE blueprint solver (trees generated, traversed and applied on the device),
    infoset-updates (= Decisions, the reference's `infos` counter) per second; the reference's batch of 128 trees beside the
    GPU-sized one; the CPU oracle on one host thread as the baseline.  On several GPUs the trees of an epoch are sharded by
    rank and the per-infoset entries exchanged by key (robopoker_amd.parallel.ShardedNlhe), weak scaling."""
    import torch

    from robopoker_amd.nlhe import NlheSolver

    sharded = world > 1 or args.force_sharded
    dist = None
    if sharded:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        init_rccl(rank, world, args.dist_backend)

    def run(batch, steps, warmup, profile=False, sampling=None, hyper=None, exact=False):
        s = NlheSolver(cap_log2=args.nlhe_cap, regret="linear", weight="linear", batch=batch, seed=args.seed, device=local_rank,
                       sampling=sampling or args.sampling, hyper=hyper)
        if exact:
            s.set_exact(True)
        if sharded:
            from robopoker_amd.parallel import ShardedNlhe

            sh = ShardedNlhe(s, device="cuda")
            one = sh.step
        else:
            one = lambda: s.step(args.update)  # noqa: E731
        for _ in range(warmup):
            one()
        n0, i0, _ = s.counters()
        if sharded:
            torch.cuda.synchronize()
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        n1, i1, keys = s.counters()  # counters() synchronises the stream
        dt = time.perf_counter() - t0
        infos, nodes = i1 - i0, n1 - n0
        if sharded:
            torch.cuda.synchronize()
            dist.barrier()
            t = torch.tensor([time.perf_counter() - t0, 0.0, 0.0], dtype=torch.float64, device="cuda")
            dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
            t[1], t[2] = float(infos), float(nodes)  # each rank counts the Decisions of its own trees
            dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
            dt, infos, nodes = float(t[0]), int(t[1]), int(t[2])
        prof = None
        if profile and not sharded:
            # a second pass of the same number of steps with a HIP event pair around every kernel group (the event packets cost
            # a few microseconds per launch, so they stay out of the timed region above)
            s.profile(True)
            for _ in range(steps):
                one()
            prof = {"groups": s.kernel_times(), "census": s.census(), "levels": s.last_shape()[0], "steps": steps}
            s.profile(False)
        s.close()
        return {"infos": infos, "nodes": nodes, "dt": dt, "keys": keys, "prof": prof}

    big = run(args.nlhe_batch, args.steps, args.warmup, profile=True)
    # RP_BENCH_NO_REF=1 (profiling runs: scripts/r3_nlhe_traffic.sh): every dispatch of the process belongs to the big batch
    ref_steps = max(args.steps, 100)  # 0.36 ms a step: a hundred of them, behind ten that fill the table's first infosets
    ref = run(128, ref_steps, 10) if not os.environ.get("RP_BENCH_NO_REF") else {"infos": 0, "dt": 1.0}
    if rank != 0:
        dist.destroy_process_group()
        return
    trees = args.nlhe_batch * args.steps * world
    line = {
        "metric": "mccfr_infoset_updates_per_sec", "value": big["infos"] / big["dt"], "unit": "infoset-updates/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": big["dt"] / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "heads-up NLHE blueprint MCCFR (BASELINE configs[3] on one GPU): trees generated on the device, "
                               "hash encoder (no trained abstraction), linear regret / linear weight; level-synchronous traversal",
                   "batch_per_gpu": args.nlhe_batch, "global_batch": args.nlhe_batch * world, "table_rows": 1 << args.nlhe_cap,
                   "max_actions": 9, "update": "composed, exchanged by infoset key" if sharded else args.update,
                   "infosets_in_table": big["keys"], "parallelism": f"tree-sharded x{world}"},
        "trees_per_s": trees / big["dt"], "nodes_per_s": big["nodes"] / big["dt"],
        "nodes_per_tree": big["nodes"] / trees, "infos_per_tree": big["infos"] / trees,
        "reference_batch_128": {"value": ref["infos"] / ref["dt"], "unit": "infoset-updates/s",
                                "ms_per_step": ref["dt"] / ref_steps * 1e3, "steps": ref_steps, "warmup": 10,
                                "note": "nlhe/src/solver.rs:11 batch_size = 128: one tree per workgroup of 512, the traversal in one launch "
                                        "(k_nl_tree, ~0.29 ms = the slowest tree: ~13 us per tree level of dependent, lane-serial work; "
                                        "values in the reference's own order = bit-exact Decisions); its last workgroup posts the "
                                        "Decisions count to pinned host memory; four more launches (emit, k_prep_one, block maps, fold); "
                                        "latency bound by construction — a step depends on the previous one (DESIGN 3c, round 6)"},
    }
    line["config"]["sampling"] = args.sampling
    if not sharded and not os.environ.get("RP_BENCH_NO_REF"):
        # the regime a training run spends all but its first minutes in: PluribusSampling past its warm-up (sample/pluribus.rs:72-101,
        # hyperparams/pruning.rs:45-51: 16 384 epochs, threshold -3e5, explore 0.05).  A fresh table has no regret below the
        # reference's threshold, so the leg sets the warm-up to 0 and a threshold that bites on the default regrets (fold 100,
        # check/call 50, raise 10, shove 0 — kicker/src/edge.rs:61-72): what is timed is the pruned traversal's machinery (coin,
        # masks, the terminal-child exemption, smaller trees), not a trained blueprint's pruning rate
        import ctypes

        from robopoker_amd import _lib as rp_lib

        hp = rp_lib.Hyper()
        rp_lib.load().rp_hyper_default(ctypes.byref(hp))
        hp.prune_warmup, hp.prune_threshold, hp.prune_explore = args.prune_warmup, args.prune_threshold, args.prune_explore
        pr_run = run(args.nlhe_batch, args.steps, args.warmup, sampling="pluribus", hyper=hp)
        line["pruned_regime"] = {
            "value": pr_run["infos"] / pr_run["dt"], "unit": "infoset-updates/s", "ms_per_step": pr_run["dt"] / args.steps * 1e3,
            "sampling": "pluribus", "prune_warmup": args.prune_warmup, "prune_threshold": args.prune_threshold,
            "prune_explore": args.prune_explore, "nodes_per_tree": pr_run["nodes"] / trees, "infos_per_tree": pr_run["infos"] / trees,
            "trees_per_s": trees / pr_run["dt"],
            "note": "same batch, PluribusSampling past its warm-up with a threshold that bites on a fresh table's default regrets"}
    if not sharded and not os.environ.get("RP_BENCH_NO_REF") and args.nlhe_batch > 2048:
        # the same batch with the regret vectors in the reference's own float order on the batch-wide kernels (rp_nlhe_set_exact: bit-exact
        # Decisions; 192 more bytes per node for the per-ancestor reach rows).  Batches up to 2 048 trees are always evaluated that way
        ex = run(args.nlhe_batch, max(2, args.steps // 2), 2, exact=True)
        line["exact_order"] = {"value": ex["infos"] / ex["dt"], "unit": "infoset-updates/s", "ms_per_step": ex["dt"] / max(2, args.steps // 2) * 1e3,
                               "note": "rp_nlhe_set_exact(1): recursed_value / ancestor_reach in the reference's order (flow.rs:166-216) at this batch"}
    pr = big["prof"]
    if pr:
        c, g, k = pr["census"], pr["groups"], pr["steps"]
        n_all = c["terminal"] + c["chance"] + c["walker"] + c["opponent"]
        # what k_nl_expand has to move per node — every node's meta once (the kind sort); a decision node's game
        # state (48 B), key slot (32 B), row (regrets 36 B, + weights 36 B at an opponent node), its node record (24 / 20 B) and
        # 8 B (link, factor) per child; a chance node 12 B + 8 B
        alg = (4 * n_all + c["walker"] * (48 + 32 + 36 + 24 + 4) + 8 * c["walker_children"] + c["opponent"] * (48 + 32 + 72 + 20 + 8)
               + c["chance"] * 20)
        ms = g["expand"][0]
        ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else None
        line["roofline"] = {"bound": "hbm", "kernel": "k_nl_expand", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": ach / HBM_PEAK_GBPS if ach else None, "traffic": None,
                            "algorithmic_bytes_per_launch": alg / max(g["expand"][1], 1),
                            "avg_launch_us": ms * 1e3 / max(g["expand"][1], 1), "launches_per_step": g["expand"][1] / k,
                            "note": "one launch per tree level; latency bound (random 32-B key probes and 144-B rows of a "
                                    f"2^{args.nlhe_cap}-row table, dependent on each other) — profiles/r03_nlhe_*"}
        line["kernel_ms_per_step"] = {name: v[0] / k for name, v in g.items()}
        line["nodes_by_kind_per_step"] = {name: v / k for name, v in c.items()}
        line["levels"] = pr["levels"]
    else:
        line["roofline"] = {"bound": "hbm", "kernel": "k_nl_expand", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": None, "traffic": None, "note": "not profiled in the sharded run"}
    if args.cpu_seconds > 0:
        import oracle_nlmc

        o = oracle_nlmc.OracleNlhe(cap_log2=min(args.nlhe_cap, 22), regret="linear", weight="linear", batch=128, seed=args.seed)
        o.step()
        _, i0, _ = o.counters()
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < args.cpu_seconds:
            o.step()
            steps += 1
        dtc = time.perf_counter() - t0
        _, i1, _ = o.counters()
        line["cpu_baseline"] = {"value": (i1 - i0) / dtc, "unit": "infoset-updates/s", "cores": 1, "kind": "port",
                                "sample": f"oracle/rp_oracle_nlmc.c, batch 128, {steps * 128} trees in {dtc:.1f} s on 1 host thread"}
        line["cpu_baseline_all_cores"] = nlhe_cpu_all_cores(args.seed, min(5.0, args.cpu_seconds))
    else:
        line["cpu_baseline"] = None
    if sharded:
        line["rccl_nranks"] = RCCL_NRANKS
    print(json.dumps(line), flush=True)
    if sharded:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.no_extras:
        args.no_kmeans, args.cpu_seconds = True, 0.0
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if args.scaling is None:
        args.scaling = "strong" if world > 1 else "weak"
    if args.window is None:
        args.window = 4 if world > 1 else 1
    if args.workload == "nlhe-synth":
        return nlhe_synth(args, rank, world, local_rank)
    if args.workload == "nlhe":
        return nlhe_real(args, rank, world, local_rank)

    from robopoker_amd import Game

    dist = None
    torch = None
    sharded_mode = world > 1 or args.force_sharded
    if sharded_mode and args.update == "ordered":
        raise SystemExit("the ordered update is single-GPU only (sharding exchanges composed maps)")
    if sharded_mode:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        init_rccl(rank, world, args.dist_backend)

    g = Game(args.game)
    A = g.max_actions

    def per_gpu_batch(mode):
        return max(64, args.batch // world) if mode == "strong" else args.batch

    batch = per_gpu_batch(args.scaling)
    m = timed_mccfr(args, g, batch, rank, world, local_rank, sharded_mode, torch, dist)
    other_scaling = None
    if world > 1:  # the other scaling mode's measurement rides in the same line
        om = "weak" if args.scaling == "strong" else "strong"
        o = timed_mccfr(args, g, per_gpu_batch(om), rank, world, local_rank, sharded_mode, torch, dist)
        other_scaling = {"scaling": om, "value": o["infos"] / o["dt"], "unit": "infoset-updates/s",
                         "ms_per_step": o["dt"] / args.steps * 1e3, "batch_per_gpu": o["batch"]}
    infos, dt = m["infos"], m["dt"]

    if rank == 0:
        # dominant kernel vs the HBM roofline.  Algorithmic bytes per infoset-update = key 24 B + A*16 B read
        # + A*16 B written = 24 + 32*A; the kernel that realises the update is "update".
        upd_avg_ms, trav_avg_ms = m["update_ms"], m["traverse_ms"]
        per_launch_updates = m["infos_local"] / max(args.steps, 1)
        dom, dom_ms = ("update", upd_avg_ms) if upd_avg_ms >= trav_avg_ms else ("traverse", trav_avg_ms)
        bytes_per_update = 24 + 32 * A
        achieved = per_launch_updates * bytes_per_update / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic, traffic_src = profiled_traffic((dom, args.update), batch)
        other = "ordered" if args.update == "composed" else "composed"
        args.batch = batch
        other_rate = side_rate(args, g, local_rank, other) if world == 1 and not args.force_sharded and not args.no_extras else None
        valu_instr, valu_src = profiled_valu(dom)
        hbm_alg = {"achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                   "bytes_per_update": bytes_per_update,
                   "note": "SURVEY §8d's 24 + 32 A algorithmic bytes per infoset-update x the updates of one launch / the "
                           "dominant kernel's event-timed duration: a nominal figure — these bytes never reach HBM here "
                           "(see hbm_frac_measured)"}
        hbm_measured = traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if traffic and dom_ms > 0 else None
        composed_small = args.update == "composed" and dom == "traverse"
        if composed_small and valu_instr:
            # Leduc's tables are 3.8 KB (L2 resident) and in the composed mode the Decisions never reach HBM (traversal + block
            # maps are one kernel): the kernel is priced against what binds it, VALU issue
            rate = valu_instr / (dom_ms * 1e-3)
            roofline_obj = {"bound": "valu", "kernel": dom, "achieved": rate, "peak": VALU_PEAK_WAVE_INSTR,
                            "unit": "wave-instructions/s", "frac": rate / VALU_PEAK_WAVE_INSTR,
                            "valu_instructions_per_launch": valu_instr, "valu_source": valu_src,
                            "valu_note": "SQ_INSTS_VALU per launch from the committed rocprofv3 --pmc reduction named in "
                                         "valu_source (a separate profiling run of this command at this batch) / this run's "
                                         "event-timed launch duration; peak = 256 CU x 4 SIMD x 2.4 GHz / 2 cycles per wave64 "
                                         "instruction",
                            "lane_instructions_per_update": valu_instr * 64.0 / per_launch_updates if per_launch_updates else None}
        else:
            roofline_obj = {"bound": "hbm", "kernel": dom, **{k: hbm_alg[k] for k in ("achieved", "peak", "unit", "frac")}}
        roofline_obj.update({
            "traffic": traffic, "traffic_source": traffic_src, "hbm_frac_measured": hbm_measured,
            "traffic_note": "PMC bytes per launch read from the committed rocprofv3 --pmc reduction named in "
                            "traffic_source (a separate profiling run at this batch), not measured in this run; "
                            "hbm_frac_measured = traffic / avg_launch_ms / 8 TB/s",
            "hbm_algorithmic": hbm_alg, "updates_per_launch": per_launch_updates, "avg_launch_ms": dom_ms,
            "kernels_ms": {"traverse": trav_avg_ms, "compact": m["compact_ms"], "update": upd_avg_ms},
            "traversal_kernel": m.get("variant"),
        })
        line = {
            "metric": "mccfr_infoset_updates_per_sec",
            "value": infos / dt,
            "unit": "infoset-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.game}-holdem external-sampling MCCFR, tables resident in HBM (BASELINE configs[1])",
                "regret": args.regret, "weight": args.weight, "sampling": args.sampling,
                "batch_per_gpu": batch, "global_batch": batch * world, "infosets": g.n_infos,
                "actions": A, "update": args.update + ("+allgather" if sharded_mode else ""),
                "exchange_window": args.window if sharded_mode else None,
                "update_tolerance": "composed: regret/weight/payoff within rtol 1e-4 per step of the sequential "
                                    "order (tests/test_gpu_mccfr.py), visits exact; ordered: bit-exact",
                "parallelism": f"tree-sharded x{world}",
            },
            "roofline": roofline_obj,
            "other_update_mode": {"update": other, "value": other_rate, "unit": "infoset-updates/s"},
        }
        if other_scaling is not None:
            line["other_scaling"] = other_scaling
        if world == 1 and not args.force_sharded and args.game == "leduc" and not args.no_extras:
            line["convergence"] = convergence(args, g, local_rank)
            if args.projection:  # opt-in, under a key that cannot be mistaken for a multi-GPU result
                try:
                    line["unmeasured"] = {"strong_scaling_projection": strong_scaling_projection(args, local_rank, dt / args.steps * 1e3)}
                except Exception as exc:  # noqa: BLE001  (a reported extra, never fatal)
                    line["unmeasured"] = {"strong_scaling_projection": {"error": f"{type(exc).__name__}: {exc}"}}
            try:
                line["time_to_exploitability"] = convergence_times(args, g, local_rank)
            except Exception as exc:  # noqa: BLE001  (a reported extra, never fatal)
                line["time_to_exploitability"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args)
            try:
                line["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args, seconds=min(5.0, args.cpu_seconds))
            except Exception as exc:  # noqa: BLE001  (a reported extra, never fatal)
                line["cpu_baseline_all_cores"] = {"error": f"{type(exc).__name__}: {exc}"}
        else:
            line["cpu_baseline"] = None
        if world == 1 and not args.force_sharded and not args.no_kmeans:
            km = kmeans_secondary(args)
            if km is not None:
                line["kmeans"] = km
        if world == 1 and not args.force_sharded and not args.no_extras:
            try:
                line["nlhe"] = nlhe_extra(args, local_rank)
            except Exception as exc:  # noqa: BLE001  (a reported extra, never fatal)
                line["nlhe"] = {"error": f"{type(exc).__name__}: {exc}"}
            try:
                line["abstraction_inputs"] = abstraction_inputs(args, local_rank)
            except Exception as exc:  # noqa: BLE001  (a reported extra, never fatal)
                line["abstraction_inputs"] = {"error": f"{type(exc).__name__}: {exc}"}
        if sharded_mode:
            line["rccl_nranks"] = RCCL_NRANKS
        try:  # the whole detail beside the line (scratch: gpurun_out/ travels back from the GPU box)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as fh:
                json.dump(line, fh)
        except OSError:
            pass
        print(json.dumps(line if args.verbose else compact_line(line)), flush=True)

    if sharded_mode and not args.no_kmeans:
        # The k-means exchange on the same ranks, AFTER the contract line is out (stdout carries exactly one JSON
        # line; this result goes to stderr).  A watchdog ends the process cleanly if a collective ever hangs.
        import threading

        guard = threading.Timer(240.0, lambda: os._exit(0))
        guard.daemon = True
        guard.start()
        try:
            km = kmeans_sharded(rank, world, local_rank)
        except Exception as exc:  # noqa: BLE001
            km = {"error": f"{type(exc).__name__}: {exc}"}
        if rank == 0:
            print("kmeans_sharded: " + json.dumps(km), file=sys.stderr, flush=True)
        guard.cancel()
    if sharded_mode:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

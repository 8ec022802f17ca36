"""The hierarchical clustering pipeline on one GPU: the order of ``forge::PreTraining::cluster``
(crates/forge/src/pretraining.rs:37-44) and the steps of ``lloyd::Layer::cluster`` (crates/lloyd/src/layer.rs:195-248).

    river   Lookup::grow(Street::Rive): every river isomorphism -> its quantised equity         (deuce.river_equity)
    turn    points = river lookup projected onto the turn isomorphisms (histograms over the 101 equity buckets);
            k-means++ / init_bounds / 32 Elkan iterations under Equity::variation; lookup, metric, future
    flop    points = turn lookup projected onto the flop isomorphisms (histograms over the 256 turn clusters);
            the same steps under Sinkhorn EMD over the turn layer's metric
    preflop Lookup::grow(Street::Pref): isomorphism k -> abstraction k (lookup.rs:179-184)

Everything between the streets stays in HBM: the projection writes the points in the layout
``rp_kmeans_create_device`` reads.  ``Artifacts`` mirrors lloyd/src/artifacts.rs:11-18 {lookup, metric, future}.
Seeds are this build's own by default; ``Layer.set_rng("reference", street)`` seeds k-means++ the reference's way
(SipHash(street) -> SmallRng -> WeightedIndex<f32>, layer.rs:156-166; include/rp_refrng.h).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import deuce
from .lloyd import Layer

K_CLUSTERS = {"flop": 256, "turn": 256}   # pokerkit KMEANS_{FLOP,TURN}_CLUSTER_COUNT (lib.rs:187-189)
ITERATIONS = {"flop": 32, "turn": 32}     # KmeansHyperParams::DEFAULT (lloyd/src/hyperparams/kmeans.rs:18-24)


@dataclass
class Artifacts:
    """lloyd::Artifacts for one street.  lookup: (obs int64[n] on the device, abstraction index uint8[n]);
    metric: normalised triangular distances float32[K(K-1)/2] (None on the river / preflop);
    future: centroid histograms uint32[K][bins] with their weights (None on the river / preflop)."""
    street: str
    obs: torch.Tensor
    abstraction: torch.Tensor
    metric: np.ndarray | None = None
    future: np.ndarray | None = None
    future_weight: np.ndarray | None = None
    timings: dict = field(default_factory=dict)


def cluster_river(device=0) -> Artifacts:
    t0 = time.perf_counter()
    obs = deuce.isomorphisms("rive", device=device)
    t_iso = deuce.kernel_ms()
    _, bucket = deuce.river_equity(obs)
    t_eq = deuce.kernel_ms()
    return Artifacts("rive", obs, bucket, timings={"isomorphisms_ms": t_iso, "equity_ms": t_eq, "wall_s": time.perf_counter() - t0})


def cluster_preflop(device=0, flop: Artifacts | None = None, flop_metric=None) -> Artifacts:
    """PrefLayer::cluster (forge/src/pretraining.rs:40-43; K = N = 169, no k-means iterations): the lookup is
    Lookup::grow(Street::Pref) (isomorphism k -> abstraction k, lookup.rs:179-184).  With the flop artifacts the layer's
    metric() and future() are produced too (layer.rs:85-114): the 169 preflop histograms are the projections of the flop
    lookup over all 19 600 flops of a pocket (Lookup::future: children enumerated here, bucket lookups on the device
    through rp_lookup_get), their pairwise distance (emd(x, y) + emd(y, x)) / 2 under the flop layer's metric
    (rp_sinkhorn_divergence), normalised by the maximum (Metric::from, metric.rs:127-141)."""
    import itertools

    from .lloyd import sinkhorn_divergence
    obs = deuce.isomorphisms("pref", device=device)
    art = Artifacts("pref", obs, torch.arange(obs.numel(), dtype=torch.uint8, device=obs.device))
    if flop is None:
        return art
    t0 = time.perf_counter()
    bins = int(flop.abstraction.max().item()) + 1
    table = deuce.Lookup("flop", flop.obs, flop.abstraction)
    combos = np.array(list(itertools.combinations(range(50), 3)), dtype=np.int64)  # boards as indices into the 50 other cards
    # all 169 x 19 600 children in ONE device lookup (a host loop of 169 round trips before)
    pockets = obs.cpu().tolist()
    children = np.empty((len(pockets), combos.shape[0]), dtype=np.int64)
    for k, o in enumerate(pockets):
        p1, p0 = (o & 0xff) - 1, ((o >> 8) & 0xff) - 1  # pocket cards, ascending: p0 < p1 (observation.rs:132-141)
        rest = np.array([c for c in range(52) if c not in (p0, p1)], dtype=np.int64)
        c = rest[combos] + 1  # (19600, 3), each row ascending
        children[k] = (c[:, 0] << 32) | (c[:, 1] << 24) | (c[:, 2] << 16) | ((p0 + 1) << 8) | (p1 + 1)
    buckets = table.lookup(torch.from_numpy(children.reshape(-1)).to(obs.device)).cpu().numpy().reshape(children.shape)
    hist = np.stack([np.bincount(b, minlength=bins) for b in buckets]).astype(np.uint32)
    table.close()
    K = obs.numel()
    tri = np.zeros(K * (K - 1) // 2, dtype=np.float32)
    if flop_metric is not None:
        hi, lo = np.array([(i, j) for i in range(K) for j in range(i)], dtype=np.int64).T
        d = (sinkhorn_divergence(hist[hi], hist[lo], flop_metric, device=device)
             + sinkhorn_divergence(hist[lo], hist[hi], flop_metric, device=device)) / np.float32(2.0)
        tri[hi * (hi - 1) // 2 + lo] = d  # Pair::merge (pair.rs:36-39)
        # Metric::from normalises by fold(f32::MIN_POSITIVE, f32::max) (metric.rs:127-141): an all-zero distance set stays 0, never NaN
        tri = tri / max(tri.max(), np.finfo(np.float32).tiny)
    art.metric, art.future, art.future_weight = tri, hist, hist.sum(axis=1).astype(np.uint64)
    art.timings = {"preflop_s": time.perf_counter() - t0}
    return art


def cluster_layer(street: str, below: Artifacts, tri=None, K=None, iterations=None, seed=None, log=None, limit=None, libm="contract",
                  rng="counter") -> Artifacts:
    """Layer::cluster for the turn (below = river artifacts) or the flop (below = turn artifacts, tri = its metric).
    libm="glibc" / rng="reference": the reference's own exp / ln and k-means++ draw (rp_kmeans_set_libm / rp_kmeans_set_rng)."""
    assert (street, below.street) in (("turn", "rive"), ("flop", "turn"))
    K = K or K_CLUSTERS[street]
    iterations = ITERATIONS[street] if iterations is None else iterations
    bins = deuce.RIVER_BUCKETS if street == "turn" else int(below.abstraction.max().item()) + 1
    kind = "variation" if street == "turn" else "sinkhorn"
    dev = below.obs.device
    say = log or (lambda m: None)
    tm = {}
    t0 = time.perf_counter()
    obs = deuce.isomorphisms(street, device=dev.index or 0)
    if limit:
        obs = obs[:limit].contiguous()
    table = deuce.Lookup(below.street, below.obs, below.abstraction)
    points = table.projections(obs, bins)
    tm["project_ms"] = deuce.kernel_ms()
    table.close()
    tm["hydrate_s"] = time.perf_counter() - t0
    say(f"{street}: {obs.numel()} points x {bins} bins projected in {tm['project_ms']:.1f} ms")
    t0 = time.perf_counter()
    layer = Layer(K, None, kind, tri, seed=deuce.STREETS[street] if seed is None else seed, device=dev.index or 0,
                  counts_dev_ptr=points.data_ptr(), shape=tuple(points.shape))
    if libm != "contract":
        layer.set_libm(libm)
    if rng != "counter":
        layer.set_rng(rng, deuce.STREETS[street])
    tm["create_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    layer.init_centroids()
    tm["init_s"] = time.perf_counter() - t0
    say(f"{street}: k-means++ {tm['init_s']:.2f} s")
    t0 = time.perf_counter()
    layer.init_bounds()
    tm["bound_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    moved = []
    for it in range(iterations):
        t1 = time.perf_counter()
        _, _, re = layer.step()
        moved.append(float(re))
        say(f"{street}: iteration {it} {time.perf_counter() - t1:.3f} s reassigned {re:.5f}")
    tm["iterate_s"] = time.perf_counter() - t0
    tm["reassigned"] = moved
    t0 = time.perf_counter()
    bucket, _ = layer.lookup()
    tm["lookup_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    metric = layer.metric()
    tm["metric_s"] = time.perf_counter() - t0
    future, weight = layer.centroids()
    tm["rms"] = layer.rms()
    tm["distances"], tm["sinkhorn_iterations"] = layer.stats()
    if kind == "sinkhorn":
        import os

        tm["prune"] = layer.prune_stats()
        tm["refresh"] = layer.refresh_stats()
        n_margin = int(os.environ.get("RP_LLOYD_MARGIN_SAMPLE", "0"))
        if n_margin:  # scripts/mfma_audit.py: the bound's room around the exact divergence on a sample of THESE points
            from .lloyd import margin_audit

            idx = torch.linspace(0, points.shape[0] - 1, n_margin, device=points.device).long()
            tm["margins"] = margin_audit(points[idx].cpu().numpy(), future, tri, device=dev.index or 0)
    layer.close()
    del points
    return Artifacts(street, obs, torch.from_numpy(bucket).to(dev), metric, future, weight, tm)


def run(device=0, log=None, flop_iterations=None, turn_iterations=None, libm="contract", rng="counter") -> dict[str, Artifacts]:
    """PreTraining::run's clustering order: river, turn, flop, preflop (pretraining.rs:24-44, Street::all().rev())."""
    out = {}
    out["rive"] = cluster_river(device)
    if log:
        log(f"river: {out['rive'].obs.numel()} isomorphisms, equity {out['rive'].timings['equity_ms']:.0f} ms")
    out["turn"] = cluster_layer("turn", out["rive"], iterations=turn_iterations, log=log, libm=libm, rng=rng)
    out["flop"] = cluster_layer("flop", out["turn"], tri=out["turn"].metric, iterations=flop_iterations, log=log, libm=libm, rng=rng)
    out["pref"] = cluster_preflop(device, out["flop"], out["flop"].metric)
    return out


# ---- one process per GPU (BASELINE configs[4]: point-sharded k-means across the GPUs of a node) ---------
def _slice(n: int, rank: int, world: int) -> tuple[int, int, int]:
    """Contiguous, equal-width index slices of an isomorphism list: (lo, hi, width); the last may be short."""
    width = (n + world - 1) // world
    lo = min(n, rank * width)
    return lo, min(n, lo + width), width


def _gather_u8(mine: torch.Tensor, width: int, n: int, group) -> torch.Tensor:
    """All-gather equal-width uint8 slices (the last padded) back into the list's order."""
    import torch.distributed as dist

    from .parallel import _all_gather_bytes
    world = dist.get_world_size(group)
    pad = torch.zeros(width, dtype=torch.uint8, device=mine.device)
    pad[: mine.numel()] = mine
    out = torch.empty(width * world, dtype=torch.uint8, device=mine.device)
    _all_gather_bytes(out, pad, group)
    return out[:n].contiguous()


def cluster_river_sharded(device, group=None) -> Artifacts:
    """Every rank lists the river isomorphisms (24 ms), evaluates ITS slice, and the 1-byte buckets are all-gathered."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    obs = deuce.isomorphisms("rive", device=device)
    lo, hi, width = _slice(obs.numel(), rank, world)
    _, mine = deuce.river_equity(obs[lo:hi].contiguous())
    t_eq = deuce.kernel_ms()
    return Artifacts("rive", obs, _gather_u8(mine, width, obs.numel(), group), timings={"equity_ms": t_eq, "slice": [lo, hi]})


def cluster_layer_sharded(street: str, below: Artifacts, tri=None, K=None, iterations=None, seed=None, log=None, group=None,
                          limit=None) -> Artifacts:
    """Layer::cluster with the points (and their Elkan bounds) sharded by rank: the table of the street below is
    replicated, each rank projects and owns a contiguous slice of this street's isomorphisms, centroid sums are
    all-reduced every iteration (robopoker_amd.parallel.ShardedLayer), the final assignments are all-gathered."""
    import torch.distributed as dist

    from .parallel import ShardedLayer
    assert (street, below.street) in (("turn", "rive"), ("flop", "turn"))
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    K = K or K_CLUSTERS[street]
    iterations = ITERATIONS[street] if iterations is None else iterations
    bins = deuce.RIVER_BUCKETS if street == "turn" else int(below.abstraction.max().item()) + 1
    kind = "variation" if street == "turn" else "sinkhorn"
    dev = below.obs.device
    say = (lambda m: log(f"[rank {rank}] {m}")) if log else (lambda m: None)
    tm = {}
    obs = deuce.isomorphisms(street, device=dev.index or 0)
    if limit:
        obs = obs[:limit].contiguous()
    lo, hi, width = _slice(obs.numel(), rank, world)
    table = deuce.Lookup(below.street, below.obs, below.abstraction)
    points = table.projections(obs[lo:hi].contiguous(), bins)
    tm["project_ms"] = deuce.kernel_ms()
    table.close()
    seed = deuce.STREETS[street] if seed is None else seed
    engine = Layer(K, None, kind, tri, seed=seed, device=dev.index or 0, counts_dev_ptr=points.data_ptr(), shape=tuple(points.shape))
    layer = ShardedLayer(engine, K, bins, seed, device=str(dev), group=group)
    t0 = time.perf_counter()
    layer.init_centroids()
    tm["init_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    layer.init_bounds()
    tm["bound_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for it in range(iterations):
        t1 = time.perf_counter()
        _, _, re = layer.step()
        say(f"{street}: iteration {it} {time.perf_counter() - t1:.3f} s reassigned {re:.5f}")
    tm["iterate_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    mine, _ = engine.lookup()
    bucket = _gather_u8(torch.from_numpy(mine).to(dev), width, obs.numel(), group)
    tm["lookup_s"] = time.perf_counter() - t0
    metric = engine.metric()
    future, weight = engine.centroids()
    engine.close()
    return Artifacts(street, obs, bucket, metric, future, weight, tm)


def run_sharded(device=0, group=None, log=None, flop_iterations=None, turn_iterations=None) -> dict[str, Artifacts]:
    """``run`` with one process per GPU (launch with torch.distributed.run; backend nccl = RCCL over xGMI).  Every rank
    returns the same artifacts."""
    out = {"rive": cluster_river_sharded(device, group)}
    out["turn"] = cluster_layer_sharded("turn", out["rive"], iterations=turn_iterations, log=log, group=group)
    out["flop"] = cluster_layer_sharded("flop", out["turn"], tri=out["turn"].metric, iterations=flop_iterations, log=log, group=group)
    out["pref"] = cluster_preflop(device, out["flop"], out["flop"].metric)  # 169 points: every rank computes them
    return out

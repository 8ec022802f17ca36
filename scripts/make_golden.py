#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the CPU oracle.

The reference (Rust) holds no golden vectors for these paths and cannot run here, so the committed vectors are
ORACLE outputs on seeded inputs: they freeze the arithmetic contract (rp_math.h), the RNG
definitions and the operation order across rounds — a change in any of them shows up as a diff here.
f32 values are stored as their raw u32 bit patterns.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from lloyd_fixtures import flop_hist, flop_like_points, flop_metric, smooth_metric, turn_like_points  # noqa: E402
from robopoker_amd import Game  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32).tolist()


def mccfr_cases():
    cases = []
    for game, regret, weight, sampling, batch, steps, seed in [
        ("kuhn", "floored", "linear", "external", 64, 6, 1),
        ("kuhn", "discounted", "quadratic", "pluribus", 33, 8, 2),
        ("leduc", "floored", "linear", "external", 200, 5, 3),
        ("leduc", "linear", "linear", "prunable", 77, 6, 4),
        ("leduc", "asymmetric", "exponential", "external", 50, 6, 5),
        ("rps", "summed", "constant", "external", 40, 8, 6),
    ]:
        g = Game(game)
        hp = oracle.default_hyper()
        hp.prune_warmup = 2
        hp.prune_threshold = -1.5
        s = oracle.OracleSolver(g, regret, weight, sampling, batch=batch, seed=seed, hyper=hp)
        for _ in range(steps):
            s.step()
        rows = s.export()
        cases.append(dict(game=game, regret=regret, weight=weight, sampling=sampling, batch=batch, steps=steps,
                          seed=seed, prune_warmup=2, prune_threshold=-1.5, counters=list(s.counters()),
                          regret_bits=bits(rows["regret"]), weight_bits=bits(rows["weight"]),
                          payoff_bits=bits(rows["payoff"]), visits=rows["visits"].tolist(),
                          exploitability_bits=bits([s.exploitability()])[0]))
    return cases


def sinkhorn_cases():
    tri = flop_metric()
    pairs = [([(0, 3), (5, 1), (12, 4)], [(2, 2), (8, 5), (20, 1), (24, 3)]),
             ([(0, 3), (5, 1), (12, 4), (24, 2)], [(0, 3), (5, 1), (12, 4), (24, 2)]),
             ([(1, 7)], [(30, 2), (31, 9)]),
             ([(3, 1), (4, 1), (5, 1), (6, 1)], [(3, 1), (4, 2), (5, 3), (6, 4), (7, 5)])]
    out = []
    for a, b in pairs:
        mu, nu = flop_hist(a), flop_hist(b)
        c, it = oracle.sinkhorn_cost(mu, nu, tri)
        out.append(dict(mu=a, nu=b, cost_bits=bits([c])[0], iterations=it,
                        divergence_bits=bits([oracle.sinkhorn_divergence(mu, nu, tri)])[0]))
    pts = turn_like_points(8, bins=101, mass=46, seed=11).astype(np.uint32)
    var = [dict(i=i, j=j, bits=bits([oracle.equity_variation(pts[i], pts[j])])[0]) for i in range(4) for j in range(4, 8)]
    return dict(metric="flop_metric(32): d(i,j)=(((7i+13j)%97)+1)/100 (crates/lloyd/src/sinkhorn.rs:252-262)",
                sinkhorn=out, variation_points="turn_like_points(8, bins=101, mass=46, seed=11)", variation=var)


def kmeans_cases():
    out = []
    for kind, K, N, bins, mass, seed, iters in [("sinkhorn", 5, 120, 32, 20, 21, 16), ("variation", 8, 400, 101, 46, 22, 0)]:
        if kind == "sinkhorn":
            pts, tri = flop_like_points(N, bins=bins, mass=mass, seed=seed), smooth_metric(bins, seed)
        else:
            pts, tri = turn_like_points(N, bins=bins, mass=mass, seed=seed), None
        hp = oracle.default_sinkhorn()
        if iters:
            hp.iterations = iters
        km = oracle.OracleKmeans(K, pts, kind, tri, hp=hp, seed=seed)
        chosen = km.init_centroids()
        km.init_bounds()
        drifts = []
        for _ in range(3):
            d, sizes, moved = km.step()
            drifts.append(bits(d))
        b, dist = km.assign()
        c, w = km.centroids()
        out.append(dict(kind=kind, K=K, N=N, bins=bins, mass=mass, seed=seed, sinkhorn_iterations=iters or 128,
                        chosen=chosen.tolist(), drift_bits=drifts, sizes=sizes.tolist(), buckets=b.tolist(),
                        distance_bits=bits(dist), centroid_weight=w.tolist(), centroid_checksum=int(c.astype(np.uint64).sum()),
                        metric_bits=bits(km.metric()), rms_bits=bits([km.rms()])[0]))
    return out


def deuce_cases():
    """Abstraction inputs: strength keys, canonical forms, river equities (raw f32 bits) and buckets, the head and a
    checksum of each street's isomorphism list, a few turn histograms."""
    import random

    import oracle_deuce as od
    rng = random.Random(2024)
    hands = [sum(1 << c for c in rng.sample(range(52), k)) for k in (5, 6, 7) for _ in range(40)]
    obs = []
    for n_board in (0, 3, 4, 5):
        for _ in range(25):
            cards = rng.sample(range(52), 2 + n_board)
            obs.append(od.obs_i64(sum(1 << c for c in cards[:2]), sum(1 << c for c in cards[2:])))
    river = [o for o in obs if len([b for b in range(8) if (o >> (8 * b)) & 0xff]) == 7]
    eq = [od.river_equity(*od.obs_from_i64(o)) for o in river]
    lists = {}
    for street, hi in (("pref", 1326), ("flop", 1326), ("turn", 160)):
        v = od.isomorphisms(street, 0, hi)
        lists[street] = dict(pockets=[0, hi], n=int(v.size), head=v[:12].tolist(), tail=v[-4:].tolist(),
                             xor=int(np.bitwise_xor.reduce(v)), sum_mod=int(v.astype(np.uint64).sum() % (1 << 61)))
    turn = od.isomorphisms("turn", 1200, 1210)[:6]
    return dict(hands=hands, strength_keys=[od.strength_key(h) for h in hands], obs=obs,
                canonical=[od.obs_i64(*od.isomorphism(*od.obs_from_i64(o))) for o in obs], river=river,
                equity_bits=bits([e[0] for e in eq]), won=[e[1] for e in eq], total=[e[2] for e in eq],
                bucket=[od.quantize(e[0]) for e in eq], lists=lists, turn=turn.tolist(),
                turn_histograms=od.project_river(turn).tolist())


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, data in [("mccfr_tables.json", mccfr_cases()), ("sinkhorn.json", sinkhorn_cases()), ("kmeans.json", kmeans_cases()),
                       ("deuce.json", deuce_cases())]:
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(data, f, separators=(",", ":"))
        print(name, os.path.getsize(os.path.join(OUT, name)), "bytes")


if __name__ == "__main__":
    main()

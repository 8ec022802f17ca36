#!/usr/bin/env python3
"""tests/golden/{kmeans_k256,mccfr_composed_big}.json: ORACLE outputs (the reference holds no vectors and cannot run here,
see scripts/make_golden.py) at the CONFIGURED shapes, which the per-test oracle runs cannot afford:

  kmeans_k256         Elkan over Sinkhorn EMD at K = 256, bins = 256 (flop-street shape), N = 2048: init_bounds, three
                      iterations, lookup — ~1.1 M full-size Sinkhorn solves, ~10 minutes on one host core
  mccfr_composed_big  Leduc, batch 2^18, composed update, three steps: 1024 chunks x 16 fold groups per infoset, the
                      two-level k_combine fold far beyond the 24 chunks the per-test cases reach

f32 values are stored as raw u32 bit patterns, large arrays as sha256 of their bytes."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from robopoker_amd import Game  # noqa: E402
from robopoker_amd.fixtures import flop_like_points, smooth_metric  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32).tolist()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def kmeans_k256():
    N, K, bins, seed = 2048, 256, 256, 0xF10F
    pts = flop_like_points(N, bins=bins, mass=47, seed=seed)
    tri = smooth_metric(bins, 1)
    km = oracle.OracleKmeans(K, pts, "sinkhorn", tri, seed=seed)
    start = np.random.default_rng(7).choice(N, size=K, replace=False).astype(np.uint64)
    km.set_centroids(start)
    t0 = time.time()
    km.init_bounds()
    j0, u0, l0 = km.bounds()
    print("init_bounds", time.time() - t0, flush=True)
    steps = []
    for _ in range(3):
        d, sizes, moved = km.step()
        j, u, lo = km.bounds()
        steps.append(dict(drift_bits=bits(d), sizes=sizes.tolist(), moved=moved, j_sha=sha(j), u_sha=sha(u.view(np.uint32)),
                          lower_sha=sha(lo.view(np.uint32))))
        print("step", time.time() - t0, flush=True)
    b, dist = km.assign()
    print("lookup", time.time() - t0, flush=True)
    c, w = km.centroids()
    return dict(points="flop_like_points(2048, bins=256, mass=47, seed=0xF10F)", metric="smooth_metric(256, 1)", N=N, K=K, bins=bins,
                seed=seed, start=start.tolist(), init_j=j0.tolist(), init_u_bits=bits(u0), init_lower_sha=sha(l0.view(np.uint32)),
                steps=steps, buckets=b.tolist(), distance_bits=bits(dist), centroid_weight=w.tolist(),
                centroid_sha=sha(c.astype(np.uint32)), rms_bits=bits([km.rms()])[0])


def mccfr_composed_big():
    g = Game("leduc")
    out = []
    for regret, weight, batch, steps, seed in [("floored", "linear", 1 << 18, 3, 2026), ("linear", "linear", 1 << 17, 2, 7)]:
        s = oracle.OracleSolver(g, regret, weight, "external", batch=batch, seed=seed)
        for _ in range(steps):
            s.step_world(1)  # the composed update's association (ora_mccfr_step_local + step_apply)
        rows = s.export()
        out.append(dict(game="leduc", regret=regret, weight=weight, sampling="external", batch=batch, steps=steps, seed=seed,
                        counters=list(s.counters()), regret_bits=bits(rows["regret"]), weight_bits=bits(rows["weight"]),
                        payoff_bits=bits(rows["payoff"]), visits=rows["visits"].tolist()))
    return out


def main():
    which = sys.argv[1:] or ["mccfr_composed_big", "kmeans_k256"]
    for name in which:
        data = {"kmeans_k256": kmeans_k256, "mccfr_composed_big": mccfr_composed_big}[name]()
        path = os.path.join(OUT, name + ".json")
        with open(path, "w") as f:
            json.dump(data, f, separators=(",", ":"))
        print(name, os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    main()

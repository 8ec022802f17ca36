#!/usr/bin/env python3
"""Full-size k-means configurations of BASELINE.json on ONE MI355X:
  flop  (configs[2]): N = 1 286 792 histograms, K = 256, bins = 256, mass 47, Sinkhorn EMD, k-means++ init,
                      init_bounds, --iters Elkan iterations, final lookup
  turn  (configs[4], one GPU's 1/8 share): N = 1 745 006, K = 256, bins = 101, mass 46, Equity::variation
Prints one JSON line with per-phase wall times and rates.  usage: full_kmeans.py flop|turn [iters] [N]"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np  # noqa: E402

from robopoker_amd import lloyd  # noqa: E402
from lloyd_fixtures import flop_like_points, smooth_metric, turn_like_points  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "flop"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 32
K = 256
if which == "flop":
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 1286792
    bins, kind = 256, "sinkhorn"
    pts = flop_like_points(N, bins=bins, mass=47, seed=0xF10F)
    tri = smooth_metric(bins, 1)
    bytes_per_point = 2320
else:
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 1745006
    bins, kind = 101, "variation"
    pts = turn_like_points(N, bins=bins, mass=46, seed=5)
    tri = None
    bytes_per_point = 2165

out = {"workload": which, "N": N, "K": K, "bins": bins, "metric": kind, "iterations": iters}
t0 = time.perf_counter()
L = lloyd.Layer(K, pts, kind, tri, seed=1)
out["create_s"] = time.perf_counter() - t0  # upload + point masses + memoised OT(p,p)
t0 = time.perf_counter()
L.init_centroids()
out["kmeanspp_s"] = time.perf_counter() - t0
d0, i0 = L.stats()
t0 = time.perf_counter()
L.init_bounds()
out["init_bounds_s"] = time.perf_counter() - t0
d1, i1 = L.stats()
out["init_bounds_distances_per_s"] = (d1 - d0) / out["init_bounds_s"]
per_iter = []
t_all = time.perf_counter()
for it in range(iters):
    da, ia = L.stats()
    t0 = time.perf_counter()
    drift, sizes, moved = L.step()
    dt = time.perf_counter() - t0
    db, ib = L.stats()
    per_iter.append({"s": round(dt, 4), "distances": db - da, "moved": round(float(moved), 5)})
    print(f"iter {it}: {dt:.3f}s distances={db - da} moved={moved:.4f}", file=sys.stderr, flush=True)
total = time.perf_counter() - t_all
out["elkan_total_s"] = total
out["points_per_s"] = N * iters / total
out["algorithmic_GBps"] = N * iters * bytes_per_point / total / 1e9
out["hbm_frac"] = out["algorithmic_GBps"] / 8000.0
out["per_iteration"] = per_iter
t0 = time.perf_counter()
L.lookup()
out["lookup_s"] = time.perf_counter() - t0
out["rms"] = L.rms()
d2, i2 = L.stats()
out["distances_total"] = d2
out["sinkhorn_iterations_total"] = i2
print(json.dumps(out), flush=True)

#!/usr/bin/env python3
"""Full-size audit of the interval-decided refresh (csrc/refresh_bound.hpp): the flop layer (BASELINE configs[2]) clustered twice on
the same points and the same draw — with the refresh bound, and with RP_LLOYD_NO_REFRESH_BOUND=1 (every refresh the bit-faithful
solve) — compared after EVERY Elkan iteration: assignments, drift bits, sizes; then the lookup's buckets and distance bits.
usage: r6_refresh_audit.py [iters] [N]      RP_FULL_LIBM=glibc / RP_FULL_RNG=reference as in full_kmeans.py"""
import json
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from robopoker_amd import lloyd  # noqa: E402
from robopoker_amd.fixtures import flop_like_points, smooth_metric  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1286792
libm, rng = os.environ.get("RP_FULL_LIBM", "contract"), os.environ.get("RP_FULL_RNG", "counter")
K, bins = 256, 256
pts = flop_like_points(N, bins=bins, mass=47, seed=0xF10F)
tri = smooth_metric(bins, 1)


def make(with_bound):
    if with_bound:
        os.environ.pop("RP_LLOYD_NO_REFRESH_BOUND", None)
    else:
        os.environ["RP_LLOYD_NO_REFRESH_BOUND"] = "1"
    layer = lloyd.Layer(K, pts, "sinkhorn", tri, seed=0)
    if libm != "contract":
        layer.set_libm(libm)
    if rng != "counter":
        layer.set_rng(rng, 1)
    return layer


a, b = make(True), make(False)
os.environ.pop("RP_LLOYD_NO_REFRESH_BOUND", None)
assert a.refresh_stats()["enabled"] == 1 and b.refresh_stats()["enabled"] == 0
pa, pb = a.init_centroids(), b.init_centroids()
assert np.array_equal(pa, pb)
a.init_bounds()
b.init_bounds()
rows, bad = [], 0
ta = tb = 0.0
for it in range(iters):
    t0 = time.perf_counter()
    da, sa, ma = a.step()
    t1 = time.perf_counter()
    db, sb, mb = b.step()
    t2 = time.perf_counter()
    ta, tb = ta + (t1 - t0), tb + (t2 - t1)
    ja, _, _ = a.bounds(lower=False)
    jb, _, _ = b.bounds(lower=False)
    _, uiv = a.upper_interval()
    diff = int((ja != jb).sum())
    drift_same = bool(np.array_equal(da.view(np.uint32), db.view(np.uint32)))
    bad += diff + (0 if drift_same else 1) + (0 if np.array_equal(sa, sb) else 1)
    rows.append({"iteration": it, "assignments_differing": diff, "drift_bits_equal": drift_same, "sizes_equal": bool(np.array_equal(sa, sb)),
                 "moved_equal": ma == mb, "interval_valued_points": int((uiv != 0).sum())})
    print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
ba, dda = a.lookup()
bb, ddb = b.lookup()
out = {"workload": f"flop layer N={N} K={K} bins={bins}, libm={libm}, rng={rng}, {iters} Elkan iterations: with the refresh bound vs every "
                   "refresh bit-faithful (RP_LLOYD_NO_REFRESH_BOUND=1)",
       "per_iteration": rows, "total_disagreements_over_iterations": bad,
       "lookup_buckets_differing": int((ba != bb).sum()), "lookup_distance_bits_differing": int((dda.view(np.uint32) != ddb.view(np.uint32)).sum()),
       "elkan_seconds_with_bound": ta, "elkan_seconds_without": tb, "refresh_stats": a.refresh_stats(),
       "distances_with": a.stats_ex(), "distances_without": b.stats_ex()}
print(json.dumps(out))

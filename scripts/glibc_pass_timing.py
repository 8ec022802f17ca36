#!/usr/bin/env python3
"""What the lm_glibc pass of the lloyd kernels costs beside the contract pass: a flop-like slice (K = bins = 256, mass 47), unpruned in
both (RP_LLOYD_NO_MFMA_BOUND / RP_LLOYD_NO_KPP_BOUND for the contract layer, so that the two run the same solves), init_bounds +
one Elkan iteration, wall clock around synchronous calls.  No torch: starts in a second on a fresh box.

    python scripts/glibc_pass_timing.py [N [K]]      -> one JSON line"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from robopoker_amd import lloyd  # noqa: E402
from robopoker_amd.fixtures import flop_like_points, smooth_metric  # noqa: E402


def run(N, libm, K=256, bins=256, seed=0xF10F):
    pts = flop_like_points(N, bins=bins, mass=47, seed=seed)
    layer = lloyd.Layer(K, pts, "sinkhorn", smooth_metric(bins, 1), seed=seed)
    if libm == "glibc":
        layer.set_libm("glibc")
    layer.set_centroids(np.random.default_rng(seed).choice(N, size=K, replace=False).astype(np.uint64))
    d0, i0 = layer.stats()
    t0 = time.perf_counter()
    layer.init_bounds()
    t1 = time.perf_counter()
    layer.step()
    t2 = time.perf_counter()
    d1, i1 = layer.stats()
    j, _, _ = layer.bounds()
    layer.close()
    return {"init_bounds_s": t1 - t0, "step_s": t2 - t1, "solves": int(d1 - d0), "sinkhorn_iterations": int(i1 - i0)}, j


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    os.environ["RP_LLOYD_NO_MFMA_BOUND"] = "1"
    os.environ["RP_LLOYD_NO_KPP_BOUND"] = "1"
    run(max(K, 64), "contract", K=K)  # warm the context
    a, ja = run(N, "contract", K=K)
    b, jb = run(N, "glibc", K=K)
    print(json.dumps({"workload": f"flop-like slice N={N}, K={K}, bins=256, mass 47, unpruned: init_bounds + one Elkan iteration",
                      "contract": a, "glibc": b, "glibc_over_contract_init_bounds": b["init_bounds_s"] / a["init_bounds_s"],
                      "buckets_that_differ": int((ja != jb).sum()), "points": N}))

"""Timing of the neighbor passes with and without the MFMA Sinkhorn bound on a slice of the flop configuration
(run on the GPU box): python scripts/quick_bound.py [N] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lloyd_fixtures import flop_like_points, smooth_metric  # noqa: E402

from robopoker_amd import lloyd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
K, bins = 256, 256
pts = flop_like_points(N, bins=bins, mass=47, seed=0xF10F)
tri = smooth_metric(bins, 1)
start = np.random.default_rng(5).choice(N, size=K, replace=False).astype(np.uint64)
results = {}
for label, env in (("bound", None), ("exact", "RP_LLOYD_NO_MFMA_BOUND")):
    if env:
        os.environ[env] = "1"
    layer = lloyd.Layer(K, pts, "sinkhorn", tri, seed=5)
    layer.set_centroids(start)
    layer.profile(True)
    t0 = time.perf_counter()
    layer.init_bounds()
    t_ib = time.perf_counter() - t0
    for _ in range(steps):
        layer.step()
    t0 = time.perf_counter()
    b, d = layer.lookup()
    t_lk = time.perf_counter() - t0
    mb_ms, mb_n = layer.kernel_time("mfma_bound")
    nb_ms, nb_n = layer.kernel_time("neighbor")
    st = layer.prune_stats()
    results[label] = (b, d)
    print(f"{label}: init_bounds {t_ib:.3f}s lookup {t_lk:.3f}s  kernels: mfma_bound {mb_ms:.1f} ms / {mb_n}, neighbor {nb_ms:.1f} ms / {nb_n}")
    if st["enabled"]:
        # one block iteration = 16 x-tiles x NT y-tiles x 4 steps x 2 contractions of 2048 flop; NT taken as 3 (<= 48 bins)
        flops = st["block_iterations"] * 16 * 3 * 4 * 2 * 2048 + st["cost_passes"] * 16 * 3 * 4 * 2048
        print(f"   prune: {st}  survivors/point {st['survivors'] / max(st['points'], 1):.3f}  "
              f"MFMA {flops / (mb_ms * 1e-3) / 1e12:.1f} TFLOP/s (peak 157.3)")
    layer.close()
    if env:
        del os.environ[env]
same = np.array_equal(results["bound"][0], results["exact"][0]) and np.array_equal(
    results["bound"][1].view(np.uint32), results["exact"][1].view(np.uint32))
print("pruned lookup == unpruned lookup:", same)

"""Two Elkan iterations of the full flop layer (N = 1 286 792, K = 256) for the HBM-traffic counters of k_bounds_update."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robopoker_amd import lloyd  # noqa: E402
from robopoker_amd.fixtures import flop_like_points, smooth_metric  # noqa: E402

N = 1286792
pts = flop_like_points(N, bins=256, mass=47, seed=0xF10F)
layer = lloyd.Layer(256, pts, "sinkhorn", smooth_metric(256, 1), seed=1)
layer.set_centroids(np.random.default_rng(1).choice(N, size=256, replace=False).astype(np.uint64))
layer.init_bounds()
for _ in range(2):
    layer.step()
print(layer.prune_stats())

"""Kernel-time split of the flop layer's Elkan iterations on the REAL points (full N)."""
import json
import sys
import time

import torch

from robopoker_amd import deuce, pretraining
from robopoker_amd.lloyd import Layer

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
riv = pretraining.cluster_river(0)
turn = pretraining.cluster_layer("turn", riv)
flop = deuce.isomorphisms("flop")
table = deuce.Lookup("turn", turn.obs, turn.abstraction)
pts = table.projections(flop, 256)
layer = Layer(256, None, "sinkhorn", turn.metric, seed=1, counts_dev_ptr=pts.data_ptr(), shape=tuple(pts.shape))
layer.init_centroids()
layer.init_bounds()
out = []
for it in range(iters):
    layer.profile(True)
    d0, i0 = layer.stats()
    t0 = time.perf_counter()
    _, _, moved = layer.step()
    dt = time.perf_counter() - t0
    d1, i1 = layer.stats()
    row = {"iteration": it, "s": round(dt, 3), "distances": d1 - d0, "moved": moved}
    for name in ("pairwise", "step", "recompute", "bounds", "drift", "selfcost"):
        ms, n = layer.kernel_time(name)
        row[name + "_ms"] = round(ms, 2)
    layer.profile(False)
    out.append(row)
print(json.dumps(out, indent=1))

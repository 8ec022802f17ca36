"""Phase profile of the flop layer on a slice of the REAL flop points (river -> turn clustering -> flop projection)."""
import json
import sys
import time

import torch

from robopoker_amd import deuce, pretraining
from robopoker_amd.lloyd import Layer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
riv = pretraining.cluster_river(0)
turn = pretraining.cluster_layer("turn", riv)
flop = deuce.isomorphisms("flop")
idx = torch.arange(0, flop.numel(), flop.numel() // n, device="cuda")[:n]
obs = flop[idx].contiguous()
table = deuce.Lookup("turn", turn.obs, turn.abstraction)
pts = table.projections(obs, 256)
supp = (pts > 0).sum(dim=1)
out = {"n": n, "support_mean": float(supp.float().mean()), "support_max": int(supp.max()), "support_le32": float((supp <= 32).float().mean())}
layer = Layer(256, None, "sinkhorn", turn.metric, seed=1, counts_dev_ptr=pts.data_ptr(), shape=tuple(pts.shape))


def phase(name, fn):
    d0, i0 = layer.stats()
    e0 = layer.exp_evals()
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    d1, i1 = layer.stats()
    e1 = layer.exp_evals()
    out[name] = {"s": round(dt, 3), "distances": d1 - d0, "iters_per_distance": (i1 - i0) / max(d1 - d0, 1),
                 "exps_per_distance": (e1 - e0) / max(d1 - d0, 1), "exps_per_s": (e1 - e0) / dt, "distances_per_s": (d1 - d0) / dt}


phase("kmeanspp", layer.init_centroids)
phase("init_bounds", layer.init_bounds)
phase("iterations_x8", lambda: [layer.step() for _ in range(8)])
phase("lookup", layer.lookup)
cent, w = layer.centroids()
out["centroid_support_mean"] = float((cent > 0).sum(axis=1).mean())
print(json.dumps(out, indent=1))

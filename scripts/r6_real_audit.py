#!/usr/bin/env python3
"""The interval-decided refresh (csrc/refresh_bound.hpp) audited on the REAL flop layer: river equities -> turn layer -> the 1 286 792
flop isomorphisms projected onto it (robopoker_amd.pretraining), then the flop layer clustered twice — with the refresh bound, and
with RP_LLOYD_NO_REFRESH_BOUND=1 — in the arithmetic named by RP_FULL_LIBM / RP_FULL_RNG, RP_LLOYD_AUDIT=1 on both (the unpruned search
behind init_bounds and lookup).  Compared: buckets, centroids, metric, rms, the per-iteration reassigned fractions.
    python scripts/r6_real_audit.py            -> one JSON object on stdout"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["RP_LLOYD_AUDIT"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402

from robopoker_amd import pretraining  # noqa: E402

say = lambda m: print(m, file=sys.stderr, flush=True)  # noqa: E731
libm, rng = os.environ.get("RP_FULL_LIBM", "contract"), os.environ.get("RP_FULL_RNG", "counter")
torch.cuda.set_device(0)
t0 = time.perf_counter()
rive = pretraining.cluster_river(0)
turn = pretraining.cluster_layer("turn", rive, log=say, libm=libm, rng=rng)
runs = {}
for name, env in (("with_refresh_bound", None), ("every_refresh_exact", "1")):
    if env:
        os.environ["RP_LLOYD_NO_REFRESH_BOUND"] = env
    else:
        os.environ.pop("RP_LLOYD_NO_REFRESH_BOUND", None)
    t1 = time.perf_counter()
    runs[name] = pretraining.cluster_layer("flop", turn, tri=turn.metric, log=say, libm=libm, rng=rng)
    runs[name].timings["wall_s"] = time.perf_counter() - t1
os.environ.pop("RP_LLOYD_NO_REFRESH_BOUND", None)
a, b = runs["with_refresh_bound"], runs["every_refresh_exact"]
ta, tb = a.timings, b.timings
out = {"points": "the real flop layer: 1 286 792 isomorphisms projected onto the clustered turn layer (pretraining)", "libm": libm, "rng": rng,
       "N": int(a.obs.numel()), "bins": int(turn.abstraction.max().item()) + 1,
       "buckets_differing": int((a.abstraction != b.abstraction).sum().item()),
       "centroids_equal": bool(np.array_equal(a.future, b.future) and np.array_equal(a.future_weight, b.future_weight)),
       "metric_bits_equal": bool(np.array_equal(np.asarray(a.metric).view(np.uint32), np.asarray(b.metric).view(np.uint32))),
       "rms_equal": ta["rms"] == tb["rms"], "rms": ta["rms"],
       "reassigned_per_iteration_equal": ta["reassigned"] == tb["reassigned"],
       "refresh": ta.get("refresh"), "iterate_s": {"with_refresh_bound": ta["iterate_s"], "every_refresh_exact": tb["iterate_s"]},
       "layer_s": {"with_refresh_bound": ta["wall_s"], "every_refresh_exact": tb["wall_s"]},
       "prune_audit": {k: {"audited_points": t["prune"]["audited_points"], "audit_mismatches": t["prune"]["audit_mismatches"]}
                       for k, t in (("with_refresh_bound", ta), ("every_refresh_exact", tb))},
       "distances": {"with_refresh_bound": ta["distances"], "every_refresh_exact": tb["distances"]}, "wall_s": time.perf_counter() - t0}
print(json.dumps(out), flush=True)

"""Round 6: where k_nl_tree's time goes at the reference's batch (128 trees, nlhe/src/solver.rs:11).
Needs the diagnostic build (-DNL_TREE_PROF, scripts/r6_nltree_prof.sh): 10 ns ticks of tid 0 per phase, summed over trees."""
import ctypes
import json
import sys
import time

import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch  # noqa: F401

from robopoker_amd import _lib
from robopoker_amd.nlhe import NlheSolver

lib = _lib.load()
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
s = NlheSolver(cap_log2=27, regret="linear", weight="linear", batch=batch, seed=1, device=0, sampling="external")
for _ in range(10):
    s.step("composed")
out = (ctypes.c_uint64 * 16)()
lib.rp_nl_tree_prof.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
lib.rp_nl_tree_prof(out, 1)
n0, i0, _ = s.counters()
t0 = time.perf_counter()
for _ in range(steps):
    s.step("composed")
n1, i1, _ = s.counters()
dt = time.perf_counter() - t0
lib.rp_nl_tree_prof(out, 0)
v = np.array(list(out), dtype=np.float64)
names = ["root", "expand+scan", "place_children", "make_child", "up", "ex_walker", "down", "fill", "group"]
trees = v[9]
res = {"batch": batch, "steps": steps, "ms_per_step": dt / steps * 1e3, "infos_per_s": (i1 - i0) / dt,
       "levels_per_tree": v[10] / trees, "nodes_per_tree": v[11] / trees,
       "us_per_tree_by_phase": {n: v[k] * 0.01 / trees for k, n in enumerate(names)}}
res["us_per_tree_total"] = sum(res["us_per_tree_by_phase"].values())
rec = (ctypes.c_uint32 * (batch * 16))()
lib.rp_nl_tree_rec.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32]
lib.rp_nl_tree_rec(rec, batch)
r = np.array(list(rec), dtype=np.float64).reshape(batch, 16)
tot = r[:, :9].sum(axis=1) * 0.01
order = np.argsort(-tot)
res["last_step_trees_by_time"] = [
    {"us": round(float(tot[i]), 1), "levels": int(r[i, 9]), "nodes": int(r[i, 10]), "walker_nodes": int(r[i, 11]),
     "phases_us": {n: round(float(r[i, k]) * 0.01, 1) for k, n in enumerate(names)}}
    for i in list(order[:6]) + list(order[batch // 2: batch // 2 + 2]) + list(order[-2:])]
A = np.stack([np.ones(batch), r[:, 9], r[:, 10]], axis=1)
coef, *_ = np.linalg.lstsq(A, tot, rcond=None)
res["fit_us"] = {"const": float(coef[0]), "per_level": float(coef[1]), "per_node": float(coef[2]),
                 "max_tree_us": float(tot.max()), "mean_tree_us": float(tot.mean())}
print(json.dumps(res, indent=1))

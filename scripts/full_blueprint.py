#!/usr/bin/env python3
"""One measured end-to-end blueprint configuration on the TRAINED abstraction:

    pretraining.run (river equities -> turn k-means -> flop k-means -> preflop; forge/src/pretraining.rs:24-50)
      -> the four isomorphism -> abstraction Lookup tables (nlhe/src/encoder.rs:30-36)
      -> NlheSolver(tables=...) = Flagship<LinearRegret, LinearWeight, PluribusSampling> for a fixed number of steps
      -> NlheProfile::rows streamed to a blueprint file (COPY format) -> Hydrate into a fresh solver
      -> the next batch of both solvers is identical (every key, mask and the regret / policy bits).

usage: full_blueprint.py [batch] [steps] [cap_log2]      -> one JSON object on stdout (stage times in seconds)"""
import json
import os
import sys
import tempfile
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from robopoker_amd import deuce, formats, pretraining  # noqa: E402
from robopoker_amd.nlhe import NlheSolver  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 26
say = lambda m: print(m, file=sys.stderr, flush=True)  # noqa: E731
out = {"batch": batch, "steps": steps, "table_rows": 1 << cap, "solver": "Nlhe<LinearRegret, LinearWeight, PluribusSampling> (nlhe/src/lib.rs:86-90)"}
t0 = time.perf_counter()
art = pretraining.run(0, log=say)
torch.cuda.synchronize()
out["abstraction_s"] = time.perf_counter() - t0
out["abstraction_layers"] = {k: {"isomorphisms": int(a.obs.numel()), "buckets": int(a.abstraction.max().item()) + 1} for k, a in art.items()}
t0 = time.perf_counter()
tables = [deuce.Lookup(name, art[name].obs, art[name].abstraction) for name in ("pref", "flop", "turn", "rive")]
out["lookup_tables_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
a = NlheSolver(cap_log2=cap, regret="linear", weight="linear", batch=batch, seed=2026, tables=tables, sampling="pluribus")
out["solver_create_s"] = time.perf_counter() - t0
per_step = []
for s in range(steps):
    t1 = time.perf_counter()
    a.step("composed")
    n, i, k = a.counters()  # synchronises
    per_step.append({"s": round(time.perf_counter() - t1, 4), "nodes": n, "infos": i, "infosets_in_table": k})
    say(f"step {s}: {per_step[-1]}")
out["steps_s"] = sum(p["s"] for p in per_step)
out["per_step"] = per_step
nodes, infos, keys = a.counters()
late = per_step[len(per_step) // 2:]
out["infoset_updates_per_s_second_half"] = (late[-1]["infos"] - per_step[len(per_step) // 2 - 1]["infos"]) / sum(p["s"] for p in late)
out["infosets_in_table"] = keys
t0 = time.perf_counter()
exp = a.export()
out["export_s"] = time.perf_counter() - t0
path = os.path.join(tempfile.gettempdir(), "rp_blueprint.pgcopy")
t0 = time.perf_counter()
rows = formats.write_blueprint(path, *exp, only_visited=False)
out["blueprint_write_s"] = time.perf_counter() - t0
out["blueprint_rows"] = int(rows)
out["blueprint_bytes"] = os.path.getsize(path)
t0 = time.perf_counter()
back = formats.read_blueprint(path)
out["blueprint_read_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
b = NlheSolver(cap_log2=cap, regret="linear", weight="linear", batch=batch, seed=2026, tables=tables, sampling="pluribus")
b.load(*back, epoch=a.epoch)
out["hydrate_s"] = time.perf_counter() - t0
os.remove(path)
da, db = a.batch(), b.batch()
same = da["n"] == db["n"]
for f in ("tree", "past", "present", "choices", "n_actions", "expanded"):
    same = same and bool(np.array_equal(da[f], db[f]))
for f in ("regret", "policy", "payoff"):
    same = same and bool(np.array_equal(da[f].view(np.uint32), db[f].view(np.uint32)))
out["next_batch_identical_after_hydrate"] = bool(same)
out["next_batch_decisions"] = int(da["n"])
a.close()
b.close()
for t in tables:
    t.close()
print(json.dumps(out), flush=True)

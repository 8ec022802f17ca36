#!/usr/bin/env python3
"""The whole abstraction pipeline at full size on ONE MI355X (robopoker_amd.pretraining.run): river equities, turn and
flop clustering on the real point sets, preflop.  usage: full_abstraction.py [flop_iterations] [turn_iterations]"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from robopoker_amd import pretraining  # noqa: E402

fi = int(sys.argv[1]) if len(sys.argv) > 1 else None
ti = int(sys.argv[2]) if len(sys.argv) > 2 else None
t0 = time.perf_counter()
art = pretraining.run(0, log=lambda m: print(m, file=sys.stderr, flush=True), flop_iterations=fi, turn_iterations=ti)
total = time.perf_counter() - t0
out = {"total_s": total}
for street, a in art.items():
    sizes = torch.bincount(a.abstraction.to(torch.int64)).cpu().numpy()
    t = {k: v for k, v in a.timings.items() if k != "reassigned"}
    if "reassigned" in a.timings:
        t["reassigned_first_last"] = [a.timings["reassigned"][0], a.timings["reassigned"][-1]] if a.timings["reassigned"] else []
    out[street] = {"n": a.obs.numel(), "abstractions": int(len(sizes)), "empty_clusters": int((sizes == 0).sum()),
                   "largest_cluster": int(sizes.max()), "timings": t}
    if a.metric is not None:
        out[street]["metric_max"] = float(np.max(a.metric))
        out[street]["metric_min"] = float(np.min(a.metric))
print(json.dumps(out), flush=True)

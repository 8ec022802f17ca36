#!/usr/bin/env python3
"""Round 6, verdict item 2 (interval-decided refresh), measured BEFORE building it: on a flop-layer slice, at every Elkan iteration,
how many of the stale-bound refreshes (elkan.rs:113-117) an interval of the scaling-domain bound would settle — i.e. the refreshed
u = d(x, c(x)) is certainly <= min_k max(l[k], P[j][k] / 2), so the candidate loop finds nothing whatever u's last bits are — and
how wide the intervals are against the exact value.   usage: r6_refresh_study.py [N] [iters]"""
import json
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from robopoker_amd import lloyd  # noqa: E402
from robopoker_amd.fixtures import flop_like_points, smooth_metric  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
K, bins = 256, 256
pts = flop_like_points(N, bins=bins, mass=47, seed=0xF10F)
layer = lloyd.Layer(K, pts, "sinkhorn", smooth_metric(bins, 1), seed=0)
if os.environ.get("RP_FULL_LIBM") == "glibc":
    layer.set_libm("glibc")
layer.init_centroids()
layer.init_bounds()
rows = []
for it in range(iters):
    j, u, low = layer.bounds()  # the state the step starts from: u carries the last drift, every bound is stale after iteration 0
    lo, hi = layer.bound_intervals()
    drift, _, _ = layer.step()
    j2, u2, _ = layer.bounds()
    pw = layer.pairwise_last()  # of the centroids this step began with: the ones the intervals above were computed against
    half = 0.5 * pw
    np.fill_diagonal(half, np.inf)
    mid = half.min(axis=1)
    need = u > mid[j]
    idx = np.nonzero(need)[0]
    jj = j[idx].astype(np.int64)
    ilo, ihi = lo[idx, jj], hi[idx, jj]
    th = np.maximum(low[idx], half[jj])  # per candidate k: u must exceed both to fire
    th[np.arange(idx.size), jj] = np.inf
    thr = th.min(axis=1)
    settled = ihi <= thr
    fires = ilo > thr  # certainly has a candidate
    stay = j2[idx] == j[idx]
    width = (ihi - ilo) / np.maximum(ilo, 1e-9)
    # containment: a refreshed point that stayed has u2 = RN(d(x, c_j) + drift[j]) (Bounds::update); rounding is monotone
    dj = drift[jj]
    inside = ((ilo + dj).astype(np.float32) <= u2[idx]) & (u2[idx] <= (ihi + dj).astype(np.float32))
    checked = stay if it > 0 else np.zeros_like(stay)
    rows.append({"iteration": it, "refresh_candidates": int(idx.size), "settled_by_interval": int(settled.sum()),
                 "certainly_has_candidate": int(fires.sum()), "undecided": int((~settled & ~fires).sum()),
                 "infinite_hi": int(np.isinf(ihi).sum()), "median_rel_width": float(np.median(width[np.isfinite(width)])) if idx.size else None,
                 "p99_rel_width": float(np.quantile(width[np.isfinite(width)], 0.99)) if idx.size else None,
                 "points_that_moved": int((~stay).sum()), "containment_checked": int(checked.sum()),
                 "containment_violations": int((checked & ~inside).sum())})
    print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
print(json.dumps({"N": N, "K": K, "libm": os.environ.get("RP_FULL_LIBM", "contract"), "rows": rows}))

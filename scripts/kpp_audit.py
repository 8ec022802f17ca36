#!/usr/bin/env python3
"""The k-means++ interval filter (csrc/kpp_bound.hpp) audited at FULL size: Layer::init_centroids + init_bounds of the flop layer with
the filter and without it (RP_LLOYD_NO_KPP_BOUND2=1) — the picks, the buckets and the upper-bound bits must be identical.
    python scripts/kpp_audit.py [N [contract|glibc]]      -> one JSON object on stdout"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402

from robopoker_amd import lloyd  # noqa: E402
from robopoker_amd.fixtures import flop_like_points, smooth_metric  # noqa: E402

N, K, bins = int(sys.argv[1]) if len(sys.argv) > 1 else 1286792, 256, 256
libm = sys.argv[2] if len(sys.argv) > 2 else "contract"
pts, tri = flop_like_points(N, bins=bins, mass=47, seed=0xF10F), smooth_metric(256, 1)
out = {"N": N, "K": K, "bins": bins, "libm": libm}
res = {}
for name, off in (("filtered", False), ("unfiltered", True)):
    os.environ.pop("RP_LLOYD_NO_KPP_BOUND2", None)
    if off:
        os.environ["RP_LLOYD_NO_KPP_BOUND2"] = "1"
    layer = lloyd.Layer(K, pts, "sinkhorn", tri, seed=1)
    if libm != "contract":
        layer.set_libm(libm)
    t0 = time.perf_counter()
    chosen = np.asarray(layer.init_centroids())
    t1 = time.perf_counter()
    layer.init_bounds()
    t2 = time.perf_counter()
    j, u, _ = layer.bounds()
    st = layer.prune_stats()
    res[name] = (chosen, np.asarray(j).copy(), np.asarray(u).view(np.uint32).copy())
    out[name] = {"kmeanspp_s": t1 - t0, "init_bounds_s": t2 - t1, "exact_distances": layer.stats()[0],
                 **{k: v for k, v in st.items() if k.startswith("kpp_") or k.startswith("sample")}}
    layer.close()
a, b = res["filtered"], res["unfiltered"]
out["picks_differing"] = int((a[0] != b[0]).sum())
out["buckets_differing"] = int((a[1] != b[1]).sum())
out["upper_bound_bits_differing"] = int((a[2] != b[2]).sum())
print(json.dumps(out), flush=True)

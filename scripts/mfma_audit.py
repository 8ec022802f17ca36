#!/usr/bin/env python3
"""The MFMA Sinkhorn prune audited at FULL size: RP_LLOYD_AUDIT=1 runs the unpruned search behind every
pruned pass (init_bounds — the k-means++ shortcut included —, lookup) and counts the points whose bucket or distance bits
differ; a sample of (point, centroid) pairs measures the room between the bound's intervals and the exact divergence.
    python scripts/mfma_audit.py synthetic|real [margin_sample]      -> one JSON object on stdout"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["RP_LLOYD_AUDIT"] = "1"
which = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
sample = int(sys.argv[2]) if len(sys.argv) > 2 else 768
say = lambda m: print(m, file=sys.stderr, flush=True)  # noqa: E731
import numpy as np  # noqa: E402

from robopoker_amd import lloyd  # noqa: E402

t0 = time.perf_counter()
if which == "synthetic":
    from robopoker_amd.fixtures import flop_like_points, smooth_metric

    N, K, bins = int(os.environ.get("RP_AUDIT_N", "1286792")), 256, 256
    pts, tri = flop_like_points(N, bins=bins, mass=47, seed=0xF10F), smooth_metric(256, 1)
    layer = lloyd.Layer(K, pts, "sinkhorn", tri, seed=1)
    if os.environ.get("RP_AUDIT_LIBM") == "glibc":  # the audit of the bounds in the glibc-arithmetic pass (rp_kmeans_set_libm keeps them)
        layer.set_libm("glibc")
    layer.init_centroids()
    layer.init_bounds()
    for it in range(32):
        _, _, moved = layer.step()
        say(f"iteration {it}: moved {moved:.5f}")
    layer.lookup()
    prune = layer.prune_stats()
    cents = layer.centroids()[0]
    layer.close()
    idx = np.linspace(0, N - 1, sample).astype(np.int64)
    margins = lloyd.margin_audit(pts[idx], cents, tri)
    out = {"points": "synthetic flop-like histograms (fixtures.flop_like_points, seed 0xF10F)", "N": N, "K": K, "bins": bins,
           "libm": os.environ.get("RP_AUDIT_LIBM", "contract")}
else:
    os.environ["RP_LLOYD_MARGIN_SAMPLE"] = str(sample)
    from robopoker_amd import pretraining

    art = pretraining.run(0, log=say)
    tm = art["flop"].timings
    prune, margins = tm["prune"], tm["margins"]
    out = {"points": "the real flop layer: 1 286 792 isomorphisms projected onto the clustered turn layer (pretraining.run)",
           "N": int(art["flop"].obs.numel()), "K": 256, "bins": int(art["turn"].abstraction.max().item()) + 1}
out.update({"audited_points": prune["audited_points"], "audit_mismatches": prune["audit_mismatches"],
            "passes_audited": "init_bounds (k-means++ shortcut vs the unpruned search) + the final lookup (MFMA prune vs the unpruned "
                              "search), every point",
            "survivors_per_point": prune["survivors"] / max(prune["points"], 1), "margin_sample": margins,
            "wall_s": time.perf_counter() - t0})
print(json.dumps(out), flush=True)

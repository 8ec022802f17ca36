#!/usr/bin/env python3
"""Full-size k-means configurations on ONE MI355X (robopoker_amd.lloyd.bench_full).  usage: full_kmeans.py flop|turn [iters] [N]
RP_FULL_LIBM=glibc: the layer in the kernels' glibc-arithmetic pass (RP_FULL_NO_PRUNE=1: without the filters); RP_FULL_RNG=reference:
Layer::init_centroids' own SmallRng + WeightedIndex<f32> draw."""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from robopoker_amd import lloyd  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "flop"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = int(sys.argv[3]) if len(sys.argv) > 3 else None
out = lloyd.bench_full(which, iters, n, log=lambda m: print(m, file=sys.stderr, flush=True), libm=os.environ.get("RP_FULL_LIBM", "contract"),
                       rng=os.environ.get("RP_FULL_RNG", "counter"))
out["libm"] = os.environ.get("RP_FULL_LIBM", "contract")
out.pop("_centroids", None)
print(json.dumps(out), flush=True)

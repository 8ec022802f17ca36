"""Two NLHE solvers with the same seed, a few GPU-sized steps each, in one process: tables must be equal as key -> Encounter maps
and the counters identical (the traversal's scheduling and the table's row numbering are timing, its results are not)."""
import sys
import numpy as np
import torch  # noqa: F401
from robopoker_amd.nlhe import NlheSolver

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else "composed"
res = []
for rep in range(2):
    s = NlheSolver(cap_log2=25, batch=batch, seed=2026)
    for _ in range(steps):
        s.step(mode)
    past, present, choices, enc = s.export()
    trip = np.stack([past, present.astype(np.uint64), choices], axis=1)
    print("rep", rep, "rows", len(past), "distinct keys", len(np.unique(trip, axis=0)), flush=True)
    order = np.lexsort((choices, present, past))
    res.append((s.counters(), past[order], present[order], choices[order], enc[order]))
    s.close()
a, b = res
print("counters", a[0], b[0])
same_keys = all(np.array_equal(a[i], b[i]) for i in (1, 2, 3))
print("same keys", same_keys)
if same_keys:
    for f in ("visits", "regret", "weight", "payoff"):
        x, y = a[4][f], b[4][f]
        eq = np.array_equal(x.view(np.uint32), y.view(np.uint32))
        print(f, "bitwise equal" if eq else f"DIFFER at {int((x.view(np.uint32) != y.view(np.uint32)).sum())} cells")

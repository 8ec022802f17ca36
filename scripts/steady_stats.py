#!/usr/bin/env python3
"""Per-kernel time of the STEADY-STATE steps of an NLHE bench run from a rocprofv3 --kernel-trace CSV: dispatches before the
(skip+1)-th k_nl_roots launch (the first steps insert every infoset of the hash-encoder world) and after the last big-batch step
are left out.  usage: steady_stats.py <kernel_trace.csv> <skip_steps> <n_steps> <out.txt> <title>"""
import csv
import sys
from collections import defaultdict

path, skip, nsteps, out, title = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
roots = [int(r["Start_Timestamp"]) for r in rows if "k_nl_roots" in r["Kernel_Name"] or "k_nl_tree" in r["Kernel_Name"]]  # the first launch of a step
t0, t1 = roots[skip], roots[skip + nsteps] if len(roots) > skip + nsteps else int(rows[-1]["End_Timestamp"]) + 1
tot, cnt = defaultdict(float), defaultdict(int)
for r in rows:
    s = int(r["Start_Timestamp"])
    if t0 <= s < t1:
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += (int(r["End_Timestamp"]) - s) / 1e3
        cnt[k] += 1
wall = (t1 - t0) / 1e3
busy = sum(tot.values())
with open(out, "w") as f:
    f.write(f"# {title}\n# rocprofv3 --kernel-trace, steps {skip}..{skip + nsteps - 1} only; per STEP averages (microseconds)\n")
    f.write(f"# wall per step {wall / nsteps:.1f} us, kernel time per step {busy / nsteps:.1f} us\n")
    f.write(f"{'launches/step':>14} {'us/step':>12} {'avg_us':>10} {'pct':>6}  kernel\n")
    for k in sorted(tot, key=lambda k: -tot[k]):
        f.write(f"{cnt[k] / nsteps:14.1f} {tot[k] / nsteps:12.1f} {tot[k] / cnt[k]:10.2f} {100 * tot[k] / busy:6.2f}  {k[:100]}\n")
print(open(out).read())

# per-launch durations of the level kernels of ONE steady step (the last one inside the window)
if len(sys.argv) > 6:
    last = roots[skip + nsteps - 1]
    nxt = t1
    print("# per level (us): expand | children")
    ex = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if last <= int(r["Start_Timestamp"]) < nxt and "k_nl_expand" in r["Kernel_Name"]]
    ch = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if last <= int(r["Start_Timestamp"]) < nxt and "k_nl_children" in r["Kernel_Name"]]
    for l, (a, b) in enumerate(zip(ex, ch)):
        print(f"{l:3d} {a:9.1f} {b:9.1f}")

#!/bin/bash
# Round 6: per-LAUNCH durations of the Elkan iterations' kernels over the 32 iterations of the full flop layer (reference arithmetic):
# which iterations k_elkan_step's 3.6 s are spent in.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO RP_FULL_LIBM=glibc RP_FULL_RNG=reference RP_FIXTURE_CACHE=/tmp
rm -rf $OUT/kt
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python $REPO/scripts/full_kmeans.py flop 32 > $OUT/line.json 2> $OUT/kt.err
python - <<PY
import csv, collections, json, glob
f = glob.glob("$OUT/kt/**/kt_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rp::lm_glibc::", "").replace("rp::lm_contract::", "")
    per[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
out = {}
for k in ("k_elkan_step", "k_pairwise", "k_refresh_pairs", "k_refresh_interval<32u>", "k_drift_block", "k_self_block", "k_recompute", "k_bounds_update"):
    if k in per:
        out[k] = [round(x, 2) for x in per[k]]
        print(k, len(per[k]), "launches, ms:", out[k])
json.dump({"note": "per-launch kernel durations (ms) in launch order, full flop layer, glibc arithmetic + reference draw, rocprofv3 --kernel-trace", "kernels": out}, open("$OUT/r06_lloyd_per_launch_ms.json", "w"), indent=1)
PY
rm -rf $OUT/kt

#!/bin/bash
# Round 5, first GPU call: what the flop layer's kernels do with their cycles.
#   (1) rocprofv3 --kernel-trace --stats of the FULL flop layer (scripts/full_kmeans.py flop 32)
#   (2) two SQ counter groups (own runs, kernel-trace only) of a 65 536-point slice of the same layer
# every pass under its own timeout; reductions land in gpurun_out/r5prof/ (copied to profiles/r05_lloyd_* by hand).
# usage: gpurun --timeout 900 -- bash scripts/r5_lloyd_prof.sh [tag] [slice N] [slice iters]
set -u
TAG=${1:-r05}
SLICE=${2:-65536}
SIT=${3:-6}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
FULL="python $REPO/scripts/full_kmeans.py flop 32"
PART="python $REPO/scripts/full_kmeans.py flop $SIT $SLICE"
echo "== full layer, kernel trace"; date +%T
rm -rf $OUT/kt
timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $FULL > $OUT/${TAG}_lloyd_full_line.json 2> $OUT/kt.err
python $REPO/scripts/rocpd_summary.py "$(ls $OUT/kt/*.db 2>/dev/null | head -1)" $OUT/${TAG}_lloyd_full_kernel_stats.txt "$FULL" | head -30
rm -rf $OUT/kt
echo "== slice, SQ group 1"; date +%T
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/sq4 $OUT/sq5
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/sq1 -o pmc -- $PART > $OUT/sq1.log 2>&1
echo "== slice, SQ group 2"; date +%T
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD \
  --kernel-trace --output-format csv -d $OUT/sq2 -o pmc -- $PART > $OUT/sq2.log 2>&1
echo "== slice, SQ group 3"; date +%T
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/sq3 -o pmc -- $PART > $OUT/sq3.log 2>&1
# (TA_* / TCP_* groups: rocprofv3 aborts on them on this image — tried in the first call of the round, logs empty)
python $REPO/scripts/sq_reduce.py $OUT/${TAG}_lloyd_sq_counters.json "$PART" $OUT/sq1/pmc_counter_collection.csv $OUT/sq2/pmc_counter_collection.csv $OUT/sq3/pmc_counter_collection.csv | cut -c1-1200
# kernel durations of the slice from the first pass' trace (for cycles -> time)
python - <<PY
import csv, collections, json
try:
    rows = list(csv.DictReader(open("$OUT/sq1/pmc_kernel_trace.csv")))
except Exception as e:
    print("no trace", e); rows = []
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {k: {"calls": v[0], "total_us": v[1]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
json.dump(out, open("$OUT/${TAG}_lloyd_slice_kernel_us.json", "w"), indent=1)
for k, v in list(out.items())[:16]:
    print(f"{v['total_us']:14.1f} us {v['calls']:6d}  {k[:110]}")
PY
tail -3 $OUT/sq1.log $OUT/sq2.log $OUT/sq3.log | cut -c1-300
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/sq4 $OUT/sq5
date +%T

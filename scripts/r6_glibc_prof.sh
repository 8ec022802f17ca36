#!/bin/bash
# Round 6: what the lm_glibc kernels (the reference's own arithmetic) do with their cycles.
#   (0) scripts/ubench/valu_rate: issue cost of the f64 instructions glibc's expf is made of
#   (1) rocprofv3 --kernel-trace --stats of the FULL flop layer, libm = glibc, rng = reference
#   (2) three SQ counter groups (own runs, kernel-trace only) of a 65 536-point slice of the same layer
# usage: gpurun --timeout 1200 -- bash scripts/r6_glibc_prof.sh [tag] [slice N] [slice iters]
set -u
TAG=${1:-r06_before}
SLICE=${2:-65536}
SIT=${3:-6}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
export RP_FULL_LIBM=glibc RP_FULL_RNG=reference
if [ -x $REPO/scripts/ubench/valu_rate ]; then timeout 60 $REPO/scripts/ubench/valu_rate > $OUT/${TAG}_valu_issue_rates.txt 2>&1; cat $OUT/${TAG}_valu_issue_rates.txt; fi
FULL="python $REPO/scripts/full_kmeans.py flop 32"
PART="python $REPO/scripts/full_kmeans.py flop $SIT $SLICE"
echo "== full layer (glibc, reference rng), kernel trace"; date +%T
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $FULL > $OUT/${TAG}_lloyd_glibc_full_line.json 2> $OUT/kt.err
python $REPO/scripts/rocpd_summary.py "$(ls $OUT/kt/*.db 2>/dev/null | head -1)" $OUT/${TAG}_lloyd_glibc_full_kernel_stats.txt "RP_FULL_LIBM=glibc RP_FULL_RNG=reference $FULL" | head -30
rm -rf $OUT/kt
echo "== slice, SQ group 1"; date +%T
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/sq1 -o pmc -- $PART > $OUT/sq1.log 2>&1
echo "== slice, SQ group 2"; date +%T
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD \
  --kernel-trace --output-format csv -d $OUT/sq2 -o pmc -- $PART > $OUT/sq2.log 2>&1
echo "== slice, SQ group 3"; date +%T
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/sq3 -o pmc -- $PART > $OUT/sq3.log 2>&1
python $REPO/scripts/sq_reduce.py $OUT/${TAG}_lloyd_glibc_sq_counters.json "RP_FULL_LIBM=glibc RP_FULL_RNG=reference $PART" $OUT/sq1/pmc_counter_collection.csv $OUT/sq2/pmc_counter_collection.csv $OUT/sq3/pmc_counter_collection.csv | cut -c1-600
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
json.dump(out, open("$OUT/${TAG}_lloyd_glibc_slice_kernel_us.json", "w"), indent=1)
for k, v in list(out.items())[:16]:
    print(f"{v['total_us']:14.1f} us {v['calls']:6d}  {k[:110]}")
PY
python $REPO/scripts/valu_ceiling.py $OUT/${TAG}_lloyd_glibc_sq_counters.json $OUT/${TAG}_lloyd_glibc_slice_kernel_us.json $OUT/${TAG}_lloyd_glibc_valu_ceiling.json | head -20
tail -3 $OUT/sq1.log $OUT/sq2.log $OUT/sq3.log | cut -c1-300
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3
date +%T

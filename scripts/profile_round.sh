#!/bin/bash
# Collect the judged profile artifacts on the GPU box: kernel-trace stats of the default bench command, and the
# HBM-traffic PMC passes (each counter in its own run, no other trace domains).  Results land in gpurun_out/prof/.
# usage: scripts/profile_round.sh <tag>      (run through gpurun from the repo root)
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-extras --steps 40 --warmup 5"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
python $REPO/scripts/rocpd_summary.py $(ls $OUT/kt/*.db | head -1) $OUT/${TAG}_bench_kernel_stats.txt "$BENCH" > /dev/null
grep -o '{"metric.*' $OUT/kt.log > $OUT/${TAG}_bench_line_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o pmc -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o pmc -- $BENCH > $OUT/write.log 2>&1
python $REPO/scripts/pmc_traffic.py $OUT/fetch/pmc_counter_collection.csv $OUT/write/pmc_counter_collection.csv \
    $OUT/${TAG}_mccfr_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: $BENCH; FETCH_SIZE doubled (gfx950), KiB -> bytes" 1048576 composed
# the other two kernel families: k-means (flop-layer slice) and the sparse profile
rocprofv3 --kernel-trace --stats -d $OUT/kl -o kl -- python $REPO/scripts/quick_lloyd.py 8192 > $OUT/kl.log 2>&1
python $REPO/scripts/rocpd_summary.py $(ls $OUT/kl/*.db | head -1) $OUT/${TAG}_lloyd_kernel_stats.txt "python scripts/quick_lloyd.py 8192 (flop-layer slice: N=8192, K=256, bins=256, init_bounds + 2 Elkan iterations)" > /dev/null
SP="python $REPO/bench.py --workload nlhe-synth --cpu-seconds 0 --steps 40 --warmup 5"
rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks -- $SP > $OUT/ks.log 2>&1
python $REPO/scripts/rocpd_summary.py $(ls $OUT/ks/*.db | head -1) $OUT/${TAG}_sparse_kernel_stats.txt "$SP" > /dev/null
# the abstraction inputs at full size (isomorphism iterator, river equity, projections), and the VALU counters of the
# river-equity kernel (own pass, kernel-trace only)
DQ="python $REPO/scripts/quick_deuce.py"
PYTHONPATH=$REPO rocprofv3 --kernel-trace --stats -d $OUT/kd -o kd -- $DQ > $OUT/kd.log 2>&1
python $REPO/scripts/rocpd_summary.py $(ls $OUT/kd/*.db | head -1) $OUT/${TAG}_deuce_kernel_stats.txt "PYTHONPATH=. python scripts/quick_deuce.py (all four isomorphism lists; 123 156 254 river equities x2; turn and flop projections x2)" > /dev/null
PYTHONPATH=$REPO rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/pd -o pmc -- $DQ > $OUT/pd.log 2>&1
python - <<PY > $OUT/${TAG}_deuce_valu_counters.txt
import csv, collections
rows = list(csv.DictReader(open("$OUT/pd/pmc_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if "rp::" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
print("# rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY (sums over dispatches) of: PYTHONPATH=. python scripts/quick_deuce.py")
for k, v in agg.items():
    print(k, calls[k], "dispatches", {c: f"{x:.4e}" for c, x in sorted(v.items())})
PY
cat $OUT/${TAG}_bench_kernel_stats.txt

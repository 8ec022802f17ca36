#!/bin/bash
# Round 6: fresh NLHE measurements — (1) kernel trace at the reference's batch of 128, (2) one SQ pass of the 262 144-tree step
# (k_nl_expand's waits), (3) the bench line.   usage: gpurun --timeout 900 -- bash scripts/r6_nlhe_prof.sh [tag]
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6nlhe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
export RP_BENCH_NO_REF=1
echo "== batch 128 trace"; date +%T
CMD="python $REPO/bench.py --workload nlhe --nlhe-batch 128 --steps 40 --warmup 10 --cpu-seconds 0"
rm -rf $OUT/nl
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/nl -o nl -- $CMD > $OUT/nl.log 2>&1
python $REPO/scripts/steady_stats.py $OUT/nl/nl_kernel_trace.csv 10 40 $OUT/${TAG}_nlhe_kernel_stats_b128.txt "$CMD (the timed steps)" | head -40
rm -rf $OUT/nl
echo "== batch 262144 SQ pass"; date +%T
CMD2="python $REPO/bench.py --workload nlhe --nlhe-batch 262144 --steps 3 --warmup 2 --cpu-seconds 0"
rm -rf $OUT/sq
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
  --kernel-trace --output-format csv -d $OUT/sq -o pmc -- $CMD2 > $OUT/sq.log 2>&1
python $REPO/scripts/sq_reduce.py $OUT/${TAG}_nlhe_sq_counters.json "$CMD2" $OUT/sq/pmc_counter_collection.csv | grep -E "k_nl_expand|k_nl_children" | cut -c1-500
rm -rf $OUT/sq
echo "== bench line"; date +%T
unset RP_BENCH_NO_REF
cd $REPO
timeout 250 python bench.py --workload nlhe --cpu-seconds 6 --steps 8 --warmup 4 > $OUT/${TAG}_nlhe_bench_line.json 2> $OUT/nlhe.err; head -c 600 $OUT/${TAG}_nlhe_bench_line.json; echo
date +%T

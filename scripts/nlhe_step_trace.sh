#!/bin/bash
# kernel trace of a small-batch NLHE step (default 128 trees, the reference's batch): per-step kernel times -> gpurun_out/b128/
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/b128
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
CMD="python $REPO/bench.py --workload nlhe --nlhe-batch ${1:-128} --steps 40 --warmup 10 --cpu-seconds 0"
rm -rf $OUT/nl
RP_BENCH_NO_REF=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/nl -o nl -- $CMD > $OUT/nl.log 2>&1
python $REPO/scripts/steady_stats.py $OUT/nl/nl_kernel_trace.csv 10 40 $OUT/r04_nlhe_kernel_stats_b${1:-128}.txt "$CMD (the timed steps)" | head -40
rm -rf $OUT/nl

#!/bin/bash
# steady-state kernel times of the NLHE step (level-synchronous traversal) at a GPU-sized batch and at the reference's 128
set -u
TAG=${1:-r03}
BIG=${2:-262144}
EXTRA=${3:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
CMD="python $REPO/bench.py --workload nlhe --nlhe-batch $BIG --steps 8 --warmup 4 --cpu-seconds 0 $EXTRA"
rm -rf $OUT/nl
rocprofv3 --kernel-trace --output-format csv -d $OUT/nl -o nl -- $CMD > $OUT/nl.log 2>&1
grep -o '{"metric.*' $OUT/nl.log > $OUT/${TAG}_nlhe_bench_line_under_rocprof.json
python $REPO/scripts/steady_stats.py $OUT/nl/nl_kernel_trace.csv 4 8 $OUT/${TAG}_nlhe_kernel_stats_b$BIG.txt "$CMD (the $BIG-tree steps)" levels | tail -26
# the reference's batch of 128: bench.py runs it after the big one (3 warm-up + 20 timed steps)
python $REPO/scripts/steady_stats.py $OUT/nl/nl_kernel_trace.csv 15 20 $OUT/${TAG}_nlhe_kernel_stats_b128.txt "$CMD (the 128-tree steps)" | head -30
rm -rf $OUT/nl

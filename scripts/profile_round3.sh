#!/bin/bash
# Round-3 profile artifacts (run through gpurun from the repo root; the judged copies are committed under profiles/).
# Every rocprofv3 run is wrapped in its own timeout (a hung counter pass once ate 40 GPU-minutes).
#   1 kernel-trace stats of the MCCFR timed loop (bench.py --no-extras)   -> r03_bench_kernel_stats.txt (+ the line under rocprof)
#   2 steady-state kernel times of the NLHE step                          -> r03_nlhe_kernel_stats_b262144.txt / _b128.txt
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
BENCH="python $REPO/bench.py --no-extras --steps 40 --warmup 5"
rm -rf $OUT/kt
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
python $REPO/scripts/rocpd_summary.py $(ls $OUT/kt/*.db | head -1) $OUT/${TAG}_bench_kernel_stats.txt "$BENCH" | head -12
grep -o '{"metric.*' $OUT/kt.log > $OUT/${TAG}_bench_line_under_rocprof.json
rm -rf $OUT/kt
CMD="python $REPO/bench.py --workload nlhe --steps 8 --warmup 4 --cpu-seconds 0"
rm -rf $OUT/nl
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/nl -o nl -- $CMD > $OUT/nl.log 2>&1
grep -o '{"metric.*' $OUT/nl.log > $OUT/${TAG}_nlhe_bench_line_under_rocprof.json
# the run: 4 warm-up + 8 timed + 8 profiled steps of 262 144 trees, then 3 + 20 steps of 128 trees
python $REPO/scripts/steady_stats.py $OUT/nl/nl_kernel_trace.csv 4 8 $OUT/${TAG}_nlhe_kernel_stats_b262144.txt "$CMD (the timed 262144-tree steps)" levels | head -24
python $REPO/scripts/steady_stats.py $OUT/nl/nl_kernel_trace.csv 23 20 $OUT/${TAG}_nlhe_kernel_stats_b128.txt "$CMD (the timed 128-tree steps)" | head -14
rm -rf $OUT/nl

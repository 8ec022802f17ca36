#!/bin/bash
# Round 6: the whole -m gpu suite on the shipped tree, then the default bench line (compact) with its detail file.
# usage: gpurun --timeout 2400 -- bash scripts/r6_final.sh [tag]
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
echo "== gpu tests"; date +%T
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 --timeout-method=thread --durations=8 -p no:cacheprovider > $OUT/${TAG}_gpu_tests.log 2>&1; tail -14 $OUT/${TAG}_gpu_tests.log
echo "== default bench"; date +%T
timeout 500 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err; wc -c $OUT/${TAG}_bench_line.json; tail -c 2200 $OUT/${TAG}_bench_line.json; echo; tail -3 $OUT/bench.err
cp gpurun_out/bench_detail.json $OUT/${TAG}_bench_detail.json 2>/dev/null
date +%T

#!/bin/bash
# round 4: the -m gpu suite (a hung kernel ends the run at that test, by name), then the two bench lines
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4
mkdir -p $OUT
cd $REPO
timeout 700 python -m pytest tests -m gpu -q -x --timeout 90 --timeout-method=thread --durations=12 -p no:cacheprovider > $OUT/gpu_tests.log 2>&1; tail -30 $OUT/gpu_tests.log
timeout 120 python bench.py --no-extras --steps 40 --warmup 5 > $OUT/bench_noextras.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench_noextras.json
timeout 150 python bench.py --workload nlhe --cpu-seconds 0 --steps 8 --warmup 4 > $OUT/nlhe.json 2> $OUT/nlhe.err; cut -c1-300 $OUT/nlhe.json

#!/bin/bash
# usage: gpurun -- bash scripts/r5_quick.sh "<pytest -k expr>"   : selected lloyd tests, then the full flop layer's phase times
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5quick
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
date +%T
timeout 400 python -m pytest tests/test_gpu_lloyd.py -m gpu -q -x -k "$1" -p no:cacheprovider 2>&1 | tail -4
date +%T
timeout 200 python scripts/full_kmeans.py flop 32 > $OUT/r05_full_flop_kmeans.json 2> $OUT/full.err
python - <<PY
import json
d=json.load(open("$OUT/r05_full_flop_kmeans.json"))
for k in ['create_s','kmeanspp_s','init_bounds_s','elkan_total_s','lookup_s','end_to_end_s']: print(k, round(d[k],3))
print(d['kernels_ms']); print(d['mfma_bound']['sample_mismatches'], d['distances_total'], d['sinkhorn_iterations_total'])
PY
date +%T

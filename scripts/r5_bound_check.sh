#!/bin/bash
# Round 5: the MFMA bound kernel after a change — its tests, then the full flop layer (phase times + prune stats).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5bound
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
echo "== tests"; date +%T
timeout 300 python -m pytest tests/test_gpu_lloyd.py -m gpu -q -x -k "${RP_TESTS:-mfma_bound_handles or flop_config_slice}" -p no:cacheprovider 2>&1 | tail -5
echo "== full layer"; date +%T
timeout 200 python scripts/full_kmeans.py flop ${1:-32} > $OUT/r05_full_flop_kmeans.json 2> $OUT/full.err
python - <<PY
import json
d=json.load(open("$OUT/r05_full_flop_kmeans.json"))
for k in ['create_s','kmeanspp_s','init_bounds_s','elkan_total_s','lookup_s','end_to_end_s']: print(k, round(d[k],3))
print(d['kernels_ms']); print(d['mfma_bound']); print(d.get('roofline_mfma')); print('col iters per column', d['mfma_bound']['column_iterations']/d['mfma_bound']['candidates'], 'block iters per block', d['mfma_bound']['block_iterations']*16/d['mfma_bound']['candidates'])
PY
date +%T

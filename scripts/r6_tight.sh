#!/bin/bash
# Round 6: the MFMA bound's c-transform pass at several periods (RP_SB_TIGHT = wavefront iterations between two evaluations; 0 = never).
set -u
TAG=${1:-r06n}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6dual
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
timeout 900 python -m pytest tests/test_gpu_lloyd.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "mfma or pruned" 2>&1 | tail -3
for D in ${TIGHTS:-0 2 4 8}; do
RP_SB_TIGHT=$D RP_FULL_LIBM=glibc RP_FULL_RNG=reference timeout 300 python scripts/full_kmeans.py flop ${TIGHT_ITERS:-2} > $OUT/${TAG}_full_flop_tight$D.json 2> $OUT/fullt$D.err
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_full_flop_tight$D.json"))
m=d['mfma_bound']
print("TIGHT=$D", {k: round(d[k],4) for k in ['init_bounds_s','lookup_s','end_to_end_s']}, 'mfma_bound_ms', round(d['kernels_ms']['mfma_bound']['total_ms']), 'neighbor_ms', round(d['kernels_ms']['neighbor']['total_ms']),
      {k:m[k] for k in ['survivors','block_iterations','cost_passes','column_iterations']})
PY
done

#!/bin/bash
# Round 6: the c-transform pass triggered by the slots' ages (RP_SB_TRIG columns due their first evaluation) on top of the period (RP_SB_TIGHT).
set -u
TAG=${1:-r06q}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6dual
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
for CFG in ${CFGS:-16:0 16:3 16:5 16:8 32:4 32:6}; do
T=${CFG%%:*}; G=${CFG##*:}
RP_SB_TIGHT=$T RP_SB_TRIG=$G RP_FULL_LIBM=glibc RP_FULL_RNG=reference timeout 300 python scripts/full_kmeans.py flop 2 > $OUT/${TAG}_tight${T}_trig$G.json 2> $OUT/fullq.err
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_tight${T}_trig$G.json"))
m=d['mfma_bound']
print("TIGHT=$T TRIG=$G", {k: round(d[k],4) for k in ['lookup_s']}, 'mfma_bound_ms', round(d['kernels_ms']['mfma_bound']['total_ms']),
      {k:m[k] for k in ['survivors','block_iterations','column_iterations']})
PY
done

#!/bin/bash
# Round 5: the k-means++ interval filter on hardware — its two tests, the full-size audit against the unfiltered rounds, the full flop layer.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5kpp
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
echo "== tests"; date +%T
timeout 300 python -m pytest tests/test_gpu_lloyd.py -m gpu -q -x -k "kmeanspp or groupings or mfma_bound" -p no:cacheprovider 2>&1 | tail -5
echo "== audit"; date +%T
timeout 200 python scripts/kpp_audit.py > $OUT/r05_kpp_audit.json 2> $OUT/audit.err; cat $OUT/r05_kpp_audit.json; tail -2 $OUT/audit.err
echo "== full layer"; date +%T
timeout 200 python scripts/full_kmeans.py flop 32 > $OUT/r05_full_flop_kmeans.json 2> $OUT/full.err
python - <<PY
import json
d=json.load(open("$OUT/r05_full_flop_kmeans.json"))
for k in ['create_s','kmeanspp_s','init_bounds_s','elkan_total_s','lookup_s','end_to_end_s']: print(k, round(d[k],3))
print(d['kernels_ms']); print({k:v for k,v in d['mfma_bound'].items() if 'kpp' in k})
PY
date +%T

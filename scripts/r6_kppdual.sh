#!/bin/bash
# Round 6: the k-means++ filter's dual exit with the loose pair (RP_KPP_DUAL=1) and the c-transform pair (2): parity tests, then the full flop layer.
set -u
TAG=${1:-r06i}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6dual
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
timeout 900 python -m pytest tests/test_gpu_lloyd.py tests/test_gpu_z_glibc_mode.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "kmeanspp or kpp" 2>&1 | tail -4
for D in 1 2; do
RP_KPP_DUAL=$D RP_FULL_LIBM=glibc RP_FULL_RNG=reference timeout 300 python scripts/full_kmeans.py flop 32 > $OUT/${TAG}_full_flop_kppdual$D.json 2> $OUT/fullk$D.err
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_full_flop_kppdual$D.json"))
print("KPP_DUAL=$D", {k: round(d[k],4) for k in ['kmeanspp_s','init_bounds_s','elkan_total_s','lookup_s','end_to_end_s','rms']})
print({k:(round(v['total_ms']),v['launches']) for k,v in d['kernels_ms'].items()})
print({k:v for k,v in d['mfma_bound'].items() if k.startswith('kpp')})
PY
done

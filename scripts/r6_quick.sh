#!/bin/bash
# Round 6 iteration loop: the glibc-mode and lloyd parity tests, then the full flop layer in the reference's arithmetic (and, with
# "both", in the contract's), timings into gpurun_out/r6quick/.   usage: gpurun --timeout 900 -- bash scripts/r6_quick.sh [tag] [both]
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6quick
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
date +%T
timeout 500 python -m pytest tests/test_gpu_z_glibc_mode.py tests/test_gpu_lloyd.py -m gpu -q -x -p no:cacheprovider --timeout 300 2>&1 | tail -6
date +%T
RP_FULL_LIBM=glibc RP_FULL_RNG=reference timeout 300 python scripts/full_kmeans.py flop 32 > $OUT/${TAG}_full_flop_reference_arithmetic.json 2> $OUT/full.err
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_full_flop_reference_arithmetic.json"))
for k in ['create_s','kmeanspp_s','init_bounds_s','elkan_total_s','lookup_s','end_to_end_s','rms']: print(k, round(d[k],4))
print({k:(round(v['total_ms']),v['launches']) for k,v in d['kernels_ms'].items()})
PY
if [ "${2:-}" = "both" ]; then
RP_FULL_RNG=reference timeout 300 python scripts/full_kmeans.py flop 32 > $OUT/${TAG}_full_flop_contract_same_draw.json 2>> $OUT/full.err
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_full_flop_contract_same_draw.json"))
for k in ['create_s','kmeanspp_s','init_bounds_s','elkan_total_s','lookup_s','end_to_end_s','rms']: print(k, round(d[k],4))
print({k:(round(v['total_ms']),v['launches']) for k,v in d['kernels_ms'].items()})
PY
fi
date +%T

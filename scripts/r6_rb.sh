#!/bin/bash
# Round 6: the interval-decided refresh on the GPU: parity tests, the full-size audit in both arithmetics, the layer's timing.
# usage: gpurun --timeout 1500 -- bash scripts/r6_rb.sh [tag]
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6rb
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
date +%T
timeout 600 python -m pytest tests/test_gpu_lloyd.py tests/test_gpu_z_glibc_mode.py tests/test_gpu_sharded.py -m gpu -q -x -p no:cacheprovider --timeout 300 2>&1 | tail -6
date +%T
timeout 400 python scripts/r6_refresh_audit.py 32 > $OUT/${TAG}_refresh_audit_contract.json 2> $OUT/audit_contract.err; tail -3 $OUT/audit_contract.err | cut -c1-300
python -c "
import json;d=json.load(open('$OUT/${TAG}_refresh_audit_contract.json'));print({k:d[k] for k in d if k not in ('per_iteration','workload')})"
date +%T
RP_FULL_LIBM=glibc RP_FULL_RNG=reference timeout 400 python scripts/r6_refresh_audit.py 32 > $OUT/${TAG}_refresh_audit_glibc.json 2> $OUT/audit_glibc.err; tail -3 $OUT/audit_glibc.err | cut -c1-300
python -c "
import json;d=json.load(open('$OUT/${TAG}_refresh_audit_glibc.json'));print({k:d[k] for k in d if k not in ('per_iteration','workload')})"
date +%T
RP_FULL_LIBM=glibc RP_FULL_RNG=reference timeout 300 python scripts/full_kmeans.py flop 32 > $OUT/${TAG}_full_flop_reference_arithmetic.json 2> $OUT/full.err
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_full_flop_reference_arithmetic.json"))
for k in ['create_s','kmeanspp_s','init_bounds_s','elkan_total_s','lookup_s','end_to_end_s','rms']: print(k, round(d[k],4))
print({k:(round(v['total_ms']),v['launches']) for k,v in d['kernels_ms'].items()})
PY
date +%T

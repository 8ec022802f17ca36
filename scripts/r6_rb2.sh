#!/bin/bash
# Round 6: quick loop for the interval-decided refresh: its parity test, the full-size audit (contract) and the layer's timing (glibc).
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6rb
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
timeout 300 python -m pytest tests/test_gpu_lloyd.py tests/test_gpu_sharded.py tests/test_golden.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "interval_decided or elkan_iterations or two_shards or k256" 2>&1 | tail -4
timeout 400 python scripts/r6_refresh_audit.py 32 > $OUT/${TAG}_refresh_audit_contract.json 2> $OUT/audit_contract.err; tail -1 $OUT/audit_contract.err | cut -c1-300
python -c "
import json;d=json.load(open('$OUT/${TAG}_refresh_audit_contract.json'));print({k:d[k] for k in d if k not in ('per_iteration','workload')})"
RP_FULL_LIBM=glibc RP_FULL_RNG=reference timeout 300 python scripts/full_kmeans.py flop 32 > $OUT/${TAG}_full_flop_reference_arithmetic.json 2> $OUT/full.err
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_full_flop_reference_arithmetic.json"))
for k in ['create_s','kmeanspp_s','init_bounds_s','elkan_total_s','lookup_s','end_to_end_s','rms']: print(k, round(d[k],4))
print({k:(round(v['total_ms']),v['launches']) for k,v in d['kernels_ms'].items()})
print(d.get('refresh_interval'))
PY

#!/bin/bash
# Round 6: the MFMA bound after a change: its parity tests, the layer's timing, the full-size audit (RP_LLOYD_AUDIT) in glibc's arithmetic.
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6sb
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
timeout 600 python -m pytest tests/test_gpu_lloyd.py tests/test_gpu_z_glibc_mode.py tests/test_golden.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "mfma or pruned or k256 or set_prune or layer_shape or kmeans_golden" 2>&1 | tail -4
RP_FULL_LIBM=glibc RP_FULL_RNG=reference timeout 300 python scripts/full_kmeans.py flop 32 > $OUT/${TAG}_full_flop_reference_arithmetic.json 2> $OUT/full.err
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_full_flop_reference_arithmetic.json"))
for k in ['kmeanspp_s','init_bounds_s','elkan_total_s','lookup_s','end_to_end_s','rms']: print(k, round(d[k],4))
print({k:(round(v['total_ms']),v['launches']) for k,v in d['kernels_ms'].items()})
print(d['mfma_bound']); print(d['roofline_mfma'])
PY
if [ "${2:-}" = "audit" ]; then
RP_AUDIT_LIBM=glibc timeout 900 python scripts/mfma_audit.py synthetic 256 > $OUT/${TAG}_glibc_audit.json 2> $OUT/audit.err; cut -c1-900 $OUT/${TAG}_glibc_audit.json; tail -2 $OUT/audit.err
fi

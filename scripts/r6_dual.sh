#!/bin/bash
# Round 6: the MFMA bound's dual exit (RP_SB_LIP=2, the default) against the Lipschitz rule alone (RP_SB_LIP=1): parity tests, the full
# flop layer in the reference's arithmetic both ways, optionally the full-size audit.   usage: gpurun -- bash scripts/r6_dual.sh TAG [audit]
set -u
TAG=${1:-r06g}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6dual
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
timeout 900 python -m pytest tests/test_gpu_lloyd.py tests/test_gpu_z_glibc_mode.py tests/test_golden.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "${DUAL_K:-mfma or pruned or k256 or set_prune or layer_shape or kmeans_golden or kmeanspp or kpp}" 2>&1 | tail -6
for LIP in ${DUAL_LIPS:-1 2}; do
RP_SB_LIP=$LIP RP_FULL_LIBM=glibc RP_FULL_RNG=reference timeout 300 python scripts/full_kmeans.py flop 32 > $OUT/${TAG}_full_flop_lip$LIP.json 2> $OUT/full$LIP.err
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_full_flop_lip$LIP.json"))
print("LIP=$LIP", {k: round(d[k],4) for k in ['kmeanspp_s','init_bounds_s','elkan_total_s','lookup_s','end_to_end_s','rms']})
print({k:(round(v['total_ms']),v['launches']) for k,v in d['kernels_ms'].items()})
print(d['mfma_bound']); print(d['roofline_mfma']); print(d.get('prune'))
PY
done
if [ "${2:-}" = "audit" ]; then
RP_AUDIT_LIBM=glibc timeout 900 python scripts/mfma_audit.py synthetic 256 > $OUT/${TAG}_glibc_audit.json 2> $OUT/audit.err; cut -c1-1200 $OUT/${TAG}_glibc_audit.json; tail -2 $OUT/audit.err
fi

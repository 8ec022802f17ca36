#!/bin/bash
# Round 6 experiments on the MFMA bound's column queue: each argument is an environment assignment (RP_SB_ORDER=1, RP_SB_TAIL=0, ...)
# for one full flop layer in the reference's arithmetic.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6sb
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
for kv in "$@"; do
env $kv RP_FULL_LIBM=glibc RP_FULL_RNG=reference timeout 300 python scripts/full_kmeans.py flop 32 > $OUT/r06_${kv}_full_flop.json 2> $OUT/full.err
python - <<PY
import json
d=json.load(open("$OUT/r06_${kv}_full_flop.json"))
print("$kv")
for k in ['kmeanspp_s','init_bounds_s','elkan_total_s','lookup_s','end_to_end_s','rms']: print(k, round(d[k],4))
m=d['mfma_bound']; print('mfma_ms',round(d['kernels_ms']['mfma_bound']['total_ms']), 'blk_it',m['block_iterations'],'cost',m['cost_passes'],'col_it',m['column_iterations'], 'useful', m['column_iterations']/16/m['block_iterations'], 'surv', m['survivors'])
PY
done

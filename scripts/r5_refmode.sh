#!/bin/bash
# Round 5: the reference-seed k-means++ pick after its rewrite (tests + the full layer in the reference's arithmetic), and the NLHE bench line
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5ref
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
date +%T
timeout 300 python -m pytest tests/test_gpu_lloyd.py tests/test_reference_seed.py tests/test_gpu_z_glibc_mode.py -m gpu -q -x -k "reference_seed or reference" -p no:cacheprovider 2>&1 | tail -4
date +%T
RP_FULL_LIBM=glibc RP_FULL_RNG=reference timeout 300 python scripts/full_kmeans.py flop 32 > $OUT/r05_full_flop_reference_arithmetic.json 2> $OUT/full.err
python - <<PY
import json
d=json.load(open("$OUT/r05_full_flop_reference_arithmetic.json"))
for k in ['create_s','kmeanspp_s','init_bounds_s','elkan_total_s','lookup_s','end_to_end_s']: print(k, round(d[k],3))
print(d['kernels_ms'])
PY
date +%T
timeout 250 python bench.py --workload nlhe --cpu-seconds 8 --steps 8 --warmup 4 > $OUT/r05_nlhe_bench_line.json 2> $OUT/nlhe.err; head -c 400 $OUT/r05_nlhe_bench_line.json; echo; tail -2 $OUT/nlhe.err
date +%T

#!/bin/bash
# Round 6: the reference's batch (128 trees) after a change: the NLHE / sparse GPU tests, the kernel trace of the step, the bench leg.
# usage: gpurun --timeout 1500 -- bash scripts/r6_nlhe_b128.sh <tag> [notest]
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6b128
mkdir -p $OUT
cd $REPO
if [ "${2:-}" != "notest" ]; then
timeout 900 python -m pytest tests/test_gpu_nlmc.py tests/test_gpu_sparse.py tests/test_gpu_nlhe.py -m gpu -q -x -p no:cacheprovider --timeout 600 2>&1 | tail -4
fi
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
CMD="python $REPO/bench.py --workload nlhe --nlhe-batch 128 --steps 40 --warmup 10 --cpu-seconds 0"
rm -rf $OUT/nl
RP_BENCH_NO_REF=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/nl -o nl -- $CMD > $OUT/nl.log 2>&1
python $REPO/scripts/steady_stats.py $OUT/nl/nl_kernel_trace.csv 10 40 $OUT/${TAG}_nlhe_kernel_stats_b128.txt "$CMD (the timed steps)" | head -30
rm -rf $OUT/nl
cd $REPO
for i in 1 2; do
timeout 200 python bench.py --workload nlhe --nlhe-batch 128 --steps 200 --warmup 20 --cpu-seconds 0 2> $OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('batch 128, 200 steps: ms_per_step', round(d['ms_per_step'],4), 'infos/s', round(d['value']), 'ref leg', d['reference_batch_128']['value'])"
done
RP_SPARSE_NO_PREP_ONE=1 timeout 200 python bench.py --workload nlhe --nlhe-batch 128 --steps 200 --warmup 20 --cpu-seconds 0 2> $OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  tiled sort (RP_SPARSE_NO_PREP_ONE=1): ms_per_step', round(d['ms_per_step'],4), 'infos/s', round(d['value']))"
if [ "${3:-}" = "big" ]; then
timeout 400 python bench.py --workload nlhe --cpu-seconds 0 --steps 8 --warmup 4 > $OUT/${TAG}_nlhe_bench_line.json 2> $OUT/nlhe.err
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_nlhe_bench_line.json").readlines()[-1])
print("262144 trees: value", round(d["value"]), "ms_per_step", round(d["ms_per_step"],3), d["kernel_ms_per_step"], "roofline frac", round(d["roofline"]["frac"],4), "b128", d["reference_batch_128"]["value"], "pruned", d.get("pruned_regime",{}).get("value"), "exact", d.get("exact_order",{}).get("value"))
PY
fi

#!/bin/bash
# Round 5, after the last code change: the full -m gpu suite and the default bench line on the tree that ships.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/last5
mkdir -p $OUT
cd $REPO
echo "== gpu tests"; date +%T
timeout 900 python -m pytest tests -m gpu -q -x --timeout 240 --timeout-method=thread --durations=6 -p no:cacheprovider > $OUT/gpu_tests.log 2>&1; tail -10 $OUT/gpu_tests.log
echo "== smoke"; date +%T
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -9
echo "== default bench"; date +%T
timeout 420 python bench.py > $OUT/r05_bench_line.json 2> $OUT/bench.err; head -c 200 $OUT/r05_bench_line.json; echo; tail -2 $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/r05_bench_line.json"))
k = d["kmeans"]
print({x: round(k[x], 3) for x in ("create_s", "kmeanspp_s", "init_bounds_s", "elkan_total_s", "lookup_s", "end_to_end_s")})
r = k["reference_arithmetic"]
print("reference", {x: r.get(x) for x in ("kmeanspp_s", "end_to_end_s", "picks_differing_from_contract_pass", "buckets_differing_from_contract_pass", "error")})
print("nlhe", d["nlhe"]["value"], d["nlhe"]["roofline"].get("traffic"), d["nlhe"]["roofline"].get("traffic_source"), d["nlhe"]["reference_batch_128"])
print("leduc", d["value"], d["roofline"]["frac"], d["roofline"].get("valu_source"), d["roofline"].get("traffic_source"))
PY
date +%T

#!/bin/bash
# Round 5, the LAST GPU call: the full -m gpu suite on the shipped build first, then the driver's bench line, then the rocprofv3 passes
# the line's rooflines cite (every step under its own timeout, results written as they come).
#   gpurun --timeout 2400 -- bash scripts/r5_final.sh
set -u
TAG=r05
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
echo "== 1 gpu tests"; date +%T
timeout 900 python -m pytest tests -m gpu -q -x --timeout 240 --timeout-method=thread --durations=12 -p no:cacheprovider > $OUT/gpu_tests.log 2>&1; tail -20 $OUT/gpu_tests.log
echo "== 2 default bench"; date +%T
timeout 420 python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err; head -c 300 $OUT/${TAG}_bench_line.json; echo; tail -2 $OUT/bench.err
python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_bench_line.json"))
    k = d["kmeans"]
    print({x: round(k[x], 3) for x in ("create_s", "kmeanspp_s", "init_bounds_s", "elkan_total_s", "lookup_s", "end_to_end_s")})
    print("turn", {x: k["kmeans_turn"].get(x) for x in ("end_to_end_s", "points_per_s", "hbm_frac")})
    r = k["reference_arithmetic"]
    print("reference", {x: r.get(x) for x in ("end_to_end_s", "picks_differing_from_contract_pass", "buckets_differing_from_contract_pass", "error")})
except Exception as e:
    print("bench line:", e)
PY
echo "== 3 lloyd kernel trace + SQ passes of a slice"; date +%T
bash scripts/r5_lloyd_prof.sh ${TAG} 65536 6 2>&1 | grep -v "^rp::\|^void rp::" | tail -30
cp $REPO/gpurun_out/r5prof/${TAG}_lloyd_* $OUT/ 2>/dev/null
python scripts/valu_ceiling.py $OUT/${TAG}_lloyd_sq_counters.json $OUT/${TAG}_lloyd_slice_kernel_us.json $OUT/${TAG}_lloyd_valu_ceiling.json | head -16
echo "== 4 kernel trace + PMC of the Leduc loop"; date +%T
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
BENCH="python $REPO/bench.py --no-extras --steps 40 --warmup 5"
rm -rf $OUT/kt
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
python $REPO/scripts/rocpd_summary.py "$(ls $OUT/kt/*.db 2>/dev/null | head -1)" $OUT/${TAG}_bench_kernel_stats.txt "$BENCH" | head -8
rm -rf $OUT/kt $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o pmc -- $BENCH > $OUT/fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o pmc -- $BENCH > $OUT/write.log 2>&1
python $REPO/scripts/pmc_traffic.py $OUT/fetch/pmc_counter_collection.csv $OUT/write/pmc_counter_collection.csv \
    $OUT/${TAG}_mccfr_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: $BENCH; FETCH_SIZE doubled (gfx950), KiB -> bytes" 8388608 composed | tail -4
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/sq1 -o pmc -- $BENCH > $OUT/sq1.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA \
  --kernel-trace --output-format csv -d $OUT/sq2 -o pmc -- $BENCH > $OUT/sq2.log 2>&1
python $REPO/scripts/sq_reduce.py $OUT/${TAG}_mccfr_sq_counters.json "$BENCH" $OUT/sq1/pmc_counter_collection.csv $OUT/sq2/pmc_counter_collection.csv | cut -c1-300 | tail -4
rm -rf $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2
echo "== 5 nlhe bench line + the HBM traffic of its level kernels at the same batch"; date +%T
cd $REPO
timeout 200 python bench.py --workload nlhe --cpu-seconds 8 --steps 8 --warmup 4 > $OUT/${TAG}_nlhe_bench_line.json 2> $OUT/nlhe.err; head -c 300 $OUT/${TAG}_nlhe_bench_line.json; echo
cd /tmp
CMD="python $REPO/bench.py --workload nlhe --steps 3 --warmup 2 --cpu-seconds 0"
rm -rf $OUT/nf $OUT/nw
RP_BENCH_NO_REF=1 timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/nf -o pmc -- $CMD > $OUT/nf.log 2>&1
RP_BENCH_NO_REF=1 timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/nw -o pmc -- $CMD > $OUT/nw.log 2>&1
python $REPO/scripts/pmc_traffic.py $OUT/nf/pmc_counter_collection.csv $OUT/nw/pmc_counter_collection.csv \
    $OUT/${TAG}_nlhe_hbm_traffic_b262144.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: $CMD; FETCH_SIZE doubled (gfx950), KiB -> bytes; per launch" 262144 composed | tail -3
rm -rf $OUT/nf $OUT/nw
date +%T

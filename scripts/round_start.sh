#!/bin/bash
# The first GPU call of a round, in one piece (run through gpurun from the repo root, `--timeout 1500`):
#   1 the whole -m gpu suite                                   -> gpurun_out/rs/gpu_tests.log
#   2 the driver's default bench line                          -> gpurun_out/rs/<tag>_bench_line.json
#   3 the NLHE workload's line (both batches)                  -> gpurun_out/rs/<tag>_nlhe_bench_line.json
#   4 kernel-trace statistics of both timed loops              -> gpurun_out/rs/<tag>_bench_kernel_stats.txt, <tag>_nlhe_kernel_stats_*.txt
#   5 PMC HBM traffic of the NLHE level kernels                -> gpurun_out/rs/<tag>_nlhe_hbm_traffic.json
# Every step has its own `timeout` (a counter pass that hung once ate 40 GPU-minutes) and writes what it has as it goes; the
# judged copies are committed under profiles/ by hand afterwards.  About 12 GPU-minutes.
set -u
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/rs
mkdir -p $OUT
cd $REPO
echo "== 1 gpu tests"; date +%T
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
echo "== 2 default bench"; date +%T
timeout 300 python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err; head -c 300 $OUT/${TAG}_bench_line.json; echo
echo "== 3 nlhe bench"; date +%T
timeout 150 python bench.py --workload nlhe --cpu-seconds 10 > $OUT/${TAG}_nlhe_bench_line.json 2> $OUT/nlhe.err; head -c 300 $OUT/${TAG}_nlhe_bench_line.json; echo
echo "== 4 kernel traces"; date +%T
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
BENCH="python $REPO/bench.py --no-extras --steps 40 --warmup 5"
rm -rf $OUT/kt
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
python $REPO/scripts/rocpd_summary.py "$(ls $OUT/kt/*.db 2>/dev/null | head -1)" $OUT/${TAG}_bench_kernel_stats.txt "$BENCH" | head -8
rm -rf $OUT/kt
CMD="python $REPO/bench.py --workload nlhe --steps 8 --warmup 4 --cpu-seconds 0"
rm -rf $OUT/nl
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/nl -o nl -- $CMD > $OUT/nl.log 2>&1
python $REPO/scripts/steady_stats.py $OUT/nl/nl_kernel_trace.csv 4 8 $OUT/${TAG}_nlhe_kernel_stats_b262144.txt "$CMD (the timed 262144-tree steps)" levels | head -16
python $REPO/scripts/steady_stats.py $OUT/nl/nl_kernel_trace.csv 23 20 $OUT/${TAG}_nlhe_kernel_stats_b128.txt "$CMD (the timed 128-tree steps)" | head -10
rm -rf $OUT/nl
echo "== 5 nlhe traffic"; date +%T
cd $REPO
timeout 240 bash scripts/r3_nlhe_traffic.sh $TAG 65536 > $OUT/traffic.log 2>&1; tail -12 $OUT/traffic.log
cp gpurun_out/prof3/${TAG}_nlhe_hbm_traffic.json $OUT/ 2>/dev/null
echo "== 6 opt-in sparse paths (never run on a GPU before round 4): tests first, under a short timeout, then the A/B"; date +%T
RP_SS_ONEPASS=1 RP_SPARSE_APPLY_FUSED=1 timeout 60 python -m pytest tests/test_gpu_sparse.py -m gpu -q -x > $OUT/sparse_optin_tests.log 2>&1; tail -2 $OUT/sparse_optin_tests.log
if grep -q " passed" $OUT/sparse_optin_tests.log && ! grep -q "failed\|error" $OUT/sparse_optin_tests.log; then
  timeout 30 python bench.py --workload nlhe-synth --steps 60 --warmup 5 --cpu-seconds 0 > $OUT/${TAG}_sparse_default.json 2>/dev/null
  RP_SS_ONEPASS=1 timeout 30 python bench.py --workload nlhe-synth --steps 60 --warmup 5 --cpu-seconds 0 > $OUT/${TAG}_sparse_onepass.json 2>/dev/null
  RP_SS_ONEPASS=1 RP_SPARSE_APPLY_FUSED=1 timeout 30 python bench.py --workload nlhe-synth --steps 60 --warmup 5 --cpu-seconds 0 > $OUT/${TAG}_sparse_onepass_applyfused.json 2>/dev/null
  python - <<PY
import json
for f in ("default", "onepass", "onepass_applyfused"):
    try:
        d = json.load(open("$OUT/${TAG}_sparse_%s.json" % f)); print(f, round(d["value"] / 1e6), "M/s", round(d["ms_per_step"], 4), "ms", d["roofline"]["kernels_ms"])
    except Exception as e:
        print(f, "no line:", e)
PY
fi
echo "== 7 opt-in one launch per NLHE tree level: tests under a short timeout, then the A/B at both batch sizes"; date +%T
RP_NLHE_FUSED_LEVELS=1 timeout 90 python -m pytest tests/test_gpu_nlmc.py -m gpu -q -x -k "not twin and not large_batch" > $OUT/nlhe_fused_tests.log 2>&1; tail -2 $OUT/nlhe_fused_tests.log
if grep -q " passed" $OUT/nlhe_fused_tests.log && ! grep -q "failed\|error" $OUT/nlhe_fused_tests.log; then
  timeout 60 python bench.py --workload nlhe --steps 5 --warmup 3 --cpu-seconds 0 > $OUT/${TAG}_nlhe_two_launches.json 2>/dev/null
  RP_NLHE_FUSED_LEVELS=1 timeout 60 python bench.py --workload nlhe --steps 5 --warmup 3 --cpu-seconds 0 > $OUT/${TAG}_nlhe_fused_levels.json 2>/dev/null
  python - <<PY
import json
for f in ("two_launches", "fused_levels"):
    try:
        d = json.load(open("$OUT/${TAG}_nlhe_%s.json" % f)); print(f, round(d["value"] / 1e6), "M/s", d.get("kernel_ms_per_step"), "batch 128:", d.get("reference_batch_128"))
    except Exception as e:
        print(f, "no line:", e)
PY
fi
echo "== 8 the headline kernel with its cells' LDS arrays moved apart (RP_TRAV_CELL_PAD)"; date +%T
for pad in 0 3 7 13; do
  RP_TRAV_CELL_PAD=$pad timeout 60 python bench.py --no-extras --steps 40 --warmup 5 > $OUT/${TAG}_bench_cellpad_$pad.json 2>/dev/null
  python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_bench_cellpad_$pad.json")); print("pad $pad:", round(d["value"] / 1e9, 2), "G/s", d["roofline"]["kernels_ms"])
except Exception as e:
    print("pad $pad: no line:", e)
PY
done
RP_TRAV_SPLIT_PAYOFF=1 timeout 60 python bench.py --no-extras --steps 40 --warmup 5 > $OUT/${TAG}_bench_split_payoff.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/${TAG}_bench_split_payoff.json')); print('split payoff:', round(d['value']/1e9,2), 'G/s', d['roofline']['kernels_ms'])" 2>/dev/null
RP_TRAV_SPLIT_PAYOFF=1 timeout 120 python -m pytest tests/test_gpu_mccfr.py -m gpu -q -x -k "composed or static or bench_sized" 2>&1 | tail -2
date +%T
echo "== 9 the shipped Leduc loop's PMC passes (FETCH / WRITE / two SQ groups), each its own run under its own timeout"; date +%T
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o pmc -- $BENCH > $OUT/fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o pmc -- $BENCH > $OUT/write.log 2>&1
python $REPO/scripts/pmc_traffic.py $OUT/fetch/pmc_counter_collection.csv $OUT/write/pmc_counter_collection.csv \
    $OUT/${TAG}_mccfr_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: $BENCH; FETCH_SIZE doubled (gfx950), KiB -> bytes" 8388608 composed
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/sq1 -o pmc -- $BENCH > $OUT/sq1.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA \
  --kernel-trace --output-format csv -d $OUT/sq2 -o pmc -- $BENCH > $OUT/sq2.log 2>&1
python $REPO/scripts/sq_reduce.py $OUT/${TAG}_mccfr_sq_counters.json "$BENCH" $OUT/sq1/pmc_counter_collection.csv $OUT/sq2/pmc_counter_collection.csv
rm -rf $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2
date +%T

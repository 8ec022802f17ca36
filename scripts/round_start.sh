#!/bin/bash
# The first GPU call of a round (`gpurun --timeout 1500 -- bash scripts/round_start.sh rNN`): the -m gpu suite, the two bench lines and the rocprofv3 passes of the SHIPPED build (every step under its own timeout,
# results written as they come).  From the repo root through gpurun: about 14 GPU-minutes.
set -u
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
cd $REPO
echo "== 1 gpu tests"; date +%T
timeout 600 python -m pytest tests -m gpu -q -x --timeout 120 --timeout-method=thread --durations=8 -p no:cacheprovider > $OUT/gpu_tests.log 2>&1; tail -14 $OUT/gpu_tests.log
echo "== 2 default bench"; date +%T
timeout 330 python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err; head -c 400 $OUT/${TAG}_bench_line.json; echo
echo "== 3 nlhe bench"; date +%T
timeout 200 python bench.py --workload nlhe --cpu-seconds 8 --steps 8 --warmup 4 > $OUT/${TAG}_nlhe_bench_line.json 2> $OUT/nlhe.err; head -c 300 $OUT/${TAG}_nlhe_bench_line.json; echo
echo "== 3b the lloyd kernels' two arithmetic passes (lm_contract beside lm_glibc), same unpruned solves"; date +%T
timeout 120 python scripts/glibc_pass_timing.py 16384 > $OUT/${TAG}_glibc_pass_timing.json 2> $OUT/glibc.err; cat $OUT/${TAG}_glibc_pass_timing.json
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
python $REPO/scripts/sq_reduce.py $OUT/${TAG}_mccfr_sq_counters.json "$BENCH" $OUT/sq1/pmc_counter_collection.csv $OUT/sq2/pmc_counter_collection.csv | cut -c1-400 | tail -4
rm -rf $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2
echo "== 5 NLHE kernel trace"; date +%T
CMD="python $REPO/bench.py --workload nlhe --steps 8 --warmup 4 --cpu-seconds 0"
rm -rf $OUT/nl
RP_BENCH_NO_REF=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/nl -o nl -- $CMD > $OUT/nl.log 2>&1
python $REPO/scripts/steady_stats.py $OUT/nl/nl_kernel_trace.csv 4 8 $OUT/${TAG}_nlhe_kernel_stats_b262144.txt "$CMD (the timed 262144-tree steps)" levels | head -16
rm -rf $OUT/nl
date +%T
echo "== 6 the reference-batch NLHE step (128 trees)"; date +%T
cd $REPO && bash scripts/nlhe_step_trace.sh 128 | head -16
cp $REPO/gpurun_out/b128/r04_nlhe_kernel_stats_b128.txt $OUT/${TAG}_nlhe_kernel_stats_b128.txt 2>/dev/null
date +%T

#!/bin/bash
# HBM traffic of the NLHE level kernels (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes, each under its own timeout)
set -u
TAG=${1:-r03}
B=${2:-65536}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
export RP_BENCH_NO_REF=1
CMD="python $REPO/bench.py --workload nlhe --nlhe-batch $B --steps 3 --warmup 2 --cpu-seconds 0"
rm -rf $OUT/fetch $OUT/write
timeout 50 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o pmc -- $CMD > $OUT/fetch.log 2>&1
timeout 50 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o pmc -- $CMD > $OUT/write.log 2>&1
python $REPO/scripts/pmc_traffic.py $OUT/fetch/pmc_counter_collection.csv $OUT/write/pmc_counter_collection.csv \
    $OUT/${TAG}_nlhe_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: $CMD; FETCH_SIZE doubled (gfx950), KiB -> bytes; PER LAUNCH averages over the run's 8 steps of $B trees (2 warm-up + 3 timed + 3 profiled; k_nl_expand / k_nl_children: one launch per tree level, 22 per step)" $B composed | grep -E "k_nl|k_block_maps"
rm -rf $OUT/fetch $OUT/write

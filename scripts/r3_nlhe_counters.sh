#!/bin/bash
# SQ counters + HBM traffic of the NLHE step's kernels (separate rocprofv3 --pmc passes, kernel-trace only)
set -u
TAG=${1:-r03}
B=${2:-65536}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof3
mkdir -p $OUT
CMD="python $REPO/bench.py --workload nlhe --nlhe-batch $B --steps 3 --warmup 1 --cpu-seconds 0"
( cd $REPO && bash scripts/pmc_sq.sh ${TAG}nlhe $CMD ) > $OUT/sq.log 2>&1
{ echo "# scripts/pmc_sq.sh: rocprofv3 --pmc <two passes> --kernel-trace of: $CMD  (sums over all dispatches of each kernel)"; grep -E "^(void )?rp::k_nl" $OUT/sq.log; } > $OUT/${TAG}_nlhe_sq_counters.txt
cat $OUT/${TAG}_nlhe_sq_counters.txt
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
rm -rf $OUT/fetch $OUT/write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o pmc -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o pmc -- $CMD > $OUT/write.log 2>&1
python $REPO/scripts/pmc_traffic.py $OUT/fetch/pmc_counter_collection.csv $OUT/write/pmc_counter_collection.csv \
    $OUT/${TAG}_nlhe_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: $CMD; FETCH_SIZE doubled (gfx950), KiB -> bytes; PER LAUNCH averages (a step launches k_nl_expand / k_nl_children once per level)" $B composed | grep k_nl
rm -rf $OUT/fetch $OUT/write

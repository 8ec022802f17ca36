#!/bin/bash
# profiling experiments on k_nl_expand (results of ablated runs are garbage; only the kernel times matter)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ablate
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
for cfg in "0 16384 20" "0 16384 24" "0 16384 27" "2 16384 20"; do
  set -- $cfg
  export RP_NLHE_ABLATE=$1 RP_NLHE_GRID=$2
  rm -rf $OUT/run
  rocprofv3 --kernel-trace --stats -d $OUT/run -o nl -- python $REPO/bench.py --workload nlhe --nlhe-batch 262144 --nlhe-cap $3 --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/log_$1_$2_$3.txt 2>&1
  python $REPO/scripts/rocpd_summary.py $(ls $OUT/run/*.db | head -1) $OUT/stats_$1_$2_$3.txt "ablate=$1 grid=$2 cap=$3" > /dev/null
  echo "== ablate=$1 grid=$2 cap=$3"; grep -E "k_nl_expand|k_nl_children|k_nl_emit|k_permute" $OUT/stats_$1_$2_$3.txt | cut -c1-70
  grep -o '"value": [0-9.]*' $OUT/log_$1_$2_$3.txt | head -1
done
rm -rf $OUT/run

#!/bin/bash
# Round 6: A/B of environment switches at the reference's batch inside ONE box (numbers of different boxes differ by 2 - 3 %).
# usage: r6_ab.sh "<env A>" "<env B>" [reps]     e.g.  r6_ab.sh "RP_NL_TREE_GENERAL=1" "X=0" 3
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
A="$1"; B="$2"; N=${3:-3}
for i in $(seq 1 $N); do
for kv in "$A" "$B"; do
env $kv timeout 200 python bench.py --workload nlhe --nlhe-batch 128 --steps 300 --warmup 20 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$kv', 'ms_per_step', round(d['ms_per_step'],4), 'infos/s', round(d['value']))"
done
done

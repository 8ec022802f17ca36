#!/bin/bash
# Round 6: k_nl_tree's phase clocks at the reference's batch.  Uses the diagnostic library built HERE beside the product one
# (make -C robopoker_amd/csrc OBJDIR=_obj_prof OUT=../librp_mi355x_prof.so COMMON+=-DNL_TREE_PROF), swapped in on the box's scratch copy only.
# usage: r6_nltree_prof.sh <tag> [BT ...]   (workgroup sizes of k_nl_tree to try: RP_NL_TREE_BT)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6nlt
mkdir -p $OUT
cd $REPO
TAG=${1:-r06}; shift
cp robopoker_amd/librp_mi355x.so /tmp/librp_product.so
for bt in ${@:-256}; do
  # the product library first: the step as the bench times it (no clocks in the kernel)
  RP_NL_TREE_BT=$bt timeout 300 python bench.py --workload nlhe --nlhe-batch 128 --steps 40 --warmup 10 --cpu-seconds 0 2> $OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('BT $bt product: ms_per_step', round(d['ms_per_step'],4), 'infos/s', round(d['value']))"
  cp robopoker_amd/librp_mi355x_prof.so robopoker_amd/librp_mi355x.so
  RP_NL_TREE_BT=$bt timeout 300 python scripts/r6_nltree_prof.py 128 40 > $OUT/${TAG}_nltree_phases_b128_bt${bt}.json 2> $OUT/err.log
  python - <<PY
import json
d=json.load(open("$OUT/${TAG}_nltree_phases_b128_bt${bt}.json"))
print("BT $bt prof: ms_per_step", round(d["ms_per_step"],4), "infos/s", round(d["infos_per_s"]), {k: round(v,1) for k,v in d["us_per_tree_by_phase"].items()}, "total", round(d["us_per_tree_total"],1), d["fit_us"])
PY
  tail -2 $OUT/err.log
  cp /tmp/librp_product.so robopoker_amd/librp_mi355x.so
done

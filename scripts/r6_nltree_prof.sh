#!/bin/bash
# Round 6: k_nl_tree's phase clocks at the reference's batch.  Uses the diagnostic library built HERE beside the product one
# (make -C robopoker_amd/csrc OBJDIR=_obj_prof OUT=../librp_mi355x_prof.so COMMON+=-DNL_TREE_PROF), swapped in on the box's scratch copy only.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6nlt
mkdir -p $OUT
cd $REPO
cp robopoker_amd/librp_mi355x.so /tmp/librp_product.so
cp robopoker_amd/librp_mi355x_prof.so robopoker_amd/librp_mi355x.so
timeout 300 python scripts/r6_nltree_prof.py 128 40 > $OUT/${1:-r06}_nltree_phases_b128.json 2> $OUT/err.log
cat $OUT/${1:-r06}_nltree_phases_b128.json; tail -3 $OUT/err.log
cp /tmp/librp_product.so robopoker_amd/librp_mi355x.so

#!/bin/bash
# Round 5: the whole abstraction pipeline on the real point sets with this round's kernels; the audited variant of the real flop layer
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5abs
mkdir -p $OUT
cd $REPO
date +%T
timeout 400 python scripts/full_abstraction.py > $OUT/r05_full_abstraction.json 2> $OUT/r05_full_abstraction.log; tail -c 1500 $OUT/r05_full_abstraction.json; echo; tail -6 $OUT/r05_full_abstraction.log
date +%T

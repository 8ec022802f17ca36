#!/bin/bash
# usage: gpurun --timeout 900 -- bash scripts/r6_study.sh [N] [iters]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6study
mkdir -p $OUT
cd $REPO
timeout 600 python scripts/r6_refresh_study.py ${1:-131072} ${2:-10} > $OUT/r06_refresh_study_contract.json 2> $OUT/contract.err; tail -12 $OUT/contract.err | cut -c1-400
RP_FULL_LIBM=glibc timeout 600 python scripts/r6_refresh_study.py ${1:-131072} ${2:-10} > $OUT/r06_refresh_study_glibc.json 2> $OUT/glibc.err; tail -12 $OUT/glibc.err | cut -c1-400

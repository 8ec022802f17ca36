#!/bin/bash
# usage: gpurun -- bash scripts/r5_tests.sh "<pytest -k expression>" [file]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
export RP_FIXTURE_CACHE=/tmp
date +%T
timeout ${3:-600} python -m pytest ${2:-tests/test_gpu_lloyd.py} -m gpu -q -x -k "$1" -p no:cacheprovider --durations=5 2>&1 | tail -15
date +%T

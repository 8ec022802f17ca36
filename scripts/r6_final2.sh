#!/bin/bash
# Round 6, last tree: the whole -m gpu suite + the default bench line (scripts/r6_final.sh), then the rocprofv3 kernel stats and SQ counters of
# the flop layer in the reference's arithmetic (scripts/r6_glibc_prof.sh).   usage: gpurun --timeout 2700 -- bash scripts/r6_final2.sh TAG
TAG=${1:-r06k}
bash scripts/r6_final.sh $TAG
bash scripts/r6_glibc_prof.sh $TAG

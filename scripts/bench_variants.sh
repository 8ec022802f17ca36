#!/bin/bash
# A few configurations of the MCCFR bench on one GPU (through gpurun): batch sizes, the sharded path with a world of one.
run() { echo "== $*"; timeout 300 python bench.py --no-extras --steps 64 --warmup 8 "$@" 2>/dev/null | grep -o "\"value\": [0-9.]*, \|\"ms_per_step\": [0-9.]*\|kernels_ms[^}]*}" | tr "\n" " "; echo; }
run
run --batch 1048576
run --batch 131072
run --batch 1048576 --force-sharded --window 4
run --batch 1048576 --force-sharded --window 1
run --batch 131072 --force-sharded --window 4
run --batch 1048576 --force-sharded --window 4 --comm torch
run --batch 1048576 --update ordered

run() { echo "== $*"; timeout 300 python bench.py --no-extras --steps 64 --warmup 8 "$@" 2>/dev/null | grep -o "\"value\": [0-9.]*, \|\"ms_per_step\": [0-9.]*\|kernels_ms[^}]*}" | tr "\n" " "; echo; }
run
run --batch 8388608
run --force-sharded
run --force-sharded --window 4
run --force-sharded --comm torch
run --force-sharded --batch 131072
run --force-sharded --batch 131072 --window 4
run --batch 131072

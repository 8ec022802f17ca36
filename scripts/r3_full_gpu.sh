#!/bin/bash
# the whole GPU suite + the NLHE bench lines
mkdir -p gpurun_out/r3b
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r3b/tests.log
cat gpurun_out/r3b/tests.log
for s in external pluribus; do
timeout 900 python bench.py --workload nlhe --sampling $s --steps 8 --warmup 4 --cpu-seconds 10 > gpurun_out/r3b/bench_nlhe_$s.json 2> gpurun_out/r3b/bench_nlhe_$s.err
cat gpurun_out/r3b/bench_nlhe_$s.json; tail -2 gpurun_out/r3b/bench_nlhe_$s.err
done

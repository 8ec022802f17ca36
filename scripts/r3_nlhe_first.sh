#!/bin/bash
# round 3, first GPU contact of the level-synchronous NLHE traversal: parity tests, then the bench at three batch sizes
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests/test_gpu_nlmc.py tests/test_gpu_nlhe.py "tests/test_golden.py::test_device_reproduces_nlmc_golden" -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r3a/tests.log
cat gpurun_out/r3a/tests.log
for b in 128 16384 262144; do
  timeout 600 python bench.py --workload nlhe --nlhe-batch $b --steps 6 --warmup 2 --cpu-seconds 0 > gpurun_out/r3a/bench_$b.json 2> gpurun_out/r3a/bench_$b.err
  tail -c 1500 gpurun_out/r3a/bench_$b.json; tail -3 gpurun_out/r3a/bench_$b.err
done

#!/bin/bash
mkdir -p gpurun_out/audit
timeout 1500 python scripts/mfma_audit.py synthetic 768 > gpurun_out/audit/synthetic.json 2> gpurun_out/audit/synthetic.log
tail -c 1500 gpurun_out/audit/synthetic.json
timeout 1500 python scripts/mfma_audit.py real 768 > gpurun_out/audit/real.json 2> gpurun_out/audit/real.log
tail -c 1500 gpurun_out/audit/real.json
timeout 300 python -m pytest tests/test_golden.py -q -m gpu -x -k deuce 2>&1 | tail -2

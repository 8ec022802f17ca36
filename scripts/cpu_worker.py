#!/usr/bin/env python3
"""One CPU-oracle MCCFR solver for a fixed wall time; prints "<updates> <seconds>".  Used by bench.py's all-cores
baseline (one process per physical core).  usage: cpu_worker.py game regret weight sampling batch seed seconds"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import oracle  # noqa: E402
from robopoker_amd import Game  # noqa: E402

game, regret, weight, sampling = sys.argv[1:5]
batch, seed, seconds = int(sys.argv[5]), int(sys.argv[6]), float(sys.argv[7])
s = oracle.OracleSolver(Game(game), regret, weight, sampling, batch=batch, seed=seed)
s.step()
_, i0 = s.counters()
t0 = time.perf_counter()
while time.perf_counter() - t0 < seconds:
    s.step()
dt = time.perf_counter() - t0
_, i1 = s.counters()
print(i1 - i0, dt, flush=True)

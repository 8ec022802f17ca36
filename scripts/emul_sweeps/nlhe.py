import sys
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tests', 'emul')]
import harness; harness.load_emulated()
import numpy as np, oracle
import oracle_nlmc as M
import test_gpu_nlmc as T
from robopoker_amd.nlhe import NlheSolver
bad = 0
for sampling in ("external", "prunable", "pluribus"):
    for batch in (1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 257, 511, 513, 700):
        seed = batch * 7 + len(sampling)
        try:
            hp = oracle.default_hyper(); hp.prune_warmup, hp.prune_threshold, hp.prune_explore = 0, 20.0, 0.3
            dev = NlheSolver(cap_log2=17, batch=batch, seed=seed, sampling=sampling, hyper=hp)
            ora = M.OracleNlhe(cap_log2=17, batch=batch, seed=seed, sampling=sampling, hyper=hp)
            for _ in range(3):
                T._same_batch(dev.batch(), ora.batch())
                dev.step("ordered"); ora.step()
                assert dev.counters() == ora.counters(), (dev.counters(), ora.counters())
                dev.load(*ora.export(), epoch=ora.epoch)
            dev.close()
            print("ok ", sampling, batch, flush=True)
        except Exception as e:
            bad += 1; print("BAD", sampling, batch, type(e).__name__, str(e)[:200], flush=True)
print("bad", bad)

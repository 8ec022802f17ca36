import sys, os, itertools, traceback
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tests', 'emul')]
import harness; harness.load_emulated()
import numpy as np
import oracle
from robopoker_amd import lloyd
from lloyd_fixtures import flop_like_points, turn_like_points, smooth_metric
bits = lambda a: np.asarray(a, dtype=np.float32).view(np.uint32)
def run(kind, K, N, bins, mass, seed, steps=2):
    pts = flop_like_points(N, bins=bins, mass=mass, seed=seed) if kind=="sinkhorn" else turn_like_points(N, bins=bins, mass=mass, seed=seed)
    tri = smooth_metric(bins, seed) if kind=="sinkhorn" else None
    hp = oracle.default_sinkhorn(); hp.iterations = 10
    dev = lloyd.Layer(K, pts, kind, tri, hp=hp, seed=seed); ora = oracle.OracleKmeans(K, pts, kind, tri, hp=hp, seed=seed)
    assert np.array_equal(dev.init_centroids(), ora.init_centroids()), "kpp"
    dev.init_bounds(); ora.init_bounds()
    for s in range(steps):
        d1,s1,m1 = dev.step(); d2,s2,m2 = ora.step()
        assert np.array_equal(bits(d1), bits(d2)), f"drift step {s}"
        assert np.array_equal(s1, s2) and m1 == m2, f"sizes step {s}"
        j1,u1,l1 = dev.bounds(); j2,u2,l2 = ora.bounds()
        assert np.array_equal(j1,j2) and np.array_equal(bits(u1),bits(u2)) and np.array_equal(bits(l1),bits(l2)), f"bounds step {s}"
    b1, dd1 = dev.lookup(); b2, dd2 = ora.assign()
    assert np.array_equal(b1, b2) and np.array_equal(bits(dd1), bits(dd2)), "lookup"
    t1 = dev.metric(); t2 = ora.metric()
    assert np.array_equal(bits(t1), bits(t2)), "metric"
cases = []
for kind in ("sinkhorn", "variation"):
    for K, N in ((1, 5), (2, 2), (3, 70), (64, 64), (65, 130), (2, 1000)):
        for bins, mass in ((2, 3), (5, 4), (33, 7), (64, 47)) if kind=="sinkhorn" else ((101, 46), (7, 3), (2, 5)):
            cases.append((kind, K, N, bins, mass))
bad = 0
for i, c in enumerate(cases):
    try:
        run(*c, seed=100 + i)
        print("ok ", c, flush=True)
    except AssertionError as e:
        bad += 1; print("BAD", c, e, flush=True)
    except Exception as e:
        bad += 1; print("ERR", c, type(e).__name__, str(e)[:200], flush=True)
print("bad", bad)

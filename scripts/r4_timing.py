"""where does a small -m gpu test spend its time? (round 4: the suite got 10x slower somewhere)"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
t0 = time.time()
import oracle
from robopoker_amd import Game
from robopoker_amd.mccfr import Solver
print("imports", round(time.time() - t0, 2), flush=True)
for game in ("kuhn", "leduc", "leduc_wide"):
    for rng in ("counter", "reference"):
        for sampling in ("external", "pluribus"):
            t = [time.time()]
            g = Game(game); t.append(time.time())
            dev = Solver(g, "linear", "linear", sampling, batch=333, seed=42); t.append(time.time())
            dev.set_rng(rng); t.append(time.time())
            dev.step(); t.append(time.time())
            for _ in range(9): dev.step()
            t.append(time.time())
            dev.export(); t.append(time.time())
            dev.exploitability(); t.append(time.time())
            dev.close(); t.append(time.time())
            names = ["game", "create", "set_rng", "step1", "steps9", "export", "exploit", "close"]
            print(game, rng, sampling, {n: round(b - a, 3) for n, a, b in zip(names, t, t[1:])}, flush=True)

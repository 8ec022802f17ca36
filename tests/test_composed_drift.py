"""Accumulated difference between the composed update (the re-associated mode bench.py quotes and the multi-GPU exchange
uses) and the reference's sequential order, WITHOUT re-synchronising the tables between steps.

The two runs draw from the same hash stream; a weight that differs in its last bits can move a cumulative sampling threshold
across a draw, and from that tree on the runs are two different, equally valid MCCFR runs (the reference itself is not
reproducible run to run: thread-RNG root deals, RandomState maps).  A flip shows as a few cells that differ
by a whole regret contribution while the rest of the table still agrees to re-association accuracy.  So the stated
tolerance is on the BULK of the table over the first steps (90th percentile of |difference| / largest entry), and on
solution quality (exploitability) at the end of the run."""
import numpy as np
import pytest

import oracle
from robopoker_amd import Game


@pytest.mark.parametrize("regret,weight,batch", [("linear", "linear", 4096), ("floored", "linear", 512), ("summed", "constant", 512),
                                                 ("floored", "quadratic", 2048)])
def test_composed_drift_without_resync(regret, weight, batch):
    g = Game("leduc")
    steps = 64
    a = oracle.OracleSolver(g, regret, weight, "external", batch=batch, seed=9)  # the reference's order
    b = oracle.OracleSolver(g, regret, weight, "external", batch=batch, seed=9)  # composed maps (ora_mccfr_step_local/apply)
    for step in range(steps):
        a.step()
        b.step_world(1)
        if step < 12:
            ra, rb = a.export(), b.export()
            for f in ("regret", "weight", "payoff"):
                scale = max(float(np.abs(ra[f]).max()), 1e-6)
                q90 = float(np.quantile(np.abs(ra[f] - rb[f]), 0.9)) / scale
                assert q90 <= 2e-4, (f, step, q90)
    ea, eb = a.exploitability(), b.exploitability()
    assert abs(ea - eb) <= 0.15 * max(ea, eb) + 2e-3, (ea, eb)
    assert a.epoch == b.epoch == steps

import sys
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tests', 'emul')]
import harness; harness.load_emulated()
import numpy as np, oracle, torch
from robopoker_amd.sparse import DeviceBatch, SparseProfile, synthetic_batch
FIELDS = ("weight", "regret", "payoff", "visits")
bad = 0
for mode in ("ordered", "composed"):
    for n_rows in (1, 2, 37, 511, 513, 70000):
        for A in (2, 3, 9, 16):
            for n in (1, 2, 63, 64, 65, 255, 257, 1023, 1025, 5000):
                regret, weight = ("linear", "linear") if (n + A) % 2 else ("floored", "exponential")
                try:
                    g = SparseProfile(n_rows, A, regret, weight); o = oracle.OracleProfile(n_rows, A, regret, weight)
                    for e in range(2):
                        batch = synthetic_batch(n, n_rows, A, seed=n + A + e)
                        db = DeviceBatch(*batch)
                        if mode == "ordered":
                            g.apply(db, "ordered"); o.apply(batch)
                        else:
                            buf = torch.zeros(db.n * g.entry_bytes(), dtype=torch.uint8, device="cuda")
                            k = g.summarize(db, buf.data_ptr()); exp = o.summarize(batch)
                            assert k * g.entry_bytes() == exp.size and np.array_equal(buf[: exp.size].cpu().numpy(), exp), "entries"
                            g.apply(db, "composed"); o.fold(exp)
                    g.sync()
                    rows = np.arange(min(n_rows, 600))
                    a, b = g.rows(rows), o.rows(rows)
                    for f in FIELDS: assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f
                    g.close() if hasattr(g, "close") else None
                except Exception as ex:
                    bad += 1; print("BAD", mode, n_rows, A, n, type(ex).__name__, str(ex)[:160], flush=True)
print("bad", bad)

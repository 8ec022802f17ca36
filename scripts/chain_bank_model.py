"""LDS bank model of k_traverse_maps_static's chain loop: the lists of real Leduc chunks (the oracle's Decisions), 32-lane
groups of (cell, infoset) tasks in the kernel's order, one ds_read_b32 per step — cycles = the fullest bank of each group — for
every padding of the cells' value arrays (RP_TRAV_CELL_PAD).  A prediction to be checked against SQ_LDS_BANK_CONFLICT, not a
measurement."""
import sys
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests')]
import numpy as np, oracle
from robopoker_amd import Game
g = Game("leduc")
NI = g.n_infos
B = 256*8
s = oracle.OracleSolver(g, "linear", "linear", "external", batch=B, seed=5)
for _ in range(6): s.step()   # a trained-ish table, walker alternates
res = {}
for trial in range(2):
    b = s.batch()
    info = np.array([d["info"] for d in b]); tree = np.array([d["tree"] for d in b]); tree = tree - tree.min()
    s.step()
    for chunk in range(B // 256):
        sel = (tree // 256) == chunk
        cnt = np.bincount(info[sel], minlength=NI)
        lbase = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        cls = np.where(cnt > 0, 15 - np.floor(np.log2(np.maximum(cnt, 1))).astype(int) - 1, 15)
        order = np.argsort(cls, kind="stable")
        maxdec = 6
        for pad in range(0, 32):
            L = maxdec * 256 + pad
            cyc = 0
            tasks = [(t % 5, order[t // 5]) for t in range(5 * NI)]
            for g0 in range(0, len(tasks), 32):
                grp = [(c, i) for (c, i) in tasks[g0:g0 + 32] if cnt[i] > 0]
                if not grp: continue
                nmax = max(cnt[i] for _, i in grp)
                for e in range(nmax):
                    banks = np.zeros(32, int)
                    for c, i in grp:
                        if e < cnt[i]: banks[(c * L + lbase[i] + e) % 32] += 1
                    cyc += banks.max()
            res[pad] = res.get(pad, 0) + cyc
base = res[0]
for pad in sorted(res, key=lambda p: res[p])[:8]: print("pad", pad, "cycles", res[pad], "ratio", round(res[pad] / base, 3))
print("pad 0", res[0], "pad 7", res[7], "pad 13", res[13])

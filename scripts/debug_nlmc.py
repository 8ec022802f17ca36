import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_nlmc as M
from robopoker_amd.nlhe import NlheSolver
batch, seed = 64, 5
dev = NlheSolver(cap_log2=18, batch=batch, seed=seed)
ora = M.OracleNlhe(cap_log2=18, batch=batch, seed=seed)
d, o = dev.batch(), ora.batch()
print("n", d["n"], o["n"], "counters", dev.counters(), ora.counters())
dc = np.bincount(d["tree"], minlength=batch); oc = np.bincount(o["tree"].astype(np.int64), minlength=batch)
bad = np.nonzero(dc != oc)[0]
print("trees with different decision counts:", bad[:10], dc[bad[:10]], oc[bad[:10]])
past, present, choices, _ = ora.export()
# oracle rows -> keys via export order? rows are slots; rebuild via ora_nlmc_row_key
import ctypes as C
o_ = ora._o; o_.ora_nlmc_row_key.argtypes=[C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
def okey(row):
    a,b,c = C.c_uint64(), C.c_uint32(), C.c_uint64(); o_.ora_nlmc_row_key(ora._h, int(row), C.byref(a), C.byref(b), C.byref(c)); return (a.value,b.value,c.value)
def edges(p):
    out=[]
    while p and (p&31): out.append(p&31); p>>=5
    return out
for t in list(bad[:2]) + ([int(np.nonzero(dc==oc)[0][0])] if (dc==oc).any() else []):
    di = np.nonzero(d["tree"]==t)[0]; oi = np.nonzero(o["tree"]==t)[0]
    dk = [(int(d["past"][i]), int(d["present"][i]), int(d["choices"][i])) for i in di]
    ok = [okey(o["row"][i]) for i in oi]
    print("tree", t, "dev", len(dk), "ora", len(ok), "same set", set(dk)==set(ok), "same order", dk==ok)
    for k in dk:
        if k not in ok: print("  only dev:", edges(k[0]), k[1], edges(k[2]))
    for k in ok:
        if k not in dk: print("  only ora:", edges(k[0]), k[1], edges(k[2]))
    if dk==ok and len(di):
        i,j = di[0], oi[0]
        print("  first decision regret dev", d["regret"][i][:d["n_actions"][i]], "ora", o["regret"][j][:o["n_actions"][j]], "payoff", d["payoff"][i], o["payoff"][j])
same = [t for t in range(batch) if dc[t]==oc[t]]
mx=0
for t in same[:40]:
    di = np.nonzero(d["tree"]==t)[0]; oi = np.nonzero(o["tree"]==t)[0]
    if len(di): mx=max(mx, float(np.abs(d["regret"][di]-o["regret"][oi]).max()))
print("max regret diff over same-count trees", mx)

"""Per-tree comparison of the device's level-synchronous NLHE batch with the oracle's (debugging aid)."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import oracle_nlmc as M
from robopoker_amd.nlhe import NlheSolver

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = NlheSolver(cap_log2=18, batch=batch, seed=seed)
ora = M.OracleNlhe(cap_log2=18, batch=batch, seed=seed)
d, o = dev.batch(), ora.batch()
print("n", d["n"], o["n"], "shape", dev.last_shape())
op, ob, oc, _ = ora.export()
okey = {}
o_ = M.lib()
import ctypes as C
for i in range(int(o["n"])):
    kp, kb, kc = C.c_uint64(), C.c_uint32(), C.c_uint64()
    o_.ora_nlmc_row_key(ora._h, int(o["row"][i]), C.byref(kp), C.byref(kb), C.byref(kc))
    okey[i] = (kp.value, kb.value, kc.value)
for t in range(batch):
    di = np.nonzero(d["tree"] == t)[0]
    oi = np.nonzero(o["tree"] == t)[0]
    dk = [(int(d["past"][i]), int(d["present"][i]), int(d["choices"][i])) for i in di]
    ok = [okey[i] for i in oi]
    same = dk == ok
    print(f"tree {t}: dev {len(dk)} ora {len(ok)} same_order={same} same_set={set(dk) == set(ok)}")
    if not same:
        missing = [k for k in ok if k not in set(dk)]
        extra = [k for k in dk if k not in set(ok)]
        print("   missing", [(hex(a), hex(b), hex(c)) for a, b, c in missing[:6]])
        print("   extra  ", [(hex(a), hex(b), hex(c)) for a, b, c in extra[:6]])
        if len(sys.argv) > 3:
            break
ora.step()
print("oracle nodes", ora.counters())

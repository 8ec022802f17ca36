import ctypes as C, numpy as np, sys
from robopoker_amd import _lib
lib=_lib.load()
p=lambda a: a.ctypes.data
bad=0
for rep in range(5):
    n=17_800_000; bits=27
    rng=np.random.default_rng(rep)
    keys=rng.integers(0,1<<bits,size=n,dtype=np.uint64).astype(np.uint32)
    hot=rng.random(n)<0.35
    keys[hot]=(rng.integers(0,3000,size=int(hot.sum()),dtype=np.uint64).astype(np.uint32)*977+5)
    out={k:np.zeros(n,np.uint32) for k in ("keys","perm","uniq","starts","counts")}
    nr=C.c_uint32(); scan=np.zeros(n,np.uint64)
    _lib.check(lib.rp_sortscan_selftest(0,n,bits,p(keys),p(out["keys"]),p(out["perm"]),p(out["uniq"]),p(out["starts"]),p(out["counts"]),C.addressof(nr),p(scan)))
    order=np.argsort(keys,kind="stable")
    ok=np.array_equal(out["perm"],order.astype(np.uint32)) and np.array_equal(out["keys"],keys[order])
    u,s,c=np.unique(keys[order],return_index=True,return_counts=True)
    ok2=nr.value==len(u) and np.array_equal(out["uniq"][:nr.value],u) and np.array_equal(out["starts"][:nr.value],s.astype(np.uint32)) and np.array_equal(out["counts"][:nr.value],c.astype(np.uint32))
    print(rep, ok, ok2, flush=True)
    bad+= (not ok) or (not ok2)
print("bad", bad)

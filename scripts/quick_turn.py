import sys, time
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from robopoker_amd import lloyd
from lloyd_fixtures import turn_like_points
N=int(sys.argv[1]) if len(sys.argv)>1 else 1745006
K=256; bins=101
t0=time.time(); pts=turn_like_points(N,bins=bins,mass=46,seed=5); print(f"gen {time.time()-t0:.1f}s", flush=True)
t0=time.time(); L=lloyd.Layer(K,pts,"variation",seed=1); print(f"create {time.time()-t0:.3f}s", flush=True)
rng=np.random.default_rng(1)
L.set_centroids(rng.choice(N,size=K,replace=False).astype(np.uint64))
t0=time.time(); L.init_bounds(); tb=time.time()-t0
print(f"N={N} init_bounds={tb:.3f}s points/s={N/tb:.3e}", flush=True)
L.profile(True)
for it in range(3):
    t0=time.time(); drift,sizes,moved=L.step(); dt=time.time()-t0
    print(f" step{it}: {dt:.4f}s moved={moved:.3f} points/s={N/dt:.3e} algorithmic GB/s={(N*2165)/dt/1e9:.1f}", flush=True)
for nm in ["pairwise","step","recompute","bounds","drift","selfcost"]:
    ms,n=L.kernel_time(nm); print(f"  {nm}: {ms/max(n,1):.3f} ms x{n}")
t0=time.time(); b,d=L.lookup(); print(f"lookup {time.time()-t0:.3f}s")

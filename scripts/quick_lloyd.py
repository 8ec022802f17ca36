import sys, time, json
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from robopoker_amd import lloyd
from lloyd_fixtures import flop_like_points, smooth_metric
N=int(sys.argv[1]) if len(sys.argv)>1 else 8192
K=256; bins=256
pts=flop_like_points(N,bins=bins,mass=47,seed=0xF10F)
tri=smooth_metric(bins,1)
t0=time.time(); L=lloyd.Layer(K,pts,"sinkhorn",tri,seed=1); t_create=time.time()-t0
rng=np.random.default_rng(1)
L.set_centroids(rng.choice(N,size=K,replace=False).astype(np.uint64))
d0,i0=L.stats()
t0=time.time(); L.init_bounds(); tb=time.time()-t0
d1,i1=L.stats()
print(f"N={N} create(selfcost)={t_create:.3f}s init_bounds={tb:.3f}s dist/s={(d1-d0)/tb:.3e} iters/dist={(i1-i0)/(d1-d0):.1f}", flush=True)
L.profile(True)
for it in range(2):
    d0,i0=L.stats(); t0=time.time(); drift,sizes,moved=L.step(); dt=time.time()-t0; d1,i1=L.stats()
    print(f" step{it}: {dt:.3f}s dist={d1-d0} iters/dist={(i1-i0)/max(d1-d0,1):.1f} moved={moved:.3f} points/s={N/dt:.3e}", flush=True)
for nm in ["pairwise","step","recompute","bounds","drift","selfcost"]:
    ms,n=L.kernel_time(nm); print(f"  {nm}: {ms/max(n,1):.3f} ms x{n}")

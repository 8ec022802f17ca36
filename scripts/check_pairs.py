"""Debug aid: lookups and k-means++ picks with two points per wavefront vs one (RP_LLOYD_NO_PAIRS=1) must be identical.
usage: [KPP_K=32] python scripts/check_pairs.py [N]"""
import os, sys
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from robopoker_amd import lloyd
from lloyd_fixtures import flop_like_points, smooth_metric
N=int(sys.argv[1]) if len(sys.argv)>1 else 20000
K=256; bins=256
pts=flop_like_points(N,bins=bins,mass=47,seed=3)
tri=smooth_metric(bins,1)
def run(nopairs):
    if nopairs: os.environ["RP_LLOYD_NO_PAIRS"]="1"
    else: os.environ.pop("RP_LLOYD_NO_PAIRS",None)
    L=lloyd.Layer(K,pts,"sinkhorn",tri,seed=1)
    rng=np.random.default_rng(1)
    L.set_centroids(rng.choice(N,size=K,replace=False).astype(np.uint64))
    b0,d0=L.lookup()
    L.init_bounds(); L.step(); L.step()
    b1,d1=L.lookup()
    return b0,d0,b1,d1
a=run(True); b=run(False)
for nm,x,y in zip(("bucket0","dist0","bucket1","dist1"),a,b):
    x=np.asarray(x); y=np.asarray(y)
    if x.dtype.kind=='f': bad=np.flatnonzero(x.view(np.uint32)!=y.view(np.uint32))
    else: bad=np.flatnonzero(x!=y)
    print(nm, "mismatches", bad.size, bad[:10], (x[bad[:5]], y[bad[:5]]) if bad.size else "")
sup=(pts>0).sum(1)
if bad.size: print("support sizes of bad points", sup[bad[:20]])
def kpp(nopairs):
    if nopairs: os.environ["RP_LLOYD_NO_PAIRS"]="1"
    else: os.environ.pop("RP_LLOYD_NO_PAIRS",None)
    L=lloyd.Layer(int(os.environ.get("KPP_K","32")),pts,"sinkhorn",tri,seed=1)
    return np.asarray(L.init_centroids())
ka=kpp(True); kb=kpp(False)
print("kpp picks equal:", np.array_equal(ka,kb), ka[:8], kb[:8])

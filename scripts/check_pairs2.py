import os, sys
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from robopoker_amd import lloyd
from lloyd_fixtures import flop_like_points, smooth_metric
N=70000; K=256; bins=256
pts=flop_like_points(N,bins=bins,mass=47,seed=3)
tri=smooth_metric(bins,1)
rng=np.random.default_rng(1)
cidx=rng.choice(N,size=K,replace=False)
def look(P, cents, nopairs):
    if nopairs: os.environ["RP_LLOYD_NO_PAIRS"]="1"
    else: os.environ.pop("RP_LLOYD_NO_PAIRS",None)
    L=lloyd.Layer(K,P,"sinkhorn",tri,seed=1)
    L.set_centroids(np.asarray(cents,dtype=np.uint64))
    return [np.asarray(x) for x in L.lookup()]
a=look(pts,cidx,True); b=look(pts,cidx,False)
bad=np.flatnonzero(a[1].view(np.uint32)!=b[1].view(np.uint32))
print("full: bad", bad.size, bad[:6])
sup=(pts>0).sum(1)
small=np.flatnonzero(sup<=32)
pos={int(v):i for i,v in enumerate(small)}
p=int(bad[0]); q=pos[p]; partner=int(small[q^1])
print("point",p,"partner",partner,"sup",sup[p],sup[partner],"single-path bucket",a[0][p],"dist",a[1][p],"paired",b[0][p],b[1][p])
# tiny dataset: the two points first (they pair with each other), then the centroid points
D=np.concatenate([pts[[p,partner] if q%2==0 else [partner,p]], pts[cidx]])
ta=look(D, np.arange(2,2+K), True); tb=look(D, np.arange(2,2+K), False)
print("tiny: single", ta[0][:2], ta[1][:2], "paired", tb[0][:2], tb[1][:2])
# which centroid: distance of the two points to each centroid alone (K=1 layers are not allowed with kind... use K=2 with the same centroid twice)
for k in range(K):
    D2=np.concatenate([D[:2], pts[[cidx[k]]], pts[[cidx[k]]]])
    os.environ["RP_LLOYD_NO_PAIRS"]="1"; L=lloyd.Layer(1,D2,"sinkhorn",tri,seed=1); L.set_centroids(np.array([2],dtype=np.uint64)); x=np.asarray(L.lookup()[1])[:2]
    os.environ.pop("RP_LLOYD_NO_PAIRS"); L=lloyd.Layer(1,D2,"sinkhorn",tri,seed=1); L.set_centroids(np.array([2],dtype=np.uint64)); y=np.asarray(L.lookup()[1])[:2]
    if not np.array_equal(x.view(np.uint32),y.view(np.uint32)):
        print("centroid",k,"sup",sup[cidx[k]],"single",x,"paired",y); break

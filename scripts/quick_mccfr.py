import sys, time
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
from robopoker_amd import Game
from robopoker_amd.mccfr import Solver
g=Game("leduc")
for B in [1<<12, 1<<14, 1<<16, 1<<18, 1<<20]:
    s=Solver(g,"floored","linear","external",batch=B,seed=1)
    s.step_async(4); s.sync()
    n0,i0=s.counters()
    s.profile(True)
    t0=time.time(); K=20
    s.step_async(K); s.sync()
    dt=time.time()-t0
    n1,i1=s.counters()
    tr=s.kernel_time("traverse"); up=s.kernel_time("update")
    print(f"B={B} ms/step={dt/K*1e3:.3f} updates/s={(i1-i0)/dt:.3e} nodes/s={(n1-n0)/dt:.3e} traverse_ms={tr[0]/tr[1]:.3f} update_ms={up[0]/up[1]:.3f}", flush=True)
    s.close()

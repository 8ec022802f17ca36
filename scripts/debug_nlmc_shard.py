import sys, numpy as np, torch
sys.path.insert(0, "tests")
import oracle_nlmc as M
from robopoker_amd.nlhe import NlheSolver
world, batch = 2, 96
devs = [NlheSolver(cap_log2=18, batch=batch, seed=31) for _ in range(world)]
for r, d in enumerate(devs): d.set_shard(r, world)
eb, cap = devs[0].entry_bytes()
print("eb cap", eb, cap)
ora = M.OracleNlhe(cap_log2=18, batch=batch, seed=31)
oras = [M.OracleNlhe(cap_log2=18, batch=batch, seed=31) for _ in range(world)]
for r, o in enumerate(oras): o.set_shard(r, world)
mk = lambda n, dt: torch.zeros(n, dtype=dt, device="cuda")
bufs = {"ent": mk(cap*eb*world, torch.uint8), "past": mk(cap*world, torch.int64), "present": mk(cap*world, torch.int32), "choices": mk(cap*world, torch.int64)}
unit = {"ent": eb, "past": 8, "present": 4, "choices": 8}
hb = {k: np.zeros(v.numel(), dtype={"ent":np.uint8,"past":np.int64,"present":np.int32,"choices":np.int64}[k]) for k,v in bufs.items()}
for step in range(4):
    off = 0; ns=[]
    for d in devs:
        n = d.step_local(*[bufs[k].data_ptr() + off*unit[k] for k in ("ent","past","present","choices")]); off += n; ns.append(n)
    ooff=0; ons=[]
    for o in oras:
        n = o.step_local(*[hb[k].ctypes.data + ooff*unit[k] for k in ("ent","past","present","choices")]); ooff += n; ons.append(n)
    print("step", step, "dev counts", ns, "ora counts", ons)
    torch.cuda.synchronize()
    dk = set(zip(bufs["past"][:off].cpu().tolist(), bufs["present"][:off].cpu().tolist(), bufs["choices"][:off].cpu().tolist()))
    ok = set(zip(hb["past"][:ooff].tolist(), hb["present"][:ooff].tolist(), hb["choices"][:ooff].tolist()))
    print("entry keys dev", len(dk), "ora", len(ok), "sym diff", len(dk ^ ok))
    for d in devs: d.step_apply(*[bufs[k].data_ptr() for k in ("ent","past","present","choices")], off)
    for o in oras: o.step_apply(*[hb[k].ctypes.data for k in ("ent","past","present","choices")], ooff)
    ora.step_world(world)
    om = {k: v for k, v in M.as_map(*ora.export()).items() if v["visits"][0] > 0}
    for d in devs:
        dm = {k: v for k, v in M.as_map(*d.export()).items() if v["visits"][0] > 0}
        print(" dev", len(dm), "ora", len(om), "only dev", len(dm.keys()-om.keys()), "only ora", len(om.keys()-dm.keys()))
        for k in list(om.keys()-dm.keys())[:3]:
            alld = M.as_map(*d.export())
            print("   missing", k, om[k]["visits"], "in dev table:", k in alld, alld[k]["visits"] if k in alld else None)
    for d in devs: d.load(*ora.export(), epoch=ora.epoch)
    for o in oras: o.load(*ora.export())

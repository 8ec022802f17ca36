"""Times the abstraction-input kernels at full size on one GPU (device time from HIP events inside the library)."""
import json
import sys
import time

import torch

from robopoker_amd import deuce

out = {}
for street in ("pref", "flop", "turn", "rive"):
    t0 = time.time()
    obs = deuce.isomorphisms(street)
    ms = deuce.kernel_ms()
    out[f"isomorphisms_{street}"] = {"n": obs.numel(), "device_ms": round(ms, 3), "wall_s": round(time.time() - t0, 3)}
    if street == "flop":
        flop = obs
    if street == "turn":
        turn = obs
river = obs
for rep in range(2):
    e, b = deuce.river_equity(river)
    ms = deuce.kernel_ms()
out["river_equity"] = {"n": river.numel(), "device_ms": round(ms, 3), "showdowns_per_s": river.numel() * 990 / (ms * 1e-3),
                       "observations_per_s": river.numel() / (ms * 1e-3)}
table = deuce.Lookup("rive", river, b)
for rep in range(2):
    pts = table.projections(turn, 101)
    ms = deuce.kernel_ms()
out["project_turn"] = {"n": turn.numel(), "device_ms": round(ms, 3), "points_per_s": turn.numel() / (ms * 1e-3),
                       "lookups_per_s": turn.numel() * 46 / (ms * 1e-3)}
del pts
abs_ = (((turn * 2654435761) >> 20) % 200).to(torch.uint8)
t2 = deuce.Lookup("turn", turn, abs_)
for rep in range(2):
    pts = t2.projections(flop, 200)
    ms = deuce.kernel_ms()
out["project_flop"] = {"n": flop.numel(), "device_ms": round(ms, 3), "points_per_s": flop.numel() / (ms * 1e-3)}
print(json.dumps(out, indent=1))

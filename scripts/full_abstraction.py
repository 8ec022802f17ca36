#!/usr/bin/env python3
"""The whole abstraction pipeline at full size (robopoker_amd.pretraining): river equities, turn and flop clustering on
the real point sets, preflop.
    one GPU:   python scripts/full_abstraction.py [flop_iterations] [turn_iterations]
    N GPUs:    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29544 \
                   scripts/full_abstraction.py [flop_iterations] [turn_iterations]
(points sharded by rank, centroid sums all-reduced over RCCL; rank 0 prints the JSON summary)"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from robopoker_amd import pretraining  # noqa: E402

fi = int(sys.argv[1]) if len(sys.argv) > 1 else None
ti = int(sys.argv[2]) if len(sys.argv) > 2 else None
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
say = lambda m: print(m, file=sys.stderr, flush=True)  # noqa: E731
t0 = time.perf_counter()
if world > 1:
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    os.dup2(2, 1) if rank else None  # only rank 0 keeps stdout
    dist.init_process_group("nccl", rank=rank, world_size=world)
    art = pretraining.run_sharded(local, log=say if rank == 0 else None, flop_iterations=fi, turn_iterations=ti)
    dist.barrier()
    torch.cuda.synchronize()
else:
    # RP_FULL_LIBM=glibc / RP_FULL_RNG=reference: the pipeline in the reference's own arithmetic and k-means++ draw
    art = pretraining.run(local, log=say, flop_iterations=fi, turn_iterations=ti, libm=os.environ.get("RP_FULL_LIBM", "contract"),
                          rng=os.environ.get("RP_FULL_RNG", "counter"))
total = time.perf_counter() - t0
if rank != 0:
    sys.exit(0)
out = {"total_s": total, "n_gpus": world}
for street, a in art.items():
    sizes = torch.bincount(a.abstraction.to(torch.int64)).cpu().numpy()
    t = {k: v for k, v in a.timings.items() if k != "reassigned"}
    if "reassigned" in a.timings:
        t["reassigned_first_last"] = [a.timings["reassigned"][0], a.timings["reassigned"][-1]] if a.timings["reassigned"] else []
    out[street] = {"n": a.obs.numel(), "abstractions": int(len(sizes)), "empty_clusters": int((sizes == 0).sum()),
                   "largest_cluster": int(sizes.max()), "timings": t}
    if a.metric is not None:
        out[street]["metric_max"] = float(np.max(a.metric))
        out[street]["metric_min"] = float(np.min(a.metric))
print(json.dumps(out), flush=True)

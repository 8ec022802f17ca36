"""bench.py's N > 1 path, executed: two rank processes, as `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`
would start them, on a machine without a GPU.  The driver's scaling run is the only place this code would otherwise ever run
(one device per box in a round), and a crash there costs the round's only multi-GPU measurement.  The ranks load the kernels'
sources under the wave64 execution model with the RCCL stand-in (tests/emul) and use gloo for torch.distributed; the
numbers in the line mean nothing — the line's shape, the sharded step over rp_comm, the max-over-ranks / sum-over-ranks reductions
and a clean exit are what is checked."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL = os.path.join(ROOT, "tests", "emul")
CLANG = os.environ.get("RP_EMUL_CXX", "/opt/rocm/lib/llvm/bin/clang++")

RANK_MAIN = """
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {emul!r})
import harness
harness.load_emulated(build=False)
import bench
sys.argv = ["bench.py"] + {argv!r}
bench.main()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(argv, world=2, timeout=900, extra_env=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RP_EMUL_THREADS="4", **(extra_env or {}))
        code = RANK_MAIN.format(root=ROOT, emul=EMUL, argv=argv)
        procs.append(subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    return [(p.returncode, o, e) for p, (o, e) in zip(procs, outs)]


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(CLANG):
        pytest.skip(f"{CLANG} (host compiler of the execution model) is not installed")
    sys.path.insert(0, EMUL)
    import build as emul_build

    return emul_build.build(jobs=os.cpu_count() or 4)


@pytest.mark.parametrize("comm", ["native", "torch"])
def test_bench_two_ranks_prints_one_contract_line(built, comm):
    res = _launch(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4096", "--no-kmeans", "--cpu-seconds", "0",
                   "--dist-backend", "gloo", "--comm", comm])
    for rc, out, err in res:
        assert rc == 0, err[-3000:]
    lines = [ln for ln in res[0][1].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res[0][1][-2000:]
    assert not [ln for ln in res[1][1].splitlines() if ln.startswith("{")]  # rank 0 speaks alone
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "strong"
    assert line["config"]["global_batch"] == 4096 and line["config"]["batch_per_gpu"] == 2048
    assert line["value"] > 0 and line["ms_per_step"] > 0 and line["cpu_baseline"] is None
    assert line["other_scaling"]["scaling"] == "weak" and line["other_scaling"]["batch_per_gpu"] == 4096
    # whole-job units: both ranks' infoset updates over the max-over-ranks time
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 * 3 - round(line["value"] * line["ms_per_step"] * 1e-3 * 3)) < 1e-3 * line["value"]


def test_bench_nlhe_two_ranks_prints_one_contract_line(built):
    res = _launch(["--gpus", "2", "--workload", "nlhe", "--nlhe-batch", "48", "--nlhe-cap", "15", "--steps", "2", "--warmup", "1",
                   "--cpu-seconds", "0", "--dist-backend", "gloo"], extra_env={"RP_BENCH_NO_REF": "1"})
    for rc, out, err in res:
        assert rc == 0, err[-3000:]
    lines = [ln for ln in res[0][1].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res[0][1][-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["workload"]


KMEANS_MAIN = """
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, {emul!r})
import harness
harness.load_emulated(build=False)
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
bench.init_rccl(rank, world, "gloo")
out = bench.kmeans_sharded(rank, world, 0, n_points=96, K=6, bins=32, iters=2)
if rank == 0:
    print(json.dumps(out))
import torch.distributed as dist
dist.barrier()
dist.destroy_process_group()
"""


def test_bench_kmeans_exchange_two_ranks(built):
    # what the driver's N > 1 launch runs after the contract line (stderr, under a watchdog): k-means++ across the ranks, two Elkan
    # iterations with the integer all-reduce — here at a size the execution model finishes in seconds
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RP_EMUL_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, "-c", KMEANS_MAIN.format(root=ROOT, emul=EMUL)], cwd=ROOT, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    line = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"

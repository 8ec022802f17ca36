"""The library's OWN sharded entry points (rp_mccfr_step_comm, rp_nlhe_step_comm, rp_kmeans_step_comm over rp_comm, csrc/comm.cpp)
with a world of TWO — on a machine without a GPU.  On the GPU boxes a world of one is all there is (one device per box), so the
N > 1 arithmetic of these C paths (tree-id ranges per rank, gather sizes and strides, the rank-order fold, the integer all-reduce)
had never executed anywhere.  Here two processes load the kernels' sources under the wave64 execution model (tests/emul,
DESIGN.md §2b); its comm.cpp opens tests/emul/fake_rccl.cpp in place of librccl, whose collectives move host memory between the
two processes.  Expected values: the oracle's single-process world model (bit for bit; NLHE floats within the stated tolerance), and replica against
replica bit for bit."""
from __future__ import annotations

import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL = os.path.join(ROOT, "tests", "emul")
CLANG = os.environ.get("RP_EMUL_CXX", "/opt/rocm/lib/llvm/bin/clang++")
WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _enter(rank, port, world=WORLD):
    """in a rank process: gloo (carries the communicator's id and the comparisons), then the emulated library"""
    for p in (ROOT, os.path.join(ROOT, "tests"), EMUL):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import harness

    harness.load_emulated(build=False)
    return dist


def _same_everywhere(dist, arr) -> bool:
    import torch

    t = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy())
    n = torch.tensor([t.numel()])
    ref_n = n.clone()
    dist.broadcast(ref_n, src=0)
    if int(n) != int(ref_n):
        return False
    ref = t.clone()
    dist.broadcast(ref, src=0)
    return bool(torch.equal(t, ref))


def _mccfr_native(rank, port, out, window, world):
    dist = _enter(rank, port, world)
    import oracle
    from robopoker_amd import Game
    from robopoker_amd.mccfr import Solver
    from robopoker_amd.parallel import Comm

    g = Game("leduc")
    B, steps = 64, 7  # a run that ends inside a window when window = 3
    s = Solver(g, "linear", "linear", "external", batch=B, seed=33, device=0)
    s.set_update_mode("composed")
    comm = Comm.from_process_group(0)
    s.set_shard(rank, world)
    s.step_comm(comm, steps, window)
    s.sync()
    rows = s.export()
    same = _same_everywhere(dist, rows)
    ok = True
    if rank == 0:
        single = oracle.OracleSolver(g, "linear", "linear", "external", batch=B, seed=33)
        left = steps
        while left:
            w = min(window, left)
            single.window_world(world, w)
            left -= w
        exp = single.export()
        ok = all(np.array_equal(rows[f].view(np.uint32), exp[f].view(np.uint32)) for f in ("regret", "weight", "payoff", "visits"))
        ok = ok and s.epoch == steps
    out.put((f"mccfr-{rank}", bool(same and ok)))
    comm.close()
    s.close()
    dist.destroy_process_group()


def _nlhe_native(rank, port, out, sampling):
    dist = _enter(rank, port)
    import oracle
    import oracle_nlmc as M
    from robopoker_amd.nlhe import NlheSolver
    from robopoker_amd.parallel import Comm

    def hyper():
        hp = oracle.default_hyper()
        hp.prune_warmup, hp.prune_threshold, hp.prune_explore = 1, -2.0, 0.1  # pruning live from the second step on
        return hp

    s = NlheSolver(cap_log2=16, regret="linear", weight="linear", batch=24, seed=8, sampling=sampling, hyper=hyper())
    single = M.OracleNlhe(cap_log2=16, regret="linear", weight="linear", batch=24, seed=8, sampling=sampling, hyper=hyper())
    comm = Comm.from_process_group(0)
    s.set_shard(rank, WORLD)
    same = ok = True
    for step in range(3):
        s.step_comm(comm, 1)
        s.sync()
        single.step_world(WORLD)  # every rank keeps the world model: it is what the tables are resynchronised from below
        mine = {k: v for k, v in M.as_map(*s.export()).items() if v["visits"][0] > 0}
        want = {k: v for k, v in M.as_map(*single.export()).items() if v["visits"][0] > 0}
        keys = sorted(mine)
        same = same and _same_everywhere(dist, np.concatenate([np.frombuffer(mine[k].tobytes(), dtype=np.uint8) for k in keys]))
        # the device traversal sums a node's children in another order than the oracle: infosets and visits are
        # exact, floats carry the tolerance of tests/test_gpu_nlmc.py and the tables are resynchronised after every step as
        # there; the two REPLICAS are compared bit for bit
        ok = ok and want.keys() == mine.keys() and s.epoch == step + 1
        for k in want:
            if not ok:
                break
            ok = (np.array_equal(mine[k]["visits"], want[k]["visits"])
                  and np.allclose(mine[k]["regret"], want[k]["regret"], rtol=2e-4, atol=5e-3)
                  and np.allclose(mine[k]["weight"], want[k]["weight"], rtol=2e-4, atol=1e-5)
                  and np.allclose(mine[k]["payoff"], want[k]["payoff"], rtol=2e-4, atol=5e-3))
        s.load(*single.export(), epoch=single.epoch)
    ok = ok and s.counters()[1] > 0
    out.put((f"nlhe-{rank}", bool(same and ok)))
    comm.close()
    s.close()
    dist.destroy_process_group()


def _kmeans_native(rank, port, out, kind):
    dist = _enter(rank, port)
    import oracle
    from lloyd_fixtures import smooth_metric, turn_like_points
    from robopoker_amd import lloyd
    from robopoker_amd.parallel import Comm, ShardedLayer

    K, N, bins, mass, seed = 6, 200, (24 if kind == "sinkhorn" else 101), (14 if kind == "sinkhorn" else 46), 9
    pts = turn_like_points(N, bins=bins, mass=mass, seed=seed)
    tri = smooth_metric(bins, seed) if kind == "sinkhorn" else None
    hp = oracle.default_sinkhorn()
    hp.iterations = 12
    lo, hi = (0, 117) if rank == 0 else (117, N)  # ragged shards
    eng = lloyd.Layer(K, pts[lo:hi], kind, tri, hp=hp, seed=seed)
    sh = ShardedLayer(eng, K, bins, seed, device="cpu")  # k-means++ across the ranks: host logic over gloo, distances by the engine
    sh.init_centroids()
    sh.init_bounds()
    comm = Comm.from_process_group(0)
    drifts, sizes = [], None
    for _ in range(3):
        d, sizes, _ = eng.step_comm(comm)  # the library's own exchange: ncclAllReduce of the integer centroid sums
        drifts.append(np.array(d, copy=True))
    c, w = eng.centroids()
    j, _, _ = eng.bounds()
    same = _same_everywhere(dist, c)
    import torch

    other = torch.zeros(N - 117 if rank == 0 else 117, dtype=torch.uint8)
    mine = torch.from_numpy(np.ascontiguousarray(j).copy())
    if rank == 0:
        dist.recv(other, src=1)
    else:
        dist.send(mine, dst=0)
    ok = True
    if rank == 0:
        single = oracle.OracleKmeans(K, pts, kind, tri, hp=hp, seed=seed)
        single.init_centroids()
        single.init_bounds()
        sd, ssz = [], None
        for _ in range(3):
            d, ssz, _ = single.step()
            sd.append(d.copy())
        sc, sw = single.centroids()
        sj, _, _ = single.bounds()
        ok = (np.array_equal(c, sc) and np.array_equal(w, sw) and np.array_equal(sizes, ssz)
              and all(np.array_equal(np.asarray(a).view(np.uint32), b.view(np.uint32)) for a, b in zip(drifts, sd))
              and np.array_equal(np.concatenate([mine.numpy(), other.numpy()]), sj))
    out.put((f"kmeans-{rank}", bool(same and ok)))
    comm.close()
    dist.destroy_process_group()


def _guarded(fn, rank, port, out, *args):
    """a rank that raises still reports (the parent would otherwise wait for its timeout)"""
    try:
        fn(rank, port, out, *args)
    except BaseException as exc:  # noqa: BLE001
        import traceback

        traceback.print_exc()
        out.put((f"error-{rank}", f"{type(exc).__name__}: {exc}"))
        raise


@pytest.fixture(scope="module")
def rccl_dir():
    if not os.path.exists(CLANG):
        pytest.skip(f"{CLANG} (host compiler of the execution model) is not installed")
    sys.path.insert(0, EMUL)
    import build as emul_build

    emul_build.build(jobs=os.cpu_count() or 4)
    return emul_build.RCCL_DIR


def _run(rccl_dir, fn, *args, world=WORLD):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    saved = {k: os.environ.get(k) for k in ("RP_EMUL_THREADS",)}
    os.environ["RP_EMUL_THREADS"] = str(max(1, 8 // world))  # the ranks share the host's cores
    try:
        procs = [ctx.Process(target=_guarded, args=(fn, r, port, q) + args) for r in range(world)]
        for p in procs:
            p.start()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert not [k for k in results if k.startswith("error")], results
    assert all(p.exitcode == 0 for p in procs)
    return results


@pytest.mark.parametrize("world,window", [(2, 1), (2, 3), (8, 1), (8, 3)])
def test_native_sharded_mccfr_equals_the_world_model(rccl_dir, world, window):
    # eight ranks: the node the scaling run uses
    assert _run(rccl_dir, _mccfr_native, window, world, world=world) == {f"mccfr-{r}": True for r in range(world)}


@pytest.mark.parametrize("sampling", ["external", "pluribus"])
def test_native_sharded_nlhe_two_ranks_equals_the_world_model(rccl_dir, sampling):
    assert _run(rccl_dir, _nlhe_native, sampling) == {"nlhe-0": True, "nlhe-1": True}


@pytest.mark.parametrize("kind", ["sinkhorn", "variation"])
def test_native_sharded_kmeans_two_ranks_equals_single_process(rccl_dir, kind):
    assert _run(rccl_dir, _kmeans_native, kind) == {"kmeans-0": True, "kmeans-1": True}

"""world_size-2 tests of the sharded host logic (robopoker_amd/parallel.py) on CPU with the gloo backend.

The compute engine here is the CPU oracle exposing the same sharded surface as the C-ABI
(step_local / step_apply, kpp_* / step_local / step_finish); on a GPU box the engine is the HIP
Solver / Layer and the backend is RCCL.  Checked: rank sharding of tree ids and points, blob layouts,
gather/fold order, the exact-integer k-means++ draw across ranks, replica consistency.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from lloyd_fixtures import smooth_metric, turn_like_points
from robopoker_amd import Game
from robopoker_amd.parallel import ShardedLayer, ShardedProfile, ShardedSolver, rp_mulhi64, rp_stream
from robopoker_amd.sparse import synthetic_batch

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)


def _mccfr_worker(rank, port, out):
    _init(rank, port)
    g = Game("leduc")
    B, steps = 96, 5
    eng = oracle.OracleSolver(g, "linear", "linear", "external", batch=B, seed=21)
    sh = ShardedSolver(eng, device="cpu")
    for _ in range(steps):
        sh.step()
    rows = eng.export()
    # every replica must hold the same table, bit for bit
    t = torch.from_numpy(rows["regret"].view(np.int32).copy())
    ref = t.clone()
    dist.broadcast(ref, src=0)
    same = bool(torch.equal(t, ref))
    if rank == 0:
        single = oracle.OracleSolver(g, "linear", "linear", "external", batch=B, seed=21)
        for _ in range(steps):
            single.step_world(WORLD)
        exp = single.export()
        ok = all(np.array_equal(rows[f].view(np.uint32), exp[f].view(np.uint32))
                 for f in ("regret", "weight", "payoff", "visits"))
        out.put(("mccfr", same and ok and eng.epoch == steps))
    else:
        out.put(("mccfr-replica", same))
    dist.destroy_process_group()


def _mccfr_window_worker(rank, port, out, window):
    # the periodic exchange: `window` local steps per all-gather, a run that ends inside a window (flush)
    _init(rank, port)
    g = Game("leduc")
    B, steps = 64, 7
    eng = oracle.OracleSolver(g, "linear", "linear", "external", batch=B, seed=33)
    sh = ShardedSolver(eng, device="cpu", window=window)
    for _ in range(steps):
        sh.step()
    sh.flush()
    rows = eng.export()
    t = torch.from_numpy(rows["regret"].view(np.int32).copy())
    ref = t.clone()
    dist.broadcast(ref, src=0)
    same = bool(torch.equal(t, ref))
    if rank == 0:
        single = oracle.OracleSolver(g, "linear", "linear", "external", batch=B, seed=33)
        left = steps
        while left:
            w = min(window, left)
            single.window_world(WORLD, w)
            left -= w
        exp = single.export()
        ok = all(np.array_equal(rows[f].view(np.uint32), exp[f].view(np.uint32))
                 for f in ("regret", "weight", "payoff", "visits"))
        if window == 1:  # a window of one step IS step_local + step_apply
            plain = oracle.OracleSolver(g, "linear", "linear", "external", batch=B, seed=33)
            for _ in range(steps):
                plain.step_world(WORLD)
            pe = plain.export()
            ok = ok and all(np.array_equal(pe[f].view(np.uint32), exp[f].view(np.uint32))
                            for f in ("regret", "weight", "payoff", "visits"))
        out.put(("window", same and ok and eng.epoch == steps and single.epoch == steps))
    else:
        out.put(("window-replica", same and eng.epoch == steps))
    dist.destroy_process_group()


def _kmeans_worker(rank, port, out, kind):
    _init(rank, port)
    K, N, bins, mass, seed = 6, 200, (24 if kind == "sinkhorn" else 101), (14 if kind == "sinkhorn" else 46), 9
    pts = turn_like_points(N, bins=bins, mass=mass, seed=seed)
    tri = smooth_metric(bins, seed) if kind == "sinkhorn" else None
    hp = oracle.default_sinkhorn()
    hp.iterations = 12
    lo, hi = (0, 117) if rank == 0 else (117, N)  # ragged shards
    eng = oracle.OracleKmeans(K, pts[lo:hi], kind, tri, hp=hp, seed=seed)
    sh = ShardedLayer(eng, K, bins, seed, device="cpu")
    sh.init_centroids()
    sh.init_bounds()
    drifts = []
    for _ in range(3):
        d, sizes, _ = sh.step()
        drifts.append(d.copy())
    c, w = eng.centroids()
    j, u, lower = eng.bounds()
    # gather the shard results on rank 0
    gj = [torch.zeros(117, dtype=torch.uint8), torch.zeros(N - 117, dtype=torch.uint8)]
    mine = torch.from_numpy(j.copy())
    if rank == 0:
        gj[0] = mine
        dist.recv(gj[1], src=1)
    else:
        dist.send(mine, dst=0)
    if rank == 0:
        single = oracle.OracleKmeans(K, pts, kind, tri, hp=hp, seed=seed)
        single.init_centroids()
        single.init_bounds()
        sd = []
        for _ in range(3):
            d, ssz, _ = single.step()
            sd.append(d.copy())
        sc, sw = single.centroids()
        sj, _, _ = single.bounds()
        ok = (np.array_equal(c, sc) and np.array_equal(w, sw) and np.array_equal(sizes, ssz)
              and all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(drifts, sd))
              and np.array_equal(np.concatenate([gj[0].numpy(), gj[1].numpy()]), sj))
        out.put((f"kmeans-{kind}", ok))
    else:
        out.put((f"kmeans-{kind}-r1", True))
    dist.destroy_process_group()


def _profile_worker(rank, port, out):
    _init(rank, port)
    n_rows, A, steps = 300, 9, 4
    eng = oracle.OracleProfileEngine(n_rows, A, "linear", "linear")
    sh = ShardedProfile(eng, max_batch=4000, device="cpu")
    for e in range(steps):
        n = 4000 if rank == 0 else 2500 + 100 * e  # ragged shards: the gather is padded to the longest list
        sh.step(synthetic_batch(n, n_rows, A, seed=10 * e + rank))
    rows = eng.rows(np.arange(n_rows))
    t = torch.from_numpy(rows["regret"].view(np.int32).copy())
    ref = t.clone()
    dist.broadcast(ref, src=0)
    same = bool(torch.equal(t, ref))
    if rank == 0:
        single = oracle.OracleProfile(n_rows, A, "linear", "linear")
        for e in range(steps):
            blobs = [single.summarize(synthetic_batch(4000 if r == 0 else 2500 + 100 * e, n_rows, A, seed=10 * e + r))
                     for r in range(WORLD)]
            single.fold(np.concatenate(blobs))
        exp = single.rows(np.arange(n_rows))
        ok = all(np.array_equal(rows[f].view(np.uint32), exp[f].view(np.uint32)) for f in ("regret", "weight", "payoff", "visits"))
        out.put(("profile", same and ok and eng.epoch() == steps))
    else:
        out.put(("profile-replica", same))
    dist.destroy_process_group()


def _nlhe_worker(rank, port, out, sampling="external"):
    # BASELINE configs[3]'s exchange: trees sharded by rank, composed entries exchanged by infoset key (robopoker_amd.parallel.
    # ShardedNlhe) — against the single-process world model, bit for bit, and replica against replica
    import oracle_nlmc as M
    from robopoker_amd.parallel import ShardedNlhe

    _init(rank, port)

    def make():
        hp = oracle.default_hyper()
        hp.prune_warmup, hp.prune_threshold, hp.prune_explore = 1, -2.0, 0.1  # pruning live from the second step on
        return M.OracleNlhe(cap_log2=16, regret="linear", weight="linear", batch=24, seed=8, sampling=sampling, hyper=hp)

    eng = make()
    sh = ShardedNlhe(eng, device="cpu")
    for _ in range(3):
        sh.step()
    # infosets a rank merely READ during its own traversal sit in its table with their default row (the reference treats a
    # missing Encounter the same way, book.rs:93-122): replicas agree on every infoset an update has touched
    mine = {k: v for k, v in M.as_map(*eng.export()).items() if v["visits"][0] > 0}
    keys = sorted(mine)
    blob = np.concatenate([np.frombuffer(mine[k].tobytes(), dtype=np.uint8) for k in keys])
    t = torch.from_numpy(blob.copy())
    size = torch.tensor([t.numel()])
    ref_size = size.clone()
    dist.broadcast(ref_size, src=0)
    same = int(size) == int(ref_size)
    if same:
        ref = t.clone()
        dist.broadcast(ref, src=0)
        same = bool(torch.equal(t, ref))
    if rank == 0:
        single = make()
        for _ in range(3):
            single.step_world(WORLD)
        want = {k: v for k, v in M.as_map(*single.export()).items() if v["visits"][0] > 0}
        ok = want.keys() == mine.keys() and all(want[k].tobytes() == mine[k].tobytes() for k in want)
        out.put(("nlhe", same and ok and eng.epoch == 3 and eng.counters()[1] > 0))
    else:
        out.put(("nlhe-replica", same))
    dist.destroy_process_group()


def _pretraining_gather_worker(rank, port, out):
    # the list plumbing of the sharded abstraction pipeline (robopoker_amd/pretraining.py): contiguous equal-width
    # slices of an isomorphism list, the last one short; 1-byte results all-gathered back into list order
    _init(rank, port)
    from robopoker_amd import pretraining
    ok = True
    for n in (1001, 1000, 7, 2):
        whole = (torch.arange(n, dtype=torch.int64) * 37 % 251).to(torch.uint8)
        lo, hi, width = pretraining._slice(n, rank, WORLD)
        got = pretraining._gather_u8(whole[lo:hi].clone(), width, n, None)
        ok = ok and bool(torch.equal(got, whole))
    out.put((rank, ok))


def _run(fn, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=fn, args=(r, port, q) + args) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(WORLD)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return dict(results)


def test_rp_stream_and_mulhi_match_the_c_header():
    # the Python mirrors used for the cross-rank k-means++ draw must equal include/rp_math.h
    pts = turn_like_points(64, bins=101, mass=46, seed=1)
    km = oracle.OracleKmeans(4, pts, "variation", seed=77)
    km.kpp_begin()
    total = km.kpp_total()
    pick = km.kpp_pick(rp_mulhi64(rp_stream(77, 0), total))
    ref = oracle.OracleKmeans(4, pts, "variation", seed=77).init_centroids()
    assert pick == ref[0]


def test_sharded_mccfr_two_ranks_equals_world_model():
    res = _run(_mccfr_worker)
    assert res == {"mccfr": True, "mccfr-replica": True}


@pytest.mark.parametrize("window", [1, 3])
def test_sharded_mccfr_exchange_window_equals_world_model(window):
    res = _run(_mccfr_window_worker, window)
    assert res == {"window": True, "window-replica": True}


@pytest.mark.parametrize("kind", ["variation", "sinkhorn"])
def test_sharded_kmeans_two_ranks_equals_single_process(kind):
    res = _run(_kmeans_worker, kind)
    assert res[f"kmeans-{kind}"] is True


def test_sharded_sparse_profile_two_ranks_equals_world_model():
    res = _run(_profile_worker)
    assert res == {"profile": True, "profile-replica": True}


@pytest.mark.parametrize("sampling", ["external", "pluribus"])
def test_sharded_nlhe_two_ranks_exchange_by_key_equals_world_model(sampling):
    # pluribus: the Flagship type (nlhe/src/lib.rs:86-90) with pruning forced live — a rank's pruned trees still exchange by key
    res = _run(_nlhe_worker, sampling)
    assert res == {"nlhe": True, "nlhe-replica": True}


def test_sharded_pretraining_slices_and_gathers_two_ranks():
    res = _run(_pretraining_gather_worker)
    assert res == {0: True, 1: True}

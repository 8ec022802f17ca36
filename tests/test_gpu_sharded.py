"""GPU: the sharded C-ABI surface (step_local / step_apply, kpp_* / step_local / step_finish) on ONE device,
two handles playing rank 0 and rank 1, buffers exchanged by hand — against the oracle's world model, bit-exact."""
import numpy as np
import pytest

import oracle
from lloyd_fixtures import flop_like_points, smooth_metric, turn_like_points
from robopoker_amd import Game, lloyd
from robopoker_amd.mccfr import Solver
from robopoker_amd.parallel import rp_mulhi64, rp_stream

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,world", [(500, 2), (700, 8)])
def test_mccfr_shards_on_one_gpu_match_world_model(gpu, B, world):
    # `world` handles play the ranks of one node (the driver's 8-GPU run is world = 8): every replica must end each
    # step with the oracle's world-model table, bit for bit
    import torch

    g = Game("leduc")
    devs = [Solver(g, "floored", "linear", "external", batch=B, seed=33) for _ in range(world)]
    for r, d in enumerate(devs):
        d.set_shard(r, world)
    n = devs[0].summary_bytes()
    gathered = torch.zeros(n * world, dtype=torch.uint8, device="cuda")
    ora = oracle.OracleSolver(g, "floored", "linear", "external", batch=B, seed=33)
    for _ in range(5):
        for r, d in enumerate(devs):
            d.step_local(gathered.data_ptr() + r * n)
            d.sync()
        for d in devs:
            d.step_apply(gathered.data_ptr(), world)
            d.sync()
        ora.step_world(world)
        exp = ora.export()
        for d in devs:
            got = d.export()
            for f in ("visits", "regret", "weight", "payoff"):
                assert np.array_equal(got[f].view(np.uint32), exp[f].view(np.uint32)), f
    assert devs[0].epoch == 5


@pytest.mark.parametrize("B,world,window,regret", [(300, 2, 3, "linear"), (520, 4, 4, "floored"), (200, 2, 1, "linear")])
def test_mccfr_exchange_window_on_one_gpu_matches_world_model(gpu, B, world, window, regret):
    # the periodic exchange: every rank folds the composed maps of `window` local steps (table frozen, epoch advancing)
    # into one summary; one gather + apply per window.  Against ora_mccfr_window_world, bit for bit, over 3 windows.
    import torch

    g = Game("leduc")
    devs = [Solver(g, regret, "linear", "external", batch=B, seed=12) for _ in range(world)]
    for r, d in enumerate(devs):
        d.set_shard(r, world)
    n = devs[0].summary_bytes()
    gathered = torch.zeros(n * world, dtype=torch.uint8, device="cuda")
    ora = oracle.OracleSolver(g, regret, "linear", "external", batch=B, seed=12)
    for _ in range(3):
        for r, d in enumerate(devs):
            for s in range(window):
                d.window_local(gathered.data_ptr() + r * n, s == 0)
            d.sync()
        for d in devs:
            d.window_apply(gathered.data_ptr(), world)
            d.sync()
        ora.window_world(world, window)
        exp = ora.export()
        for d in devs:
            got = d.export()
            for f in ("visits", "regret", "weight", "payoff"):
                assert np.array_equal(got[f].view(np.uint32), exp[f].view(np.uint32)), f
    assert devs[0].epoch == 3 * window == ora.epoch


@pytest.mark.parametrize("kind", ["sinkhorn", "variation"])
def test_kmeans_two_shards_on_one_gpu_match_single(gpu, kind):
    import torch

    K, N, seed = 7, 300, 4
    if kind == "sinkhorn":
        bins, pts, tri = 32, flop_like_points(N, bins=32, mass=20, seed=seed), smooth_metric(32, seed)
    else:
        bins, pts, tri = 101, turn_like_points(N, bins=101, mass=46, seed=seed), None
    hp = oracle.default_sinkhorn()
    hp.iterations = 12
    cuts = [0, 131, N]
    shards = [lloyd.Layer(K, pts[cuts[r]:cuts[r + 1]], kind, tri, hp=hp, seed=seed) for r in range(2)]
    single = oracle.OracleKmeans(K, pts, kind, tri, hp=hp, seed=seed)
    # k-means++ across shards with the exact integer prefix over ranks
    for s in shards:
        s.kpp_begin()
    picks = []
    for k in range(K):
        totals = [s.kpp_total() for s in shards]
        r = rp_mulhi64(rp_stream(seed, k), sum(totals))
        owner, before = 0, 0
        while r >= before + totals[owner]:
            before += totals[owner]
            owner += 1
        idx = shards[owner].kpp_pick(r - before)
        hist = shards[owner].get_point(idx)
        picks.append(cuts[owner] + idx)
        for s in shards:
            s.set_centroid(k, hist)
            s.kpp_update(k)
    assert np.array_equal(np.array(picks, dtype=np.uint64), single.init_centroids())
    for s in shards:
        s.init_bounds()
    single.init_bounds()
    nb = shards[0].partial_bytes()
    bufs = [torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(2)]
    words32 = K * bins + K
    off64 = (words32 * 4 + 7) & ~7
    for _ in range(3):
        for s, b in zip(shards, bufs):
            s.step_local(b.data_ptr())
        torch.cuda.synchronize()
        red = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        red[: words32 * 4].view(torch.int32).copy_(bufs[0][: words32 * 4].view(torch.int32) + bufs[1][: words32 * 4].view(torch.int32))
        red[off64:].view(torch.int64).copy_(bufs[0][off64:].view(torch.int64) + bufs[1][off64:].view(torch.int64))
        torch.cuda.synchronize()
        outs = [s.step_finish(red.data_ptr()) for s in shards]
        d, sizes, _ = single.step()
        for od, osz, _ in outs:
            assert np.array_equal(od.view(np.uint32), d.view(np.uint32)) and np.array_equal(osz, sizes)
    sc, sw = single.centroids()
    sj, su, _ = single.bounds()
    for r, s in enumerate(shards):
        c, w = s.centroids()
        assert np.array_equal(c, sc) and np.array_equal(w, sw)
        j, u, _ = s.bounds()
        assert np.array_equal(j, sj[cuts[r]:cuts[r + 1]])
        # upper bounds: the reference's bits, or (interval-decided refresh, csrc/refresh_bound.hpp) an interval that contains them
        ref = su[cuts[r]:cuts[r + 1]]
        ulo, uiv = s.upper_interval()
        iv = uiv != 0
        assert np.array_equal(u[~iv].view(np.uint32), ref[~iv].view(np.uint32))
        assert np.all(ulo[iv] <= ref[iv]) and np.all(ref[iv] <= u[iv])


def test_native_rccl_comm_world_of_one(gpu):
    # rp_comm (csrc/comm.cpp): the library's own RCCL communicator, as a non-Python host would use it — unique id, join,
    # ncclAllGather / ncclAllReduce on the solver's and the layer's streams.  One GPU here, so a world of one rank: the
    # collectives are real RCCL calls, the results must be the world-model's (oracle) resp. the unsharded step's, bitwise.
    from robopoker_amd.parallel import Comm

    comm = Comm(0, 1, 0, Comm.unique_id())
    g = Game("leduc")
    dev = Solver(g, "linear", "linear", "external", batch=640, seed=3)
    ora = oracle.OracleSolver(g, "linear", "linear", "external", batch=640, seed=3)
    dev.step_comm(comm, steps=7, window=3)  # windows of 3, 3 and a trailing 1
    dev.sync()
    for w in (3, 3, 1):
        ora.window_world(1, w)
    got, exp = dev.export(), ora.export()
    for f in ("visits", "regret", "weight", "payoff"):
        assert np.array_equal(got[f].view(np.uint32), exp[f].view(np.uint32)), f
    assert dev.epoch == 7 and dev.counters() == ora.counters()
    # k-means: step_comm == step
    pts = flop_like_points(400, bins=32, mass=20, seed=2)
    tri = smooth_metric(32, 2)
    hp = oracle.default_sinkhorn()
    hp.iterations = 12
    a = lloyd.Layer(6, pts, "sinkhorn", tri, hp=hp, seed=2)
    b = lloyd.Layer(6, pts, "sinkhorn", tri, hp=hp, seed=2)
    for km in (a, b):
        km.init_centroids()
        km.init_bounds()
    for _ in range(3):
        d1, s1, m1 = a.step_comm(comm)
        d2, s2, m2 = b.step()
        assert np.array_equal(d1.view(np.uint32), d2.view(np.uint32)) and np.array_equal(s1, s2) and m1 == m2
    # NLHE: rp_nlhe_step_comm (step_local, the count + padded entry / key gathers, rank-major packing, step_apply inside one C
    # call) == the Python ShardedNlhe exchange == the oracle's world model, for a world of one: composed entries folded by key
    import oracle_nlmc as M
    from robopoker_amd.nlhe import NlheSolver

    n1 = NlheSolver(cap_log2=18, batch=96, seed=31)
    o1 = M.OracleNlhe(cap_log2=18, batch=96, seed=31)
    for step in range(3):
        n1.step_comm(comm, 1)
        n1.sync()
        o1.step_world(1)
        dm = {k: v for k, v in M.as_map(*n1.export()).items() if v["visits"][0] > 0}
        om = {k: v for k, v in M.as_map(*o1.export()).items() if v["visits"][0] > 0}
        assert dm.keys() == om.keys()
        for k in om:
            assert np.array_equal(dm[k]["visits"], om[k]["visits"]), (step, k)
            np.testing.assert_allclose(dm[k]["regret"], om[k]["regret"], rtol=2e-4, atol=5e-3)
            np.testing.assert_allclose(dm[k]["weight"], om[k]["weight"], rtol=2e-4, atol=1e-5)
        assert n1.epoch == o1.epoch == step + 1
        n1.load(*o1.export(), epoch=o1.epoch)
    n1.close()
    comm.close()

"""GPU parity of the NLHE MCCFR traversal (robopoker_amd/csrc/nlmc.hip, rp_nlhe_*) against oracle/rp_oracle_nlmc.c.

Integer state — which trees are sampled (hole cards, boards, opponent actions), every infoset key, the Decisions' order,
action counts and expanded masks, the counters, the visits — must be IDENTICAL.  The policy vector is the same float
operations on both sides: identical bits.  Regret vectors and infoset values: a batch of at most 2 048 trees (the reference's is
128) is evaluated by k_nl_tree in the reference's own order — per-leaf reach products from each walker node's child down
(flow.rs:166-216) — and equals the oracle BIT FOR BIT (test_small_batch_decisions_are_bit_exact); larger batches go through the
level-synchronous kernels' factorised evaluation D(node) = sum f(edge) D(child): the same real number, a different f32
association — stated tolerance rtol 2e-4 / atol 2e-3 chips (_same_batch is written for that path and holds for both)."""
import numpy as np
import pytest

import oracle_nlmc as M
from robopoker_amd.nlhe import NlheSolver

pytestmark = pytest.mark.gpu


def _same_batch(d, o):
    assert d["n"] == o["n"]
    n = d["n"]
    assert np.array_equal(d["tree"][:n], o["tree"][:n].astype(np.uint32))
    assert np.array_equal(d["n_actions"], o["n_actions"]) and np.array_equal(d["expanded"], o["expanded"])
    assert np.array_equal(d["policy"].view(np.uint32), o["policy"].view(np.uint32))
    np.testing.assert_allclose(d["regret"], o["regret"], rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(d["payoff"], o["payoff"], rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize("batch,seed", [(64, 5), (300, 12)])
def test_first_batch_equals_the_oracle(gpu, batch, seed):
    dev = NlheSolver(cap_log2=18, batch=batch, seed=seed)
    ora = M.OracleNlhe(cap_log2=18, batch=batch, seed=seed)
    d, o = dev.batch(), ora.batch()
    # rows differ between the two tables (insertion order); the infosets behind them must not
    past, present, choices, _ = ora.export()
    # the oracle's batch carries rows of ITS table: translate through its export order (slot order = row order)
    _same_batch(d, o)
    okeys = sorted(zip(past.tolist(), present.tolist(), choices.tolist()))
    dp, db, dc, _ = dev.export()
    assert sorted(zip(dp.tolist(), db.tolist(), dc.tolist())) == okeys
    # and Decision by Decision the same infoset
    omap = {}
    for k, (p, b, c) in enumerate(zip(past, present, choices)):
        omap[k] = (int(p), int(b), int(c))
    assert dev.counters()[2] == ora.counters()[2]


def _pruning_hyper(warmup=2, threshold=-5.0, explore=0.05):
    import oracle

    hp = oracle.default_hyper()
    hp.prune_warmup, hp.prune_threshold, hp.prune_explore = warmup, threshold, explore
    return hp


@pytest.mark.parametrize("sampling", ["pluribus", "prunable"])
def test_pruned_sampling_schemes_equal_the_oracle(gpu, sampling):
    # Flagship = Nlhe<LinearRegret, LinearWeight, PluribusSampling> (nlhe/src/lib.rs:86-90).  Pruning is forced to bite
    # (warm-up 2 epochs, threshold -5 instead of 16 384 / -3e5): the expanded masks — which walker edges survived, the explore
    # draws, the terminal-child exemption, the keep-all fallback (sample/pluribus.rs:72-101, pruning.rs:44-66) — the trees and
    # every key are the oracle's; regrets within the stated tolerance, with resynchronisation as above.
    batch = 160
    dev = NlheSolver(cap_log2=18, batch=batch, seed=33, sampling=sampling, hyper=_pruning_hyper())
    ora = M.OracleNlhe(cap_log2=18, batch=batch, seed=33, sampling=sampling, hyper=_pruning_hyper())
    pruned = 0
    for step in range(6):
        d, o = dev.batch(), ora.batch()
        _same_batch(d, o)
        full = (1 << d["n_actions"].astype(np.uint32)) - 1
        pruned += int((d["expanded"] != full).sum())
        dev.step("ordered")
        ora.step()
        assert dev.counters() == ora.counters()
        dev.load(*ora.export(), epoch=ora.epoch)
    assert pruned > 50


@pytest.mark.parametrize("sampling,batch", [("external", 128), ("pluribus", 300), ("external", 1)])
def test_tree_per_workgroup_traversal_equals_the_level_synchronous_one(gpu, monkeypatch, sampling, batch):
    # a batch of at most 2 048 trees is traversed by k_nl_tree (one tree per workgroup, one launch: the reference's batch of 128);
    # RP_NLHE_NODE_BUDGET keeps a handle on the level-synchronous kernels.  Same trees, keys, masks and policies bit for bit; the
    # regret vectors and payoffs within the batch-wide path's stated tolerance (k_nl_tree evaluates them in the reference's own
    # order — bit-exact against the oracle, test_small_batch_decisions_are_bit_exact — the level kernels in the factorised one)
    a = NlheSolver(cap_log2=18, batch=batch, seed=61, sampling=sampling, hyper=_pruning_hyper())
    monkeypatch.setenv("RP_NLHE_NODE_BUDGET", "4096")
    b = NlheSolver(cap_log2=18, batch=batch, seed=61, sampling=sampling, hyper=_pruning_hyper())
    monkeypatch.delenv("RP_NLHE_NODE_BUDGET")
    for step in range(4):
        x, y = a.batch(), b.batch()
        assert x["n"] == y["n"] and x["n"] > 0
        for k in ("tree", "past", "present", "choices", "n_actions", "expanded"):
            assert np.array_equal(x[k], y[k]), k
        assert np.array_equal(x["policy"].view(np.uint32), y["policy"].view(np.uint32))
        np.testing.assert_allclose(x["regret"], y["regret"], rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(x["payoff"], y["payoff"], rtol=2e-4, atol=2e-3)
        a.step("ordered")
        b.step("ordered")
        assert a.counters() == b.counters()
        assert a.last_shape() == b.last_shape()
        b.load(*a.export(), epoch=a.epoch)  # the two tables differ in the last bits from here on: resynchronise
    pa, pb = a.export(), b.export()
    ka = sorted(zip(pa[0].tolist(), pa[1].tolist(), pa[2].tolist()))
    kb = sorted(zip(pb[0].tolist(), pb[1].tolist(), pb[2].tolist()))
    assert ka == kb


@pytest.mark.parametrize("sampling,batch,rng", [("external", 128, "counter"), ("pluribus", 200, "counter"), ("pluribus", 128, "reference"),
                                                 ("prunable", 33, "counter")])
def test_small_batch_decisions_are_bit_exact(gpu, sampling, batch, rng):
    # At the reference's batch (one tree per workgroup, k_nl_tree) the values are computed in the reference's own order: per walker
    # node and edge reach * recursed_value with the (rel, smp) products carried from the node's child down to every leaf
    # (flow.rs:166-216).  EVERY Decisions field equals the oracle's bit for bit — the regret vectors and payoffs too — and so do
    # the tables after each step, with NO resynchronisation over eight steps
    dev = NlheSolver(cap_log2=18, batch=batch, seed=71, sampling=sampling, hyper=_pruning_hyper())
    ora = M.OracleNlhe(cap_log2=18, batch=batch, seed=71, sampling=sampling, hyper=_pruning_hyper())
    dev.set_rng(rng)
    ora.set_rng(rng)
    for step in range(8):
        d, o = dev.batch(), ora.batch()
        assert d["n"] == o["n"] and d["n"] > 0
        n = d["n"]
        assert np.array_equal(d["tree"][:n], o["tree"][:n].astype(np.uint32))
        assert np.array_equal(d["n_actions"], o["n_actions"]) and np.array_equal(d["expanded"], o["expanded"])
        for k in ("policy", "regret", "payoff"):
            assert np.array_equal(d[k].view(np.uint32), o[k].view(np.uint32)), (step, k)
        dev.step("ordered")
        ora.step()
        assert dev.counters() == ora.counters()
    dp, db, dc, de = dev.export()
    op, ob, oc, oe = ora.export()
    dk = {(int(a), int(b), int(c)): i for i, (a, b, c) in enumerate(zip(dp.tolist(), db.tolist(), dc.tolist()))}
    assert len(dk) == len(op)
    for j, key in enumerate(zip(op.tolist(), ob.tolist(), oc.tolist())):
        i = dk[(int(key[0]), int(key[1]), int(key[2]))]
        nact = (int(key[2]).bit_length() + 4) // 5  # Path: five bits per edge (kicker/src/path.rs:27-29)
        assert de[i][:nact].tobytes() == oe[j][:nact].tobytes(), key


@pytest.mark.parametrize("sampling,batch,budget", [("external", 300, "4096"), ("pluribus", 160, "4096"), ("external", 2500, None)])
def test_exact_order_on_the_level_synchronous_kernels(gpu, monkeypatch, sampling, batch, budget):
    # rp_nlhe_set_exact(1): the batch-wide kernels carry the per-ancestor reach rows too (nl_ex_child in k_nl_children, nl_ex_up_kids in
    # k_nl_up, nl_ex_walker in k_nl_fill) and their Decisions equal the oracle's bit for bit as well — forced onto those kernels by a
    # node budget at small batches, taken by size at 2 500 trees
    if budget:
        monkeypatch.setenv("RP_NLHE_NODE_BUDGET", budget)
    dev = NlheSolver(cap_log2=20, batch=batch, seed=83, sampling=sampling, hyper=_pruning_hyper())
    if budget:
        monkeypatch.delenv("RP_NLHE_NODE_BUDGET")
    dev.set_exact(True)
    ora = M.OracleNlhe(cap_log2=20, batch=batch, seed=83, sampling=sampling, hyper=_pruning_hyper())
    for step in range(3 if batch > 1000 else 5):
        d, o = dev.batch(), ora.batch()
        assert d["n"] == o["n"] and d["n"] > 0
        assert np.array_equal(d["n_actions"], o["n_actions"]) and np.array_equal(d["expanded"], o["expanded"])
        for k in ("policy", "regret", "payoff"):
            assert np.array_equal(d[k].view(np.uint32), o[k].view(np.uint32)), (step, k)
        dev.step("ordered")
        ora.step()
        assert dev.counters() == ora.counters()


@pytest.mark.parametrize("sampling", ["external", "pluribus"])
def test_reference_seed_mode_equals_the_oracle(gpu, sampling):
    # rp_nlhe_set_rng(RP_RNG_REFERENCE): the opponent's WeightedIndex draw and Pluribus' coin come from DefaultHasher(t, NlheInfo,
    # tree id) -> SmallRng (flow.rs:285-295; include/rp_refrng.h) on both sides: trees, keys, masks as the oracle's
    batch = 160
    dev = NlheSolver(cap_log2=18, batch=batch, seed=35, sampling=sampling, hyper=_pruning_hyper())
    ora = M.OracleNlhe(cap_log2=18, batch=batch, seed=35, sampling=sampling, hyper=_pruning_hyper())
    cnt = NlheSolver(cap_log2=18, batch=batch, seed=35, sampling=sampling, hyper=_pruning_hyper())
    dev.set_rng("reference")
    ora.set_rng("reference")
    differs = False
    for step in range(4):
        d, o, c = dev.batch(), ora.batch(), cnt.batch()
        _same_batch(d, o)
        differs = differs or c["n"] != d["n"] or not np.array_equal(c["past"], d["past"]) or not np.array_equal(c["expanded"], d["expanded"])
        dev.step("ordered")
        ora.step()
        cnt.step("ordered")
        assert dev.counters() == ora.counters()
        dev.load(*ora.export(), epoch=ora.epoch)
        cnt.load(*ora.export(), epoch=ora.epoch)
    assert differs  # and it is not the counter hash that drew them


def test_steps_without_resynchronisation_stay_close_to_the_oracle(gpu):
    # the NLHE counterpart of test_composed_drift.py, on the device: NO resynchronisation.  While the two runs still sample the
    # same trees (every Decisions of a batch on the same infoset) the visits agree exactly.  Regrets and weights carry the
    # per-step re-association error (2e-4) forward through sigma and q of every edge on a path, and a regret is a sum with
    # cancellation, so the statement is per infoset ROW, relative to the row's largest magnitude: 99 % of the entries within
    # 5e-4 x steps, 99.9 % within 5e-3 x steps; the few beyond are rows whose positive regrets nearly cancel (regret matching
    # divides by their sum: ill-conditioned there, for the reference's own arithmetic too; observed up to 0.4 of the row's
    # scale after six steps, no bound is claimed).  The first step at which a sampled edge flips on a rounding difference
    # ends the comparison (from then on they are two different, equally valid runs) — it must not be among the first two.
    dev = NlheSolver(cap_log2=18, regret="linear", weight="linear", batch=128, seed=44)
    ora = M.OracleNlhe(cap_log2=18, regret="linear", weight="linear", batch=128, seed=44)
    import ctypes as C

    def oracle_keys(o):
        out = []
        for r in o["row"]:
            kp, kb, kc = C.c_uint64(), C.c_uint32(), C.c_uint64()
            assert M.lib().ora_nlmc_row_key(ora._h, int(r), C.byref(kp), C.byref(kb), C.byref(kc)) == 0
            out.append((kp.value, kb.value, kc.value))
        return out

    agreed = 0
    for step in range(6):
        # the same trees?  every Decisions of the batch on the same infoset, in the same order (a flipped opponent action changes
        # the subgame path of every infoset below it)
        d, o = dev.batch(), ora.batch()
        if d["n"] != o["n"] or list(zip(d["past"].tolist(), d["present"].tolist(), d["choices"].tolist())) != oracle_keys(o):
            break
        dev.step("ordered")
        ora.step()
        assert dev.counters() == ora.counters()
        dm, om = M.as_map(*dev.export()), M.as_map(*ora.export())
        assert dm.keys() == om.keys()
        rel = []
        for k in om:
            assert np.array_equal(dm[k]["visits"], om[k]["visits"]), (step, k)
            for f in ("regret", "weight"):
                scale = float(np.abs(om[k][f]).max()) + (1.0 if f == "regret" else 1e-3)
                rel.append(np.abs(dm[k][f] - om[k][f]) / scale)
        rel = np.concatenate(rel)
        assert np.quantile(rel, 0.99) <= 5e-4 * (step + 1), (step, float(np.quantile(rel, 0.99)))
        assert np.quantile(rel, 0.999) <= 5e-3 * (step + 1), (step, float(np.quantile(rel, 0.999)))
        agreed += 1
    assert agreed >= 2


def test_every_applied_action_is_legal_in_the_checking_mode(gpu, monkeypatch):
    # Game::apply panics on an illegal action (kicker/src/game.rs:234-247).  The device evaluates is_allowed on every applied
    # action only when RP_NLHE_CHECK_LEGAL=1 (it is a third of the traversal's time and can only fire on an engine bug): with
    # the check on, two steps at a batch of a few thousand trees raise no error and produce the unchecked run's table
    monkeypatch.setenv("RP_NLHE_CHECK_LEGAL", "1")
    a = NlheSolver(cap_log2=20, batch=4096, seed=7)
    monkeypatch.delenv("RP_NLHE_CHECK_LEGAL")
    b = NlheSolver(cap_log2=20, batch=4096, seed=7)
    for _ in range(2):
        a.step("ordered")
        b.step("ordered")
    assert a.counters() == b.counters()
    am, bm = M.as_map(*a.export()), M.as_map(*b.export())
    assert am.keys() == bm.keys() and all(am[k].tobytes() == bm[k].tobytes() for k in am)


def test_steps_with_resynchronisation_track_the_oracle(gpu):
    # four Solver::steps (both walkers twice).  After each: the same infosets, identical visits, regrets / weights / payoffs
    # within the stated tolerance; then the device table is overwritten with the oracle's so the NEXT step samples the same
    # opponent actions on both sides (a sampled edge is a threshold test on f32 weights).
    batch = 128  # the reference's batch_size (nlhe/src/solver.rs:11)
    dev = NlheSolver(cap_log2=18, regret="linear", weight="linear", batch=batch, seed=21)
    ora = M.OracleNlhe(cap_log2=18, regret="linear", weight="linear", batch=batch, seed=21)
    for step in range(4):
        _same_batch(dev.batch(), ora.batch())
        dev.step("ordered")
        ora.step()
        dm, om = M.as_map(*dev.export()), M.as_map(*ora.export())
        assert dm.keys() == om.keys()
        for k in om:
            assert np.array_equal(dm[k]["visits"], om[k]["visits"]), (step, k)
            np.testing.assert_allclose(dm[k]["regret"], om[k]["regret"], rtol=2e-4, atol=5e-3)
            np.testing.assert_allclose(dm[k]["weight"], om[k]["weight"], rtol=2e-4, atol=1e-5)
            np.testing.assert_allclose(dm[k]["payoff"], om[k]["payoff"], rtol=2e-4, atol=5e-3)
        assert dev.counters() == ora.counters() and dev.epoch == ora.epoch == step + 1
        dev.load(*ora.export(), epoch=ora.epoch)


def test_large_batch_runs_and_conserves_visits(gpu):
    # a GPU-sized batch (the reference's 128 trees fill two wavefronts): every Decisions lands on its infoset exactly once
    dev = NlheSolver(cap_log2=21, batch=8192, seed=2)
    for _ in range(2):
        dev.step("composed")
    nodes, infos, keys = dev.counters()
    past, present, choices, enc = dev.export()
    assert len(past) == keys and int(enc["visits"][:, 0].sum()) == infos
    assert 100 * 8192 * 2 < nodes < 1500 * 8192 * 2 and dev.epoch == 2


def _hash_buckets(obs, street):
    """oracle/rp_oracle_nlmc.c ora_nlmc_hash_bucket on a CUDA int64 tensor of canonical observations (two's complement
    arithmetic wraps like u64; shifts made logical by masking)"""
    import torch

    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    def signed(c):
        return c - (1 << 64) if c >= (1 << 63) else c

    def mix64(z):
        z = z ^ lsr(z, 30)
        z = z * signed(0xbf58476d1ce4e5b9)
        z = z ^ lsr(z, 27)
        z = z * signed(0x94d049bb133111eb)
        return z ^ lsr(z, 31)

    nb = (169, 256, 256, 101)[street]
    z = mix64(obs ^ signed((0x51ed270b5 * (street + 1)) & ((1 << 64) - 1)))
    half = lsr(z, 1)  # floor(u / 2) of the unsigned value
    return (((half % nb) * 2 + (z & 1)) % nb).to(torch.uint8)


def test_table_encoder_equals_the_hash_encoder_on_hash_tables(gpu):
    # NlheEncoder over the pipeline's Lookup tables (all four streets: 169 / 1 286 792 / 13 960 050 / 123 156 254
    # isomorphisms): filled with the hash encoder's buckets, the table-driven traversal must produce the hash-driven one
    import torch  # noqa: F401

    from robopoker_amd import deuce

    tables = []
    for street, name in enumerate(("pref", "flop", "turn", "rive")):
        obs = deuce.isomorphisms(name)
        tables.append(deuce.Lookup(name, obs, _hash_buckets(obs, street)))
        del obs
    a = NlheSolver(cap_log2=18, batch=256, seed=9)
    b = NlheSolver(cap_log2=18, batch=256, seed=9, tables=tables)
    da, db = a.batch(), b.batch()
    assert da["n"] == db["n"] > 0
    for f in ("tree", "past", "present", "choices", "n_actions", "expanded"):
        assert np.array_equal(da[f], db[f]), f
    for f in ("regret", "policy", "payoff"):
        assert np.array_equal(da[f].view(np.uint32), db[f].view(np.uint32)), f
    a.close()
    b.close()
    for t in tables:
        t.close()


def test_blueprint_file_roundtrip_through_the_device(gpu, tmp_path):
    # train a little, stream the profile in the reference's COPY row format, hydrate a fresh solver from the file: the next
    # batch (which depends on every regret and weight through sampling and regret matching) must be identical
    from robopoker_amd import formats

    a = NlheSolver(cap_log2=18, batch=256, seed=4)
    for _ in range(3):
        a.step("ordered")
    path = str(tmp_path / "blueprint.pgcopy")
    rows = formats.write_blueprint(path, *a.export(), only_visited=False)
    assert rows > 1000
    b = NlheSolver(cap_log2=18, batch=256, seed=4)
    b.load(*formats.read_blueprint(path), epoch=a.epoch)
    da, db = a.batch(), b.batch()
    assert da["n"] == db["n"]
    for f in ("tree", "past", "present", "choices", "n_actions", "expanded"):
        assert np.array_equal(da[f], db[f]), f
    for f in ("regret", "policy", "payoff"):
        assert np.array_equal(da[f].view(np.uint32), db[f].view(np.uint32)), f


@pytest.mark.parametrize("world,batch", [(2, 96), (4, 64)])
def test_tree_shards_exchanged_by_key_match_the_world_model(gpu, world, batch):
    # BASELINE configs[3] on several GPUs: rank r traverses trees [r*B, (r+1)*B) of a world*B-tree epoch, the per-infoset
    # entries travel BY KEY (each table numbers its rows in its own insertion order) and every replica folds all of them in
    # rank order.  `world` handles on one device, buffers concatenated by hand; against ora_nlmc_step_world.  Same tolerance
    # and the same resynchronisation as the single-rank test above.
    import torch

    devs = [NlheSolver(cap_log2=18, batch=batch, seed=31) for _ in range(world)]
    for r, d in enumerate(devs):
        d.set_shard(r, world)
    eb, cap = devs[0].entry_bytes()
    ora = M.OracleNlhe(cap_log2=18, batch=batch, seed=31)
    mk = lambda n, dt: torch.zeros(n, dtype=dt, device="cuda")  # noqa: E731
    bufs = {"ent": mk(cap * eb * world, torch.uint8), "past": mk(cap * world, torch.int64),
            "present": mk(cap * world, torch.int32), "choices": mk(cap * world, torch.int64)}
    unit = {"ent": eb, "past": 8, "present": 4, "choices": 8}
    for step in range(3):
        off = 0
        for d in devs:  # rank-major: rank r's entries right behind rank r-1's
            n = d.step_local(*[bufs[k].data_ptr() + off * unit[k] for k in ("ent", "past", "present", "choices")])
            off += n
            d.sync()  # each handle runs on a stream of its own; in a real job the collective's stream orders the ranks
        assert off > 0
        for d in devs:  # step_apply rewrites the entries' row fields for ITS table, in place: one replica after the other
            d.step_apply(*[bufs[k].data_ptr() for k in ("ent", "past", "present", "choices")], off)
            d.sync()
        ora.step_world(world)
        om = {k: v for k, v in M.as_map(*ora.export()).items() if v["visits"][0] > 0}
        for d in devs:
            dm = {k: v for k, v in M.as_map(*d.export()).items() if v["visits"][0] > 0}
            assert dm.keys() == om.keys()
            for k in om:
                assert np.array_equal(dm[k]["visits"], om[k]["visits"]), (step, k)
                np.testing.assert_allclose(dm[k]["regret"], om[k]["regret"], rtol=2e-4, atol=5e-3)
                np.testing.assert_allclose(dm[k]["weight"], om[k]["weight"], rtol=2e-4, atol=1e-5)
                np.testing.assert_allclose(dm[k]["payoff"], om[k]["payoff"], rtol=2e-4, atol=5e-3)
            assert d.epoch == ora.epoch == step + 1
        for d in devs:
            d.load(*ora.export(), epoch=ora.epoch)
    for d in devs:
        d.close()


def test_twin_solvers_at_a_gpu_sized_batch_are_identical(gpu):
    # two solvers, same seed, 131 072 trees per step: which lane meets which table slot first (and which index a node gets) is timing,
    # the tables must not be — every infoset, visit, regret, weight and payoff bit for bit.  (This is the size at which a handle
    # whose zero-initialisation was still running when its first step started lost infosets: the create functions now
    # synchronise the device.)
    res = []
    for _ in range(2):
        s = NlheSolver(cap_log2=24, batch=131072, seed=77)
        for _ in range(2):
            s.step("composed")
        past, present, choices, enc = s.export()
        order = np.lexsort((choices, present, past))
        res.append((s.counters(), past[order], present[order], choices[order], enc[order]))
        s.close()
    a, b = res
    assert a[0] == b[0] and a[0][1] > 131072 * 2 * 30
    for i in (1, 2, 3):
        assert np.array_equal(a[i], b[i])
    for f in ("visits", "regret", "weight", "payoff"):
        assert np.array_equal(a[4][f].view(np.uint32), b[4][f].view(np.uint32)), f


@pytest.mark.parametrize("batch,seed", [(1, 2), (65, 3), (257, 4)])
def test_ragged_batches_equal_the_oracle(gpu, batch, seed):
    # one tree (a single lane of a single tile per level), one tree past a wavefront, one past a workgroup: two steps each,
    # both walkers, with resynchronisation — the level-synchronous kernels' ragged edges
    dev = NlheSolver(cap_log2=16, batch=batch, seed=seed)
    ora = M.OracleNlhe(cap_log2=16, batch=batch, seed=seed)
    for _ in range(2):
        _same_batch(dev.batch(), ora.batch())
        dev.step("ordered")
        ora.step()
        assert dev.counters() == ora.counters()
        dev.load(*ora.export(), epoch=ora.epoch)


def test_a_full_infoset_table_fails_the_step_and_the_error_does_not_stick(gpu):
    # 256 rows cannot hold the infosets of 200 trees: the step reports RP_ERR_CAPACITY (flag 32) instead of hanging in the probe
    # loop or corrupting rows; the error belongs to that launch — a solver with room, created afterwards in the same process,
    # and the failed handle's own counters are unaffected
    from robopoker_amd import _lib

    small = NlheSolver(cap_log2=8, batch=200, seed=1)
    with pytest.raises(_lib.RpError) as e:
        small.step("ordered")
    assert e.value.code == 5 and "32" in str(e.value)
    assert small.counters()[:2] == (0, 0)
    ok = NlheSolver(cap_log2=18, batch=200, seed=1)
    ok.step("ordered")
    assert ok.counters()[1] > 200 * 20
    small.close()
    ok.close()


def test_trainer_loop_over_the_nlhe_solver(gpu):
    # Trainer::train (forge/src/trainer.rs:18-66) over the Flagship solver type: a checkpoint every step (log interval 0),
    # Checkpoint's display line (metrics/checkpoint.rs:39-50), rate = new infos / max(1, whole seconds), Progress::summary; the
    # table is the one plain step() calls produce; the interrupt flag stops the loop after the step in flight
    import ctypes

    a = NlheSolver(cap_log2=18, batch=128, seed=6, sampling="pluribus")
    b = NlheSolver(cap_log2=18, batch=128, seed=6, sampling="pluribus")
    seen, flushes = [], []
    summary = a.train("composed", max_steps=5, log_interval=0.0, flush_interval=0.0,
                      on_checkpoint=lambda cp, line: seen.append((cp, line)), on_flush=lambda cp: flushes.append(cp))
    for _ in range(5):
        b.step("composed")
    assert a.counters() == b.counters() and a.epoch == b.epoch == 5
    am, bm = M.as_map(*a.export()), M.as_map(*b.export())
    assert am.keys() == bm.keys() and all(am[k].tobytes() == bm[k].tobytes() for k in am)
    assert len(seen) == 5 and len(flushes) == 5 and [cp["epoch"] for cp, _ in seen] == [1, 2, 3, 4, 5]
    prev = 0
    for cp, line in seen:
        assert cp["rate"] == float(cp["infos"] - prev)
        prev = cp["infos"]
        assert line == "".join(f"{x:<20}" for x in (f"batch {cp['epoch']}", f"nodes {cp['nodes']}", f"infos {cp['infos']}", f"I/sec {cp['rate']:.1f}"))
    nodes, infos, _ = a.counters()
    assert summary == "training stopped\n" + "".join(f"{x:<20}" for x in ("batch 5", f"nodes {nodes}", f"infos {infos}", f"I/sec {float(infos):.1f}"))
    stop = ctypes.c_int(1)
    a.train("composed", interrupt=stop)
    assert a.epoch == 6


def test_a_batch_traversed_in_several_passes_is_the_same_batch(gpu, monkeypatch):
    # The node arrays hold 1 536 nodes per tree of the batch.  Trees grow with training; a pass that runs out of nodes makes the
    # library traverse the batch in twice as many passes from then on (a pass = a contiguous range of tree ids) instead of failing
    # the step.  Forced here two ways — RP_NLHE_CHUNKS=3 from the start, and a budget of 200 nodes per tree (below the average
    # tree: the first step overflows and retries) — against a solver with room: every Decisions, in order, bit for bit; the
    # counters; the tables after three steps.
    monkeypatch.setenv("RP_NLHE_NODE_BUDGET", "4096")  # room to spare, on the level-synchronous kernels like the other three (a
    room = NlheSolver(cap_log2=18, batch=300, seed=19)  # default handle of this size would take k_nl_tree, whose values are exact)
    monkeypatch.setenv("RP_NLHE_CHUNKS", "3")
    three = NlheSolver(cap_log2=18, batch=300, seed=19)
    monkeypatch.delenv("RP_NLHE_CHUNKS")
    monkeypatch.setenv("RP_NLHE_NODE_BUDGET", "200")
    tight = NlheSolver(cap_log2=18, batch=300, seed=19)
    # ... and with nodes to spare but walker-node arrays (wl / ws / gdesc) a 64th of them: the partition kernels raise the same
    # flag instead of writing past those arrays (round 4; before, only the host's check AFTER the launches refused the pass)
    monkeypatch.setenv("RP_NLHE_NODE_BUDGET", "1536,64")
    narrow = NlheSolver(cap_log2=18, batch=300, seed=19)
    monkeypatch.delenv("RP_NLHE_NODE_BUDGET")
    for step in range(3):
        a = room.batch()
        for other in (three, tight, narrow):
            b = other.batch()
            assert a["n"] == b["n"]
            for f in ("tree", "past", "present", "choices", "n_actions", "expanded"):
                assert np.array_equal(a[f], b[f]), (step, f)
            for f in ("regret", "policy", "payoff"):
                assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), (step, f)
        for s in (room, three, tight, narrow):
            s.step("ordered")
        assert room.counters() == three.counters() == tight.counters() == narrow.counters()
    am = M.as_map(*room.export())
    for other in (three, tight, narrow):
        bm = M.as_map(*other.export())
        assert am.keys() == bm.keys() and all(am[k].tobytes() == bm[k].tobytes() for k in am)

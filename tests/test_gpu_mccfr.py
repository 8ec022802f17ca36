"""GPU parity: the HIP MCCFR path (through the C-ABI) against the CPU oracle — bit-exact.

The ordered update applies every touch of a table cell sequentially in tree-id order, so the regret /
weight / payoff / visit tables must be IDENTICAL to the oracle's, not merely close.  Parity at
BASELINE's full batch is covered through size-independent properties (visit conservation, determinism,
convergence thresholds of the reference's own tests).
"""
import ctypes as C

import numpy as np
import pytest

import oracle
from robopoker_amd import Game, _lib
from robopoker_amd.mccfr import Solver

pytestmark = pytest.mark.gpu


def assert_tables_equal(a: np.ndarray, b: np.ndarray):
    for f in ("visits", "regret", "weight", "payoff"):
        if a[f].dtype.kind == "f":  # equal bits would also be equal NaNs: a table never holds one
            assert not np.isnan(a[f]).any(), f"NaN in {f}"
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f"{f} differs bitwise"


def test_arithmetic_contract_on_device(gpu):
    # rp_math.h primitives, IEEE division / sqrt / fma and u32->f32 conversion: device == host, bit for bit
    rng = np.random.default_rng(0)
    n = 1 << 18
    x = np.concatenate([rng.uniform(-90, 90, n // 2), rng.standard_normal(n // 4) * 1e-3,
                        np.exp(rng.uniform(-80, 80, n // 4))]).astype(np.float32)
    y = np.concatenate([rng.uniform(-5, 5, n // 2), np.exp(rng.uniform(-60, 60, n // 2))]).astype(np.float32)
    y[y == 0] = 1.0
    dev = np.zeros(6 * n, dtype=np.float32)
    _lib.check(_lib.load().rp_math_selftest(0, n, x.ctypes.data, y.ctypes.data, dev.ctypes.data))
    host = np.zeros(6 * n, dtype=np.float32)
    o = oracle.load()
    o.ora_math_selftest.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    o.ora_math_selftest(n, x.ctypes.data, y.ctypes.data, host.ctypes.data)
    names = ["expf", "logf", "div", "sqrt", "fma", "u32->f32"]
    for k, nm in enumerate(names):
        d, h = dev[k * n:(k + 1) * n].view(np.uint32), host[k * n:(k + 1) * n].view(np.uint32)
        assert np.array_equal(d, h), f"{nm}: {np.count_nonzero(d != h)} of {n} differ"


def test_device_exp_spellings_equal_the_spec_for_every_f32(gpu):
    # rp_expf / rp_exp_floor / rp_exp_floor2 (v_med3, v_rndne, v_ldexp, packed fma) vs the contract's spec sequence,
    # all 2^32 bit patterns evaluated on the device
    bad = (C.c_uint64 * 4)()
    _lib.check(_lib.load().rp_math_exp_sweep(0, bad))
    assert list(bad)[:3] == [0, 0, 0], f"mismatches {list(bad)[:3]}, first at bits {bad[3]:#010x}"


@pytest.mark.parametrize("game", ["kuhn", "leduc", "rps"])
@pytest.mark.parametrize("regret,weight,sampling", [
    ("floored", "linear", "external"), ("linear", "linear", "pluribus"), ("summed", "constant", "external"),
    ("discounted", "quadratic", "prunable"), ("asymmetric", "exponential", "external")])
def test_tables_bit_exact_vs_oracle(gpu, game, regret, weight, sampling):
    g = Game(game)
    hp = oracle.default_hyper()
    hp.prune_warmup = 3          # let Pluribus pruning engage inside the test
    hp.prune_threshold = -2.0
    B, steps = 333, 12           # ragged batch: not a multiple of the wave / chunk sizes
    dev = Solver(g, regret, weight, sampling, batch=B, seed=42, hyper=hp)
    ora = oracle.OracleSolver(g, regret, weight, sampling, batch=B, seed=42, hyper=hp)
    for s in range(steps):
        dev.step()
        ora.step()
        assert_tables_equal(dev.export(), ora.export())
    assert dev.epoch == ora.epoch == steps
    assert dev.counters() == ora.counters()
    assert dev.exploitability() == ora.exploitability()
    assert dev.sum_regret() == ora.sum_regret()
    for info in range(g.n_infos):
        for kind in ("iterated", "averaged", "sampling"):
            assert np.array_equal(dev.policy(info, kind), ora.policy(info, kind))


def test_hbm_scratch_traversal_variant_is_identical(gpu, monkeypatch):
    # the generic traversal kernel (per-tree scratch in HBM, any game size) and the LDS-resident one used for
    # small games must produce the same Decisions
    g = Game("leduc")
    hp = oracle.default_hyper()
    hp.prune_warmup = 2
    hp.prune_threshold = -2.0
    a = Solver(g, "linear", "linear", "pluribus", batch=999, seed=77, hyper=hp)
    monkeypatch.setenv("RP_MCCFR_HBM_SCRATCH", "1")
    b = Solver(g, "linear", "linear", "pluribus", batch=999, seed=77, hyper=hp)
    monkeypatch.delenv("RP_MCCFR_HBM_SCRATCH")
    ora = oracle.OracleSolver(g, "linear", "linear", "pluribus", batch=999, seed=77, hyper=hp)
    for _ in range(8):
        a.step()
        b.step()
        ora.step()
        assert_tables_equal(a.export(), ora.export())
        assert_tables_equal(b.export(), ora.export())
    assert a.counters() == b.counters() == ora.counters()


@pytest.mark.parametrize("game,batch", [("leduc", 1500), ("kuhn", 700), ("leduc_wide", 900)])
@pytest.mark.parametrize("regret,weight", [("linear", "linear"), ("floored", "quadratic")])
def test_static_skeleton_traversal_equals_generic_and_oracle(gpu, monkeypatch, game, batch, regret, weight):
    # csrc/traverse_static.hpp: Kuhn's and Leduc's traversal is instantiated over the game's compile-time action skeleton
    # (external sampling); RP_TRAV_GENERIC=1 keeps the per-lane DFS kernel.  Both against the oracle, bit for bit, both
    # walkers, long enough for the opponent's sampling weights to leave their defaults.
    g = Game(game)
    a = Solver(g, regret, weight, "external", batch=batch, seed=41)
    monkeypatch.setenv("RP_TRAV_GENERIC", "1")
    b = Solver(g, regret, weight, "external", batch=batch, seed=41)
    monkeypatch.delenv("RP_TRAV_GENERIC")
    ora = oracle.OracleSolver(g, regret, weight, "external", batch=batch, seed=41)
    for _ in range(10):
        a.step()
        b.step()
        ora.step()
        assert_tables_equal(a.export(), ora.export())
        assert_tables_equal(b.export(), ora.export())
    assert a.counters() == b.counters() == ora.counters()
    assert a.kernel_variant() == "static" and b.kernel_variant() != "static"


@pytest.mark.parametrize("game,batch", [("leduc", 1500), ("kuhn", 300), ("leduc", 256), ("kuhn", 5000)])
@pytest.mark.parametrize("regret,weight", [("linear", "linear"), ("floored", "quadratic"), ("summed", "exponential")])
def test_fused_traversal_and_block_maps_equal_the_unfused_kernels_and_the_oracle(gpu, monkeypatch, game, batch, regret, weight):
    # composed update: k_traverse_maps_static composes a chunk's block maps from the Decisions while they are still in
    # registers / LDS (nothing goes through HBM); RP_TRAV_UNFUSED=1 keeps k_traverse_static + k_chunk_maps, RP_TRAV_GENERIC=1
    # the per-lane DFS + k_chunk_maps.  All three against the oracle's blocked composition, bit for bit; ragged last chunk,
    # a single full chunk, many chunks.
    g = Game(game)
    fused = Solver(g, regret, weight, "external", batch=batch, seed=17)
    monkeypatch.setenv("RP_TRAV_UNFUSED", "1")
    unfused = Solver(g, regret, weight, "external", batch=batch, seed=17)
    monkeypatch.delenv("RP_TRAV_UNFUSED")
    monkeypatch.setenv("RP_TRAV_GENERIC", "1")
    generic = Solver(g, regret, weight, "external", batch=batch, seed=17)
    monkeypatch.delenv("RP_TRAV_GENERIC")
    ora = oracle.OracleSolver(g, regret, weight, "external", batch=batch, seed=17)
    devs = (fused, unfused, generic)
    for d in devs:
        d.set_update_mode("composed")
    for _ in range(8):
        ora.step_world(1)
        for d in devs:
            d.step()
            assert_tables_equal(d.export(), ora.export())
    assert fused.counters() == unfused.counters() == generic.counters() == ora.counters()


def test_bench_sized_composed_step_equals_the_oracle(gpu):
    # bench.py's default step (2^23 trees) and a little more: 36 865 chunks = 577 groups of RP_FOLD_GROUP block maps per
    # infoset, i.e. 73 parts per infoset in k_combine2 and a final fold that runs through more than one LDS tile of group
    # maps, a ragged last chunk.  Both walkers, bit for bit against the oracle's blocked composition.
    g = Game("leduc")
    batch = (1 << 23) + (1 << 20) + 77
    dev = Solver(g, "floored", "linear", "external", batch=batch, seed=2026)
    dev.set_update_mode("composed")
    ora = oracle.OracleSolver(g, "floored", "linear", "external", batch=batch, seed=2026)
    for _ in range(2):
        dev.step()
        ora.step_world(1)
        assert_tables_equal(dev.export(), ora.export())
    assert dev.counters() == ora.counters() and dev.kernel_variant() == "static"


@pytest.mark.parametrize("mode", ["ordered", "composed"])
@pytest.mark.parametrize("overrides", [
    dict(temperature=0.5, smoothing=0.25, curiosity=0.2),         # sampling distribution far from the defaults
    dict(temperature=4.0, smoothing=8.0, curiosity=0.001),
    dict(prune_threshold=-5.0, prune_explore=0.5, prune_warmup=2),  # pruning bites within a few epochs
    dict(regret_min=-2.0),                                          # the Linear/Discounted floor is hit
])
def test_hyper_parameter_corners_bit_exact(gpu, mode, overrides):
    # every constant of rp_hyper (flow.rs:24-59 sampling, pruning.rs / pluribus.rs, regret/mod.rs floor) moves the tables
    g = Game("leduc")
    hp = oracle.default_hyper()
    for k, v in overrides.items():
        setattr(hp, k, v)
    regret, sampling = ("linear", "pluribus") if "prune_threshold" in overrides else ("linear", "external")
    dev = Solver(g, regret, "linear", sampling, batch=700, seed=9, hyper=hp)
    ora = oracle.OracleSolver(g, regret, "linear", sampling, batch=700, seed=9, hyper=hp)
    if mode == "composed":
        dev.set_update_mode("composed")
    for _ in range(8):
        dev.step()
        if mode == "composed":
            ora.step_world(1)
        else:
            ora.step()
    a, b = dev.export(), ora.export()
    for f in ("visits", "regret", "weight", "payoff"):
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f


@pytest.mark.parametrize("mode", ["ordered", "composed"])
@pytest.mark.parametrize("game,regret,weight,sampling", [("leduc", "discounted", "linear", "pluribus"),
                                                        ("kuhn", "floored", "quadratic", "prunable"),
                                                        ("leduc", "linear", "exponential", "external")])
def test_long_run_stays_bit_exact(gpu, mode, game, regret, weight, sampling):
    # 300 epochs with a ragged batch: tables diverge chaotically after the first differing bit, so equality after
    # hundreds of epochs means every sampled tree, every regret vector and every update agreed along the way
    if mode == "composed" and regret == "discounted":
        pytest.skip("sign-dependent discount: ordered mode only")
    g = Game(game)
    hp = oracle.default_hyper()
    hp.prune_warmup = 40
    dev = Solver(g, regret, weight, sampling, batch=257, seed=1234, hyper=hp)
    ora = oracle.OracleSolver(g, regret, weight, sampling, batch=257, seed=1234, hyper=hp)
    if mode == "composed":
        dev.set_update_mode("composed")
    for _ in range(300):
        dev.step_async(1)
        if mode == "composed":
            ora.step_world(1)
        else:
            ora.step()
    dev.sync()
    a, b = dev.export(), ora.export()
    for f in ("visits", "regret", "weight", "payoff"):
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f
    assert dev.counters() == ora.counters()


@pytest.mark.parametrize("mode", ["ordered", "composed"])
@pytest.mark.parametrize("regret,weight,sampling", [("floored", "linear", "external"), ("linear", "quadratic", "pluribus")])
def test_wide_leduc_takes_the_large_game_path_bit_exact(gpu, mode, regret, weight, sampling):
    # 616 infosets (> 256): per-infoset slot map, k_count / k_compact, k_block_maps over (infoset, chunk) groups
    g = Game("leduc_wide")
    assert g.n_infos == 616
    hp = oracle.default_hyper()
    hp.prune_warmup = 3
    dev = Solver(g, regret, weight, sampling, batch=1500, seed=77, hyper=hp)
    ora = oracle.OracleSolver(g, regret, weight, sampling, batch=1500, seed=77, hyper=hp)
    if mode == "composed":
        dev.set_update_mode("composed")
    for _ in range(10):
        dev.step()
        if mode == "composed":
            ora.step_world(1)
        else:
            ora.step()
    a, b = dev.export(), ora.export()
    for f in ("visits", "regret", "weight", "payoff"):
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f
    assert dev.counters() == ora.counters()


def test_slotmap_sort_variant_for_large_games_is_identical(gpu, monkeypatch):
    # games with more than 256 infosets sort their Decisions through a per-infoset slot map; force that path on Leduc
    g = Game("leduc")
    monkeypatch.setenv("RP_MCCFR_SLOTMAP", "1")
    a = Solver(g, "linear", "linear", "pluribus", batch=3000, seed=5)
    monkeypatch.delenv("RP_MCCFR_SLOTMAP")
    b = Solver(g, "linear", "linear", "pluribus", batch=3000, seed=5)
    for k in range(6):
        if k == 3:  # both update modes go through the sorted segments on the large-game path
            a.set_update_mode("composed")
            b.set_update_mode("composed")
        a.step()
        b.step()
    ra, rb = a.export(), b.export()
    for f in ("visits", "regret", "weight", "payoff"):
        assert np.array_equal(ra[f].view(np.uint32), rb[f].view(np.uint32)), f


@pytest.mark.parametrize("batch", [1, 63, 64, 65, 1024, 1025, 5000])
def test_batch_size_edges(gpu, batch):
    g = Game("leduc")
    dev = Solver(g, "linear", "linear", "external", batch=batch, seed=7)
    ora = oracle.OracleSolver(g, "linear", "linear", "external", batch=batch, seed=7)
    for _ in range(3):
        dev.step()
        ora.step()
    assert_tables_equal(dev.export(), ora.export())


def test_import_export_roundtrip_and_resume(gpu):
    g = Game("kuhn")
    a = Solver(g, "floored", "linear", "external", batch=128, seed=3)
    a.solve(128 * 20)
    rows, epoch = a.export(), a.epoch
    b = Solver(g, "floored", "linear", "external", batch=128, seed=3)
    b.load_rows(rows, epoch)          # Flagship::hydrate shape (nlhe/src/profile.rs:97-140)
    assert_tables_equal(b.export(), rows)
    a.solve(128 * 5)
    b.solve(128 * 5)
    assert_tables_equal(a.export(), b.export())
    e = a.get(g.info_id("K|B"), 1)
    assert e.visits == rows["visits"][g.info_id("K|B") * 2 + 1] + (a.export()["visits"] - rows["visits"])[g.info_id("K|B") * 2 + 1]


@pytest.mark.parametrize("game,regret,weight,sampling,batch", [
    ("leduc", "linear", "linear", "external", 777), ("leduc", "floored", "linear", "external", 6000),
    ("leduc", "summed", "exponential", "pluribus", 3001), ("rps", "floored", "quadratic", "external", 2000),
    ("kuhn", "linear", "constant", "prunable", 4097)])
def test_composed_mode_matches_oracle_world_semantics(gpu, game, regret, weight, sampling, batch):
    # single-GPU composed update == the oracle's model of the blocked composition (world = 1), bit for bit;
    # batches of several RP_COMPOSE_CHUNK-tree chunks: every infoset folds several blocks
    g = Game(game)
    hp = oracle.default_hyper()
    hp.prune_warmup = 2
    hp.prune_threshold = -2.0
    dev = Solver(g, regret, weight, sampling, batch=batch, seed=11, hyper=hp)
    dev.set_update_mode("composed")
    ora = oracle.OracleSolver(g, regret, weight, sampling, batch=batch, seed=11, hyper=hp)
    for _ in range(6):
        dev.step()
        ora.step_world(1)
        assert_tables_equal(dev.export(), ora.export())


def test_composed_mode_is_within_tolerance_of_the_reference_order(gpu):
    # the composed update re-associates the reference's sequential per-touch update; per step, from identical
    # tables, the difference stays at f32 round-off: rtol 1e-4 (regret, weight), 2e-4 (payoff mean), visits exact
    g = Game("leduc")
    B = 1 << 14
    dev = Solver(g, "linear", "linear", "external", batch=B, seed=4)
    dev.set_update_mode("composed")
    ref = oracle.OracleSolver(g, "linear", "linear", "external", batch=B, seed=4)
    for _ in range(5):
        dev.load_rows(ref.export(), ref.epoch)
        dev.step()
        ref.step()
        a, b = dev.export(), ref.export()
        assert np.array_equal(a["visits"], b["visits"])
        np.testing.assert_allclose(a["regret"], b["regret"], rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(a["weight"], b["weight"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(a["payoff"], b["payoff"], rtol=2e-4, atol=2e-5)


def test_composed_mode_rejects_sign_dependent_discount(gpu):
    g = Game("kuhn")
    dev = Solver(g, "discounted", "linear", "external", batch=64, seed=1)
    dev.set_update_mode("composed")
    with pytest.raises(_lib.RpError) as ei:
        dev.step()
    assert ei.value.code == _lib.RP_ERR_UNSUPPORTED


def test_full_batch_properties(gpu):
    # BASELINE config 2 at the bench batch: visit conservation and determinism are size independent
    g = Game("leduc")
    B = 1 << 16
    a = Solver(g, "floored", "linear", "external", batch=B, seed=5)
    b = Solver(g, "floored", "linear", "external", batch=B, seed=5)
    for _ in range(4):
        a.step()
        b.step()
    ra, rb = a.export(), b.export()
    assert_tables_equal(ra, rb)
    nodes, infos = a.counters()
    # every Decisions increments the visits of each of its infoset's 2 edges once (solver.rs:187-192)
    assert int(ra["visits"].sum()) == 2 * infos
    roots = [g.info_id(n) for n in ("J|", "Q|", "K|")]
    # walker alternates (book.rs:142-144): P0's root infosets are visited by every tree of epochs 0 and 2
    assert sum(int(ra["visits"][i * 2]) for i in roots) == 2 * B
    assert np.isfinite(ra["regret"]).all() and np.isfinite(ra["weight"]).all()


def test_leduc_converges_like_the_reference(gpu):
    # crates/leduc/src/solver.rs:105-123 threshold with a GPU-sized batch (epoch-stale regrets, as NLHE's 128)
    g = Game("leduc")
    dev = Solver(g, "floored", "linear", "external", batch=4096, seed=18)
    dev.solve(4096 * 1024)
    assert dev.exploitability() < 0.080


@pytest.mark.parametrize("batch,epochs", [(4096, 1024), (1 << 18, 256)])
def test_leduc_converges_in_composed_mode(gpu, batch, epochs):
    # the re-associated update solves the game too: same threshold as the reference's test, at the batch sizes the
    # benchmark runs (an epoch = one batch; large batches need far fewer epochs than the reference's batch of 1)
    g = Game("leduc")
    dev = Solver(g, "floored", "linear", "external", batch=batch, seed=18)
    dev.set_update_mode("composed")
    dev.solve(batch * epochs)
    assert dev.epoch == epochs
    assert dev.exploitability() < 0.080


def test_kuhn_nash_on_device(gpu):
    g = Game("kuhn")
    dev = Solver(g, "floored", "linear", "external", batch=1024, seed=7)
    dev.solve(1024 * 2048)
    pol = lambda name, a: float(dev.policy(g.info_id(name), "averaged")[a])  # noqa: E731
    assert dev.exploitability() < 0.020
    assert pol("J|B", 0) > 0.95 and pol("K|B", 1) > 0.95 and pol("K|X", 1) > 0.95
    assert abs(pol("J|", 1) - 9 / 31) < 0.05 and abs(pol("K|", 1) - 27 / 31) < 0.05
    assert abs(pol("Q|XB", 1) - 23 / 31) < 0.05 and abs(pol("J|X", 1) - 9 / 31) < 0.05


def test_trainer_loop_checkpoints_and_summary(gpu):
    # Trainer::train over the solver (trainer.rs:18-66): a checkpoint every step (log interval 0), the Checkpoint
    # display of metrics/checkpoint.rs:39-50, rate = new infos / max(1, whole seconds) (metrics/mod.rs:67-80),
    # Progress::summary at the end (progress.rs:24-26); the table is the one plain step() calls produce
    g = Game("leduc")
    a = Solver(g, "floored", "linear", "external", batch=256, seed=5)
    b = Solver(g, "floored", "linear", "external", batch=256, seed=5)
    seen, flushes = [], []
    summary = a.train(max_steps=12, log_interval=0.0, flush_interval=0.0,
                      on_checkpoint=lambda cp, line: seen.append((cp, line)), on_flush=lambda cp: flushes.append(cp))
    for _ in range(12):
        b.step()
    ea, eb = a.export(), b.export()
    for f in ("visits", "regret", "weight", "payoff"):
        assert np.array_equal(ea[f].view(np.uint32), eb[f].view(np.uint32))
    assert len(seen) == 12 and len(flushes) == 12 and [cp["epoch"] for cp, _ in seen] == list(range(1, 13))
    prev = 0
    for cp, line in seen:
        assert cp["infos"] >= prev and cp["rate"] == float(cp["infos"] - prev)  # whole seconds < 1 -> divided by 1
        prev = cp["infos"]
        want = "".join(f"{x:<20}" for x in (f"batch {cp['epoch']}", f"nodes {cp['nodes']}", f"infos {cp['infos']}",
                                            f"I/sec {cp['rate']:.1f}"))
        assert line == want
    nodes, infos = a.counters()
    assert summary == "training stopped\n" + "".join(
        f"{x:<20}" for x in (f"batch 12", f"nodes {nodes}", f"infos {infos}", f"I/sec {float(infos):.1f}"))
    # the interrupt flag stops the loop after the step in flight
    import ctypes
    stop = ctypes.c_int(1)
    a.train(interrupt=stop)
    assert a.epoch == 13


def test_update_mode_chosen_at_creation(gpu):
    # rp_mccfr_create_mode: a caller that wants the fast, shardable composed update says so once; the result is the solver that
    # rp_mccfr_create + rp_mccfr_set_update_mode builds, and a schedule the composed mode cannot express is refused at creation
    g = Game("leduc")
    a = Solver(g, "linear", "linear", "external", batch=4096, seed=3, mode="composed")
    b = Solver(g, "linear", "linear", "external", batch=4096, seed=3)
    b.set_update_mode("composed")
    for _ in range(3):
        a.step()
        b.step()
    ea, eb = a.export(), b.export()
    for f in ("visits", "regret", "weight", "payoff"):
        assert np.array_equal(ea[f].view(np.uint32), eb[f].view(np.uint32)), f
    with pytest.raises(_lib.RpError):
        Solver(g, "discounted", "linear", "external", batch=64, seed=3, mode="composed")
    c = Solver(g, "discounted", "linear", "external", batch=64, seed=3, mode="ordered")
    c.step()


@pytest.mark.parametrize("mode", ["ordered", "composed"])
@pytest.mark.parametrize("game,sampling,batch", [("leduc", "pluribus", 1500), ("leduc", "prunable", 700), ("kuhn", "pluribus", 900)])
def test_static_skeleton_traversal_of_the_pruned_schemes(gpu, monkeypatch, mode, game, sampling, batch):
    # PrunableSampling / PluribusSampling on the compile-time skeleton (traverse_static.hpp, PRUNED = true): a pruned walker edge
    # is a dead skeleton node.  Pruning is forced to bite (warm-up 2, threshold just below zero).  Against the per-lane DFS
    # kernel (RP_TRAV_GENERIC=1) and the oracle, bit for bit, in both update modes (the fused traversal + block maps
    # kernel skips the list entries of pruned edges like k_chunk_maps<true>).
    g = Game(game)
    hp = oracle.default_hyper()
    hp.prune_warmup, hp.prune_threshold, hp.prune_explore = 2, -0.05, 0.1
    a = Solver(g, "linear", "linear", sampling, batch=batch, seed=23, hyper=hp)
    monkeypatch.setenv("RP_TRAV_GENERIC", "1")
    b = Solver(g, "linear", "linear", sampling, batch=batch, seed=23, hyper=hp)
    monkeypatch.delenv("RP_TRAV_GENERIC")
    ora = oracle.OracleSolver(g, "linear", "linear", sampling, batch=batch, seed=23, hyper=hp)
    assert a.kernel_variant() == "static" and b.kernel_variant() != "static"
    if mode == "composed":
        a.set_update_mode("composed")
        b.set_update_mode("composed")
    for _ in range(12):
        a.step()
        b.step()
        if mode == "composed":
            ora.step_world(1)
        else:
            ora.step()
        assert_tables_equal(a.export(), ora.export())
        assert_tables_equal(b.export(), ora.export())
    assert a.counters() == b.counters() == ora.counters()
    # pruning did bite: fewer nodes than external sampling grows from the same seed
    e = Solver(g, "linear", "linear", "external", batch=batch, seed=23, hyper=hp)
    for _ in range(12):
        e.step()
    assert a.counters()[0] < e.counters()[0]


# ---- reference-seed mode (rp_rng_kind RP_RNG_REFERENCE): the reference's DefaultHasher -> SmallRng chain draws the branches
# (flow.rs:285-295; include/rp_refrng.h; the CPU side of it in tests/test_reference_seed.py) -----------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("game", ["kuhn", "leduc", "rps", "leduc_wide"])
@pytest.mark.parametrize("regret,weight,sampling", [
    ("linear", "linear", "external"), ("linear", "linear", "pluribus"), ("discounted", "quadratic", "prunable")])
def test_reference_seed_tables_bit_exact_vs_oracle(gpu, game, regret, weight, sampling):
    g = Game(game)
    hp = oracle.default_hyper()
    hp.prune_warmup = 3
    hp.prune_threshold = -2.0
    hp.prune_explore = 0.3
    B, steps = 333, 10
    dev = Solver(g, regret, weight, sampling, batch=B, seed=42, hyper=hp)
    ora = oracle.OracleSolver(g, regret, weight, sampling, batch=B, seed=42, hyper=hp)
    dev.set_rng("reference")
    ora.set_rng("reference")
    for s in range(steps):
        dev.step()
        ora.step()
        assert_tables_equal(dev.export(), ora.export())
    assert dev.counters() == ora.counters()
    assert dev.exploitability() == ora.exploitability()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["RP_TRAV_GENERIC", "RP_MCCFR_HBM_SCRATCH"])
def test_reference_seed_in_the_generic_traversals(gpu, monkeypatch, variant):
    g = Game("leduc")
    hp = oracle.default_hyper()
    hp.prune_warmup = 2
    hp.prune_threshold = -2.0
    monkeypatch.setenv(variant, "1")
    dev = Solver(g, "linear", "linear", "pluribus", batch=700, seed=9, hyper=hp)
    monkeypatch.delenv(variant)
    ora = oracle.OracleSolver(g, "linear", "linear", "pluribus", batch=700, seed=9, hyper=hp)
    dev.set_rng("reference")
    ora.set_rng("reference")
    for _ in range(6):
        dev.step()
        ora.step()
        assert_tables_equal(dev.export(), ora.export())


@pytest.mark.gpu
@pytest.mark.parametrize("sampling", ["external", "pluribus"])
def test_reference_seed_composed_step_equals_the_oracle(gpu, sampling):
    # the headline path (k_traverse_maps_static + k_combine2) with the reference chain drawing the branches
    g = Game("leduc")
    hp = oracle.default_hyper()
    hp.prune_warmup = 2
    hp.prune_threshold = -2.0
    B = 4096 + 77
    dev = Solver(g, "linear", "linear", sampling, batch=B, seed=3, hyper=hp)
    ora = oracle.OracleSolver(g, "linear", "linear", sampling, batch=B, seed=3, hyper=hp)
    dev.set_update_mode("composed")
    dev.set_rng("reference")
    ora.set_rng("reference")
    for _ in range(6):
        dev.step()
        ora.step_world(1)
        assert_tables_equal(dev.export(), ora.export())
    assert dev.counters() == ora.counters()


@pytest.mark.gpu
def test_reference_seed_rejects_missing_streams(gpu):
    g = Game("kuhn")
    dev = Solver(g, "linear", "linear", "external", batch=8, seed=1)
    lib = _lib.load()
    assert lib.rp_mccfr_set_rng(dev._h, 1, None) != 0
    bad = g.hash_streams()
    bad.n_infos = 5
    assert lib.rp_mccfr_set_rng(dev._h, 1, C.byref(bad)) != 0
    dev.set_rng("counter")
    dev.step()

"""Pins the MCCFR oracle against every known-answer test the reference holds for this path.

Reference tests restated here (the reference has no golden vectors for MCCFR — SURVEY.md §8c):
  crates/kuhn/src/solver.rs:141-153,234-277   44 (sampling x regret x weight) exploitability thresholds @ 2^18
  crates/kuhn/src/solver.rs:176-203           analytic Nash in 31sts + pure strategies + k/a ~ 3
  crates/kuhn/src/solver.rs:163-174           sampling_distribution sums to 1
  crates/leduc/src/solver.rs:105-123          3 exploitability thresholds < 0.080 @ 2^18
  crates/roshambo/src/solver.rs:157-166,205-257  averaged policy (0.4,0.4,0.2), exploitability < 0.03 @ 2^16
"""
import numpy as np
import pytest

import oracle
from robopoker_amd import Game, _lib

N18 = 1 << 18
N16 = 1 << 16

KUHN_MATRIX = [
    ("external", "summed", "constant", 0.020), ("external", "summed", "linear", 0.025),
    ("external", "summed", "quadratic", 0.025), ("external", "summed", "exponential", 0.030),
    ("external", "linear", "constant", 0.020), ("external", "linear", "linear", 0.020),
    ("external", "linear", "quadratic", 0.030), ("external", "linear", "exponential", 0.025),
    ("external", "floored", "constant", 0.020), ("external", "floored", "linear", 0.020),
    ("external", "floored", "quadratic", 0.020), ("external", "floored", "exponential", 0.020),
    ("external", "asymmetric", "constant", 0.020), ("external", "asymmetric", "linear", 0.020),
    ("external", "asymmetric", "quadratic", 0.035), ("external", "asymmetric", "exponential", 0.030),
    ("external", "discounted", "constant", 0.020), ("external", "discounted", "linear", 0.020),
    ("external", "discounted", "quadratic", 0.020), ("external", "discounted", "exponential", 0.020),
    ("prunable", "floored", "constant", 0.020), ("prunable", "floored", "linear", 0.020),
    ("prunable", "floored", "quadratic", 0.020), ("prunable", "floored", "exponential", 0.020),
    ("prunable", "asymmetric", "constant", 0.020), ("prunable", "asymmetric", "linear", 0.020),
    ("prunable", "asymmetric", "quadratic", 0.030), ("prunable", "asymmetric", "exponential", 0.025),
    ("prunable", "discounted", "constant", 0.020), ("prunable", "discounted", "linear", 0.020),
    ("prunable", "discounted", "quadratic", 0.020), ("prunable", "discounted", "exponential", 0.020),
    ("pluribus", "floored", "constant", 0.020), ("pluribus", "floored", "linear", 0.020),
    ("pluribus", "floored", "quadratic", 0.020), ("pluribus", "floored", "exponential", 0.020),
    ("pluribus", "asymmetric", "constant", 0.020), ("pluribus", "asymmetric", "linear", 0.020),
    ("pluribus", "asymmetric", "quadratic", 0.035), ("pluribus", "asymmetric", "exponential", 0.035),
    ("pluribus", "discounted", "constant", 0.020), ("pluribus", "discounted", "linear", 0.020),
    ("pluribus", "discounted", "quadratic", 0.020), ("pluribus", "discounted", "exponential", 0.020),
]


@pytest.fixture(scope="module")
def kuhn():
    return Game("kuhn")


@pytest.fixture(scope="module")
def leduc():
    return Game("leduc")


def test_game_tables_shape(kuhn, leduc):
    # 6-card Kuhn: 12 infosets, 30 deals, 277-node exploitability tree
    assert kuhn.table.n_infos == 12 and kuhn.table.n_states == 277 and kuhn.table.max_actions == 2
    # Leduc: 120 decision infosets x 2 actions
    assert leduc.table.n_infos == 120 and leduc.table.max_actions == 2
    rps = Game("rps")
    assert rps.table.n_infos == 2 and rps.table.max_actions == 3 and rps.table.n_states == 13
    for g in (kuhn, leduc, rps):
        assert _lib.load().rp_game_table_check(g.table) == 0


def test_kuhn_payoffs_are_zero_sum_and_bounded(kuhn):
    t = kuhn.table
    pay = np.ctypeslib.as_array(t.payoffs, shape=(t.n_terminals, 2))
    assert np.all(pay[:, 0] == -pay[:, 1])
    assert set(np.abs(pay[:, 0]).tolist()) <= {0.0, 1.0, 2.0}  # kuhn/src/game.rs:35-64


def test_leduc_payoffs(leduc):
    t = leduc.table
    pay = np.ctypeslib.as_array(t.payoffs, shape=(t.n_terminals, 2))
    assert np.all(pay[:, 0] == -pay[:, 1])
    assert set(np.abs(pay[:, 0]).tolist()) <= {0.0, 1.0, 3.0, 5.0, 7.0}  # leduc/src/game.rs:57-110


@pytest.mark.parametrize("sampling,regret,weight,tol", KUHN_MATRIX)
def test_kuhn_exploitability_matrix(kuhn, sampling, regret, weight, tol):
    s = oracle.OracleSolver(kuhn, regret, weight, sampling, batch=1, seed=18).solve(N18)
    e = s.exploitability()
    assert e < tol, f"{sampling}+{regret}+{weight}: exploitability {e:.4f} >= {tol}"


def test_kuhn_nash_equilibrium(kuhn):
    s = oracle.OracleSolver(kuhn, "floored", "linear", "external", batch=1, seed=7).solve(N18)

    def pol(name, a):
        return float(s.policy(kuhn.info_id(name), "averaged")[a])

    FOLD, CALL, CHECK, BET = 0, 1, 0, 1
    assert pol("J|B", FOLD) > 0.95 and pol("J|XB", FOLD) > 0.95
    assert pol("K|B", CALL) > 0.95 and pol("K|XB", CALL) > 0.95
    assert pol("K|X", BET) > 0.95
    assert pol("Q|", CHECK) > 0.85
    near = lambda v, t, tol: abs(v - t) < tol  # noqa: E731
    assert near(pol("J|", BET), 9 / 31, 0.05)
    assert near(pol("K|", BET), 27 / 31, 0.05)
    assert near(pol("Q|B", CALL), 17 / 31, 0.08)
    assert near(pol("Q|XB", CALL), 23 / 31, 0.05)
    assert near(pol("J|X", BET), 9 / 31, 0.05)
    assert near(pol("Q|X", BET), 8 / 31, 0.18)
    assert near(pol("K|", BET) / pol("J|", BET), 3.0, 0.4)


def test_kuhn_sampling_distribution_is_normalized(kuhn):
    s = oracle.OracleSolver(kuhn, "floored", "linear", "external", batch=1, seed=3).solve(1 << 12)
    for info in range(kuhn.n_infos):
        assert abs(float(s.policy(info, "sampling").sum()) - 1.0) < 1e-4


@pytest.mark.parametrize("sampling,regret,weight", [("external", "floored", "linear"),
                                                    ("external", "discounted", "linear"),
                                                    ("prunable", "floored", "linear")])
def test_leduc_exploitability(leduc, sampling, regret, weight):
    s = oracle.OracleSolver(leduc, regret, weight, sampling, batch=1, seed=18).solve(N18)
    e = s.exploitability()
    assert e < 0.080, f"exploitability {e:.4f}"


@pytest.mark.parametrize("sampling,regret,weight,tol", [
    ("external", "floored", "linear", 0.05), ("external", "linear", "linear", 0.05),
    ("external", "summed", "constant", 0.05), ("external", "discounted", "linear", 0.05),
    ("external", "asymmetric", "linear", 0.05), ("prunable", "floored", "linear", 0.05),
    ("pluribus", "floored", "linear", 0.05)])
def test_rps_equilibrium(sampling, regret, weight, tol):
    g = Game("rps")
    s = oracle.OracleSolver(g, regret, weight, sampling, batch=1, seed=16).solve(N16)
    for info in (0, 1):
        p = s.policy(info, "averaged")
        assert abs(p[0] - 0.40) < tol and abs(p[1] - 0.40) < tol and abs(p[2] - 0.20) < tol


def test_rps_exploitability():
    g = Game("rps")
    s = oracle.OracleSolver(g, "floored", "linear", "external", batch=1, seed=16).solve(N16)
    assert s.exploitability() < 0.03


def test_batched_epochs_still_converge(leduc):
    # batch_size > 1 = the reference's NLHE configuration (nlhe/src/solver.rs:11, batch 128):
    # every tree of an epoch samples against the same profile, updates apply in tree order.
    s = oracle.OracleSolver(leduc, "floored", "linear", "external", batch=256, seed=5).solve(1 << 19)
    assert s.epoch == (1 << 19) // 256
    assert s.exploitability() < 0.12


def test_oracle_is_deterministic(kuhn):
    a = oracle.OracleSolver(kuhn, "linear", "linear", "pluribus", batch=4, seed=9).solve(4096).export()
    b = oracle.OracleSolver(kuhn, "linear", "linear", "pluribus", batch=4, seed=9).solve(4096).export()
    assert a.tobytes() == b.tobytes()
    c = oracle.OracleSolver(kuhn, "linear", "linear", "pluribus", batch=4, seed=10).solve(4096).export()
    assert a.tobytes() != c.tobytes()


def test_first_epoch_semantics(kuhn):
    # at t = 0 LinearRegret's discount is 0 (R <- delta) and LinearWeight adds
    # sigma * 0, so weights stay at the EPSILON floor after the first step.
    s = oracle.OracleSolver(kuhn, "linear", "linear", "external", batch=1, seed=1)
    s.step()
    rows = s.export()
    touched = rows["visits"] > 0
    assert touched.any()
    assert np.all(rows["weight"][touched] == np.float32(1.17549435e-38))


def test_composed_world_update_matches_ordered_within_tolerance(leduc):
    # the multi-GPU exchange semantics (ora_mccfr_step_world) against plain ordered steps on the same
    # world*B trees: identical sampling, fp32-reassociation-level difference in the tables
    B, world = 1500, 4  # several RP_COMPOSE_CHUNK-tree chunks (= blocks per infoset) per rank
    a = oracle.OracleSolver(leduc, "linear", "linear", "external", batch=B * world, seed=11)
    b = oracle.OracleSolver(leduc, "linear", "linear", "external", batch=B, seed=11)
    for _ in range(6):
        # resynchronise before every step: sampling reads the tables, so an ulp of difference would
        # eventually change a sampled branch and the two trajectories would stop being comparable
        b.load_rows(a.export(), a.epoch)
        a.step()
        b.step_world(world)
        ra, rb = a.export(), b.export()
        assert np.array_equal(ra["visits"], rb["visits"])
        np.testing.assert_allclose(ra["regret"], rb["regret"], rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(ra["weight"], rb["weight"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ra["payoff"], rb["payoff"], rtol=2e-4, atol=2e-5)


def test_division_by_reciprocal_is_exact():
    # rp_math.h's rp_div_by_recip (used on the GPU to keep the Welford chain short) == IEEE division
    import ctypes as C
    o = oracle.load()
    o.ora_div_by_recip_mismatches.restype = C.c_uint64
    o.ora_div_by_recip_mismatches.argtypes = [C.c_uint64, C.c_uint64]
    assert o.ora_div_by_recip_mismatches(40_000_000, 12345) == 0
    o.ora_div_unproven.restype = C.c_uint64
    # the single-correction quotient carries an exact proof; "not proven" (-> plain division) must stay rare
    assert o.ora_div_unproven() < 40_000_000 * 1e-3


def test_tree_parallel_step_equals_the_sequential_step():
    # ora_mccfr_step_mt (bench.py's all-core CPU baseline: the reference's rayon batch() + sequential update) must be the
    # sequential oracle bit for bit, whatever the thread count
    g = Game("leduc")
    a = oracle.OracleSolver(g, "linear", "linear", "pluribus", batch=777, seed=3)
    b = oracle.OracleSolver(g, "linear", "linear", "pluribus", batch=777, seed=3)
    for threads in (2, 3, 8, 5):
        a.step()
        b.step_mt(threads)
    ra, rb = a.export(), b.export()
    for f in ("regret", "weight", "payoff", "visits"):
        assert np.array_equal(ra[f].view(np.uint32), rb[f].view(np.uint32)), f
    assert a.counters() == b.counters() and a.epoch == b.epoch

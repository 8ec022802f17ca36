"""The sparse-profile oracle (ora_profile_*) is pinned to the dense MCCFR oracle, which is pinned to the reference's
known-answer tests: feeding Leduc's own Decisions through the row-addressed path must reproduce Solver::step."""
import numpy as np
import pytest

import oracle
from robopoker_amd import Game
from robopoker_amd.sparse import synthetic_batch


def leduc_decisions(solver, A):
    ds = solver.batch()  # this epoch's Decisions (Solver::batch is pure w.r.t. the profile)
    n = len(ds)
    row = np.array([d["info"] for d in ds], dtype=np.uint32)
    nact = np.array([d["n"] for d in ds], dtype=np.uint8)
    exp = np.array([d["expanded"] for d in ds], dtype=np.uint16)
    reg = np.zeros((n, A), dtype=np.float32)
    pol = np.zeros((n, A), dtype=np.float32)
    for i, d in enumerate(ds):
        reg[i, : d["n"]] = d["regret"]
        pol[i, : d["n"]] = d["policy"]
    pay = np.array([d["payoff"] for d in ds], dtype=np.float32)
    return row, nact, exp, reg, pol, pay


@pytest.mark.parametrize("regret,weight,sampling", [("floored", "linear", "external"), ("linear", "linear", "pluribus"),
                                                     ("discounted", "quadratic", "prunable")])
def test_sparse_apply_equals_solver_step_on_leduc_decisions(regret, weight, sampling):
    g = Game("leduc")
    dense = oracle.OracleSolver(g, regret, weight, sampling, batch=64, seed=11)
    A = g.max_actions
    sparse = oracle.OracleProfile(g.n_infos, A, regret, weight)
    for step in range(12):
        batch = leduc_decisions(dense, A)  # before they are applied
        dense.step()
        sparse.apply(batch)
        assert sparse.epoch() == dense.epoch
    rows = dense.export().reshape(g.n_infos, A)
    got = sparse.rows(np.arange(g.n_infos))
    for f in ("weight", "regret", "payoff", "visits"):
        assert np.array_equal(rows[f].view(np.uint32), got[f].view(np.uint32)), f


def test_summarize_then_fold_matches_apply_within_tolerance_and_counts_exactly():
    batch = synthetic_batch(5000, n_rows=300, max_actions=9, seed=3)
    a = oracle.OracleProfile(300, 9, "linear", "linear")
    b = oracle.OracleProfile(300, 9, "linear", "linear")
    for e in range(4):
        a.apply(batch)
        b.fold(b.summarize(batch))
    ra, rb = a.rows(np.arange(300)), b.rows(np.arange(300))
    assert np.array_equal(ra["visits"], rb["visits"])
    for f in ("weight", "regret", "payoff"):
        assert np.allclose(ra[f], rb[f], rtol=2e-4, atol=1e-3), f


def test_world_fold_in_rank_order_equals_one_rank_with_the_concatenated_batch_when_blocks_align():
    # two ranks with RP_SPARSE_BLOCK-aligned per-row touch counts would be identical; in general the association
    # differs, so the check here is the structural one: folding [rank0 entries, rank1 entries] touches every row both
    # ranks touched, in rank order, and counts visits exactly
    b0 = synthetic_batch(3000, n_rows=200, max_actions=6, seed=1)
    b1 = synthetic_batch(3000, n_rows=200, max_actions=6, seed=2)
    w = oracle.OracleProfile(200, 6, "floored", "linear")
    e0, e1 = w.summarize(b0), w.summarize(b1)
    w.fold(np.concatenate([e0, e1]))
    seq = oracle.OracleProfile(200, 6, "floored", "linear")
    seq.apply(tuple(np.concatenate([x, y]) for x, y in zip(b0, b1)))
    rw, rs = w.rows(np.arange(200)), seq.rows(np.arange(200))
    assert np.array_equal(rw["visits"], rs["visits"])
    assert w.epoch() == seq.epoch() == 1
    for f in ("weight", "regret", "payoff"):
        assert np.allclose(rw[f], rs[f], rtol=2e-4, atol=1e-3), f


def test_composed_rejects_sign_dependent_discount():
    p = oracle.OracleProfile(10, 4, "discounted", "linear")
    with pytest.raises(ValueError):
        p.summarize(synthetic_batch(10, 10, 4))

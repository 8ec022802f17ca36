"""GPU parity of the glibc-arithmetic mode of the stand-alone Sinkhorn operators (rp_sinkhorn_set_libm(RP_LIBM_GLIBC)).

In this mode exp / ln are glibc's expf / logf (include/rp_libm_glibc.h, equal to glibc 2.35's on all 2^32 inputs:
tests/test_libm_glibc.py), i.e. what f32::exp / f32::ln are in a Rust build on Linux (sinkhorn.rs:115,120-127,136; phi.rs:36).
The checker is the oracle on the same restated functions (ora_lloyd_set_libm(2)), which on a glibc host IS the oracle on the
platform's libm (mode 1).  Bit-exact: costs, iteration counts, divergences, the flow matrix.

Added after round 4's GPU minutes were spent: run on the wave64 execution model (tests/test_emul.py) before its first hardware
run, and kept in a file of its own, collected last, so a surprise here cannot hide the tests that have run on hardware."""
import ctypes as C

import numpy as np
import pytest

import oracle
from lloyd_fixtures import flop_hist, flop_like_points, flop_metric, random_metric, smooth_metric
from robopoker_amd import lloyd

pytestmark = pytest.mark.gpu


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture()
def glibc_mode(gpu):
    o = oracle.load()
    o.ora_lloyd_set_libm.argtypes = [C.c_int]
    o.ora_lloyd_set_libm(2)
    lloyd.sinkhorn_set_libm("glibc")
    yield
    lloyd.sinkhorn_set_libm("contract")
    o.ora_lloyd_set_libm(0)


def test_closed_form_fixture_in_glibc_arithmetic(glibc_mode):
    # the reference's own fixture (sinkhorn.rs:240-293): the tolerances it asserts, and the oracle's bits
    tri = flop_metric()
    mu = flop_hist([(0, 3), (5, 1), (12, 4)])
    nu = flop_hist([(2, 2), (8, 5), (20, 1), (24, 3)])
    h = flop_hist([(0, 3), (5, 1), (12, 4), (24, 2)])
    d = lloyd.sinkhorn_divergence(np.stack([mu, nu, h]), np.stack([nu, mu, h]), tri)
    exp = [oracle.sinkhorn_divergence(mu, nu, tri), oracle.sinkhorn_divergence(nu, mu, tri), oracle.sinkhorn_divergence(h, h, tri)]
    assert np.array_equal(bits(d), bits(exp))
    assert abs(d[2]) < 1e-4 and abs(d[0] - d[1]) < 1e-3 and d[0] > 0


@pytest.mark.parametrize("bins,nnz_a,nnz_b", [(32, 5, 9), (101, 30, 60), (256, 47, 256), (64, 1, 64)])
def test_random_pairs_in_glibc_arithmetic(glibc_mode, bins, nnz_a, nnz_b):
    rng = np.random.default_rng(bins * 1000 + nnz_a)
    tri = random_metric(bins, rng)
    P = 4
    mu = np.zeros((P, bins), dtype=np.uint32)
    nu = np.zeros((P, bins), dtype=np.uint32)
    for p in range(P):
        mu[p, rng.choice(bins, nnz_a, replace=False)] = rng.integers(1, 9, nnz_a)
        nu[p, rng.choice(bins, nnz_b, replace=False)] = rng.integers(1, 2000, nnz_b)
    hp = oracle.default_sinkhorn()
    hp.iterations = 24
    d = lloyd.sinkhorn_divergence(mu, nu, tri, hp)
    c, it = lloyd.sinkhorn_cost(mu, nu, tri, hp)
    for p in range(P):
        assert bits(d[p]) == bits(oracle.sinkhorn_divergence(mu[p], nu[p], tri, hp))
        ec, eit = oracle.sinkhorn_cost(mu[p], nu[p], tri, hp)
        assert bits(c[p]) == bits(ec) and it[p] == eit


def test_converging_solves_and_the_flow_in_glibc_arithmetic(glibc_mode):
    # flop-like pairs run to the tolerance (~100 iterations): the stopping iteration is part of what must agree
    bins = 64
    tri = smooth_metric(bins, 3)
    pts = flop_like_points(12, bins=bins, mass=30, seed=21).astype(np.uint32)
    c, it = lloyd.sinkhorn_cost(pts[:6], pts[6:], tri)
    for p in range(6):
        ec, eit = oracle.sinkhorn_cost(pts[p], pts[6 + p], tri)
        assert bits(c[p]) == bits(ec) and it[p] == eit
    flow, coupling = lloyd.sinkhorn_flow(pts[0], pts[6], tri)
    eflow, ecoupling = oracle.sinkhorn_flow(pts[0], pts[6], tri)[:2]
    assert np.array_equal(bits(flow), bits(eflow)) and np.array_equal(bits(coupling), bits(ecoupling))
    fold = np.float32(0.0)
    for v in flow[pts[0] > 0][:, pts[6] > 0].ravel():
        fold = np.float32(fold + v)
    assert bits(fold) == bits(c[0])


def test_the_mode_differs_from_the_contract_by_ulps_and_switches_back(gpu):
    bins = 64
    tri = smooth_metric(bins, 3)
    pts = flop_like_points(40, bins=bins, mass=30, seed=22).astype(np.uint32)
    a, ia = lloyd.sinkhorn_cost(pts[:20], pts[20:], tri)
    lloyd.sinkhorn_set_libm("glibc")
    try:
        b, ib = lloyd.sinkhorn_cost(pts[:20], pts[20:], tri)
    finally:
        lloyd.sinkhorn_set_libm("contract")
    a2, ia2 = lloyd.sinkhorn_cost(pts[:20], pts[20:], tri)
    assert np.array_equal(bits(a), bits(a2)) and np.array_equal(ia, ia2)
    same = ia == ib
    assert same.any()
    ulps = np.abs(bits(a).astype(np.int64) - bits(b).astype(np.int64))[same]
    assert ulps.max() <= 32  # measured on the CPU side: <= 7 (tests/test_oracle_lloyd.py::test_platform_libm_*)
    assert (np.abs(a - b) / np.maximum(np.abs(a), 1e-6)).max() < 1e-5
    with pytest.raises(Exception):
        from robopoker_amd import _lib
        _lib.check(_lib.load().rp_sinkhorn_set_libm(7))

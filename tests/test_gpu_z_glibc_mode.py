"""GPU parity of the glibc-arithmetic pass of the lloyd kernels: a whole layer (rp_kmeans_set_libm(RP_LIBM_GLIBC)) and the stand-alone
Sinkhorn operators (rp_sinkhorn_set_libm(RP_LIBM_GLIBC)).

In this mode exp / ln are glibc's expf / logf (include/rp_libm_glibc.h, equal to glibc 2.35's on all 2^32 inputs:
tests/test_libm_glibc.py), i.e. what f32::exp / f32::ln are in a Rust build on Linux (sinkhorn.rs:115,120-127,136; phi.rs:36).
The checker is the oracle on the same restated functions (ora_lloyd_set_libm(2)), which on a glibc host IS the oracle on the
platform's libm (mode 1).  Bit-exact: k-means++ picks, bounds, drift, sizes, buckets, lookup distances and the layer's metric;
costs, iteration counts, divergences and the flow matrix of single solves.

Written against the wave64 execution model (tests/test_emul.py) when round 4's GPU minutes were all but spent; first hardware run
with the last of them: all passed (this file + tests/test_reference_kat.py, profiles/r04_glibc_pass_first_hw_run.log; the 2^32 sweep
in profiles/r04_glibc_device_sweep.txt), and the pass
costs 2.3 x the contract's on the same unpruned solves (profiles/r04_glibc_pass_timing.json)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from lloyd_fixtures import flop_hist, flop_like_points, flop_metric, random_metric, smooth_metric
from robopoker_amd import _lib, lloyd

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


def _layer_pair(K, N, bins, mass, seed, iters):
    pts = flop_like_points(N, bins=bins, mass=mass, seed=seed)
    tri = smooth_metric(bins, seed)
    hp = oracle.default_sinkhorn()
    hp.iterations = iters
    dev = lloyd.Layer(K, pts, "sinkhorn", tri, hp=hp, seed=seed)
    dev.set_libm("glibc")
    return dev, oracle.OracleKmeans(K, pts, "sinkhorn", tri, hp=hp, seed=seed)


from test_gpu_lloyd import _check_state  # noqa: E402  (exact bounds bit for bit; interval-valued ones by containment)


@pytest.fixture()
def oracle_on_glibc(gpu):
    o = oracle.load()
    o.ora_lloyd_set_libm.argtypes = [C.c_int]
    o.ora_lloyd_set_libm(2)
    yield
    o.ora_lloyd_set_libm(0)


@pytest.mark.parametrize("K,N,bins,mass,rng", [(5, 150, 32, 20, "counter"), (70, 200, 48, 24, "counter"), (12, 300, 32, 20, "reference")])
def test_a_layer_clustered_in_glibc_arithmetic_equals_the_oracle(oracle_on_glibc, K, N, bins, mass, rng):
    # Layer::cluster (layer.rs:200-240) with every exp / ln glibc's: k-means++ (counter draw, or layer.rs:155-178's own SmallRng +
    # WeightedIndex), init_bounds, four Elkan iterations, lookup, metric, rms — the grouped (four / two points per wavefront)
    # kernels included, the two bound filters off
    dev, ora = _layer_pair(K, N, bins, mass, seed=K + N, iters=16)
    if rng == "reference":
        dev.set_rng("reference", 1)
        ora.set_rng("reference", 1)
    assert np.array_equal(dev.init_centroids(), ora.init_centroids()), "k-means++ picks differ"
    dev.init_bounds()
    ora.init_bounds()
    _check_state(dev, ora)
    for _ in range(4):
        d1, s1, m1 = dev.step()
        d2, s2, m2 = ora.step()
        assert np.array_equal(bits(d1), bits(d2)), "drift differs"
        assert np.array_equal(s1, s2) and m1 == m2
        _check_state(dev, ora)
    b1, dd1 = dev.lookup()
    b2, dd2 = ora.assign()
    assert np.array_equal(b1, b2) and np.array_equal(bits(dd1), bits(dd2))
    assert np.array_equal(bits(dev.metric()), bits(ora.metric()))
    assert bits(dev.rms()) == bits(ora.rms())


def test_the_layer_mode_is_set_before_the_first_centroid_and_stays(gpu):
    pts = flop_like_points(64, bins=32, mass=20, seed=4)
    tri = smooth_metric(32, 4)
    dev = lloyd.Layer(4, pts, "sinkhorn", tri, seed=4)
    dev.init_centroids()
    with pytest.raises(Exception):
        dev.set_libm("glibc")  # centroids exist: their tables are in the other arithmetic
    dev = lloyd.Layer(4, pts, "sinkhorn", tri, seed=4)
    dev.set_libm("glibc")
    dev.set_libm("glibc")      # idempotent
    with pytest.raises(Exception):
        dev.set_libm("contract")
    var = lloyd.Layer(4, pts, "variation", seed=4)
    var.set_libm("glibc")      # no exp / ln on the variation path: accepted, nothing changes
    var.init_centroids()


def test_glibc_expf_and_logf_on_the_device_equal_the_host_on_every_float(gpu):
    # include/rp_libm_glibc.h evaluated by the GPU over ALL 2^32 bit patterns (f64 fma, table look-ups, the one rounding to f32)
    # against the host evaluation of the same header — which tests/test_libm_glibc.py holds equal to glibc's own functions on every
    # input: four order-independent checksums (the host's are committed: tests/golden/glibc_checksums.json, kept by a CPU test).
    # Under the execution model (no GPU): 2^19 patterns around 1.0, -88, the subnormals and the infinities, host side recomputed.
    import json
    import os

    from robopoker_amd import _lib

    if os.environ.get("RP_EMUL") != "1":
        want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "glibc_checksums.json")))
        dev = (C.c_uint64 * 4)()
        _lib.check(_lib.load().rp_libm_glibc_sweep(0, want["range"][0], want["range"][1], dev))
        assert [hex(x) for x in dev] == want["sums"]
        return
    o = oracle.load()
    o.ora_libm_glibc_checksums.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    for lo, hi in [(0x3F7F0000, 0x3F810000), (0xC2AF0000, 0xC2B10000), (0x00000000, 0x00020000), (0x7F7F0000, 0x7F810000)]:
        dev = (C.c_uint64 * 4)()
        _lib.check(_lib.load().rp_libm_glibc_sweep(0, lo, hi, dev))
        host = (C.c_uint64 * 4)()
        o.ora_libm_glibc_checksums(lo, hi, host)
        assert list(dev) == list(host), (hex(lo), hex(hi))


def test_the_kernels_branch_free_glibc_forms_equal_the_ladder_forms_on_every_float(gpu):
    # what the lm_glibc kernels evaluate since round 6 — rp_glibc_expf_tab / rp_glibc_exp_floor_tab / rp_glibc_logf_tab on the tables
    # in LDS (csrc/lm_glibc_dev.hpp) — against the header's ladder forms (pinned to the host by the test above), on the device, over
    # ALL 2^32 bit patterns.  Under the execution model: the same four windows as above.
    import os

    from robopoker_amd import _lib

    ranges = [(0, 1 << 32)] if os.environ.get("RP_EMUL") != "1" else [(0x3F7F0000, 0x3F810000), (0xC2AF0000, 0xC2B10000), (0x00000000, 0x00020000),
                                                                      (0x7F7F0000, 0x7F810000), (0x42B00000, 0x42B40000), (0xFF7F0000, 0xFF810000)]
    for lo, hi in ranges:
        bad = (C.c_uint64 * 4)()
        _lib.check(_lib.load().rp_libm_glibc_tab_sweep(0, lo, hi, bad))
        assert list(bad)[:3] == [0, 0, 0], (hex(lo), hex(hi), list(bad)[:3], hex(bad[3] & 0xFFFFFFFF))


def test_the_pruned_glibc_pass_equals_the_oracle(oracle_on_glibc, monkeypatch):
    # the glibc pass keeps the k-means++ column bound, its interval filter and the MFMA bound (rp_kmeans_set_libm).  Exactness does
    # not depend on them as long as every minimiser survives: picks, bounds, an iteration and the lookup against the unpruned oracle,
    # and the on-device audit (RP_LLOYD_AUDIT: the unpruned search behind every pruned pass) counts no disagreement
    monkeypatch.setenv("RP_LLOYD_AUDIT", "1")
    dev, ora = _layer_pair(24, 400, 64, 30, seed=31, iters=24)
    assert np.array_equal(dev.init_centroids(), ora.init_centroids())
    dev.init_bounds()
    ora.init_bounds()
    _check_state(dev, ora)
    dev.step()
    ora.step()
    _check_state(dev, ora)
    b1, dd1 = dev.lookup()
    b2, dd2 = ora.assign()
    assert np.array_equal(b1, b2) and np.array_equal(bits(dd1), bits(dd2))
    st = dev.prune_stats()
    assert st["enabled"] == 1 and st["audited_points"] > 0 and st["audit_mismatches"] == 0
    assert st["survivors"] < st["candidates"]  # it did prune
    assert st["kpp_bound_pairs"] > 0 and st["kpp_bound_kept"] < st["kpp_bound_pairs"]  # ... in the k-means++ rounds too


def test_set_prune_off_runs_every_distance_and_changes_nothing(gpu):
    # rp_kmeans_set_prune(h, 0): Elkan::neighbor / Layer::init_centroids as the reference loops them — K exact solves per point — in
    # either arithmetic; same picks, bounds and buckets as the filtered passes, and the mode is refused once centroids exist
    N, K, bins = 500, 12, 64
    pts = flop_like_points(N, bins=bins, mass=30, seed=77)
    tri = smooth_metric(bins, 2)
    got = {}
    for libm in ("contract", "glibc"):
        for prune in (True, False):
            layer = lloyd.Layer(K, pts, "sinkhorn", tri, seed=9)
            if libm == "glibc":
                layer.set_libm("glibc")
            if not prune:
                layer.set_prune(False)
            chosen = np.asarray(layer.init_centroids())
            layer.init_bounds()
            d_init = layer.stats()[0]
            layer.step()
            b, d = layer.lookup()
            st = layer.prune_stats()
            got[(libm, prune)] = (chosen, np.asarray(b), bits(d))
            if prune:
                assert st["enabled"] == 1 and st["kpp_bound_pairs"] > 0
            else:
                assert st["enabled"] == 0 and st["kpp_bound_pairs"] == 0
                assert d_init >= N * K  # k-means++ and init_bounds solved every pair
                with pytest.raises(_lib.RpError):
                    layer.set_prune(False)  # too late: the layer has centroids
        a, c = got[(libm, True)], got[(libm, False)]
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2])

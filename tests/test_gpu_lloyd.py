"""GPU parity: the HIP lloyd path (through the C-ABI) against the CPU oracle — bit-exact.

Every f32 the kernels produce follows the oracle's (= the reference's) operation order, so Sinkhorn costs,
divergences, Elkan bounds, drifts and bucket assignments must be IDENTICAL, not merely close.
"""
import ctypes as C

import numpy as np
import pytest

import oracle
from lloyd_fixtures import flop_hist, flop_like_points, flop_metric, random_metric, smooth_metric, turn_like_points
from robopoker_amd import lloyd

pytestmark = pytest.mark.gpu


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def test_sinkhorn_fixture_bit_exact_and_properties(gpu):
    # the reference's closed-form fixture (sinkhorn.rs:240-293)
    tri = flop_metric()
    mu = flop_hist([(0, 3), (5, 1), (12, 4)])
    nu = flop_hist([(2, 2), (8, 5), (20, 1), (24, 3)])
    h = flop_hist([(0, 3), (5, 1), (12, 4), (24, 2)])
    d = lloyd.sinkhorn_divergence(np.stack([mu, nu, h]), np.stack([nu, mu, h]), tri)
    exp = [oracle.sinkhorn_divergence(mu, nu, tri), oracle.sinkhorn_divergence(nu, mu, tri),
           oracle.sinkhorn_divergence(h, h, tri)]
    assert np.array_equal(bits(d), bits(exp))
    assert abs(d[2]) < 1e-4 and abs(d[0] - d[1]) < 1e-3 and d[0] > 0
    c, it = lloyd.sinkhorn_cost(np.stack([mu, h]), np.stack([nu, h]), tri)
    for k, (a, b) in enumerate([(mu, nu), (h, h)]):
        ec, eit = oracle.sinkhorn_cost(a, b, tri)
        assert bits(c[k]) == bits(ec) and it[k] == eit


# the last four: a side with <= 32 rows against >= 32 columns takes two lanes per row (softmin_sum_split): either side, column counts
# with every remainder class of 16 and of 8
@pytest.mark.parametrize("bins,nnz_a,nnz_b", [(32, 5, 9), (101, 30, 60), (256, 47, 256), (256, 256, 47), (64, 1, 64),
                                              (256, 256, 20), (256, 17, 250), (128, 32, 33), (256, 201, 32)])
def test_sinkhorn_random_pairs_bit_exact(gpu, bins, nnz_a, nnz_b):
    rng = np.random.default_rng(bins * 1000 + nnz_a)
    tri = random_metric(bins, rng)
    P = 6
    mu = np.zeros((P, bins), dtype=np.uint32)
    nu = np.zeros((P, bins), dtype=np.uint32)
    for p in range(P):
        mu[p, rng.choice(bins, nnz_a, replace=False)] = rng.integers(1, 9, nnz_a)
        nu[p, rng.choice(bins, nnz_b, replace=False)] = rng.integers(1, 2000, nnz_b)
    hp = oracle.default_sinkhorn()
    hp.iterations = 24  # keeps the CPU oracle quick; the cap itself is part of the checked behaviour
    d = lloyd.sinkhorn_divergence(mu, nu, tri, hp)
    c, it = lloyd.sinkhorn_cost(mu, nu, tri, hp)
    for p in range(P):
        assert bits(d[p]) == bits(oracle.sinkhorn_divergence(mu[p], nu[p], tri, hp))
        ec, eit = oracle.sinkhorn_cost(mu[p], nu[p], tri, hp)
        assert bits(c[p]) == bits(ec) and it[p] == eit


def test_empty_histogram_costs_zero(gpu):
    tri = flop_metric()
    d = lloyd.sinkhorn_divergence(flop_hist([]), flop_hist([(2, 2), (8, 5)]), tri)
    assert d[0] == 0.0
    # the raw cost of an empty support is an EMPTY f32 sum (sinkhorn.rs:211-216), which libcore folds from -0.0 since Rust 1.83
    # (the workspace asks for 1.90): sign bit set; the divergence's x - 0.5 xx - 0.5 yy and .max(0.0) bring +0.0 back
    empty, some = flop_hist([]), flop_hist([(2, 2), (8, 5)])
    c, it = lloyd.sinkhorn_cost(np.stack([empty, some, empty]), np.stack([some, empty, empty]), tri)
    assert [int(x) for x in bits(c)] == [0x80000000] * 3
    assert bits(c[0]) == bits(oracle.sinkhorn_cost(empty, some, tri)[0])
    d = lloyd.sinkhorn_divergence(np.stack([empty, empty]), np.stack([some, empty]), tri)
    assert [int(x) for x in bits(d)] == [0, 0]


def test_equity_variation_bit_exact(gpu):
    pts = turn_like_points(64, bins=101, mass=46, seed=3).astype(np.uint32)
    x, y = pts[:32], pts[32:]
    d = lloyd.equity_variation(x, y)
    exp = [oracle.equity_variation(x[i], y[i]) for i in range(32)]
    assert np.array_equal(bits(d), bits(exp))
    assert np.array_equal(bits(lloyd.equity_variation(y, x)), bits(d))  # exactly symmetric (emd.rs:72-80)
    assert np.all(lloyd.equity_variation(x, x) == 0.0)


def _pair(kind, K, N, bins, mass, seed, iters=None):
    if kind == "sinkhorn":
        pts = flop_like_points(N, bins=bins, mass=mass, seed=seed)
        tri = smooth_metric(bins, seed)
    else:
        pts = turn_like_points(N, bins=bins, mass=mass, seed=seed)
        tri = None
    hp = oracle.default_sinkhorn()
    if iters:
        hp.iterations = iters
    return (lloyd.Layer(K, pts, kind, tri, hp=hp, seed=seed), oracle.OracleKmeans(K, pts, kind, tri, hp=hp, seed=seed))


def _check_state(dev, ora):
    """assignments, centroids and every bound the device holds exactly: bit for bit.  Where the interval-decided refresh
    (csrc/refresh_bound.hpp) settled a point, the device holds an interval instead of Bounds::error: [ulo, u] must contain the
    oracle's value, and l[assigned centroid] (the reference sets it to u at the refresh) must lie in it too.  Returns the number
    of interval-valued points."""
    j1, u1, l1 = dev.bounds()
    j2, u2, l2 = ora.bounds()
    assert np.array_equal(j1, j2), "assignments differ"
    ulo, uiv = dev.upper_interval()
    iv = uiv != 0
    assert np.array_equal(bits(u1[~iv]), bits(u2[~iv])), "upper bounds differ"
    assert np.all(ulo[iv] <= u2[iv]) and np.all(u2[iv] <= u1[iv]), "an interval does not contain the reference's upper bound"
    own = np.zeros(l1.shape, dtype=bool)
    own[np.nonzero(iv)[0], j1[iv].astype(np.int64)] = True
    assert np.array_equal(bits(l1[~own]), bits(l2[~own])), "lower bounds differ"
    assert np.all(l1[own] <= l2[own]), "the lower bound of an interval-valued point's own centroid exceeds the reference's"
    c1, w1 = dev.centroids()
    c2, w2 = ora.centroids()
    assert np.array_equal(c1, c2) and np.array_equal(w1, w2), "centroids differ"
    return int(iv.sum())


@pytest.mark.parametrize("kind,K,N,bins,mass,street", [("sinkhorn", 12, 1500, 32, 20, 1), ("variation", 40, 70001, 101, 46, 2),
                                                       ("variation", 7, 1024, 64, 30, 2), ("variation", 3, 5, 64, 30, 1)])
def test_reference_seed_kmeanspp_picks_equal_the_oracle(gpu, kind, K, N, bins, mass, street):
    # rp_kmeans_set_rng(RP_RNG_REFERENCE): Layer::init_centroids' own chain (layer.rs:155-178) — DefaultHasher(street) ->
    # SmallRng, one generator for the K picks, WeightedIndex<f32> over f32 running sums in index order (k_kpp_ref_pick: one
    # wavefront, 1024 potentials per LDS chunk; N = 70001 and 1024 sit on the chunk edges) — picks, then the state after a step
    dev, ora = _pair(kind, K, N, bins, mass, seed=21)
    dev.set_rng("reference", street)
    ora.set_rng("reference", street)
    assert np.array_equal(dev.init_centroids(), ora.init_centroids()), "k-means++ picks differ"
    dev.init_bounds()
    ora.init_bounds()
    dev.step()
    ora.step()
    _check_state(dev, ora)


def _host_weighted_index(w, v01):
    """rand 0.9.2 WeightedIndex<f32>::new + sample for a given value0_1: sequential f32 running sums, x = v01 * scale, partition_point"""
    cum = np.add.accumulate(w, dtype=np.float32)
    total = cum[-1]
    if not total > 0:
        return len(w), total
    scale = total
    while np.float32(np.float32(scale * np.float32(1 - 2.0 ** -23)) + np.float32(0)) >= total:
        scale = np.nextafter(scale, np.float32(0), dtype=np.float32)
    x = np.float32(np.float32(v01) * scale)
    return min(int(np.searchsorted(cum[:-1], x, side="right")), len(w) - 1), total


def _weight_sets(big):
    rng = np.random.default_rng(5)
    n = 300_000 if big else 3_000
    d2 = (rng.gamma(2.0, 0.02, n) ** 2).astype(np.float32)  # what potentials look like: squared distances
    yield "squared distances", d2
    yield "ones (the first pick)", np.ones(n, dtype=np.float32)
    z = d2.copy()
    z[rng.random(n) < 0.3] = 0
    yield "a third zeros", z
    t = np.ones(n, dtype=np.float32)
    t[0] = np.float32(2.0 ** 24)  # every later 1.0 is half an ulp of the sum: a tie at every term
    yield "ties at every term", t
    h = d2.copy()
    h[n // 2] = np.float32(1e9)  # one term that jumps thirty binades
    h[n // 3] = np.float32(2.0 ** -140)  # a subnormal
    yield "a huge term and a subnormal", h
    g = (2.0 ** rng.integers(-30, 8, n)).astype(np.float32)  # powers of two: exact sums, binade edges hit exactly
    yield "powers of two", g
    yield "short", d2[:5]
    yield "one chunk and one term", d2[:257]
    yield "all zero", np.zeros(700, dtype=np.float32)


def test_the_chunked_weighted_index_equals_a_term_by_term_walk(gpu):
    # csrc/kpp_refpick.hpp: the reference-seed draw's N dependent f32 additions replaced by one exact integer addition per 256-term
    # chunk wherever the running sum stays inside a binade and no term is a tie, walked term by term elsewhere — index AND total against
    # a host loop (numpy's accumulate is sequential) and against round 5's one-wavefront kernel, on weights built to hit the exceptions
    import os

    from robopoker_amd import _lib

    lib = _lib.load()
    big = os.environ.get("RP_EMUL") != "1"
    for name, w in _weight_sets(big):
        for v01 in (0.0, 0.37, 0.5, 0.9999999):
            want_i, want_t = _host_weighted_index(w, v01)
            for mode in (0, 1):
                out = (C.c_uint64 * 3)()
                _lib.check(lib.rp_weighted_index_probe(0, len(w), w.ctypes.data_as(C.c_void_p), C.c_float(v01), mode, out))
                assert int(out[0]) == want_i and int(out[1]) == int(np.float32(want_t).view(np.uint32)), (name, v01, mode, list(out), want_i)
            if name == "squared distances" and big:
                assert out[2] < 120, out[2]  # of 1172 chunks: the first, the binade crossings, the ties


@pytest.mark.parametrize("kind,K,N,bins,mass", [("sinkhorn", 5, 150, 32, 20), ("sinkhorn", 70, 200, 48, 24),
                                                ("variation", 8, 2048, 101, 46), ("variation", 130, 700, 101, 46),
                                                ("variation", 20, 500, 64, 30)])
def test_elkan_iterations_bit_exact(gpu, kind, K, N, bins, mass):
    dev, ora = _pair(kind, K, N, bins, mass, seed=K + N, iters=16)
    assert np.array_equal(dev.init_centroids(), ora.init_centroids()), "k-means++ picks differ"
    dev.init_bounds()
    ora.init_bounds()
    _check_state(dev, ora)
    for _ in range(4):
        e0, o0, r0 = dev.stats_ex(), oracle.lloyd_stats()[0], dev.refresh_stats()
        d1, s1, m1 = dev.step()
        d2, s2, m2 = ora.step()
        assert np.array_equal(bits(d1), bits(d2)), "drift differs"
        assert np.array_equal(s1, s2) and m1 == m2
        _check_state(dev, ora)
        # the distances Elkan's rule evaluates in this step (has_shifted BEFORE the distance, bounds.rs:57-61): the reference's count —
        # solved here, remembered from an earlier step (same centroid content), or a refresh settled by its interval (csrc/
        # refresh_bound.hpp; the exactify solves are extra and counted apart); the turn kernels compute whole tiles on top
        e1, o1, r1 = dev.stats_ex(), oracle.lloyd_stats()[0], dev.refresh_stats()
        assert (e1["evaluated"] - e0["evaluated"]) + (e1["remembered"] - e0["remembered"]) + (r1["settled"] - r0["settled"]) == o1 - o0, \
            (kind, e0, e1, r0, r1, o0, o1)
    b1, dd1 = dev.lookup()
    b2, dd2 = ora.assign()
    assert np.array_equal(b1, b2) and np.array_equal(bits(dd1), bits(dd2))
    assert np.array_equal(bits(dev.metric()), bits(ora.metric()))
    assert bits(dev.rms()) == bits(ora.rms())


def test_recompute_shared_by_several_workgroups_per_centroid(gpu):
    # from 2^17 points on Elkan::recompute (elkan.rs:128-142) runs eight workgroups per centroid that add their integer sums into zeroed
    # outputs (k_recompute, gridDim.y): centroids, sizes, drift bits and every bound of two steps against the oracle
    dev, ora = _pair("variation", 6, 140000, 24, 12, seed=5, iters=16)
    assert np.array_equal(dev.init_centroids(), ora.init_centroids()), "k-means++ picks differ"
    dev.init_bounds()
    ora.init_bounds()
    _check_state(dev, ora)
    for _ in range(2):
        d1, s1, m1 = dev.step()
        d2, s2, m2 = ora.step()
        assert np.array_equal(bits(d1), bits(d2)), "drift differs"
        assert np.array_equal(s1, s2) and m1 == m2 and int(np.sum(s1)) == 140000
        _check_state(dev, ora)
    assert bits(dev.rms()) == bits(ora.rms())


@pytest.mark.parametrize("libm", ["contract", "glibc"])
def test_interval_decided_refresh_keeps_the_reference_state(gpu, libm):
    # csrc/refresh_bound.hpp: from the second iteration on most stale-bound refreshes (elkan.rs:113-117) are settled by a scaling-domain
    # interval instead of a bit-faithful solve.  Per step against the oracle: assignments, drift bits, sizes, centroids and every exact
    # bound identical; interval-valued upper bounds contain the oracle's value (_check_state); the interval kernel really ran and
    # settled points, intervals that could not decide the next filter were replaced by exact values (exactify), and the reference's
    # distance count is evaluated + remembered + settled.  The same layer with rp_kmeans_set_prune(0) holds no interval at any step.
    o = oracle.load()
    o.ora_lloyd_set_libm.argtypes = [C.c_int]
    if libm == "glibc":
        o.ora_lloyd_set_libm(2)  # before the oracle's layer exists: its OT(p, p) terms are computed at creation
    try:
        dev, ora = _pair("sinkhorn", 12, 600, 32, 20, seed=7, iters=16)
        if libm == "glibc":
            dev.set_libm("glibc")
        assert np.array_equal(dev.init_centroids(), ora.init_centroids()), "k-means++ picks differ"
        dev.init_bounds()
        ora.init_bounds()
        assert _check_state(dev, ora) == 0
        held = 0
        for it in range(6):
            e0, o0, r0 = dev.stats_ex(), oracle.lloyd_stats()[0], dev.refresh_stats()
            d1, s1, m1 = dev.step()
            d2, s2, m2 = ora.step()
            assert np.array_equal(bits(d1), bits(d2)), ("drift differs", it)
            assert np.array_equal(s1, s2) and m1 == m2
            held += _check_state(dev, ora)
            e1, o1, r1 = dev.stats_ex(), oracle.lloyd_stats()[0], dev.refresh_stats()
            assert (e1["evaluated"] - e0["evaluated"]) + (e1["remembered"] - e0["remembered"]) + (r1["settled"] - r0["settled"]) == o1 - o0
        st = dev.refresh_stats()
        assert st["enabled"] == 1 and st["settled"] > 300 and st["exactify_solves"] > 0 and held > 300, st
        b1, dd1 = dev.lookup()
        b2, dd2 = ora.assign()
        assert np.array_equal(b1, b2) and np.array_equal(bits(dd1), bits(dd2))
        # the exact-by-construction mode: no interval anywhere, every bound the oracle's
        plain, ora2 = _pair("sinkhorn", 12, 600, 32, 20, seed=7, iters=16)
        if libm == "glibc":
            plain.set_libm("glibc")
        plain.set_prune(False)
        plain.init_centroids()
        ora2.init_centroids()
        plain.init_bounds()
        ora2.init_bounds()
        for _ in range(3):
            plain.step()
            ora2.step()
            assert _check_state(plain, ora2) == 0
        assert plain.refresh_stats()["enabled"] == 0
    finally:
        o.ora_lloyd_set_libm(0)


@pytest.mark.parametrize("kind,K,N,bins,mass", [("sinkhorn", 1, 5, 2, 3), ("sinkhorn", 2, 2, 5, 4), ("sinkhorn", 65, 130, 33, 7),
                                                ("sinkhorn", 2, 1000, 64, 47), ("variation", 1, 5, 7, 3), ("variation", 64, 64, 7, 3),
                                                ("variation", 65, 130, 2, 5), ("variation", 2, 2, 101, 46)])
def test_layer_shape_corners_bit_exact(gpu, kind, K, N, bins, mass):
    # one cluster, as many clusters as points, more clusters than DISTINCT points (empty clusters: 0/0 densities, NaN distances
    # that f32::max skips when the metric is normalised, metric.rs:127-141), two bins, a bin count that is no multiple of four,
    # 65 clusters (one past a wavefront of lower bounds): picks, bounds, drift, sizes, lookup and metric against the oracle
    dev, ora = _pair(kind, K, N, bins, mass, seed=K + N, iters=10)
    assert np.array_equal(dev.init_centroids(), ora.init_centroids())
    dev.init_bounds()
    ora.init_bounds()
    for _ in range(2):
        d1, s1, m1 = dev.step()
        d2, s2, m2 = ora.step()
        assert np.array_equal(bits(d1), bits(d2)) and np.array_equal(s1, s2) and m1 == m2
        _check_state(dev, ora)
    b1, dd1 = dev.lookup()
    b2, dd2 = ora.assign()
    assert np.array_equal(b1, b2) and np.array_equal(bits(dd1), bits(dd2))
    assert np.array_equal(bits(dev.metric()), bits(ora.metric()))


def test_elkan_equals_naive_on_device(gpu):
    # crates/lloyd/src/tests.rs:148-161 (variation layer, K=8, N=2048, 8 iterations) on the GPU
    pts = turn_like_points(2048, bins=101, mass=46, seed=1)
    start = np.random.default_rng(1).choice(2048, size=8, replace=False).astype(np.uint64)
    e = lloyd.Layer(8, pts, "variation", seed=1)
    n = lloyd.Layer(8, pts, "variation", seed=1)
    e.set_centroids(start)
    n.set_centroids(start)
    e.init_bounds()
    for _ in range(8):
        e.step()
        n.step_naive()
        ce, we = e.centroids()
        cn, wn = n.centroids()
        assert np.array_equal(ce, cn) and np.array_equal(we, wn)


def test_step_naive_and_set_centroids_bit_exact(gpu):
    dev, ora = _pair("sinkhorn", 6, 120, 32, 16, seed=4, iters=12)
    start = np.arange(0, 120, 20, dtype=np.uint64)
    dev.set_centroids(start)
    ora.set_centroids(start)
    for _ in range(2):
        dev.step_naive()
        ora.step_naive()
        c1, w1 = dev.centroids()
        c2, w2 = ora.centroids()
        assert np.array_equal(c1, c2) and np.array_equal(w1, w2)


def test_empty_cluster_attracts_like_the_reference(gpu):
    # duplicate seeds leave one cluster empty after the first step; an empty centroid
    # has divergence 0 to everything.  The device must reproduce the oracle here too.
    dev, ora = _pair("sinkhorn", 3, 60, 32, 16, seed=8, iters=12)
    start = np.array([5, 5, 17], dtype=np.uint64)
    for km in (dev, ora):
        km.set_centroids(start)
        km.init_bounds()
    for _ in range(3):
        d1, s1, _ = dev.step()
        d2, s2, _ = ora.step()
        assert np.array_equal(bits(d1), bits(d2)) and np.array_equal(s1, s2)
    b1, _ = dev.lookup()
    b2, _ = ora.assign()
    assert np.array_equal(b1, b2)


def test_preflop_shape_k_equals_n_metric_only(gpu):
    # Pref layer: K = N, zero iterations, only Layer::metric runs Sinkhorn
    pts = flop_like_points(24, bins=40, mass=30, seed=2)
    tri = smooth_metric(40, 2)
    hp = oracle.default_sinkhorn()
    hp.iterations = 16
    dev = lloyd.Layer(24, pts, "sinkhorn", tri, hp=hp)
    ora = oracle.OracleKmeans(24, pts, "sinkhorn", tri, hp=hp)
    idx = np.arange(24, dtype=np.uint64)
    dev.set_centroids(idx)
    ora.set_centroids(idx)
    assert np.array_equal(bits(dev.metric()), bits(ora.metric()))


def test_flop_config_slice_properties(gpu):
    # BASELINE config 3 shape (K=256, 256 bins, mass 47) on a slice: size-independent properties
    N, K = 1024, 256
    pts = flop_like_points(N, bins=256, mass=47, seed=0xF10F)
    tri = smooth_metric(256, 1)
    dev = lloyd.Layer(K, pts, "sinkhorn", tri, seed=1)
    dev.set_centroids(np.arange(0, N, N // K, dtype=np.uint64)[:K])
    dev.init_bounds()
    j0, u0, _ = dev.bounds()
    assert np.all(u0 >= 0) and np.all(u0[np.arange(0, N, N // K)[:K]] == 0.0)  # a seed point is at distance 0 of itself
    drift, sizes, moved = dev.step()
    c, w = dev.centroids()
    assert sizes.sum() == N and w.sum() == N * 47 and c.sum() == N * 47  # centroids are exact integer sums
    assert np.array_equal(c.sum(axis=1), w)
    j1, _, lo = dev.bounds()
    for k in range(K):  # recompute() == sum of members, checked on the host
        assert np.array_equal(c[k], pts[j1 == k].astype(np.uint32).sum(axis=0))
    assert np.all(lo >= 0) and np.all(drift >= 0) and 0.0 <= moved <= 1.0


@pytest.mark.parametrize("small_supports", [False, True])
def test_points_per_wavefront_groupings_agree(gpu, monkeypatch, small_supports):
    # Points with <= 16 support bins are solved four per wavefront against a shared centroid, those with <= 32 two per
    # wavefront (k_neighborG / k_kpp_updateG, k_refresh_pairs), the rest one; RP_LLOYD_GROUPING=pairs / none /
    # norefresh switch the groupings off.  Every grouping performs the same float operations per solve: identical picks, buckets and
    # distances.  Large enough that the lists are built while the GPU is busy (a missing stream sync once
    # mis-classified late points).  small_supports: like the real flop points (<= 27 bins, ~11 on average), so
    # that most points take the four-per-wavefront path and centroids the shared centroid-row pass.
    N, K, bins = 60000, 48, 256
    if small_supports:
        rng = np.random.default_rng(8)
        pts = np.zeros((N, bins), dtype=np.uint8)
        for i in range(N):
            k = int(rng.integers(1, 28))
            sup = rng.choice(bins, size=k, replace=False)
            pts[i, sup] = 1
            extra = rng.choice(sup, size=47 - k, replace=True)
            np.add.at(pts[i], extra, 1)
        assert (pts.sum(axis=1) == 47).all()
    else:
        pts = flop_like_points(N, bins=bins, mass=47, seed=3)
    tri = smooth_metric(bins, 1)

    def run(env):
        for v in ("RP_LLOYD_GROUPING", "RP_LLOYD_NO_KPP_BOUND", "RP_LLOYD_NO_MFMA_BOUND"):
            monkeypatch.delenv(v, raising=False)
        if env:
            monkeypatch.setenv(*env)
        layer = lloyd.Layer(K, pts, "sinkhorn", tri, seed=11)
        chosen = np.asarray(layer.init_centroids())
        layer.init_bounds()
        layer.step()
        layer.step()  # the second iteration starts from stale bounds: the grouped refresh pass runs
        bucket, dist = layer.lookup()
        return chosen, np.asarray(bucket), np.asarray(dist)

    c1, b1, d1 = run(("RP_LLOYD_GROUPING", "none"))
    # ... and the two prunes (k-means++ column-marginal bound, MFMA bound of the neighbor passes) against their absence
    for env in (None, ("RP_LLOYD_GROUPING", "pairs"), ("RP_LLOYD_GROUPING", "norefresh"), ("RP_LLOYD_NO_KPP_BOUND", "1"),
                ("RP_LLOYD_NO_MFMA_BOUND", "1")):
        c2, b2, d2 = run(env)
        assert np.array_equal(c1, c2), "k-means++ picks differ"
        assert np.array_equal(b1, b2)
        assert np.array_equal(d1.view(np.uint32), d2.view(np.uint32))


@pytest.mark.parametrize("temperature,iterations,tolerance", [(0.025, 128, 5e-4), (0.1, 64, 1e-3), (0.01, 200, 1e-5),
                                                              (0.5, 8, 1e-9)])
def test_sinkhorn_hyper_parameter_corners_bit_exact(gpu, temperature, iterations, tolerance):
    # T, the iteration cap and the tolerance of Sinkhorn (sinkhorn.rs:77-92,129-139): cost, divergence and the
    # iteration count itself must match the oracle bit for bit; T = 0.01 drives exp arguments to -100 (the
    # MIN_POSITIVE clamp), 8 iterations with a tiny tolerance exercises the cap
    rng = np.random.default_rng(5)
    bins, pairs = 64, 24
    tri = random_metric(bins, rng)
    mu = np.zeros((pairs, bins), dtype=np.uint32)
    nu = np.zeros((pairs, bins), dtype=np.uint32)
    for p in range(pairs):
        for h in (mu, nu):
            sup = rng.choice(bins, size=rng.integers(1, 40), replace=False)
            h[p, sup] = rng.integers(1, 9, size=sup.size)
    hp = oracle.default_sinkhorn()
    hp.temperature, hp.iterations, hp.tolerance = temperature, iterations, tolerance
    d = lloyd.sinkhorn_divergence(mu, nu, tri, hp)
    c, it = lloyd.sinkhorn_cost(mu, nu, tri, hp)
    for p in range(pairs):
        assert bits(d[p]) == bits(oracle.sinkhorn_divergence(mu[p], nu[p], tri, hp)), p
        ec, eit = oracle.sinkhorn_cost(mu[p], nu[p], tri, hp)
        assert bits(c[p]) == bits(ec) and it[p] == eit, p


# ---- the MFMA Sinkhorn bound in front of the neighbor passes (sinkhorn_bound.hpp) --------------------
def _dense_centroid_layers(N, K, bins, mass, seed, iters=None):
    """a device layer and an oracle layer whose centroids are SUMS of point groups (broad supports, like converged
    centroids) for the even k and single points for the odd k (sparse, like k-means++ seeds)"""
    pts = flop_like_points(N, bins=bins, mass=mass, seed=seed)
    tri = smooth_metric(bins, seed)
    hp = oracle.default_sinkhorn()
    if iters:
        hp.iterations = iters
    dev = lloyd.Layer(K, pts, "sinkhorn", tri, hp=hp, seed=seed)
    ora = oracle.OracleKmeans(K, pts, "sinkhorn", tri, hp=hp, seed=seed)
    rng = np.random.default_rng(seed)
    start = rng.choice(N, size=K, replace=False).astype(np.uint64)
    cents = np.zeros((K, bins), dtype=np.uint32)
    order = np.argsort(pts.astype(np.int64) @ np.arange(bins))  # neighbours in the 1-d embedding
    for k in range(K):
        if k % 2:
            cents[k] = pts[start[k]]
        else:
            lo = int(rng.integers(0, N - 12))
            cents[k] = pts[order[lo:lo + 12]].astype(np.uint32).sum(axis=0)
    for km in (dev, ora):
        km.set_centroids(start)
        for k in range(K):
            km.set_centroid(k, cents[k])
    return dev, ora, pts, cents, tri, hp


@pytest.mark.parametrize("drop", [True, False])
@pytest.mark.parametrize("N,K,bins,mass,iters", [(64, 40, 64, 30, None), (40, 37, 256, 47, None), (48, 20, 101, 20, 12)])
def test_mfma_bound_intervals_contain_the_oracle(gpu, monkeypatch, N, K, bins, mass, iters, drop):
    # every interval of the scaling-domain bound must contain the value the bit-faithful solve returns (the oracle's
    # Sinkhorn::divergence, centroid first as in Elkan::neighbor), be tight for typical pairs, and leave few survivors
    # drop = True: columns are ordered by the rigorous column-marginal bound and dropped ([bound, inf)) once it exceeds a
    # published upper bound; RP_SB_NO_LB0 follows every column to the end of its stopping window
    if not drop:
        monkeypatch.setenv("RP_SB_NO_LB0", "1")
    dev, ora, pts, cents, tri, hp = _dense_centroid_layers(N, K, bins, mass, seed=N + K, iters=iters)
    lo, hi = dev.bound_intervals()
    assert lo.shape == (N, K) and np.all(lo >= 0) and np.all(hi >= lo)
    exact = np.zeros((N, K), dtype=np.float32)
    for i in range(N):
        for k in range(K):
            exact[i, k] = oracle.sinkhorn_divergence(cents[k], pts[i].astype(np.uint32), tri, hp, bins)
    bad = np.argwhere((exact < lo) | (exact > hi))
    assert bad.size == 0, f"{len(bad)} intervals miss the exact value, first {bad[:3]}: " \
                          f"{[(lo[i, k], exact[i, k], hi[i, k]) for i, k in bad[:3]]}"
    finite = np.isfinite(hi)
    assert finite.mean() > (0.1 if drop else 0.99)  # (round 6: far columns leave by the dual bound — 0.15 of K = 37 are followed to the end)
    assert np.all(finite[np.arange(N), exact.argmin(axis=1)])  # the nearest centroid is always followed to the end
    # typical width of the intervals that matter — the centroids within four times the smallest upper bound —: the margin, not the
    # stopping window (a far column's window iterates may be taken from a Lipschitz bound: wider by design, round 6)
    close = finite & (lo <= 4.0 * hi.min(axis=1, keepdims=True))
    assert np.median((hi - lo)[close]) < 2e-4
    survivors = (lo <= hi.min(axis=1, keepdims=True)).sum(axis=1)
    assert survivors.mean() < 3.0, survivors
    # the pruned passes themselves: identical buckets and distances
    b1, d1 = dev.lookup()
    b2, d2 = ora.assign()
    assert np.array_equal(b1, b2) and np.array_equal(bits(d1), bits(d2))
    st = dev.prune_stats()
    assert st["enabled"] == 1 and st["survivors"] < st["candidates"] // 4


@pytest.mark.parametrize("N,K,bins,mass,iters", [(300, 6, 256, 47, None), (200, 5, 64, 20, None), (160, 4, 101, 30, 12)])
def test_kmeanspp_interval_filter_lower_bounds_hold(gpu, N, K, bins, mass, iters):
    # the second k-means++ filter (kpp_bound.hpp): for every (point-like centroid, point) pair its lower bound must not exceed the
    # distance the bit-faithful solve returns (oracle: Sinkhorn::divergence, centroid first as in layer.rs:170-178), and it must be
    # tight enough to be worth running: within the margin for most pairs
    pts = flop_like_points(N, bins=bins, mass=mass, seed=N + bins)
    tri = smooth_metric(bins, K)
    hp = oracle.default_sinkhorn()
    if iters:
        hp.iterations = iters
    dev = lloyd.Layer(K, pts, "sinkhorn", tri, hp=hp, seed=3)
    start = np.random.default_rng(N).choice(N, size=K, replace=False).astype(np.uint64)
    dev.set_centroids(start)
    tight = []
    for k in range(K):
        lo = dev.kpp_bound_probe(k)
        assert lo.shape == (N,) and np.all(lo >= 0) and np.all(np.isfinite(lo))
        c = pts[start[k]].astype(np.uint32)
        exact = np.array([oracle.sinkhorn_divergence(c, pts[i].astype(np.uint32), tri, hp, bins) for i in range(N)], dtype=np.float32)
        bad = np.flatnonzero(lo > exact)
        assert bad.size == 0, f"centroid {k}: {bad.size} lower bounds above the exact distance, first {[(lo[i], exact[i]) for i in bad[:3]]}"
        got = lo > 0
        tight.append(np.mean((exact[got] - lo[got]) <= 1e-4 + 1e-3 * exact[got]) if got.any() else 0.0)
        assert got.mean() > 0.5  # a bound for most pairs (0 = no bound: outside the tile, or a window that did not close)
        # the production rule against one potential (round 6: a pair may leave by the Kantorovich dual bound before its window closes):
        # whatever a pair leaves with is a lower bound of the exact distance and reaches the potential; most pairs beyond it do leave
        for q in (0.1, 0.5):
            pd = float(np.quantile(exact, q))
            lo2 = dev.kpp_bound_probe(k, potential=pd * pd)
            left = lo2 > 0
            bad2 = np.flatnonzero(lo2 > exact)
            assert bad2.size == 0, f"centroid {k}, potential {pd}^2: {[(lo2[i], exact[i]) for i in bad2[:3]]}"
            assert np.all(lo2[left].astype(np.float32) ** 2 >= np.float32(pd * pd) * np.float32(0.999999))
            beyond = exact > 1.5 * pd + 1e-3
            assert beyond.sum() == 0 or left[beyond].mean() > 0.8, (k, q, left[beyond].mean())
    assert np.mean(tight) > 0.8, tight


def test_kmeanspp_interval_filter_keeps_the_picks(gpu, monkeypatch):
    # k-means++ with and without the interval filter: identical picks, potentials' consequences (init_bounds from the notes) and buckets;
    # the filter must actually discard most of what the column bound lets through
    N, K, bins = 6000, 24, 256
    pts = flop_like_points(N, bins=bins, mass=47, seed=21)
    tri = smooth_metric(bins, 1)

    def run(off):
        monkeypatch.delenv("RP_LLOYD_NO_KPP_BOUND2", raising=False)
        if off:
            monkeypatch.setenv("RP_LLOYD_NO_KPP_BOUND2", "1")
        layer = lloyd.Layer(K, pts, "sinkhorn", tri, seed=4)
        chosen = np.asarray(layer.init_centroids())
        layer.init_bounds()
        j, u, _ = layer.bounds()
        return chosen, np.asarray(j), np.asarray(u), layer.prune_stats(), layer.stats()[0]

    c0, j0, u0, st0, d0 = run(True)
    c1, j1, u1, st1, d1 = run(False)
    assert np.array_equal(c0, c1), "k-means++ picks differ"
    assert np.array_equal(j0, j1) and np.array_equal(bits(u0), bits(u1))
    assert st0["kpp_bound_pairs"] == 0 and st1["kpp_bound_pairs"] > 0
    assert st1["kpp_bound_kept"] < st1["kpp_bound_pairs"] // 2, st1
    assert d1 < d0  # fewer bit-faithful solves


def test_kmeanspp_rounds_tripwire(gpu, monkeypatch):
    # every 521st point the k-means++ interval filter drops is solved anyway and the bound it was dropped with compared with the exact
    # distance (Metric::kpp_claim, kpp_note).  On a healthy layer nothing is counted and results are those without the wire; with the claims
    # scaled by RP_KPP_CLAIM_TEST (a bound 1000 x too high) the next call that hands results out fails with RP_ERR_INTERNAL.
    from robopoker_amd import _lib
    N, K, bins = 6000, 24, 256
    pts = flop_like_points(N, bins=bins, mass=47, seed=21)
    tri = smooth_metric(bins, 1)

    def run(scale):
        monkeypatch.delenv("RP_KPP_CLAIM_TEST", raising=False)
        if scale:
            monkeypatch.setenv("RP_KPP_CLAIM_TEST", scale)
        layer = lloyd.Layer(K, pts, "sinkhorn", tri, seed=4)
        chosen = np.asarray(layer.init_centroids())
        layer.init_bounds()
        return chosen, layer.prune_stats()

    chosen, st = run(None)
    assert st["kpp_bound_pairs"] > 0 and st["kpp_bound_kept"] < st["kpp_bound_pairs"] and st["sample_mismatches"] == 0
    monkeypatch.setenv("RP_LLOYD_NO_SAMPLE_CHECK", "1")
    chosen0, _ = run(None)
    monkeypatch.delenv("RP_LLOYD_NO_SAMPLE_CHECK")
    assert np.array_equal(chosen, chosen0)
    with pytest.raises(_lib.RpError) as err:
        run("1000")
    assert "sampled points" in str(err.value)


def test_mfma_bound_audit_against_the_unpruned_pass(gpu, monkeypatch):
    # RP_LLOYD_AUDIT=1: every pruned neighbor pass is followed by the unpruned one and compared point by point on the
    # device (the "debug build" of the prune); k-means++ picks, init_bounds, two Elkan iterations, lookup at the
    # configured shape K = 256, bins = 256
    monkeypatch.setenv("RP_LLOYD_AUDIT", "1")
    N, K, bins = 3000, 256, 256
    pts = flop_like_points(N, bins=bins, mass=47, seed=0xF10F)
    tri = smooth_metric(bins, 1)
    dev = lloyd.Layer(K, pts, "sinkhorn", tri, seed=5)
    dev.set_centroids(np.random.default_rng(5).choice(N, size=K, replace=False).astype(np.uint64))
    dev.init_bounds()
    dev.step()
    dev.step()
    dev.lookup()
    st = dev.prune_stats()
    assert st["audited_points"] == 2 * N and st["audit_mismatches"] == 0, st
    assert st["survivors"] < 2 * st["points"], st  # ~1 survivor per point
    monkeypatch.delenv("RP_LLOYD_AUDIT")
    monkeypatch.setenv("RP_LLOYD_NO_MFMA_BOUND", "1")
    ref = lloyd.Layer(K, pts, "sinkhorn", tri, seed=5)
    assert ref.prune_stats()["enabled"] == 0


def test_mfma_bound_handles_degenerate_inputs(gpu):
    # K not a multiple of 16, fewer bins than a tile, a point with a single bin, an empty centroid, a hot temperature
    # (K ~ 1) and a cold one whose exp(-C/T) underflows (intervals open up, nothing may be pruned wrongly)
    for temperature in (0.025, 2.0, 0.0005):
        rng = np.random.default_rng(3)
        bins, K, N = 9, 5, 33
        pts = np.zeros((N, bins), dtype=np.uint8)
        for i in range(N):
            sup = rng.choice(bins, size=int(rng.integers(1, bins + 1)), replace=False)
            pts[i, sup] = rng.integers(1, 6, size=sup.size)
        tri = random_metric(bins, rng)
        hp = oracle.default_sinkhorn()
        hp.temperature = temperature
        dev = lloyd.Layer(K, pts, "sinkhorn", tri, hp=hp, seed=1)
        ora = oracle.OracleKmeans(K, pts, "sinkhorn", tri, hp=hp, seed=1)
        start = np.array([0, 1, 2, 3, 3], dtype=np.uint64)
        for km in (dev, ora):
            km.set_centroids(start)
            km.set_centroid(4, np.zeros(bins, dtype=np.uint32))  # empty cluster: divergence 0 to everything
        b1, d1 = dev.lookup()
        b2, d2 = ora.assign()
        assert np.array_equal(b1, b2) and np.array_equal(bits(d1), bits(d2)), temperature


def test_coupling_flow_bit_exact(gpu):
    # rp_sinkhorn_flow (monge Coupling::flow as Sinkhorn implements it): every cell of flow and coupling
    bins = 40
    tri = smooth_metric(bins, 4)
    pts = flop_like_points(6, bins=bins, mass=25, seed=6).astype(np.uint32)
    for a, b in ((pts[0], pts[1]), (pts[2], pts[2]), (pts[3], pts[5])):
        f1, c1 = lloyd.sinkhorn_flow(a, b, tri)
        f2, c2 = oracle.sinkhorn_flow(a, b, tri)
        assert np.array_equal(bits(f1), bits(f2)) and np.array_equal(bits(c1), bits(c2))

"""Host mirror of the reference's ``elkan::Elkan`` / ``lloyd::Layer`` / ``lloyd::Sinkhorn`` surface over the C-ABI.

``Layer`` follows crates/lloyd/src/layer.rs + kmeans.rs (``init_centroids`` = k-means++, ``init_bounds``,
``step`` = one ``Kmeans::next``, ``lookup`` = ``Layer::lookup``, ``metric`` = ``Layer::metric``), the free functions
follow ``Sinkhorn::divergence`` / ``Coupling::cost`` / ``Equity::variation``.  All compute runs in librp_mi355x.so's
HIP kernels; there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import _lib


def default_sinkhorn() -> _lib.SinkhornHP:
    hp = _lib.SinkhornHP()
    _lib.load().rp_sinkhorn_hp_default(C.byref(hp))
    return hp


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Layer:
    """``Layer<K, N>`` (layer.rs:23-33): N histogram points, K centroids, Elkan bounds — resident in HBM."""

    def __init__(self, K: int, counts, kind="sinkhorn", tri=None, hp=None, seed=0, device=0, counts_dev_ptr=None,
                 shape=None):
        self._lib = _lib.load()
        self.hp = hp or default_sinkhorn()
        self.K = K
        self._tri = np.ascontiguousarray(tri, dtype=np.float32) if tri is not None else None
        self._h = C.c_void_p()
        if counts_dev_ptr is not None:
            self.N, self.bins = shape
            _lib.check(self._lib.rp_kmeans_create_device(K, self.N, self.bins, C.c_void_p(counts_dev_ptr),
                                                         _lib.METRIC[kind], _p(self._tri), C.byref(self.hp), seed,
                                                         device, C.byref(self._h)))
        else:
            counts = np.ascontiguousarray(counts, dtype=np.uint8)
            self.N, self.bins = counts.shape
            _lib.check(self._lib.rp_kmeans_create(K, self.N, self.bins, _p(counts), _lib.METRIC[kind], _p(self._tri),
                                                  C.byref(self.hp), seed, device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rp_kmeans_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- Elkan / Layer --------------------------------------------------------------------------
    def set_rng(self, kind: str, street: int = 1):
        """"counter" (default: the fixed-point draw) or "reference": Layer::init_centroids' own SmallRng + WeightedIndex<f32>
        (layer.rs:155-178); street = the Street discriminant hashed into the seed (1 = Flop)"""
        _lib.check(self._lib.rp_kmeans_set_rng(self._h, _lib.RNG[kind], street))

    def set_libm(self, kind: str):
        """``"glibc"``: every exp / ln of the layer's Sinkhorn distances is glibc's ``expf`` / ``logf`` (what ``f32::exp`` / ``f32::ln``
        are in a Rust build on Linux), evaluated in double on the device: distances, bounds, drift and buckets are the reference's
        bit for bit.  Before the first centroid; the exact solves cost 2.3 x the default f32 contract's, the filters in front of them
        (audited in both arithmetics) stay."""
        _lib.check(self._lib.rp_kmeans_set_libm(self._h, LIBM[kind]))

    def set_prune(self, enable: bool):
        """``False``: no filter in front of the exact solves — every (point, centroid) distance through the bit-faithful kernel, as
        the reference loops them (the yardstick of the audits).  Before the first centroid."""
        _lib.check(self._lib.rp_kmeans_set_prune(self._h, 1 if enable else 0))

    def init_centroids(self) -> np.ndarray:
        chosen = np.zeros(self.K, dtype=np.uint64)
        _lib.check(self._lib.rp_kmeans_init_centroids(self._h, _p(chosen)))
        return chosen

    def set_centroids(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        _lib.check(self._lib.rp_kmeans_set_centroids(self._h, _p(idx)))

    def set_centroid(self, k: int, counts):
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        _lib.check(self._lib.rp_kmeans_set_centroid(self._h, k, _p(counts)))

    def get_point(self, index: int) -> np.ndarray:
        out = np.zeros(self.bins, dtype=np.uint32)
        _lib.check(self._lib.rp_kmeans_get_point(self._h, index, _p(out)))
        return out

    # k-means++ primitives (a point-sharded job interleaves collectives between them)
    def kpp_begin(self):
        _lib.check(self._lib.rp_kmeans_kpp_begin(self._h))

    def kpp_total(self) -> int:
        t = C.c_uint64()
        _lib.check(self._lib.rp_kmeans_kpp_total(self._h, C.byref(t)))
        return t.value

    def kpp_pick(self, r: int) -> int:
        i = C.c_uint64()
        _lib.check(self._lib.rp_kmeans_kpp_pick(self._h, r, C.byref(i)))
        return i.value

    def kpp_update(self, k: int):
        _lib.check(self._lib.rp_kmeans_kpp_update(self._h, k))

    def init_bounds(self):
        _lib.check(self._lib.rp_kmeans_init_bounds(self._h))

    def step(self):
        drift = np.zeros(self.K, dtype=np.float32)
        sizes = np.zeros(self.K, dtype=np.uint64)
        re = C.c_double()
        _lib.check(self._lib.rp_kmeans_step(self._h, _p(drift), _p(sizes), C.byref(re)))
        return drift, sizes, re.value

    def step_naive(self):
        _lib.check(self._lib.rp_kmeans_step_naive(self._h))

    def lookup(self):
        b = np.zeros(self.N, dtype=np.uint8)
        d = np.zeros(self.N, dtype=np.float32)
        _lib.check(self._lib.rp_kmeans_assign(self._h, _p(b), _p(d)))
        return b, d

    assign = lookup

    def bounds(self, lower=True):
        j = np.zeros(self.N, dtype=np.uint8)
        u = np.zeros(self.N, dtype=np.float32)
        lo = np.zeros((self.N, self.K), dtype=np.float32) if lower else None
        _lib.check(self._lib.rp_kmeans_bounds(self._h, _p(j), _p(u), _p(lo) if lower else None))
        return j, u, lo

    def centroids(self):
        c = np.zeros((self.K, self.bins), dtype=np.uint32)
        w = np.zeros(self.K, dtype=np.uint64)
        _lib.check(self._lib.rp_kmeans_centroids(self._h, _p(c), _p(w)))
        return c, w

    def metric(self) -> np.ndarray:
        t = np.zeros(self.K * (self.K - 1) // 2, dtype=np.float32)
        _lib.check(self._lib.rp_kmeans_metric(self._h, _p(t)))
        return t

    def rms(self) -> float:
        out = C.c_float()
        _lib.check(self._lib.rp_kmeans_rms(self._h, C.byref(out)))
        return out.value

    def stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(self._lib.rp_kmeans_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stats_ex(self) -> dict:
        """raw counters (rp_mi355x_diag.h): evaluated + remembered over Elkan steps is the reference's own distance count"""
        a = np.zeros(5, dtype=np.uint64)
        _lib.check(self._lib.rp_kmeans_stats_ex(self._h, _p(a)))
        return {"evaluated": int(a[0]), "sinkhorn_iterations": int(a[1]), "terms": int(a[2]), "remembered": int(a[3]),
                "computed_beyond": int(a[4])}

    def exp_evals(self) -> int:
        v = C.c_uint64()
        _lib.check(self._lib.rp_kmeans_exp_evals(self._h, C.byref(v)))
        return v.value

    def prune_stats(self) -> dict:
        """what the MFMA Sinkhorn bound in front of the neighbor passes discarded (rp_prune_stats)"""
        st = _lib.PruneStats()
        _lib.check(self._lib.rp_kmeans_prune_stats(self._h, C.byref(st)))
        return {name: getattr(st, name) for name, _ in st._fields_ if name != "reserved"}

    def bound_intervals(self):
        """(lo, hi), each (N, K): the bound's interval for distance(centroid k, point i) against the current centroids"""
        lo = np.zeros((self.N, self.K), dtype=np.float32)
        hi = np.zeros((self.N, self.K), dtype=np.float32)
        _lib.check(self._lib.rp_kmeans_bound_intervals(self._h, _p(lo), _p(hi)))
        return lo, hi

    def pairwise_last(self) -> np.ndarray:
        """(K, K): the centroid-to-centroid distances the last step worked with (rp_mi355x_diag.h)"""
        pw = np.zeros((self.K, self.K), dtype=np.float32)
        _lib.check(self._lib.rp_kmeans_pairwise_last(self._h, _p(pw)))
        return pw

    def refresh_stats(self) -> dict:
        """the interval-decided refresh of the Elkan iterations (rp_mi355x_diag.h)"""
        a = np.zeros(8, dtype=np.uint64)
        _lib.check(self._lib.rp_kmeans_refresh_stats(self._h, _p(a)))
        return {"examined": int(a[0]), "settled": int(a[1]), "pair_iterations": int(a[2]), "cost_passes": int(a[3]),
                "exactify_solves": int(a[4]), "enabled": int(a[5]), "remembered_intervals": int(a[6])}

    def upper_interval(self):
        """(ulo, uiv): where uiv != 0, [ulo, bounds()[1]] contains the reference's upper bound (rp_mi355x_diag.h)"""
        ulo = np.zeros(self.N, dtype=np.float32)
        uiv = np.zeros(self.N, dtype=np.uint8)
        _lib.check(self._lib.rp_kmeans_upper_interval(self._h, _p(ulo), _p(uiv)))
        return ulo, uiv

    def kpp_bound_probe(self, k: int, potential: float | None = None) -> np.ndarray:
        """lo[N]: the second k-means++ filter's lower bound of distance(centroid k, point i) (rp_mi355x_diag.h); with a potential: the
        production rule against it (dual exit included), 0 where the pair was kept for the solve"""
        lo = np.zeros(self.N, dtype=np.float32)
        if potential is None:
            _lib.check(self._lib.rp_kmeans_kpp_bound_probe(self._h, C.c_uint32(k), _p(lo)))
        else:
            _lib.check(self._lib.rp_kmeans_kpp_bound_probe_at(self._h, C.c_uint32(k), C.c_float(potential), _p(lo)))
        return lo

    # ---- multi-GPU exchange --------------------------------------------------------
    def partial_bytes(self) -> int:
        n = C.c_size_t()
        _lib.check(self._lib.rp_kmeans_partial_bytes(self._h, C.byref(n)))
        return n.value

    def step_local(self, partial_dev_ptr: int):
        _lib.check(self._lib.rp_kmeans_step_local(self._h, C.c_void_p(partial_dev_ptr)))

    def step_finish(self, reduced_dev_ptr: int):
        drift = np.zeros(self.K, dtype=np.float32)
        sizes = np.zeros(self.K, dtype=np.uint64)
        re = C.c_double()
        _lib.check(self._lib.rp_kmeans_step_finish(self._h, C.c_void_p(reduced_dev_ptr), _p(drift), _p(sizes),
                                                   C.byref(re)))
        return drift, sizes, re.value

    def step_comm(self, comm):
        """Kmeans::next of a point-sharded job over an ``rp_comm``: integer centroid sums all-reduced by the library"""
        drift = np.zeros(self.K, dtype=np.float32)
        sizes = np.zeros(self.K, dtype=np.uint64)
        re = C.c_double()
        _lib.check(self._lib.rp_kmeans_step_comm(self._h, comm.handle, _p(drift), _p(sizes), C.byref(re)))
        return drift, sizes, re.value

    def set_stream(self, hip_stream_ptr):
        _lib.check(self._lib.rp_kmeans_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    # ---- profiling hooks ------------------------------------------------------------------------
    def profile(self, enable=True):
        _lib.check(self._lib.rp_kmeans_profile(self._h, 1 if enable else 0))

    def kernel_time(self, name: str):
        ms, n = C.c_double(), C.c_uint64()
        _lib.check(self._lib.rp_kmeans_kernel_time(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def margin_audit(points, centroids, tri, hp=None, device=0, chunk=64) -> dict:
    """How much room the MFMA bound's intervals (csrc/sinkhorn_bound.hpp) leave around the exact divergence: for every
    (point, centroid) pair of ``points`` (S, bins) x ``centroids`` (K, bins) the interval [lo, hi] of the bound and the
    bit-faithful ``distance(centroid, point)``.  slack = min(exact - lo, hi - exact) >= 0 or the interval missed; reported
    absolute and relative to the cost margin the kernel applies (dc_abs + dc_rel * d)."""
    points = np.ascontiguousarray(points, dtype=np.uint8)
    centroids = np.ascontiguousarray(centroids, dtype=np.uint32)
    S, K = points.shape[0], centroids.shape[0]
    sub = Layer(K, points, "sinkhorn", tri, hp=hp, seed=1, device=device)
    sub.set_centroids(np.arange(K, dtype=np.uint64) % S)
    for k in range(K):
        sub.set_centroid(k, centroids[k])
    lo, hi = sub.bound_intervals()
    sub.close()
    exact = np.zeros((S, K), dtype=np.float32)
    for a in range(0, S, chunk):
        b = min(S, a + chunk)
        mu = np.repeat(centroids[None, :, :], b - a, axis=0).reshape(-1, centroids.shape[1])
        nu = np.repeat(points[a:b].astype(np.uint32), K, axis=0)
        exact[a:b] = sinkhorn_divergence(mu, nu, tri, hp, device).reshape(b - a, K)
    finite = np.isfinite(hi)
    miss = (exact < lo) | (exact > hi)
    slack = np.minimum(exact - lo, np.where(finite, hi - exact, np.inf))
    margin = 4e-6 + 4e-5 * exact
    nearest = exact.argmin(axis=1)
    return {"pairs": int(S * K), "closed_intervals": int(finite.sum()), "missed": int(miss.sum()),
            "min_slack": float(slack.min()), "min_slack_over_margin": float((slack / margin).min()),
            "median_slack_over_margin": float(np.median((slack / margin)[finite])),
            "nearest_always_closed": bool(finite[np.arange(S), nearest].all()),
            "survivors_per_point": float((lo <= hi.min(axis=1, keepdims=True)).sum(axis=1).mean())}


LIBM = {"contract": 0, "glibc": 1}


def sinkhorn_set_libm(kind: str = "contract") -> None:
    """Which exp / ln ``sinkhorn_cost`` / ``sinkhorn_divergence`` / ``sinkhorn_flow`` compute with (``rp_sinkhorn_set_libm``):
    ``"contract"`` = the build's f32 functions (what the clustering uses), ``"glibc"`` = glibc's ``expf`` / ``logf`` restated
    (include/rp_libm_glibc.h): one solve as a Rust build on Linux computes it, bit for bit."""
    _lib.check(_lib.load().rp_sinkhorn_set_libm(LIBM[kind]))


def sinkhorn_divergence(mu, nu, tri, hp=None, device=0) -> np.ndarray:
    """``Sinkhorn::divergence`` (sinkhorn.rs:166-171) for P pairs: mu, nu are (P, bins) u32 counts."""
    mu = np.ascontiguousarray(np.atleast_2d(mu), dtype=np.uint32)
    nu = np.ascontiguousarray(np.atleast_2d(nu), dtype=np.uint32)
    tri = np.ascontiguousarray(tri, dtype=np.float32)
    hp = hp or default_sinkhorn()
    out = np.zeros(mu.shape[0], dtype=np.float32)
    _lib.check(_lib.load().rp_sinkhorn_divergence(mu.shape[1], mu.shape[0], _p(mu), _p(nu), _p(tri), C.byref(hp),
                                                  device, _p(out)))
    return out


def sinkhorn_cost(mu, nu, tri, hp=None, device=0):
    """``Coupling::minimize().cost()`` (sinkhorn.rs:194-218) for P pairs, and the iterations each solve used."""
    mu = np.ascontiguousarray(np.atleast_2d(mu), dtype=np.uint32)
    nu = np.ascontiguousarray(np.atleast_2d(nu), dtype=np.uint32)
    tri = np.ascontiguousarray(tri, dtype=np.float32)
    hp = hp or default_sinkhorn()
    out = np.zeros(mu.shape[0], dtype=np.float32)
    it = np.zeros(mu.shape[0], dtype=np.uint32)
    _lib.check(_lib.load().rp_sinkhorn_cost(mu.shape[1], mu.shape[0], _p(mu), _p(nu), _p(tri), C.byref(hp), device,
                                            _p(out), _p(it)))
    return out, it


def sinkhorn_flow(mu, nu, tri, hp=None, device=0):
    """``impl Coupling for Sinkhorn`` (sinkhorn.rs:194-218): minimise one pair, return (flow, coupling), each (bins, bins);
    the x-major left fold of ``flow`` is ``cost()``."""
    mu = np.ascontiguousarray(mu, dtype=np.uint32)
    nu = np.ascontiguousarray(nu, dtype=np.uint32)
    tri = np.ascontiguousarray(tri, dtype=np.float32)
    hp = hp or default_sinkhorn()
    bins = mu.size
    flow = np.zeros((bins, bins), dtype=np.float32)
    coupling = np.zeros((bins, bins), dtype=np.float32)
    _lib.check(_lib.load().rp_sinkhorn_flow(bins, _p(mu), _p(nu), _p(tri), C.byref(hp), device, _p(flow), _p(coupling)))
    return flow, coupling


def equity_variation(x, y, device=0) -> np.ndarray:
    """``Equity::variation`` (equity.rs:41-53) for P pairs of equal-width histograms."""
    x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.uint32)
    y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.uint32)
    out = np.zeros(x.shape[0], dtype=np.float32)
    _lib.check(_lib.load().rp_equity_variation(x.shape[1], x.shape[0], _p(x), _p(y), device, _p(out)))
    return out


def smoke(oracle) -> None:
    """Tiny k-means on device 0 checked bit for bit against the CPU oracle (called by __graft_entry__.smoke)."""
    from .fixtures import flop_like_points, smooth_metric

    bins, K, N = 32, 4, 96
    pts = flop_like_points(N, bins=bins, mass=20, seed=5)
    tri = smooth_metric(bins, 5)
    dev = Layer(K, pts, "sinkhorn", tri, seed=9)
    ora = oracle.OracleKmeans(K, pts, "sinkhorn", tri, seed=9)
    assert np.array_equal(dev.init_centroids(), ora.init_centroids()), "k-means++ picks differ"
    dev.init_bounds()
    ora.init_bounds()
    for _ in range(2):
        d1, s1, m1 = dev.step()
        d2, s2, m2 = ora.step()
        assert np.array_equal(d1.view(np.uint32), d2.view(np.uint32)) and np.array_equal(s1, s2) and m1 == m2
    b1, _ = dev.lookup()
    b2, _ = ora.assign()
    assert np.array_equal(b1, b2), "bucket assignments differ"
    print(f"smoke lloyd ok: K={K} N={N} bins={bins} distances={dev.stats()[0]}")
    dev.close()


def bench_slice(n_points: int = 16384, K: int = 256, bins: int = 256, iters: int = 2, seed: int = 0xF10F, libm: str = "contract"):
    """points/sec of Elkan iterations on a bounded slice of the flop-street configuration (BASELINE configs[2]):
    K = 256 centroids, 256-bin histograms of mass 47, Sinkhorn T=0.025 / <=128 iterations / tol 5e-4.

    Reports the Sinkhorn phase against the VALU roofline (it is exp-bound, not HBM-bound: SURVEY §8d) and the
    bound-update phase against the HBM roofline (2 * 4 * K bytes of lower bounds per point, read + written)."""
    from .fixtures import flop_like_points, smooth_metric

    pts = flop_like_points(n_points, bins=bins, mass=47, seed=seed)
    tri = smooth_metric(bins, 1)
    layer = Layer(K, pts, "sinkhorn", tri, seed=seed)
    if libm != "contract":
        layer.set_libm(libm)  # the lm_glibc pass of the kernels, unpruned (the rooflines below are the contract pass's)
    rng = np.random.default_rng(seed)
    layer.set_centroids(rng.choice(n_points, size=K, replace=False).astype(np.uint64))
    t0 = time.perf_counter()
    layer.init_bounds()
    t_bounds = time.perf_counter() - t0
    d0, i0 = layer.stats()
    e0 = layer.exp_evals()
    layer.profile(True)
    t0 = time.perf_counter()
    for _ in range(iters):
        layer.step()
    dt = time.perf_counter() - t0
    d1, i1 = layer.stats()
    e1 = layer.exp_evals()
    step_ms, step_n = layer.kernel_time("step")
    pw_ms, pw_n = layer.kernel_time("pairwise")
    bd_ms, bd_n = layer.kernel_time("bounds")
    dr_ms, dr_n = layer.kernel_time("drift")
    sc_ms, sc_n = layer.kernel_time("selfcost")
    layer.profile(False)
    sinkhorn_s = (step_ms + pw_ms + dr_ms + sc_ms) * 1e-3
    # VALU issue roofline from the chip's MEASURED sustained rates (profiles/r01_valu_issue_rates.txt, scripts/ubench):
    # 8.2e11 plain wave64 VALU instructions/s; a packed f32 instruction costs 2.12 plain ones.  One block of 8 softmin
    # terms is 106 VALU instructions in the shipped ISA, 44 of them packed (sub, clamp, exp polynomial, ldexp, floor,
    # sequential add) = 155.3 plain-equivalents, i.e. 2.70e12 terms/s with every lane busy.
    valu_peak_exps = 8.2e11 * 64 * 8 / (44 * 2.12 + 62)
    exps = e1 - e0
    bd_avg_s = bd_ms / max(bd_n, 1) * 1e-3
    out = {
        "metric": "kmeans_points_per_sec",
        "value": n_points * iters / dt,
        "unit": "points/s",
        "workload": f"flop-layer slice: N={n_points} of 1286792, K={K}, bins={bins}, mass 47, Sinkhorn EMD, "
                    f"{iters} Elkan iterations after init_bounds",
        "init_bounds_points_per_sec": n_points / t_bounds,
        "init_bounds_distances_per_sec": n_points * K / t_bounds,
        "distances": d1 - d0,
        "sinkhorn_iterations": i1 - i0,
        "exp_evals": exps,
        "roofline_sinkhorn": {"bound": "valu-exp", "achieved": exps / sinkhorn_s if sinkhorn_s > 0 else 0.0,
                              "peak": valu_peak_exps, "unit": "exp/s",
                              "frac": (exps / sinkhorn_s) / valu_peak_exps if sinkhorn_s > 0 else 0.0,
                              "note": "bit-reproducible software exp, 106 VALU instr (44 packed) per 8 softmin terms against the measured VALU issue rate; "
                                      "lanes are rows of one support (<= 47 of 64 busy on the point side)"},
        "roofline_bounds": {"bound": "hbm", "achieved": (n_points * K * 8) / bd_avg_s / 1e9 if bd_avg_s > 0 else 0.0,
                            "peak": 8000.0, "unit": "GB/s",
                            "frac": (n_points * K * 8) / bd_avg_s / 1e9 / 8000.0 if bd_avg_s > 0 else 0.0},
        "kernels_ms": {"step": step_ms / max(step_n, 1), "pairwise": pw_ms / max(pw_n, 1),
                       "bounds": bd_ms / max(bd_n, 1), "drift": dr_ms / max(dr_n, 1)},
    }
    layer.close()
    return out


def cpu_baseline_full(oracle, gpu_out, centroids, threads=None):
    """The CPU oracle with the reference's parallel structure (rayon par_iter over points and over centroid pairs ->
    OpenMP, oracle/rp_oracle_lloyd.c) on all host cores, on a BOUNDED sample of the full flop configuration, composed into
    the full-size Elkan iteration it stands for:

      rate  = point-centroid Sinkhorn solves per second (init_bounds of 768 sample points against 48 of the converged
              centroids the GPU run ended with);
      t_pw  = the pairwise pass of those 48 centroids (48 x 47 solves), scaled by (K (K - 1)) / (48 x 47) to K = 256;
      one full-size iteration = t_pw + (distances the GPU run's steady-state iteration evaluated) / rate.

    The algorithm and the distance counts are the GPU run's (the two are bit-identical), only the clock is the CPU's."""
    import os

    from .fixtures import flop_like_points, smooth_metric

    threads = threads or max(1, (os.cpu_count() or 2) // 2)
    K, Ks, Ns, bins = gpu_out["K"], 48, 768, gpu_out["bins"]
    pts = flop_like_points(Ns, bins=bins, mass=47, seed=0xF10F)
    oracle.lloyd_set_threads(threads)
    try:
        km = oracle.OracleKmeans(Ks, pts, "sinkhorn", smooth_metric(bins, 1), seed=1)
        km.set_centroids(np.arange(Ks, dtype=np.uint64))
        pick = np.linspace(0, K - 1, Ks).astype(int)
        for k, src in enumerate(pick):
            km.set_centroid(k, centroids[src])
        t0 = time.perf_counter()
        km.init_bounds()
        t_ib = time.perf_counter() - t0
        rate = Ns * Ks / t_ib
        d0, _ = oracle.lloyd_stats()
        t0 = time.perf_counter()
        km.step()
        t_step = time.perf_counter() - t0
        d1, _ = oracle.lloyd_stats()
    finally:
        oracle.lloyd_set_threads(1)
    pair_solves = Ks * (Ks - 1)
    t_pw_sample = max(t_step - max(d1 - d0 - pair_solves - Ks, 0) / rate, 0.0)
    t_pw = t_pw_sample * (K * (K - 1)) / pair_solves
    per_iter = gpu_out["per_iteration"]
    d_iter = float(np.median([it["distances"] for it in per_iter[len(per_iter) // 2:]])) - K * (K - 1) - K if per_iter else 0.0
    t_iter = t_pw + max(d_iter, 0.0) / rate
    return {"value": gpu_out["N"] / t_iter if t_iter > 0 else 0.0, "unit": "points/s", "cores": threads, "kind": "port",
            "estimate": True,  # composed from a measured solve rate and the GPU run's distance counts; no full-size CPU run stands behind it
            "distances_per_s": rate, "pairwise_s_at_K256": t_pw, "iteration_s_at_full_size": t_iter,
            "sample": f"oracle/rp_oracle_lloyd.c, OpenMP x{threads}: init_bounds of {Ns} points x {Ks} converged centroids "
                      f"({Ns * Ks} solves in {t_ib:.1f} s) and one Elkan step ({pair_solves} centroid-pair solves, {t_step:.1f} s); "
                      f"composed to a full-size iteration: pairwise x{K * (K - 1) / pair_solves:.1f} + {int(max(d_iter, 0))} "
                      f"point-centroid solves (the GPU run's steady-state count) at the measured rate"}


# rooflines of the lloyd kernels (MI355X_MICROARCH.md): HBM 8 TB/s; f32 MFMA 157.3 TFLOP/s = 64 flop/clk/SIMD; VALU: one
# wave64 instruction per 2 cycles per SIMD = 256 CU x 4 SIMD x 2.4 GHz / 2 = 1.23e12 wave-instructions/s
HBM_PEAK_GBPS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 2.0
SOFTMIN_INSTR_PER_8_TERMS = 106  # VALU instructions per 8 softmin terms per lane in the shipped ISA, contract arithmetic
VALU_SUSTAINED_PLAIN = 8.2e11    # plain (unpacked) wave64 VALU instructions/s the chip sustains (profiles/r01_valu_issue_rates.txt)
SOFTMIN_CONTRACT = (106, 457.0)  # (VALU instructions, SIMD-cycles of issue) per 8 terms: 155 plain-equivalent x 2.95 cycles
SOFTMIN_GLIBC = (175, 700.0)     # 88 f64 instructions (5.1 / 4.2 - 4.5 cycles) + 87 f32 / integer (profiles/r06_valu_issue_rates.txt)
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9


def bench_full(which: str = "flop", iters: int = 32, n_points: int | None = None, seed: int = 1, log=None, libm: str = "contract",
               rng: str = "counter", pts=None, keep: dict | None = None):
    """A full-size k-means configuration of BASELINE.json on one GPU:

    flop (configs[2]): N = 1 286 792 histograms, K = 256, bins = 256, mass 47, Sinkhorn EMD, k-means++,
                       init_bounds, `iters` Elkan iterations, final lookup;
    turn (configs[4], one GPU's 1/8 share): N = 1 745 006, K = 256, bins = 101, mass 46, Equity::variation.
    Returns per-phase wall times, rates and the rooflines of the three kernel families (a dict): the bit-faithful
    Sinkhorn solves against the VALU issue peak, the MFMA bound against the f32 MFMA peak, the Elkan bound update
    against the HBM peak — all from HIP events recorded by the library on its launch stream over the whole run.
    libm / rng: the layer's arithmetic ("glibc": the reference's own exp / ln) and k-means++ draw ("reference": Layer::init_centroids'
    SmallRng + WeightedIndex<f32>); pts: the points, when the caller already holds them; keep: receives the picks and the buckets."""
    from .fixtures import flop_like_points, smooth_metric, turn_like_points

    K = 256
    if which == "flop":
        N = n_points or 1286792
        bins, kind, tri, bytes_per_point = 256, "sinkhorn", smooth_metric(256, 1), 2320
        pts = flop_like_points(N, bins=bins, mass=47, seed=0xF10F) if pts is None else pts
    elif which == "turn":
        N = n_points or 1745006
        bins, kind, tri, bytes_per_point = 101, "variation", None, 2165
        pts = turn_like_points(N, bins=bins, mass=46, seed=5)
    else:
        raise ValueError(which)
    out = {"metric": "kmeans_points_per_sec", "unit": "points/s",
           "workload": f"{which}-street layer, FULL size: N={N} points, K={K}, bins={bins}, {kind} distance, k-means++, init_bounds, "
                       f"{iters} Elkan iterations, lookup (BASELINE configs[{2 if which == 'flop' else 4}]"
                       f"{'' if which == 'flop' else ', one GPU of eight'}); synthetic points (SURVEY 8d)",
           "N": N, "K": K, "bins": bins, "distance": kind, "iterations": iters}
    t0 = time.perf_counter()
    layer = Layer(K, pts, kind, tri, seed=seed)
    if libm != "contract":
        layer.set_libm(libm)  # the kernels' lm_glibc pass (filters kept; RP_FULL_NO_PRUNE=1: rp_kmeans_set_prune(h, 0))
        if os.environ.get("RP_FULL_NO_PRUNE"):
            layer.set_prune(False)
    if rng != "counter":
        layer.set_rng(rng, 1 if which == "flop" else 2)  # Street::Flop / Street::Turn: the seed of init_centroids' SmallRng (layer.rs:155-157)
    out["libm"], out["rng"] = libm, rng
    out["create_s"] = time.perf_counter() - t0  # upload + point masses + memoised OT(p,p)
    layer.profile(True)
    e0 = layer.exp_evals()
    t0 = time.perf_counter()
    chosen = layer.init_centroids()
    out["kmeanspp_s"] = time.perf_counter() - t0
    e_kpp = layer.exp_evals() - e0
    d0, _ = layer.stats()
    t0 = time.perf_counter()
    layer.init_bounds()
    out["init_bounds_s"] = time.perf_counter() - t0
    d1, _ = layer.stats()
    out["init_bounds_exact_distances"] = d1 - d0
    per_iter = []
    t_all = time.perf_counter()
    for it in range(iters):
        da, _ = layer.stats()
        t0 = time.perf_counter()
        _, _, moved = layer.step()
        dt = time.perf_counter() - t0
        db, _ = layer.stats()
        per_iter.append({"s": round(dt, 4), "distances": db - da, "moved": round(float(moved), 5)})
        if log:
            log(f"iter {it}: {dt:.3f}s distances={db - da} moved={moved:.4f}")
    total = time.perf_counter() - t_all
    out["elkan_total_s"] = total
    out["value"] = out["points_per_s"] = N * iters / total if iters else 0.0
    out["algorithmic_GBps"] = N * iters * bytes_per_point / total / 1e9 if iters else 0.0
    out["hbm_frac"] = out["algorithmic_GBps"] / HBM_PEAK_GBPS
    out["per_iteration"] = per_iter
    t0 = time.perf_counter()
    buckets, _ = layer.lookup()
    out["lookup_s"] = time.perf_counter() - t0
    if keep is not None:
        keep["picks"], keep["buckets"] = np.asarray(chosen).copy(), np.asarray(buckets).copy()
    out["end_to_end_s"] = out["create_s"] + out["kmeanspp_s"] + out["init_bounds_s"] + total + out["lookup_s"]
    out["rms"] = layer.rms()
    d2, i2 = layer.stats()
    ex = layer.stats_ex()
    out["distances_total"] = d2
    out["distances_remembered"] = ex["remembered"]  # evaluated by the reference at that point, reused here (same centroid content)
    out["sinkhorn_iterations_total"] = i2
    ms = {name: layer.kernel_time(name) for name in ("pairwise", "step", "recompute", "bounds", "neighbor", "selfcost", "kpp",
                                                     "drift", "mfma_bound", "kpp_bound", "refresh_bound")}
    out["kernels_ms"] = {k: {"total_ms": round(v[0], 3), "launches": v[1]} for k, v in ms.items()}
    bd_ms, bd_n = ms["bounds"]
    if bd_n:
        gbps = N * K * 8 / (bd_ms / bd_n * 1e-3) / 1e9
        out["roofline_bounds"] = {"bound": "hbm", "kernel": "k_bounds_update", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": gbps / HBM_PEAK_GBPS, "bytes_per_launch": N * K * 8, "avg_launch_ms": bd_ms / bd_n,
                                  "note": f"lower bounds f32[N][K] read + written once per Elkan iteration: {N * K * 8 / 1e9:.2f} GB per launch"
                                          " (past the 256 MiB Infinity Cache at full size)"}
    if kind == "variation":
        # the turn kernels (k_elkan_step_var / k_neighbor_var) keep 64 centroid CDFs per wavefront in registers and compute whole
        # (point, 64 centroids) tiles — 2 VALU instructions per bin per tile — then replay Elkan's rule on the tile: what the rule
        # EVALUATES is `distances_total` (the reference's count, tests/test_gpu_lloyd.py::test_elkan_iterations_bit_exact), what was
        # computed beyond it is counted here
        st_ms = ms["step"][0] + ms["neighbor"][0]
        tiles = (ex["computed_beyond"] + N * K * 2) / 64.0  # init_bounds and lookup evaluate all K for every point
        winstr = tiles * bins * 2.0
        out["distances_computed_beyond_rule"] = ex["computed_beyond"]
        out["roofline_variation"] = {"bound": "valu", "kernel": "k_elkan_step_var + k_neighbor_var", "achieved": winstr / (st_ms * 1e-3) if st_ms else 0.0,
                                     "peak": VALU_PEAK_WAVE_INSTR, "unit": "wave-instructions/s",
                                     "frac": winstr / (st_ms * 1e-3) / VALU_PEAK_WAVE_INSTR if st_ms else 0.0,
                                     "note": "(point, 64-centroid) tiles x bins x 2 VALU instructions (t = cx - CY, s += |t|: equity.rs:41-53) over "
                                             "the event-timed step and neighbor kernels; the CDF build and the rule's replay are not counted"}
    if kind == "sinkhorn":
        exps = layer.exp_evals() - e0
        valu_s = (ms["pairwise"][0] + ms["step"][0] + ms["neighbor"][0] + ms["selfcost"][0] + ms["kpp"][0] + ms["drift"][0]) * 1e-3
        # per 8 softmin terms per lane, from the shipped ISA (scripts/isa_summary.py): VALU instructions, and the
        # SIMD-cycles they take to issue at the measured per-instruction costs (profiles/r06_valu_issue_rates.txt):
        #   contract  106 instructions, 44 of them packed f32 (two issue slots each): 155 plain x 2.95 cycles = 457
        #   glibc     175 instructions, 88 of them f64 (5.1 cycles for fma / mul, 4.2 - 4.5 for cvt / rndne): 700
        per8, cyc8 = SOFTMIN_GLIBC if libm == "glibc" else SOFTMIN_CONTRACT
        instr = exps / 64.0 / 8.0 * per8  # wave64 VALU instructions if every lane carried a term
        cyc = exps / 64.0 / 8.0 * cyc8
        out["roofline_sinkhorn"] = {"bound": "valu", "kernel": "wave_sinkhorn_cost (softmin)", "arithmetic": libm,
                                    "achieved": instr / valu_s if valu_s else 0.0,
                                    "peak": VALU_PEAK_WAVE_INSTR, "unit": "wave-instructions/s",
                                    "frac": instr / valu_s / VALU_PEAK_WAVE_INSTR if valu_s else 0.0, "exp_terms": exps,
                                    "instructions_per_8_terms": per8,
                                    "note": "bit-reproducible exp: VALU instructions per 8 softmin terms per lane from the shipped ISA; achieved = "
                                            "terms / 64 / 8 x that / kernel time, i.e. counts only lanes that carry a term (a point fills "
                                            "<= 47 of 64 lanes); peak = one wave64 instruction per 2 cycles per SIMD (nominal)",
                                    # against what the pipe can actually issue: the measured cost of each instruction kind
                                    "issue": {"simd_cycles_per_8_terms": cyc8, "achieved": cyc / valu_s if valu_s else 0.0,
                                              "peak": SIMD_CYCLES_PER_S, "unit": "SIMD-cycles/s of VALU issue",
                                              "frac": cyc / valu_s / SIMD_CYCLES_PER_S if valu_s else 0.0,
                                              "counters": ("profiles/r06_lloyd_glibc_valu_ceiling.json" if libm == "glibc" else
                                                           "profiles/r05_lloyd_valu_ceiling.json") +
                                                          ": SQ_ACTIVE_INST_VALU per wave-cycle x wavefronts in flight; the gap to this figure "
                                                          "is lanes that carry no term"}}
        kpp_ms, kpp_n = ms["kpp"]
        if kpp_ms > 0:  # k-means++ on its own: K rounds of (column-marginal filter, Sinkhorn solves of the survivors against ONE new centroid)
            ki = e_kpp / 64.0 / 8.0 * per8
            out["roofline_kmeanspp"] = {"bound": "valu", "kernel": "k_kpp_filter + k_kpp_update* (softmin)", "achieved": ki / (kpp_ms * 1e-3),
                                        "peak": VALU_PEAK_WAVE_INSTR, "unit": "wave-instructions/s",
                                        "frac": ki / (kpp_ms * 1e-3) / VALU_PEAK_WAVE_INSTR, "exp_terms": e_kpp, "launches": kpp_n,
                                        "kernel_s": kpp_ms * 1e-3, "wall_s": out["kmeanspp_s"],
                                        "note": "the same accounting as roofline_sinkhorn, for the k-means++ rounds alone: softmin terms of "
                                                "the solves the column-marginal bound let through, against the kernels' event time (filter "
                                                "launches included); K sequential rounds, each against one new sparse centroid"}
        st = layer.prune_stats()
        kb_ms, kb_n = ms["kpp_bound"]
        if kb_ms > 0:  # the k-means++ interval filter (csrc/kpp_bound.hpp): FMA-bound scaling-domain iterations, registers only
            out["kmeanspp_interval_filter"] = {"pairs": st["kpp_bound_pairs"], "kept": st["kpp_bound_kept"],
                                               "pair_iterations": st["kpp_bound_iterations"], "cost_passes": st["kpp_bound_cost_passes"],
                                               "kernel_s": kb_ms * 1e-3, "launches": kb_n,
                                               "pair_iterations_per_s": st["kpp_bound_iterations"] / (kb_ms * 1e-3),
                                               "note": "the (new centroid, point) pairs the column-marginal bound let through, examined by a "
                                                       "scaling-domain interval; `kept` went on to the bit-faithful solve"}
        if st.get("ref_pick_chunks"):  # the reference-seed draw: WeightedIndex<f32>'s sequential sums (csrc/kpp_refpick.hpp)
            out["reference_seed_draw"] = {"chunks": st["ref_pick_chunks"], "walked_term_by_term": st["ref_pick_walked"],
                                          "note": "256-term chunks over the K draws; a chunk inside one binade without a tie is one exact "
                                                  "addition, the others are walked as the reference does"}
        rs = layer.refresh_stats()
        if rs["enabled"]:  # the interval-decided refresh of the Elkan iterations (csrc/refresh_bound.hpp)
            rb_ms = ms["refresh_bound"][0]
            out["refresh_interval"] = {**rs, "kernel_s": rb_ms * 1e-3,
                                       "pair_iterations_per_s": rs["pair_iterations"] / (rb_ms * 1e-3) if rb_ms else 0.0,
                                       "note": "stale-bound refreshes examined by a scaling-domain interval; `settled` needed no bit-faithful "
                                               "solve; `exactify_solves` replaced an interval by its exact value before the next filter"}
        out["mfma_bound"] = st
        mb_ms, mb_n = ms["mfma_bound"]
        if st["enabled"] and mb_ms > 0:
            tf = st["mfma_instructions"] * 2048 / (mb_ms * 1e-3) / 1e12
            # useful = the share of the issued column slots that held a column still iterating (16 slots per wavefront iteration)
            live = st.get("column_iterations", 0) / max(16 * st.get("block_iterations", 0), 1)
            out["roofline_mfma"] = {"bound": "mfma", "kernel": "k_sinkhorn_bound", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS,
                                    "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS, "dtype": "f32 (v_mfma_f32_16x16x4_f32)",
                                    "live_slot_frac": live, "achieved_useful": tf * live, "frac_useful": tf * live / MFMA_F32_PEAK_TFLOPS,
                                    "cost_pass_share": st.get("cost_passes", 0) / max(2 * st.get("block_iterations", 0) + st.get("cost_passes", 0), 1),
                                    "survivors_per_point": st["survivors"] / max(st["points"], 1),
                                    "note": "scaling-domain Sinkhorn bound in front of init_bounds and lookup; flops = MFMA "
                                            "instructions issued x 2048; *_useful counts the column slots that held a live column; "
                                            "cost_pass_share = the K.*C contractions' share of the issued MFMAs"}
        out["_centroids"] = layer.centroids()[0]  # for the CPU baseline's pairwise sample; callers pop it before printing
    layer.close()
    return out


def cpu_baseline_slice(oracle, seconds: float = 8.0, K: int = 256, bins: int = 256, seed: int = 0xF10F):
    """The CPU oracle (oracle/rp_oracle_lloyd.c, 1 thread) on a bounded sample of the flop-layer workload:
    init_bounds-style full distances point -> centroid, as many points as fit in `seconds`."""
    from .fixtures import flop_like_points, smooth_metric

    pts = flop_like_points(K + 64, bins=bins, mass=47, seed=seed)
    tri = smooth_metric(bins, 1)
    cents, probe = pts[:K].astype(np.uint32), pts[K:].astype(np.uint32)
    done = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        i, k = divmod(done, K)
        if i >= probe.shape[0]:
            break
        oracle.sinkhorn_cost(cents[k], probe[i], tri, bins=bins)  # OT(centroid, point); self terms are memoised (sinkhorn.rs:166-171)
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "distances/s", "cores": 1, "kind": "port",
            "sample": f"oracle/rp_oracle_lloyd.c Sinkhorn solve (self terms memoised as in the reference), {done} point-centroid distances of the flop-layer "
                      f"init_bounds (K={K}, bins={bins}, mass 47) in {dt:.1f} s on 1 host thread"}

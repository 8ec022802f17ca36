"""Independent float64 restatements of the reference's numeric definitions, written from the Rust sources (not from
oracle/*.c) in a different style (vectorised numpy, float64): the C oracle must agree with them to float32 accuracy.
They pin the oracle from a second side; the oracle stays the bit-exact checker of the HIP path.

  sinkhorn_*      crates/lloyd/src/sinkhorn.rs:77-139,166-171,194-218, phi.rs:34-39, bins.rs:58-60,84-88
  variation       crates/lloyd/src/equity.rs:41-53
  regret_matching crates/mccfr/src/strategy/profile.rs:31-51, flow.rs:18-59
"""
from __future__ import annotations

import numpy as np

MIN_POSITIVE = float(np.finfo(np.float32).tiny)


def dense_cost(tri, bins: int) -> np.ndarray:
    """Metric::raw_distance (metric.rs:41-55): 0 on the diagonal, tri[hi(hi-1)/2 + lo] elsewhere (pair.rs:58-65)."""
    C = np.zeros((bins, bins), dtype=np.float64)
    hi, lo = np.tril_indices(bins, -1)
    C[hi, lo] = C[lo, hi] = np.asarray(tri, dtype=np.float64)[hi * (hi - 1) // 2 + lo]
    return C


def sinkhorn_f64(mu, nu, C, temperature=0.025, iterations=128, tolerance=5e-4, trace=False):
    """Sinkhorn::from(mu, nu, metric).minimize().cost() in float64.

    lhs/rhs start uniform, ln(1/|support|) (phi.rs:34-39); one iteration = lhs update from the old rhs, then the rhs
    update from the NEW lhs (sinkhorn.rs:80-87); stop when the L1 change of exp(potential) on both sides sums below
    the tolerance (sinkhorn.rs:134-139).  Returns (cost, iterations[, errs, costs])."""
    mu = np.asarray(mu, dtype=np.float64)
    nu = np.asarray(nu, dtype=np.float64)
    sx, sy = np.flatnonzero(mu), np.flatnonzero(nu)
    if len(sx) == 0 or len(sy) == 0:
        return (0.0, 0, np.zeros(0), np.zeros(0)) if trace else (0.0, 0)
    la, lb = np.log(mu[sx] / mu.sum()), np.log(nu[sy] / nu.sum())
    R = C[np.ix_(sx, sy)] / temperature
    f = np.full(len(sx), np.log(1.0 / len(sx)))
    g = np.full(len(sy), np.log(1.0 / len(sy)))
    errs, costs, stop = [], [], None

    def cost_of(f, g):
        return float((np.exp(f[:, None] + g[None, :] - R) * C[np.ix_(sx, sy)]).sum())

    for t in range(iterations):
        nf = la - np.log(np.maximum(np.exp(g[None, :] - R), MIN_POSITIVE).sum(1))
        ef = np.abs(np.exp(nf) - np.exp(f)).sum()
        f = nf
        ng = lb - np.log(np.maximum(np.exp(f[:, None] - R), MIN_POSITIVE).sum(0))
        eg = np.abs(np.exp(ng) - np.exp(g)).sum()
        g = ng
        if trace:
            errs.append(ef + eg)
            costs.append(cost_of(f, g))
            if stop is None and ef + eg < tolerance:
                stop = (costs[-1], t + 1)
            continue
        if ef + eg < tolerance:
            return cost_of(f, g), t + 1
    if trace:
        c, it = stop if stop else (costs[-1], iterations)
        return c, it, np.array(errs), np.array(costs)
    return cost_of(f, g), iterations


def sinkhorn_divergence_f64(mu, nu, C, **hp):
    """Sinkhorn::divergence (sinkhorn.rs:166-171): max(0, OT(mu,nu) - OT(mu,mu)/2 - OT(nu,nu)/2)."""
    xy = sinkhorn_f64(mu, nu, C, **hp)[0]
    xx = sinkhorn_f64(mu, mu, C, **hp)[0]
    yy = sinkhorn_f64(nu, nu, C, **hp)[0]
    return max(xy - 0.5 * xx - 0.5 * yy, 0.0)


def variation_f64(x, y) -> float:
    """Equity::variation (equity.rs:41-53): mean absolute difference of the two CDFs over the buckets."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    return float(np.abs(np.cumsum(x / x.sum()) - np.cumsum(y / y.sum())).sum() / len(x))


def variation_exact(x, y):
    """the same in exact rational arithmetic (fractions): the closed form the float results must round to"""
    from fractions import Fraction

    wx, wy = int(np.sum(x)), int(np.sum(y))
    cx = cy = Fraction(0)
    tot = Fraction(0)
    for a, b in zip(x, y):
        cx += Fraction(int(a), wx)
        cy += Fraction(int(b), wy)
        tot += abs(cx - cy)
    return tot / len(x)


def regret_matching_f64(regrets) -> np.ndarray:
    """RefProf::iterated_distribution (profile.rs:47-51): max(R, eps) / sum max(R, eps)."""
    r = np.maximum(np.asarray(regrets, dtype=np.float64), MIN_POSITIVE)
    return r / r.sum()


def averaged_f64(weights) -> np.ndarray:
    """RefProf::averaged_distribution (profile.rs:40-44)."""
    w = np.maximum(np.asarray(weights, dtype=np.float64), MIN_POSITIVE)
    return w / w.sum()


def sampling_f64(weights, temperature=1.0, smoothing=2.0, curiosity=0.05) -> np.ndarray:
    """sampling_distribution (flow.rs:24-59): q(a) = max(curiosity, (W(a)/tau + beta) / (sum W + beta)), normalised."""
    w = np.maximum(np.asarray(weights, dtype=np.float64), MIN_POSITIVE)
    q = np.maximum((w / temperature + smoothing) / (w.sum() + smoothing), curiosity)
    return q / q.sum()

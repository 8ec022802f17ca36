"""Numerical study behind the MFMA Sinkhorn bound: how closely does a float32 scaling-domain
iteration u = a / (K v), v = b / (K^T u) with K = exp(-C/T) track the reference's log-domain Gauss-Seidel solve
(oracle/rp_oracle_lloyd.c), how wide is the interval once the stopping iteration is uncertain, and how many of the
256 centroids survive the prune per point.  CPU only (numpy + the oracle); not part of the product."""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from lloyd_fixtures import flop_like_points, smooth_metric, tri_index  # noqa: E402

T = np.float32(0.025)
TOL = np.float32(5e-4)
ITERS = 128


def dense_metric(tri, bins):
    C = np.zeros((bins, bins), dtype=np.float32)
    for j in range(1, bins):
        for i in range(j):
            C[i, j] = C[j, i] = tri[tri_index(i, j)]
    return C


def w1_centroids(pts, K, rounds=6, seed=0):
    """cheap stand-in for converged centroids: Lloyd under the 1-d CDF distance (the metric is a 1-d embedding)."""
    rng = np.random.default_rng(seed)
    P = pts.astype(np.float64)
    P /= P.sum(1, keepdims=True)
    cdf = np.cumsum(P, 1)
    idx = rng.choice(len(pts), K, replace=False)
    cen = cdf[idx]
    for _ in range(rounds):
        d = np.abs(cdf[:, None, :] - cen[None, :, :]).sum(2)
        a = d.argmin(1)
        for k in range(K):
            if (a == k).any():
                cen[k] = cdf[a == k].mean(0)
    sums = np.zeros((K, pts.shape[1]), dtype=np.uint32)
    for k in range(K):
        sums[k] = pts[a == k].sum(0) if (a == k).any() else pts[idx[k]]
    return sums


def scaling_trace(mu, nu, Kmat, Cmat):
    """float32 scaling-domain trace: err_t and cost_t for t = 1..ITERS (cost after the v update of iteration t)."""
    sx, sy = np.nonzero(mu)[0], np.nonzero(nu)[0]
    a = (mu[sx] / np.float32(mu.sum())).astype(np.float32)
    b = (nu[sy] / np.float32(nu.sum())).astype(np.float32)
    Ks = Kmat[np.ix_(sx, sy)]
    KC = (Ks * Cmat[np.ix_(sx, sy)]).astype(np.float32)
    u = np.full(len(sx), np.float32(1.0) / np.float32(len(sx)), dtype=np.float32)
    v = np.full(len(sy), np.float32(1.0) / np.float32(len(sy)), dtype=np.float32)
    errs, costs = [], []
    for _ in range(ITERS):
        un = (a / (Ks @ v)).astype(np.float32)
        eu = np.abs(un - u).sum(dtype=np.float32)
        u = un
        vn = (b / (Ks.T @ u)).astype(np.float32)
        ev = np.abs(vn - v).sum(dtype=np.float32)
        v = vn
        errs.append(eu + ev)
        costs.append(np.float32(u @ (KC @ v)))
    return np.array(errs), np.array(costs)


def main():
    bins, K = 256, 256
    n_pts = int(os.environ.get("PROTO_POINTS", "24"))
    pts = flop_like_points(4096, bins=bins, mass=47, seed=0xF10F)
    tri = smooth_metric(bins, 1)
    Cm = dense_metric(tri, bins)
    Km = np.exp(-(Cm / T)).astype(np.float32)
    cents = w1_centroids(pts, K)
    print("centroid support sizes: mean", (cents > 0).sum(1).mean(), "max", (cents > 0).sum(1).max())
    hp = oracle.default_sinkhorn()
    rng = np.random.default_rng(1)
    sample = rng.choice(len(pts), n_pts, replace=False)
    rho = float(os.environ.get("PROTO_RHO", "0.02"))
    dc = float(os.environ.get("PROTO_DC", "2e-6"))
    worst_dev, widths, surv, viol = 0.0, [], [], 0
    t0 = time.time()
    for pi in sample:
        p = pts[pi].astype(np.uint32)
        sp = oracle.sinkhorn_cost(p, p, tri, hp, bins)[0]
        lo, hi, ex = np.zeros(K), np.zeros(K), np.zeros(K)
        for k in range(K):
            c = cents[k]
            exact, it = oracle.sinkhorn_cost(c, p, tri, hp, bins)
            sc = oracle.sinkhorn_cost(c, c, tri, hp, bins)[0]
            errs, costs = scaling_trace(c.astype(np.float32), p.astype(np.float32), Km, Cm)
            below_hi = np.nonzero(errs < TOL * (1 + rho))[0]
            below_lo = np.nonzero(errs < TOL * (1 - rho))[0]
            t_first = below_hi[0] if len(below_hi) else ITERS - 1
            t_last = below_lo[0] if len(below_lo) else ITERS - 1
            win = costs[t_first:t_last + 1]
            c_lo, c_hi = win.min() - dc, win.max() + dc
            if not (c_lo <= exact <= c_hi):
                viol += 1
                print("VIOLATION", pi, k, exact, c_lo, c_hi, it, t_first + 1, t_last + 1)
            worst_dev = max(worst_dev, abs(float(costs[it - 1]) - float(exact)))
            widths.append(c_hi - c_lo)
            f32 = np.float32
            ex[k] = max(f32(f32(exact - f32(0.5) * sc) - f32(0.5) * sp), 0)
            lo[k] = max(f32(f32(f32(c_lo) - f32(0.5) * sc) - f32(0.5) * sp), 0)
            hi[k] = max(f32(f32(f32(c_hi) - f32(0.5) * sc) - f32(0.5) * sp), 0)
        ub = hi.min()
        s = int((lo <= ub).sum())
        surv.append(s)
        srt = np.sort(ex)
        print(f"point {pi}: nnz={int((p > 0).sum())} best={srt[0]:.5f} second={srt[1]:.5f} survivors={s} "
              f"({time.time() - t0:.0f}s)", flush=True)
    print(f"worst |scaling - exact| at the exact's own iteration: {worst_dev:.3e}")
    print(f"interval width: mean {np.mean(widths):.3e} max {np.max(widths):.3e}")
    print(f"survivors per point: mean {np.mean(surv):.2f} max {np.max(surv)}; violations {viol}")


if __name__ == "__main__":
    main()

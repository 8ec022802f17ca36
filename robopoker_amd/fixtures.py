"""Seeded synthetic inputs for the lloyd path (shared by the oracle tests, the GPU parity tests, smoke() and bench.py).

flop_metric / flop_hist restate the reference's closed-form Sinkhorn fixture
(crates/lloyd/src/sinkhorn.rs:240-262); the point generators follow SURVEY.md §8d's synthetic-input table.
"""
from __future__ import annotations

import numpy as np


def tri_index(i: int, j: int) -> int:
    lo, hi = (i, j) if i < j else (j, i)
    return 0 if hi == 0 else hi * (hi - 1) // 2 + lo


def flop_metric(bins: int = 32) -> np.ndarray:
    """d(i,j) = (((7i + 13j) % 97) + 1) / 100 for i < j (sinkhorn.rs:252-262)."""
    tri = np.zeros(bins * (bins - 1) // 2, dtype=np.float32)
    for i in range(bins):
        for j in range(i + 1, bins):
            tri[tri_index(i, j)] = np.float32(((i * 7 + j * 13) % 97) + 1) / np.float32(100.0)
    return tri


def flop_hist(entries, bins: int = 32) -> np.ndarray:
    h = np.zeros(bins, dtype=np.uint32)
    for idx, count in entries:
        h[idx] = count
    return h


def random_metric(bins: int, rng) -> np.ndarray:
    """symmetric random ground cost in (0, 1], normalised by its max (Metric::from, metric.rs:127-141)."""
    tri = rng.random(bins * (bins - 1) // 2, dtype=np.float32) + np.float32(1e-3)
    return (tri / tri.max()).astype(np.float32)


def smooth_metric(bins: int, seed: int = 0) -> np.ndarray:
    """|e_i - e_j| over a random 1-d embedding of the bins: a true metric, like EMD between clusters."""
    rng = np.random.default_rng(seed)
    e = np.sort(rng.random(bins)).astype(np.float32)
    tri = np.zeros(bins * (bins - 1) // 2, dtype=np.float32)
    for j in range(1, bins):
        for i in range(j):
            tri[tri_index(i, j)] = abs(e[j] - e[i])
    return (tri / tri.max()).astype(np.float32)


def turn_like_points(n: int, bins: int = 101, mass: int = 46, seed: int = 0, spread: float = 0.08) -> np.ndarray:
    """n histograms of `mass` draws from a unimodal profile over `bins` buckets (u8 counts).

    Mimics Histogram::from(Observation::from(Street::Turn)): 46 river cards binned by equity."""
    rng = np.random.default_rng(seed)
    centers = rng.random(n)
    widths = spread * (0.5 + rng.random(n))
    draws = centers[:, None] + widths[:, None] * rng.standard_normal((n, mass))
    idx = np.clip(np.rint(draws * (bins - 1)), 0, bins - 1).astype(np.int64)
    out = np.zeros((n, bins), dtype=np.uint8)
    rows = np.repeat(np.arange(n), mass)
    np.add.at(out, (rows, idx.ravel()), 1)
    return out


def flop_like_points(n: int, bins: int = 256, mass: int = 47, seed: int = 0xF10F) -> np.ndarray:
    """n histograms of `mass` draws over a neighbourhood of s ~ U{8..47} of the `bins` turn buckets.

    RP_FIXTURE_CACHE=<dir>: the array is kept there as .npy (the full flop layer takes 30 s of host time to draw; scripts that
    start several processes on it in one GPU call set this)."""
    import os

    cache = os.environ.get("RP_FIXTURE_CACHE")
    path = os.path.join(cache, f"flop_like_{n}_{bins}_{mass}_{seed}.npy") if cache else None
    if path and os.path.exists(path):
        return np.load(path)
    rng = np.random.default_rng(seed)
    out = np.zeros((n, bins), dtype=np.uint8)
    centers = rng.integers(0, bins, size=n)
    sizes = rng.integers(8, min(48, bins) + 1, size=n) if bins >= 48 else rng.integers(2, bins + 1, size=n)
    for i in range(n):
        s = int(sizes[i])
        support = (centers[i] + rng.choice(min(bins, 3 * s), size=min(s, bins), replace=False)) % bins
        draws = rng.choice(support, size=mass)
        np.add.at(out[i], draws, 1)
    if path:
        tmp = path + f".{os.getpid()}.tmp.npy"
        np.save(tmp, out)
        os.replace(tmp, path)
    return out

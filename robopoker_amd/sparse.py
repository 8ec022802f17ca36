"""Row-addressed regret/strategy table in HBM (include/rp_mi355x.h, rp_profile_*): the update half of
``Solver::step`` (crates/mccfr/src/solver/solver.rs:96-105,143-192) at NLHE scale.

Batches live in device memory; this module only carries raw pointers across the C-ABI.  ``DeviceBatch`` uploads numpy
arrays through torch (device memory plumbing only) for tests and the benchmark.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch  # imported BEFORE the first HIP call of librp_mi355x.so: one HIP runtime per process (torch bundles its own)

from . import _lib
from .mccfr import default_hyper


class DeviceBatch:
    """Decisions of one step, uploaded to the GPU.  row u32[n], n_actions u8[n], expanded u16[n],
    regret/policy f32[n][A], payoff f32[n] — in application (tree-id) order."""

    def __init__(self, row, n_actions, expanded, regret, policy, payoff, device=0):
        dev = torch.device("cuda", device)
        self.n = int(len(row))
        self._t = [
            torch.from_numpy(np.ascontiguousarray(row, dtype=np.uint32).view(np.int32)).to(dev),
            torch.from_numpy(np.ascontiguousarray(n_actions, dtype=np.uint8)).to(dev),
            torch.from_numpy(np.ascontiguousarray(expanded, dtype=np.uint16).view(np.int16)).to(dev),
            torch.from_numpy(np.ascontiguousarray(regret, dtype=np.float32)).to(dev),
            torch.from_numpy(np.ascontiguousarray(policy, dtype=np.float32)).to(dev),
            torch.from_numpy(np.ascontiguousarray(payoff, dtype=np.float32)).to(dev),
        ]
        torch.cuda.synchronize(dev)
        self.c = _lib.Decisions(self.n, *[t.data_ptr() if t.numel() else None for t in self._t])


class SparseProfile:
    def __init__(self, n_rows: int, max_actions: int, regret="linear", weight="linear", hyper=None, default_regret=None,
                 max_batch=0, device=0):
        self._lib = _lib.load()
        self.n_rows, self.A, self.device = int(n_rows), int(max_actions), device
        hp = hyper or default_hyper()
        dr = None
        if default_regret is not None:
            dr = np.ascontiguousarray(default_regret, dtype=np.float32)
            assert dr.size == self.A
        h = C.c_void_p()
        _lib.check(self._lib.rp_profile_create(device, self.n_rows, self.A, _lib.REGRET[regret], _lib.WEIGHT[weight],
                                               C.byref(hp), dr.ctypes.data if dr is not None else None, max_batch,
                                               C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rp_profile_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def apply(self, batch: DeviceBatch, mode="ordered"):
        _lib.check(self._lib.rp_profile_apply(self._h, C.byref(batch.c), _lib.UPDATE[mode]))

    def sync(self):
        _lib.check(self._lib.rp_profile_sync(self._h))

    def epoch(self) -> int:
        e = C.c_uint64()
        _lib.check(self._lib.rp_profile_epoch(self._h, C.byref(e)))
        return e.value

    def set_epoch(self, e: int):
        _lib.check(self._lib.rp_profile_set_epoch(self._h, e))

    def rows(self, rows) -> np.ndarray:
        """Encounters of the given rows: structured array [len(rows)][A] of (weight, regret, payoff, visits)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        out = np.zeros((rows.size, self.A), dtype=[("weight", "<f4"), ("regret", "<f4"), ("payoff", "<f4"), ("visits", "<u4")])
        _lib.check(self._lib.rp_profile_get_rows(self._h, rows.size, rows.ctypes.data, out.ctypes.data))
        return out

    def set_stream(self, ptr):
        _lib.check(self._lib.rp_profile_set_stream(self._h, ptr))

    def entry_bytes(self) -> int:
        n = C.c_size_t()
        _lib.check(self._lib.rp_profile_entry_bytes(self._h, C.byref(n)))
        return n.value

    def summarize(self, batch: DeviceBatch, entries_dev_ptr: int) -> int:
        n = C.c_uint32()
        _lib.check(self._lib.rp_profile_summarize(self._h, C.byref(batch.c), entries_dev_ptr, C.byref(n)))
        return n.value

    def fold(self, entries_dev_ptr: int, n_entries: int):
        _lib.check(self._lib.rp_profile_fold(self._h, entries_dev_ptr, n_entries))

    def profile(self, enable=True):
        _lib.check(self._lib.rp_profile_profile(self._h, int(enable)))

    def kernel_time(self, name: str):
        ms, n = C.c_double(), C.c_uint64()
        _lib.check(self._lib.rp_profile_kernel_time(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def synthetic_batch(n: int, n_rows: int, max_actions: int = 9, zipf: float = 1.1, seed: int = 0):
    """Decisions with Zipf(zipf) row popularity over n_rows rows, |choices| in 2..max_actions
    fixed per row, regret deltas ~ N(0, 50^2), a random policy vector, payoff ~ N(0, 100).  Returns numpy arrays."""
    rng = np.random.default_rng(seed)
    # inverse-CDF sampling of a continuous power law, clipped to the table: rank ~ u^(-1/(zipf-1)) is heavy tailed;
    # a bounded Zipf over n_rows ranks via the generalized harmonic approximation
    u = rng.random(n)
    s = zipf
    hmax = (n_rows ** (1.0 - s) - 1.0) / (1.0 - s)
    rank = np.floor(((u * hmax) * (1.0 - s) + 1.0) ** (1.0 / (1.0 - s))).astype(np.int64)
    rank = np.clip(rank, 1, n_rows) - 1
    row = ((rank * 2654435761) % n_rows).astype(np.uint32)  # scatter the popular ranks over the table
    nact = (2 + (row.astype(np.uint64) * 0x9E3779B97F4A7C15 >> np.uint64(40)) % np.uint64(max_actions - 1)).astype(np.uint8)
    regret = (rng.standard_normal((n, max_actions)) * 50.0).astype(np.float32)
    policy = rng.random((n, max_actions)).astype(np.float32)
    lanes = np.arange(max_actions)[None, :] < nact[:, None]
    policy = np.where(lanes, policy, 0.0).astype(np.float32)
    policy /= policy.sum(axis=1, keepdims=True)
    regret = np.where(lanes, regret, 0.0).astype(np.float32)
    full = ((1 << nact.astype(np.uint32)) - 1).astype(np.uint16)
    pruned = rng.random(n) < 0.1  # a tenth of the touches carry a pruned regret vector
    drop = (1 << (rng.integers(0, 16, n) % nact)).astype(np.uint16)
    expanded = np.where(pruned, full & ~drop, full).astype(np.uint16)
    payoff = (rng.standard_normal(n) * 100.0).astype(np.float32)
    return row, nact, expanded, regret, policy.astype(np.float32), payoff

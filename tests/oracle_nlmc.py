"""ctypes wrapper of the CPU oracle's NLHE external-sampling MCCFR (oracle/rp_oracle_nlmc.c).  TEST INFRASTRUCTURE ONLY.

Mirrors ``mccfr!(Nlhe, NlheEncoder, ..., 128)`` (crates/nlhe/src/solver.rs:11): ``step`` = Solver::step, ``batch`` = the
Decisions of the current epoch, ``export`` = the profile's rows keyed by NlheInfo (past, present, choices)."""
from __future__ import annotations

import ctypes as C

import numpy as np

import oracle
from robopoker_amd import _lib

A = 9
ENC = np.dtype([("weight", "<f4"), ("regret", "<f4"), ("payoff", "<f4"), ("visits", "<u4")])
_o = None


def lib():
    global _o
    if _o is None:
        o = oracle.load()
        vp = C.c_void_p
        o.ora_nlmc_create.restype = vp
        o.ora_nlmc_create.argtypes = [C.c_uint32, C.c_int, C.c_int, C.POINTER(_lib.Hyper), C.c_uint64, C.c_uint32]
        o.ora_nlmc_destroy.argtypes = [vp]
        o.ora_nlmc_set_sampling.argtypes = [vp, C.c_int]
        o.ora_nlmc_set_rng.argtypes = [vp, C.c_int]
        o.ora_nlmc_set_table.argtypes = [vp, C.c_int, vp, vp, C.c_uint64]
        o.ora_nlmc_step.argtypes = [vp]
        o.ora_nlmc_batch.restype = C.c_uint64
        o.ora_nlmc_batch.argtypes = [vp] + [C.POINTER(vp)] * 7
        o.ora_nlmc_epoch.restype = C.c_uint64
        o.ora_nlmc_epoch.argtypes = [vp]
        o.ora_nlmc_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        o.ora_nlmc_last_tree_nodes.restype = C.c_uint32
        o.ora_nlmc_last_tree_nodes.argtypes = [vp]
        o.ora_nlmc_export.restype = C.c_uint64
        o.ora_nlmc_export.argtypes = [vp, C.c_uint64, vp, vp, vp, vp]
        o.ora_nlmc_import.argtypes = [vp, C.c_uint64, vp, vp, vp, vp]
        o.ora_nlmc_set_shard.argtypes = [vp, C.c_uint32, C.c_uint32]
        o.ora_nlmc_entry_bytes.restype = C.c_size_t
        o.ora_nlmc_entry_bytes.argtypes = [vp]
        o.ora_nlmc_step_local.restype = C.c_int64
        o.ora_nlmc_step_local.argtypes = [vp, vp, vp, vp, vp]
        o.ora_nlmc_step_apply.argtypes = [vp, vp, vp, vp, vp, C.c_uint64]
        o.ora_nlmc_step_world.restype = C.c_int
        o.ora_nlmc_step_world.argtypes = [vp, C.c_uint32]
        o.ora_nlmc_row_key.restype = C.c_int
        o.ora_nlmc_row_key.argtypes = [vp, C.c_uint32, vp, vp, vp]
        o.ora_nlmc_hash_bucket.restype = C.c_uint32
        o.ora_nlmc_hash_bucket.argtypes = [C.c_int, C.c_int64]
        _o = o
    return _o


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleNlhe:
    def __init__(self, cap_log2=16, regret="linear", weight="linear", batch=128, seed=0, hyper=None, sampling="external"):
        self._o = lib()
        self.hp = hyper or oracle.default_hyper()
        self.batch_size = batch
        self._h = C.c_void_p(self._o.ora_nlmc_create(cap_log2, _lib.REGRET[regret], _lib.WEIGHT[weight], C.byref(self.hp), seed, batch))
        self._o.ora_nlmc_set_sampling(self._h, _lib.SAMPLING[sampling])

    def set_rng(self, kind: str):
        self._o.ora_nlmc_set_rng(self._h, _lib.RNG[kind])

    def __del__(self):
        if getattr(self, "_h", None):
            self._o.ora_nlmc_destroy(self._h)
            self._h = None

    def step(self):
        self._o.ora_nlmc_step(self._h)

    # ---- the sharded surface of the C-ABI (rp_nlhe_set_shard / step_local / step_apply), host pointers ----
    def set_shard(self, rank: int, world: int):
        self._o.ora_nlmc_set_shard(self._h, rank, world)

    def entry_bytes(self):
        return self._o.ora_nlmc_entry_bytes(self._h), self.batch_size * 600

    def step_local(self, entries_ptr, past_ptr, present_ptr, choices_ptr) -> int:
        n = self._o.ora_nlmc_step_local(self._h, C.c_void_p(entries_ptr), C.c_void_p(past_ptr), C.c_void_p(present_ptr), C.c_void_p(choices_ptr))
        if n < 0:
            raise RuntimeError("composed update unsupported for this schedule")
        return n

    def step_apply(self, entries_ptr, past_ptr, present_ptr, choices_ptr, n: int):
        self._o.ora_nlmc_step_apply(self._h, C.c_void_p(entries_ptr), C.c_void_p(past_ptr), C.c_void_p(present_ptr), C.c_void_p(choices_ptr), n)

    def step_world(self, world: int):
        if self._o.ora_nlmc_step_world(self._h, world) != 0:
            raise RuntimeError("composed update unsupported for this schedule")

    def batch(self):
        ptr = [C.c_void_p() for _ in range(7)]
        n = self._o.ora_nlmc_batch(self._h, *[C.byref(p) for p in ptr])

        def arr(p, dt, count):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(dt)), shape=(count,)).copy() if count else np.zeros(0, dtype=dt)

        return dict(n=n, row=arr(ptr[0], C.c_uint32, n), n_actions=arr(ptr[1], C.c_uint8, n), expanded=arr(ptr[2], C.c_uint16, n),
                    regret=arr(ptr[3], C.c_float, n * A).reshape(n, A), policy=arr(ptr[4], C.c_float, n * A).reshape(n, A),
                    payoff=arr(ptr[5], C.c_float, n), tree=arr(ptr[6], C.c_uint64, n))

    @property
    def epoch(self) -> int:
        return self._o.ora_nlmc_epoch(self._h)

    def counters(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._o.ora_nlmc_counters(self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def export(self):
        """dict (past, present, choices) -> Encounter rows (A,) of every infoset in the table"""
        _, _, n = self.counters()
        past = np.zeros(n, dtype=np.uint64)
        present = np.zeros(n, dtype=np.uint32)
        choices = np.zeros(n, dtype=np.uint64)
        enc = np.zeros((n, A), dtype=ENC)
        got = self._o.ora_nlmc_export(self._h, n, _p(past), _p(present), _p(choices), _p(enc))
        assert got == n
        return past, present, choices, enc

    def load(self, past, present, choices, enc):
        past, present, choices = (np.ascontiguousarray(past, dtype=np.uint64), np.ascontiguousarray(present, dtype=np.uint32),
                                  np.ascontiguousarray(choices, dtype=np.uint64))
        enc = np.ascontiguousarray(enc, dtype=ENC)
        self._o.ora_nlmc_import(self._h, past.size, _p(past), _p(present), _p(choices), _p(enc))


def as_map(past, present, choices, enc):
    return {(int(p), int(b), int(c)): enc[i] for i, (p, b, c) in enumerate(zip(past, present, choices))}

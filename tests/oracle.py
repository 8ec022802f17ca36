"""ctypes wrapper of the CPU oracle (oracle/_build/librp_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np

from robopoker_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "_build", "librp_oracle.so")
ORACLE_BUILD = "-O3 (portable)"
MAXA = 16


class Decision(C.Structure):
    _fields_ = [
        ("info", C.c_uint32),
        ("n_actions", C.c_uint32),
        ("expanded", C.c_uint32),
        ("regret", C.c_float * MAXA),
        ("policy", C.c_float * MAXA),
        ("payoff", C.c_float),
        ("tree", C.c_uint64),
    ]


_ora = None


def load() -> C.CDLL:
    global _ora
    if _ora is not None:
        return _ora
    if not os.path.exists(ORACLE_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    path = ORACLE_PATH
    if os.environ.get("RP_ORACLE_NATIVE"):
        # bench.py's cpu_baseline legs: the same sources compiled -O3 -march=native ON THIS MACHINE; the parity tests
        # use the portable build, which travels between machines.  No compiler here = the portable build, said in the sample text
        native = os.path.join(ROOT, "oracle", "_build", "librp_oracle_native.so")
        try:
            # several rank processes may get here together: one of them builds (under a lock, into a file of its own that is renamed
            # into place: a process that has the previous file mapped keeps it), the others find its stamp — this machine's boot id
            # and the sources' newest mtime — and load what it built.  A library that travelled here from another machine has
            # another boot id in its stamp and is rebuilt (-march=native is this machine's).
            import fcntl
            import glob

            os.makedirs(os.path.dirname(native), exist_ok=True)
            try:
                boot = open("/proc/sys/kernel/random/boot_id").read().strip()
            except OSError:
                boot = "unknown"
            srcs = glob.glob(os.path.join(ROOT, "oracle", "*.[ch]")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
            want = f"{boot} {max(os.path.getmtime(f) for f in srcs):.3f}"
            stamp = native + ".stamp"
            mine = f"_build/librp_oracle_native.{os.getpid()}.so"
            with open(native + ".lock", "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                have = open(stamp).read() if os.path.exists(stamp) and os.path.exists(native) else ""
                if have != want or boot == "unknown":
                    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "native", f"NATIVE={mine}"],
                                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                    os.replace(os.path.join(ROOT, "oracle", mine), native)
                    with open(stamp, "w") as f:
                        f.write(want)
            path = native
        except (OSError, subprocess.CalledProcessError):
            pass
    global ORACLE_BUILD
    ORACLE_BUILD = "-O3 -march=native" if path != ORACLE_PATH else "-O3 (portable)"
    o = C.CDLL(path)
    vp = C.c_void_p
    o.ora_mccfr_create.restype = vp
    o.ora_mccfr_create.argtypes = [C.POINTER(_lib.GameTable), C.c_int, C.c_int, C.c_int, C.c_uint32,
                                   C.POINTER(_lib.Hyper), C.c_uint64]
    o.ora_mccfr_destroy.argtypes = [vp]
    o.ora_mccfr_step.argtypes = [vp]
    o.ora_mccfr_solve.argtypes = [vp, C.c_uint64]
    o.ora_mccfr_step_mt.argtypes = [vp, C.c_uint32]
    o.ora_lloyd_set_threads.argtypes = [C.c_int]
    o.ora_lloyd_max_threads.restype = C.c_int
    o.ora_mccfr_batch.restype = C.c_uint64
    o.ora_mccfr_batch.argtypes = [vp, C.POINTER(C.POINTER(Decision))]
    o.ora_mccfr_step_world.restype = C.c_int
    o.ora_mccfr_step_world.argtypes = [vp, C.c_uint32]
    o.ora_mccfr_summary_bytes.restype = C.c_size_t
    o.ora_mccfr_summary_bytes.argtypes = [vp]
    o.ora_mccfr_step_local.restype = C.c_int
    o.ora_mccfr_step_local.argtypes = [vp, C.c_uint32, vp]
    o.ora_mccfr_step_apply.argtypes = [vp, vp, C.c_uint32]
    o.ora_mccfr_window_local.restype = C.c_int
    o.ora_mccfr_window_local.argtypes = [vp, C.c_uint32, vp, C.c_int]
    o.ora_mccfr_window_apply.argtypes = [vp, vp, C.c_uint32]
    o.ora_mccfr_window_world.restype = C.c_int
    o.ora_mccfr_window_world.argtypes = [vp, C.c_uint32, C.c_uint32]
    o.ora_mccfr_epoch.restype = C.c_uint64
    o.ora_mccfr_epoch.argtypes = [vp]
    o.ora_mccfr_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    o.ora_mccfr_set_batch.argtypes = [vp, C.c_uint32]
    o.ora_mccfr_export.argtypes = [vp, C.POINTER(_lib.Encounter)]
    o.ora_mccfr_import.argtypes = [vp, C.POINTER(_lib.Encounter), C.c_uint64]
    o.ora_mccfr_policy.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(C.c_float)]
    o.ora_mccfr_sum_regret.restype = C.c_float
    o.ora_mccfr_sum_regret.argtypes = [vp]
    o.ora_mccfr_exploitability.restype = C.c_float
    o.ora_mccfr_exploitability.argtypes = [vp]
    # lloyd
    o.ora_sinkhorn_cost.restype = C.c_float
    o.ora_sinkhorn_cost.argtypes = [C.c_uint32, vp, vp, vp, C.POINTER(_lib.SinkhornHP), C.POINTER(C.c_uint32)]
    o.ora_sinkhorn_trace.restype = C.c_float
    o.ora_sinkhorn_trace.argtypes = [C.c_uint32, vp, vp, vp, C.POINTER(_lib.SinkhornHP), C.POINTER(C.c_uint32), vp, vp]
    o.ora_sinkhorn_flow.argtypes = [C.c_uint32, vp, vp, vp, C.POINTER(_lib.SinkhornHP), vp, vp]
    o.ora_sinkhorn_divergence.restype = C.c_float
    o.ora_sinkhorn_divergence.argtypes = [C.c_uint32, vp, vp, vp, C.POINTER(_lib.SinkhornHP)]
    o.ora_equity_variation.restype = C.c_float
    o.ora_equity_variation.argtypes = [C.c_uint32, vp, vp]
    o.ora_kmeans_create.restype = vp
    o.ora_kmeans_create.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, vp, C.c_int, vp,
                                    C.POINTER(_lib.SinkhornHP), C.c_uint64]
    o.ora_kmeans_destroy.argtypes = [vp]
    o.ora_kmeans_set_centroids.argtypes = [vp, vp]
    o.ora_kmeans_init_centroids.argtypes = [vp, vp]
    o.ora_kmeans_init_bounds.argtypes = [vp]
    o.ora_kmeans_step.argtypes = [vp, vp, vp, C.POINTER(C.c_double)]
    o.ora_kmeans_step_naive.argtypes = [vp]
    o.ora_kmeans_assign.argtypes = [vp, vp, vp]
    o.ora_kmeans_bounds.argtypes = [vp, vp, vp, vp]
    o.ora_kmeans_centroids.argtypes = [vp, vp, vp]
    o.ora_kmeans_metric.argtypes = [vp, vp]
    o.ora_kmeans_rms.restype = C.c_float
    o.ora_kmeans_rms.argtypes = [vp]
    o.ora_kmeans_kpp_begin.argtypes = [vp]
    o.ora_kmeans_kpp_total.restype = C.c_uint64
    o.ora_kmeans_kpp_total.argtypes = [vp]
    o.ora_kmeans_kpp_pick.restype = C.c_uint64
    o.ora_kmeans_kpp_pick.argtypes = [vp, C.c_uint64]
    o.ora_kmeans_kpp_update.argtypes = [vp, C.c_uint32]
    o.ora_kmeans_get_point.argtypes = [vp, C.c_uint64, vp]
    o.ora_kmeans_set_centroid.argtypes = [vp, C.c_uint32, vp]
    o.ora_kmeans_partial_bytes.restype = C.c_size_t
    o.ora_kmeans_partial_bytes.argtypes = [vp]
    o.ora_kmeans_step_local.argtypes = [vp, vp]
    o.ora_kmeans_step_finish.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_double)]
    o.ora_lloyd_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
    _ora = o
    return o


def default_hyper() -> _lib.Hyper:
    hp = _lib.Hyper()
    _lib.load().rp_hyper_default(C.byref(hp))
    return hp


def default_sinkhorn() -> _lib.SinkhornHP:
    hp = _lib.SinkhornHP()
    _lib.load().rp_sinkhorn_hp_default(C.byref(hp))
    return hp


class OracleSolver:
    """CPU oracle with the same surface as robopoker_amd.mccfr.Solver."""

    def __init__(self, game, regret="floored", weight="linear", sampling="external", batch=1, seed=0, hyper=None):
        self.game = game
        self.hp = hyper if hyper is not None else default_hyper()
        self._o = load()
        self._h = self._o.ora_mccfr_create(C.byref(game.table), _lib.REGRET[regret], _lib.WEIGHT[weight],
                                           _lib.SAMPLING[sampling], batch, C.byref(self.hp), seed)
        assert self._h, "oracle create failed"
        self.cells = game.table.n_infos * game.table.max_actions

    def __del__(self):
        if getattr(self, "_h", None):
            self._o.ora_mccfr_destroy(self._h)
            self._h = None

    def step(self):
        self._o.ora_mccfr_step(self._h)

    def solve(self, trees: int):
        self._o.ora_mccfr_solve(self._h, trees)
        return self

    def step_mt(self, threads: int):
        """Solver::step with the reference's tree-parallel batch() on `threads` host threads (CPU baseline)"""
        self._o.ora_mccfr_step_mt(self._h, threads)

    # ---- the sharded surface of the C-ABI (rp_mccfr_set_shard / step_local / step_apply), host pointers ----
    def set_rng(self, kind: str, streams=None):
        """rp_rng_kind: "counter" or "reference" (include/rp_refrng.h); streams default to the built-in game's"""
        o = load()
        o.ora_mccfr_set_rng.argtypes = [C.c_void_p, C.c_int, C.POINTER(_lib.HashStreams)]
        if kind == "reference" and streams is None:
            streams = self.game.hash_streams()
        rc = o.ora_mccfr_set_rng(self._h, _lib.RNG[kind], C.byref(streams) if streams is not None else None)
        assert rc == 0, "ora_mccfr_set_rng refused the streams"

    def set_shard(self, rank: int, world: int):
        self._rank, self._world = rank, world

    def set_stream(self, ptr):
        pass

    def summary_bytes(self) -> int:
        return self._o.ora_mccfr_summary_bytes(self._h)

    def step_local(self, ptr: int):
        if self._o.ora_mccfr_step_local(self._h, getattr(self, "_rank", 0), C.c_void_p(ptr)) != 0:
            raise RuntimeError("composed update unsupported for this schedule")

    def step_apply(self, ptr: int, world: int):
        self._o.ora_mccfr_step_apply(self._h, C.c_void_p(ptr), world)

    def step_world(self, world: int):
        rc = self._o.ora_mccfr_step_world(self._h, world)
        if rc != 0:
            raise RuntimeError("composed update unsupported for this schedule")

    def window_local(self, ptr: int, first: bool):
        if self._o.ora_mccfr_window_local(self._h, getattr(self, "_rank", 0), C.c_void_p(ptr), 1 if first else 0) != 0:
            raise RuntimeError("composed update unsupported for this schedule")

    def window_apply(self, ptr: int, world: int):
        self._o.ora_mccfr_window_apply(self._h, C.c_void_p(ptr), world)

    def window_world(self, world: int, window: int):
        if self._o.ora_mccfr_window_world(self._h, world, window) != 0:
            raise RuntimeError("composed update unsupported for this schedule")

    def batch(self):
        p = C.POINTER(Decision)()
        n = self._o.ora_mccfr_batch(self._h, C.byref(p))
        out = []
        for i in range(n):
            d = p[i]
            out.append(dict(info=d.info, n=d.n_actions, expanded=d.expanded, regret=list(d.regret)[: d.n_actions],
                            policy=list(d.policy)[: d.n_actions], payoff=d.payoff, tree=d.tree))
        return out

    @property
    def epoch(self) -> int:
        return self._o.ora_mccfr_epoch(self._h)

    def counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._o.ora_mccfr_counters(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def set_batch(self, b: int):
        self._o.ora_mccfr_set_batch(self._h, b)

    def export(self) -> np.ndarray:
        rows = (_lib.Encounter * self.cells)()
        self._o.ora_mccfr_export(self._h, rows)
        return np.frombuffer(rows, dtype=ENC_DTYPE).copy()

    def load_rows(self, rows: np.ndarray, epoch: int):
        buf = np.ascontiguousarray(rows, dtype=ENC_DTYPE)
        self._o.ora_mccfr_import(self._h, buf.ctypes.data_as(C.POINTER(_lib.Encounter)), epoch)

    def policy(self, info: int, kind="averaged") -> np.ndarray:
        out = (C.c_float * MAXA)()
        self._o.ora_mccfr_policy(self._h, info, _lib.DIST[kind], out)
        return np.array(out[: self.game.n_actions(info)], dtype=np.float32)

    def sum_regret(self) -> float:
        return self._o.ora_mccfr_sum_regret(self._h)

    def exploitability(self) -> float:
        return self._o.ora_mccfr_exploitability(self._h)


ENC_DTYPE = np.dtype([("weight", "<f4"), ("regret", "<f4"), ("payoff", "<f4"), ("visits", "<u4")])


class OracleProfile:
    """ora_profile_*: the sparse-profile oracle (oracle/rp_oracle_mccfr.c)."""

    def __init__(self, n_rows, max_actions, regret="linear", weight="linear", hyper=None, default_regret=None):
        o = load()
        vp = C.c_void_p
        o.ora_profile_create.restype = vp
        o.ora_profile_create.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.POINTER(_lib.Hyper), vp]
        o.ora_profile_destroy.argtypes = [vp]
        o.ora_profile_apply.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp, vp]
        o.ora_profile_get.argtypes = [vp, C.c_uint32, vp]
        o.ora_profile_set_epoch.argtypes = [vp, C.c_uint64]
        o.ora_profile_entry_bytes.restype = C.c_size_t
        o.ora_profile_entry_bytes.argtypes = [vp]
        o.ora_profile_summarize.restype = C.c_int64
        o.ora_profile_summarize.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp, vp, vp]
        o.ora_profile_fold.argtypes = [vp, vp, C.c_uint64]
        self.o, self.A, self.n_rows = o, max_actions, n_rows
        hp = hyper or default_hyper()
        dr = None if default_regret is None else np.ascontiguousarray(default_regret, dtype=np.float32)
        self.h = o.ora_profile_create(n_rows, max_actions, _lib.REGRET[regret], _lib.WEIGHT[weight], C.byref(hp),
                                      dr.ctypes.data if dr is not None else None)

    def __del__(self):
        if getattr(self, "h", None):
            self.o.ora_profile_destroy(self.h)
            self.h = None

    @staticmethod
    def _arrays(batch):
        row, nact, expanded, regret, policy, payoff = batch
        return (np.ascontiguousarray(row, dtype=np.uint32), np.ascontiguousarray(nact, dtype=np.uint8),
                np.ascontiguousarray(expanded, dtype=np.uint16), np.ascontiguousarray(regret, dtype=np.float32),
                np.ascontiguousarray(policy, dtype=np.float32), np.ascontiguousarray(payoff, dtype=np.float32))

    def apply(self, batch):
        a = self._arrays(batch)
        self.o.ora_profile_apply(self.h, len(a[0]), *[x.ctypes.data for x in a])

    def summarize(self, batch) -> np.ndarray:
        a = self._arrays(batch)
        eb = self.o.ora_profile_entry_bytes(self.h)
        blob = np.zeros(max(len(a[0]), 1) * eb, dtype=np.uint8)
        n = self.o.ora_profile_summarize(self.h, len(a[0]), *[x.ctypes.data for x in a], blob.ctypes.data)
        if n < 0:
            raise ValueError("composed update unsupported for this schedule")
        return blob[: n * eb].copy()

    def fold(self, blob: np.ndarray):
        eb = self.o.ora_profile_entry_bytes(self.h)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self.o.ora_profile_fold(self.h, blob.ctypes.data, blob.size // eb)

    def entry_bytes(self) -> int:
        return self.o.ora_profile_entry_bytes(self.h)

    def set_epoch(self, e: int):
        self.o.ora_profile_set_epoch(self.h, e)

    def epoch(self) -> int:
        return self.o.ora_mccfr_epoch(self.h)

    def rows(self, rows) -> np.ndarray:
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        out = np.zeros((rows.size, self.A), dtype=[("weight", "<f4"), ("regret", "<f4"), ("payoff", "<f4"), ("visits", "<u4")])
        for i, r in enumerate(rows):
            self.o.ora_profile_get(self.h, int(r), out[i].ctypes.data)
        return out


class OracleProfileEngine(OracleProfile):
    """OracleProfile behind the pointer-based sharded surface of the C-ABI (summarize / fold on raw host pointers)."""

    def set_stream(self, ptr):
        pass

    def summarize(self, batch, ptr: int) -> int:
        blob = OracleProfile.summarize(self, batch)
        C.memmove(ptr, blob.ctypes.data, blob.size)
        return blob.size // self.entry_bytes()

    def fold(self, ptr: int, n: int):
        blob = np.ctypeslib.as_array((C.c_uint8 * max(n * self.entry_bytes(), 1)).from_address(ptr))[: n * self.entry_bytes()]
        OracleProfile.fold(self, blob.copy())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def sinkhorn_cost(mu, nu, tri, hp=None, bins=None):
    hp = hp or default_sinkhorn()
    mu = np.ascontiguousarray(mu, dtype=np.uint32)
    nu = np.ascontiguousarray(nu, dtype=np.uint32)
    tri = np.ascontiguousarray(tri, dtype=np.float32)
    it = C.c_uint32()
    c = load().ora_sinkhorn_cost(bins or mu.size, _p(mu), _p(nu), _p(tri), C.byref(hp), C.byref(it))
    return c, it.value


def sinkhorn_trace(mu, nu, tri, hp=None, bins=None):
    """cost, iterations, and per iteration: the stopping statistic and the cost had the solve stopped there."""
    hp = hp or default_sinkhorn()
    mu = np.ascontiguousarray(mu, dtype=np.uint32)
    nu = np.ascontiguousarray(nu, dtype=np.uint32)
    tri = np.ascontiguousarray(tri, dtype=np.float32)
    it = C.c_uint32()
    errs = np.zeros(hp.iterations, dtype=np.float32)
    costs = np.zeros(hp.iterations, dtype=np.float32)
    c = load().ora_sinkhorn_trace(bins or mu.size, _p(mu), _p(nu), _p(tri), C.byref(hp), C.byref(it), _p(errs), _p(costs))
    return c, it.value, errs, costs


def sinkhorn_flow(mu, nu, tri, hp=None):
    hp = hp or default_sinkhorn()
    mu = np.ascontiguousarray(mu, dtype=np.uint32)
    nu = np.ascontiguousarray(nu, dtype=np.uint32)
    tri = np.ascontiguousarray(tri, dtype=np.float32)
    flow = np.zeros((mu.size, mu.size), dtype=np.float32)
    coupling = np.zeros((mu.size, mu.size), dtype=np.float32)
    load().ora_sinkhorn_flow(mu.size, _p(mu), _p(nu), _p(tri), C.byref(hp), _p(flow), _p(coupling))
    return flow, coupling


def sinkhorn_divergence(mu, nu, tri, hp=None, bins=None):
    hp = hp or default_sinkhorn()
    mu = np.ascontiguousarray(mu, dtype=np.uint32)
    nu = np.ascontiguousarray(nu, dtype=np.uint32)
    tri = np.ascontiguousarray(tri, dtype=np.float32)
    return load().ora_sinkhorn_divergence(bins or mu.size, _p(mu), _p(nu), _p(tri), C.byref(hp))


def equity_variation(x, y):
    x = np.ascontiguousarray(x, dtype=np.uint32)
    y = np.ascontiguousarray(y, dtype=np.uint32)
    return load().ora_equity_variation(x.size, _p(x), _p(y))


class OracleKmeans:
    """CPU oracle with the same surface as robopoker_amd.lloyd.Layer."""

    def __init__(self, K, counts, kind="sinkhorn", tri=None, hp=None, seed=0):
        counts = np.ascontiguousarray(counts, dtype=np.uint8)
        self.N, self.bins = counts.shape
        self.K = K
        self.hp = hp or default_sinkhorn()
        self._o = load()
        tri_arr = np.ascontiguousarray(tri, dtype=np.float32) if tri is not None else None
        self._h = self._o.ora_kmeans_create(K, self.N, self.bins, _p(counts), _lib.METRIC[kind],
                                            _p(tri_arr) if tri_arr is not None else None, C.byref(self.hp), seed)
        assert self._h, "oracle kmeans create failed"

    def __del__(self):
        if getattr(self, "_h", None):
            self._o.ora_kmeans_destroy(self._h)
            self._h = None

    def set_rng(self, kind: str, street: int = 1):
        o = load()
        o.ora_kmeans_set_rng.argtypes = [C.c_void_p, C.c_int, C.c_int]
        o.ora_kmeans_set_rng(self._h, _lib.RNG[kind], street)

    def init_centroids(self):
        chosen = np.zeros(self.K, dtype=np.uint64)
        self._o.ora_kmeans_init_centroids(self._h, _p(chosen))
        return chosen

    def set_centroids(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        self._o.ora_kmeans_set_centroids(self._h, _p(idx))

    def init_bounds(self):
        self._o.ora_kmeans_init_bounds(self._h)

    # ---- the sharded surface of the C-ABI (rp_kmeans_kpp_* / step_local / step_finish), host pointers ----
    def set_stream(self, ptr):
        pass

    def kpp_begin(self):
        self._o.ora_kmeans_kpp_begin(self._h)

    def kpp_total(self) -> int:
        return self._o.ora_kmeans_kpp_total(self._h)

    def kpp_pick(self, r: int) -> int:
        return self._o.ora_kmeans_kpp_pick(self._h, r)

    def kpp_update(self, k: int):
        self._o.ora_kmeans_kpp_update(self._h, k)

    def get_point(self, idx: int) -> np.ndarray:
        out = np.zeros(self.bins, dtype=np.uint32)
        self._o.ora_kmeans_get_point(self._h, idx, _p(out))
        return out

    def set_centroid(self, k: int, counts):
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        self._o.ora_kmeans_set_centroid(self._h, k, _p(counts))

    def partial_bytes(self) -> int:
        return self._o.ora_kmeans_partial_bytes(self._h)

    def step_local(self, ptr: int):
        self._o.ora_kmeans_step_local(self._h, C.c_void_p(ptr))

    def step_finish(self, ptr: int):
        drift = np.zeros(self.K, dtype=np.float32)
        sizes = np.zeros(self.K, dtype=np.uint64)
        re = C.c_double()
        self._o.ora_kmeans_step_finish(self._h, C.c_void_p(ptr), _p(drift), _p(sizes), C.byref(re))
        return drift, sizes, re.value

    def step(self):
        drift = np.zeros(self.K, dtype=np.float32)
        sizes = np.zeros(self.K, dtype=np.uint64)
        re = C.c_double()
        self._o.ora_kmeans_step(self._h, _p(drift), _p(sizes), C.byref(re))
        return drift, sizes, re.value

    def step_naive(self):
        self._o.ora_kmeans_step_naive(self._h)

    def assign(self):
        b = np.zeros(self.N, dtype=np.uint8)
        d = np.zeros(self.N, dtype=np.float32)
        self._o.ora_kmeans_assign(self._h, _p(b), _p(d))
        return b, d

    def bounds(self):
        j = np.zeros(self.N, dtype=np.uint8)
        u = np.zeros(self.N, dtype=np.float32)
        lo = np.zeros((self.N, self.K), dtype=np.float32)
        self._o.ora_kmeans_bounds(self._h, _p(j), _p(u), _p(lo))
        return j, u, lo

    def centroids(self):
        c = np.zeros((self.K, self.bins), dtype=np.uint32)
        w = np.zeros(self.K, dtype=np.uint64)
        self._o.ora_kmeans_centroids(self._h, _p(c), _p(w))
        return c, w

    def metric(self):
        t = np.zeros(self.K * (self.K - 1) // 2, dtype=np.float32)
        self._o.ora_kmeans_metric(self._h, _p(t))
        return t

    def rms(self) -> float:
        return self._o.ora_kmeans_rms(self._h)


def lloyd_set_threads(n: int):
    """threads of the oracle's point-parallel k-means loops (CPU baseline; results do not depend on it)"""
    load().ora_lloyd_set_threads(n)


def max_threads() -> int:
    return load().ora_lloyd_max_threads()


def lloyd_stats(reset=False):
    a, b = C.c_uint64(), C.c_uint64()
    load().ora_lloyd_stats(C.byref(a), C.byref(b), 1 if reset else 0)
    return a.value, b.value

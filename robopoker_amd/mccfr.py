"""Host mirror of the reference's ``mccfr::Solver`` surface over the MI355X C-ABI.

Names and meaning follow crates/mccfr/src/solver/solver.rs (``step``, ``solve``, ``spend``,
``exploitability``) and strategy/profile.rs (``t`` -> ``epoch``, ``cum_*`` -> ``get``,
``iterated_distribution`` / ``averaged_distribution`` -> ``policy``).  All compute happens in
librp_mi355x.so's HIP kernels; there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .games import Game

ENC_DTYPE = np.dtype([("weight", "<f4"), ("regret", "<f4"), ("payoff", "<f4"), ("visits", "<u4")])


def default_hyper() -> _lib.Hyper:
    hp = _lib.Hyper()
    _lib.load().rp_hyper_default(C.byref(hp))
    return hp


class Solver:
    """``mccfr!(..)<R, W, S>`` (strategy/macros.rs:7-151) on one MI355X.

    Parameters mirror the reference's type parameters: ``regret`` = RegretSchedule, ``weight`` =
    WeightSchedule, ``sampling`` = SamplingScheme, ``batch`` = ``Solver::batch_size()``.
    """

    def __init__(self, game: Game, regret="floored", weight="linear", sampling="external", batch=1, seed=0,
                 hyper=None, device=0, mode=None):
        self._lib = _lib.load()
        self.game = game
        self.hp = hyper if hyper is not None else default_hyper()
        self._h = C.c_void_p()
        if mode is None:  # the reference's exact order (rp_mccfr_create's default)
            _lib.check(self._lib.rp_mccfr_create(C.byref(game.table), _lib.REGRET[regret], _lib.WEIGHT[weight],
                                                 _lib.SAMPLING[sampling], batch, C.byref(self.hp), seed, device,
                                                 C.byref(self._h)))
        else:
            _lib.check(self._lib.rp_mccfr_create_mode(C.byref(game.table), _lib.REGRET[regret], _lib.WEIGHT[weight],
                                                      _lib.SAMPLING[sampling], batch, C.byref(self.hp), seed, device,
                                                      _lib.UPDATE[mode], C.byref(self._h)))
        self.batch = batch
        self.cells = game.table.n_infos * game.table.max_actions

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rp_mccfr_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- Solver ---------------------------------------------------------------------------------
    def step(self):
        _lib.check(self._lib.rp_mccfr_step(self._h))

    def step_async(self, steps=1):
        _lib.check(self._lib.rp_mccfr_step_async(self._h, steps))

    def sync(self):
        _lib.check(self._lib.rp_mccfr_sync(self._h))

    def solve(self, trees: int):
        _lib.check(self._lib.rp_mccfr_solve(self._h, trees))
        return self

    def spend(self, seconds: float):
        it, el = C.c_uint64(), C.c_double()
        _lib.check(self._lib.rp_mccfr_spend(self._h, seconds, C.byref(it), C.byref(el)))
        return it.value, el.value

    def train(self, max_steps=0, max_seconds=0.0, log_interval=60.0, flush_interval=1800.0, on_checkpoint=None,
              on_flush=None, interrupt=None):
        """``Trainer::train`` (crates/forge/src/trainer.rs:18-66): loop { step; checkpoint; flush; interrupt? } in the
        library.  ``on_checkpoint(dict, line)`` gets the Checkpoint and its display line, ``on_flush(dict)`` fires at
        the flush cadence, ``interrupt`` is a ctypes c_int the caller may set to stop.  Returns Progress::summary."""
        EVENT = C.CFUNCTYPE(None, C.c_int, C.c_void_p, C.c_char_p, C.c_void_p)

        class Checkpoint(C.Structure):
            _fields_ = [("epoch", C.c_uint64), ("nodes", C.c_uint64), ("infos", C.c_uint64), ("rate", C.c_double)]

        def cb(event, cp, line, _user):
            c = C.cast(cp, C.POINTER(Checkpoint)).contents
            d = {"epoch": c.epoch, "nodes": c.nodes, "infos": c.infos, "rate": c.rate}
            if event == 0 and on_checkpoint:
                on_checkpoint(d, line.decode())
            if event == 1 and on_flush:
                on_flush(d)

        fn = EVENT(cb)
        buf = C.create_string_buffer(256)
        _lib.check(self._lib.rp_mccfr_train(self._h, int(max_steps), float(max_seconds), float(log_interval),
                                            float(flush_interval), C.cast(fn, C.c_void_p), None,
                                            C.byref(interrupt) if interrupt is not None else None, buf, len(buf)))
        return buf.value.decode()

    def exploitability(self) -> float:
        out = C.c_float()
        _lib.check(self._lib.rp_mccfr_exploitability(self._h, C.byref(out)))
        return out.value

    # ---- RefProf / MutProf ----------------------------------------------------------------------
    @property
    def epoch(self) -> int:
        e = C.c_uint64()
        _lib.check(self._lib.rp_mccfr_epoch(self._h, C.byref(e)))
        return e.value

    def counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(self._lib.rp_mccfr_counters(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def get(self, info: int, edge: int) -> _lib.Encounter:
        e = _lib.Encounter()
        _lib.check(self._lib.rp_mccfr_get(self._h, info, edge, C.byref(e)))
        return e

    def set(self, info: int, edge: int, weight=0.0, regret=0.0, payoff=0.0, visits=0):
        e = _lib.Encounter(weight, regret, payoff, visits)
        _lib.check(self._lib.rp_mccfr_set(self._h, info, edge, C.byref(e)))

    def export(self) -> np.ndarray:
        rows = np.zeros(self.cells, dtype=ENC_DTYPE)
        _lib.check(self._lib.rp_mccfr_export(self._h, rows.ctypes.data_as(C.POINTER(_lib.Encounter)), self.cells))
        return rows

    def load_rows(self, rows: np.ndarray, epoch: int):
        buf = np.ascontiguousarray(rows, dtype=ENC_DTYPE)
        _lib.check(self._lib.rp_mccfr_import(self._h, buf.ctypes.data_as(C.POINTER(_lib.Encounter)), buf.size, epoch))

    def policy(self, info: int, kind="averaged") -> np.ndarray:
        out = (C.c_float * 16)()
        n = C.c_uint32()
        _lib.check(self._lib.rp_mccfr_policy(self._h, info, _lib.DIST[kind], out, C.byref(n)))
        return np.array(out[: n.value], dtype=np.float32)

    def sum_regret(self) -> float:
        out = C.c_float()
        _lib.check(self._lib.rp_mccfr_sum_regret(self._h, C.byref(out)))
        return out.value

    # ---- knobs ----------------------------------------------------------------------------------
    def set_batch(self, batch: int):
        _lib.check(self._lib.rp_mccfr_set_batch(self._h, batch))
        self.batch = batch

    def set_update_mode(self, mode: str):
        _lib.check(self._lib.rp_mccfr_set_update_mode(self._h, _lib.UPDATE[mode]))

    def set_rng(self, kind: str, streams: "_lib.HashStreams | None" = None):
        """"counter" (default: the build's own hash) or "reference" (the reference's DefaultHasher -> SmallRng chain,
        include/rp_refrng.h); `streams` defaults to the built-in game's (Game.hash_streams)"""
        if kind == "reference" and streams is None:
            streams = self.game.hash_streams()
        _lib.check(self._lib.rp_mccfr_set_rng(self._h, _lib.RNG[kind], C.byref(streams) if streams is not None else None))

    def set_stream(self, hip_stream_ptr):
        _lib.check(self._lib.rp_mccfr_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    # ---- multi-GPU exchange --------------------------------------------------------
    def set_shard(self, rank: int, world: int):
        _lib.check(self._lib.rp_mccfr_set_shard(self._h, rank, world))

    def summary_bytes(self) -> int:
        n = C.c_size_t()
        _lib.check(self._lib.rp_mccfr_summary_bytes(self._h, C.byref(n)))
        return n.value

    def step_local(self, summary_dev_ptr: int):
        _lib.check(self._lib.rp_mccfr_step_local(self._h, C.c_void_p(summary_dev_ptr)))

    def step_apply(self, gathered_dev_ptr: int, world: int):
        _lib.check(self._lib.rp_mccfr_step_apply(self._h, C.c_void_p(gathered_dev_ptr), world))

    def window_local(self, window_dev_ptr: int, first: bool):
        _lib.check(self._lib.rp_mccfr_window_local(self._h, C.c_void_p(window_dev_ptr), 1 if first else 0))

    def window_apply(self, gathered_dev_ptr: int, world: int):
        _lib.check(self._lib.rp_mccfr_window_apply(self._h, C.c_void_p(gathered_dev_ptr), world))

    def step_comm(self, comm, steps: int, window: int = 1):
        """``steps`` sharded steps over an ``rp_comm`` (robopoker_amd.parallel.Comm): exchange windows of ``window`` local
        steps, one ncclAllGather per window on the solver's stream, no host work in between"""
        _lib.check(self._lib.rp_mccfr_step_comm(self._h, comm.handle, steps, window))

    # ---- profiling hooks ------------------------------------------------------------------------
    def profile(self, enable=True):
        _lib.check(self._lib.rp_mccfr_profile(self._h, 1 if enable else 0))

    def kernel_variant(self) -> str:
        """which Solver::batch kernel runs: "hbm" (any game), "lds" (small games), "static" (compile-time skeleton)"""
        v = C.c_int()
        _lib.check(self._lib.rp_mccfr_traversal_variant(self._h, C.byref(v)))
        return ("hbm", "lds", "static")[v.value]

    def kernel_time(self, name: str):
        ms, n = C.c_double(), C.c_uint64()
        _lib.check(self._lib.rp_mccfr_kernel_time(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

"""The NLHE rules engine on the device (include/rp_mi355x.h, rp_nlhe_playouts; robopoker_amd/csrc/nlhe.hip): random
abstract hands played by the device-side restatement of ``kicker::GameN`` / ``NlheGame::apply`` / ``Showdown::settle``."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch  # before the first HIP call of librp_mi355x.so: one HIP runtime per process

from . import _lib

A = 9  # widest NLHE infoset (pokerkit/src/lib.rs:130-133)
ENC_DTYPE = np.dtype([("weight", "<f4"), ("regret", "<f4"), ("payoff", "<f4"), ("visits", "<u4")])


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class NlheSolver:
    """``mccfr!(Nlhe, NlheEncoder, NlheTurn, NlheEdge, NlheGame, NlheInfo, 128)`` (crates/nlhe/src/solver.rs:11) on one
    MI355X: ``step`` = ``Solver::step``, ``batch`` = ``Solver::batch`` (inspection), ``export`` / ``load`` = the blueprint
    rows by NlheInfo.  ``tables``: the encoder's four ``deuce.Lookup`` (pref, flop, turn, river); None = hash encoder.
    ``sampling``: "external" (the macro's default), "prunable", "pluribus" (the Flagship type)."""

    def __init__(self, cap_log2=20, regret="linear", weight="linear", batch=128, seed=0, hyper=None, tables=None, device=0,
                 sampling="external"):
        self._lib = _lib.load()
        self.hp = hyper
        if self.hp is None:
            self.hp = _lib.Hyper()
            self._lib.rp_hyper_default(C.byref(self.hp))
        self.batch_size = batch
        self._tables = tables
        tab = None
        if tables is not None:
            arr = (C.c_void_p * 4)(*[t._h for t in tables])
            tab = C.cast(arr, C.c_void_p)
            self._tab_arr = arr
        self._h = C.c_void_p()
        _lib.check(self._lib.rp_nlhe_create(device, cap_log2, _lib.REGRET[regret], _lib.WEIGHT[weight], C.byref(self.hp), seed, batch,
                                            tab, C.byref(self._h)))
        if sampling != "external":  # Flagship = Nlhe<LinearRegret, LinearWeight, PluribusSampling> (nlhe/src/lib.rs:86-90)
            _lib.check(self._lib.rp_nlhe_set_sampling(self._h, _lib.SAMPLING[sampling]))

    def set_rng(self, kind: str):
        """"counter" (default) or "reference": opponent draws and Pluribus' coin from the reference's DefaultHasher -> SmallRng chain"""
        _lib.check(self._lib.rp_nlhe_set_rng(self._h, _lib.RNG[kind]))

    def set_exact(self, on: bool = True):
        """regret vectors in the reference's own float order on the batch-wide kernels too (always so up to 2 048 trees per step)"""
        _lib.check(self._lib.rp_nlhe_set_exact(self._h, 1 if on else 0))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rp_nlhe_destroy(self._h)
            self._h = None

    __del__ = close

    def step(self, mode="ordered"):
        _lib.check(self._lib.rp_nlhe_step(self._h, _lib.UPDATE[mode]))

    def train(self, mode="composed", max_steps=0, max_seconds=0.0, log_interval=60.0, flush_interval=1800.0, on_checkpoint=None,
              on_flush=None, interrupt=None):
        """``Trainer::train`` (crates/forge/src/trainer.rs:18-66) over this solver; returns Progress::summary"""
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
        _lib.check(self._lib.rp_nlhe_train(self._h, _lib.UPDATE[mode], int(max_steps), float(max_seconds), float(log_interval),
                                           float(flush_interval), C.cast(fn, C.c_void_p), None,
                                           C.byref(interrupt) if interrupt is not None else None, buf, len(buf)))
        return buf.value.decode()

    def profile(self, enable: bool):
        _lib.check(self._lib.rp_nlhe_profile(self._h, 1 if enable else 0))

    def kernel_times(self):
        """{group: (total_ms, launches)} since profile(True); groups: expand, children, sweeps, decide, apply"""
        out = {}
        for name in ("expand", "children", "sweeps", "decide", "apply"):
            ms, n = C.c_double(), C.c_uint64()
            _lib.check(self._lib.rp_nlhe_kernel_time(self._h, name.encode(), C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def census(self):
        """nodes of the profiled steps by kind + children of their walker nodes"""
        k, w = (C.c_uint64 * 4)(), C.c_uint64()
        _lib.check(self._lib.rp_nlhe_census(self._h, k, C.byref(w)))
        return dict(terminal=k[0], chance=k[1], walker=k[2], opponent=k[3], walker_children=w.value)

    def last_shape(self):
        """(levels, nodes) of the last traversed batch"""
        a, b = C.c_uint32(), C.c_uint32()
        _lib.check(self._lib.rp_nlhe_last_shape(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def batch(self, cap=1 << 22):
        n = C.c_uint32()
        _lib.check(self._lib.rp_nlhe_batch(self._h, 0, C.byref(n), *([None] * 9)))
        m = min(cap, n.value)
        out = dict(n=n.value, tree=np.zeros(m, np.uint32), past=np.zeros(m, np.uint64), present=np.zeros(m, np.uint32),
                   choices=np.zeros(m, np.uint64), n_actions=np.zeros(m, np.uint8), expanded=np.zeros(m, np.uint16),
                   regret=np.zeros((m, A), np.float32), policy=np.zeros((m, A), np.float32), payoff=np.zeros(m, np.float32))
        _lib.check(self._lib.rp_nlhe_batch(self._h, m, C.byref(n), _p(out["tree"]), _p(out["past"]), _p(out["present"]), _p(out["choices"]),
                                           _p(out["n_actions"]), _p(out["expanded"]), _p(out["regret"]), _p(out["policy"]), _p(out["payoff"])))
        return out

    # ---- multi-GPU exchange by infoset key (include/rp_mi355x.h rp_nlhe_step_local / step_apply) ----
    def set_shard(self, rank: int, world: int):
        _lib.check(self._lib.rp_nlhe_set_shard(self._h, rank, world))

    def entry_bytes(self):
        b, m = C.c_size_t(), C.c_uint32()
        _lib.check(self._lib.rp_nlhe_entry_bytes(self._h, C.byref(b), C.byref(m)))
        return b.value, m.value

    def step_local(self, entries_ptr: int, past_ptr: int, present_ptr: int, choices_ptr: int) -> int:
        n = C.c_uint32()
        _lib.check(self._lib.rp_nlhe_step_local(self._h, C.c_void_p(entries_ptr), C.c_void_p(past_ptr), C.c_void_p(present_ptr),
                                                C.c_void_p(choices_ptr), C.byref(n)))
        return n.value

    def step_apply(self, entries_ptr: int, past_ptr: int, present_ptr: int, choices_ptr: int, n: int):
        _lib.check(self._lib.rp_nlhe_step_apply(self._h, C.c_void_p(entries_ptr), C.c_void_p(past_ptr), C.c_void_p(present_ptr),
                                                C.c_void_p(choices_ptr), n))

    def step_comm(self, comm, steps: int = 1):
        """`steps` sharded steps over the library's own RCCL communicator (robopoker_amd.mccfr.Comm)"""
        _lib.check(self._lib.rp_nlhe_step_comm(self._h, comm.handle, steps))

    def set_stream(self, hip_stream_ptr):
        _lib.check(self._lib.rp_nlhe_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def sync(self):
        _lib.check(self._lib.rp_nlhe_sync(self._h))

    @property
    def epoch(self) -> int:
        e = C.c_uint64()
        _lib.check(self._lib.rp_nlhe_epoch(self._h, C.byref(e)))
        return e.value

    def counters(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.check(self._lib.rp_nlhe_counters(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def export(self):
        n = C.c_uint64()
        _lib.check(self._lib.rp_nlhe_export(self._h, 0, C.byref(n), None, None, None, None))
        m = n.value
        past, present, choices = np.zeros(m, np.uint64), np.zeros(m, np.uint32), np.zeros(m, np.uint64)
        enc = np.zeros((m, A), dtype=ENC_DTYPE)
        _lib.check(self._lib.rp_nlhe_export(self._h, m, C.byref(n), _p(past), _p(present), _p(choices), _p(enc)))
        return past, present, choices, enc

    def load(self, past, present, choices, enc, epoch: int):
        past, present, choices = (np.ascontiguousarray(past, np.uint64), np.ascontiguousarray(present, np.uint32),
                                  np.ascontiguousarray(choices, np.uint64))
        enc = np.ascontiguousarray(enc, dtype=ENC_DTYPE)
        _lib.check(self._lib.rp_nlhe_import(self._h, past.size, _p(past), _p(present), _p(choices), _p(enc), epoch))



def playouts(n_players: int, n_games: int, seed: int, max_steps: int = 200, device: int = 0):
    """-> (payoffs float32[n_games][n_players], digests int64[n_games] (the u64 bit patterns), steps int32[n_games])."""
    dev = torch.device("cuda", device)
    pay = torch.empty((n_games, n_players), dtype=torch.float32, device=dev)
    dig = torch.empty(n_games, dtype=torch.int64, device=dev)
    steps = torch.empty(n_games, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    _lib.check(_lib.load().rp_nlhe_playouts(device, n_players, n_games, seed, max_steps, pay.data_ptr(), dig.data_ptr(), steps.data_ptr()))
    return pay, dig, steps

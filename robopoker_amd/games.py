"""Built-in game tables (crates/kuhn, crates/leduc, crates/roshambo) behind ``rp_game_*``."""
from __future__ import annotations

import ctypes as C

from . import _lib


class Game:
    """Owns an ``rp_game`` handle; ``table`` is the flat ``rp_game_table`` view passed to solvers."""

    def __init__(self, kind: str):
        lib = _lib.load()
        self.kind = kind
        self._h = C.c_void_p()
        _lib.check(lib.rp_game_create(_lib.GAME[kind], C.byref(self._h)))
        self.table = _lib.GameTable()
        _lib.check(lib.rp_game_view(self._h, C.byref(self.table)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _lib.load().rp_game_destroy(h)
            self._h = None

    @property
    def n_infos(self) -> int:
        return self.table.n_infos

    @property
    def max_actions(self) -> int:
        return self.table.max_actions

    def hash_streams(self) -> _lib.HashStreams:
        """the `Hash::hash` byte stream of every infoset and in-tree chance info (reference-seed mode, Solver.set_rng)"""
        out = _lib.HashStreams()
        _lib.check(_lib.load().rp_game_hash_streams(self._h, C.byref(out)))
        return out

    def skeleton(self) -> str:
        """the compile-time action skeleton this table matches (csrc/traverse_static.hpp): "kuhn", "leduc" or "" — decides
        whether the solver's traversal is the instantiated kernel or the generic per-lane DFS; host-side check, no device"""
        out = C.c_int()
        _lib.check(_lib.load().rp_game_skeleton(C.byref(self.table), C.byref(out)))
        return ("", "kuhn", "leduc")[out.value]

    def info_id(self, name: str) -> int:
        out = C.c_uint32()
        _lib.check(_lib.load().rp_game_info_id(self._h, name.encode(), C.byref(out)))
        return out.value

    def info_name(self, info: int) -> str:
        buf = C.create_string_buffer(64)
        _lib.check(_lib.load().rp_game_info_name(self._h, info, buf, 64))
        return buf.value.decode()

    def n_actions(self, info: int) -> int:
        return self.table.info_actions[info]

    def player(self, info: int) -> int:
        return self.table.info_player[info]

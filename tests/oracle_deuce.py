"""ctypes wrapper of the CPU oracle's abstraction-input functions (oracle/rp_oracle_deuce.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

import oracle

RANKS = "23456789TJQKA"
SUITS = "cdhs"
VARIANTS = ["HighCard", "OnePair", "TwoPair", "ThreeOAK", "Straight", "FullHouse", "Flush", "FourOAK", "StraightFlush"]
STREETS = {"pref": 0, "flop": 1, "turn": 2, "rive": 3}

_o = None


def lib() -> C.CDLL:
    global _o
    if _o is not None:
        return _o
    o = oracle.load()
    u64, u32, i64 = C.c_uint64, C.c_uint32, C.c_int64
    o.ora_strength_key.restype = u32
    o.ora_strength_key.argtypes = [u64]
    o.ora_strength.argtypes = [u64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(u32)]
    o.ora_hand_iter_first.restype = u64
    o.ora_hand_iter_first.argtypes = [u32, u64]
    o.ora_hand_iter_next.restype = u64
    o.ora_hand_iter_next.argtypes = [u64, u64]
    o.ora_obs_to_i64.restype = i64
    o.ora_obs_to_i64.argtypes = [u64, u64]
    o.ora_obs_from_i64.argtypes = [i64, C.POINTER(u64), C.POINTER(u64)]
    o.ora_river_equity.restype = C.c_float
    o.ora_river_equity.argtypes = [u64, u64, C.POINTER(u32), C.POINTER(u32)]
    o.ora_quantize.restype = u32
    o.ora_quantize.argtypes = [C.c_float]
    o.ora_permutation.argtypes = [u64, u64, C.POINTER(C.c_uint8)]
    o.ora_permute.argtypes = [C.POINTER(C.c_uint8), u64, C.POINTER(u64)]
    o.ora_isomorphism.argtypes = [u64, u64, C.POINTER(u64), C.POINTER(u64)]
    o.ora_is_canonical.restype = C.c_int
    o.ora_is_canonical.argtypes = [u64, u64]
    o.ora_isomorphisms.restype = u64
    o.ora_isomorphisms.argtypes = [C.c_int, u32, u32, C.c_void_p, u64]
    o.ora_lookup_index.restype = i64
    o.ora_lookup_index.argtypes = [C.c_void_p, u64, u64, u64]
    o.ora_project.restype = C.c_int
    o.ora_project.argtypes = [i64, C.c_void_p, C.c_void_p, u64, u32, C.c_void_p]
    o.ora_project_river.argtypes = [i64, C.c_void_p]
    _o = o
    return o


def card(s: str) -> int:
    return RANKS.index(s[0].upper()) * 4 + SUITS.index(s[1].lower())


def hand(s: str) -> int:
    """Hand::try_from(&str) (hand.rs:158-164): cards in any order, whitespace ignored."""
    s = "".join(s.split())
    h = 0
    for i in range(0, len(s), 2):
        h |= 1 << card(s[i:i + 2])
    return h


def strength(h: int):
    v, r1, r2, k = C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint32()
    lib().ora_strength(h, C.byref(v), C.byref(r1), C.byref(r2), C.byref(k))
    return VARIANTS[v.value], r1.value, r2.value, k.value


def strength_key(h: int) -> int:
    return lib().ora_strength_key(h)


def kick(ranks: str) -> int:
    return sum(1 << RANKS.index(c) for c in ranks)


def hand_iter(n: int, mask: int):
    o = lib()
    h = o.ora_hand_iter_first(n, mask)
    while h:
        yield h
        h = o.ora_hand_iter_next(h, mask)


def obs(s: str):
    """'AsKh~2c3d4h' -> (pocket, public) bit sets (observation.rs:224-238)."""
    po, _, pu = s.partition("~")
    return hand(po), hand(pu)


def obs_i64(pocket: int, public: int) -> int:
    return lib().ora_obs_to_i64(pocket, public)


def obs_from_i64(v: int):
    a, b = C.c_uint64(), C.c_uint64()
    lib().ora_obs_from_i64(v, C.byref(a), C.byref(b))
    return a.value, b.value


def river_equity(pocket: int, public: int):
    w, s = C.c_uint32(), C.c_uint32()
    e = lib().ora_river_equity(pocket, public, C.byref(w), C.byref(s))
    return np.float32(e), w.value, s.value


def quantize(p) -> int:
    return lib().ora_quantize(float(p))


def permutation(pocket: int, public: int):
    p = (C.c_uint8 * 4)()
    lib().ora_permutation(pocket, public, p)
    return list(p)


def permute(perm, h: int) -> int:
    p = (C.c_uint8 * 4)(*perm)
    out = C.c_uint64()
    lib().ora_permute(p, h, C.byref(out))
    return out.value


def isomorphism(pocket: int, public: int):
    a, b = C.c_uint64(), C.c_uint64()
    lib().ora_isomorphism(pocket, public, C.byref(a), C.byref(b))
    return a.value, b.value


def is_canonical(pocket: int, public: int) -> bool:
    return bool(lib().ora_is_canonical(pocket, public))


def isomorphisms(street, pocket_lo=0, pocket_hi=1326, count_only=False):
    st = STREETS[street] if isinstance(street, str) else street
    n = lib().ora_isomorphisms(st, pocket_lo, pocket_hi, None, 0)
    if count_only:
        return n
    out = np.zeros(n, dtype=np.int64)
    lib().ora_isomorphisms(st, pocket_lo, pocket_hi, out.ctypes.data, n)
    return out


def project(obs_list, keys, abs_, bins):
    keys = np.ascontiguousarray(keys, dtype=np.int64)
    abs_ = np.ascontiguousarray(abs_, dtype=np.uint8)
    out = np.zeros((len(obs_list), bins), dtype=np.uint32)
    for i, o in enumerate(obs_list):
        rc = lib().ora_project(int(o), keys.ctypes.data, abs_.ctypes.data, len(keys), bins, out[i].ctypes.data)
        if rc != 0:
            raise KeyError(f"child of observation {int(o):#x} missing from the table")
    return out


def project_river(turn_obs):
    out = np.zeros((len(turn_obs), 101), dtype=np.uint32)
    for i, o in enumerate(turn_obs):
        lib().ora_project_river(int(o), out[i].ctypes.data)
    return out

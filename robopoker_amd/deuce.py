"""Abstraction inputs on the GPU (include/rp_mi355x.h, "abstraction inputs"): the mirror of the reference's
``deuce`` surface that feeds the k-means layers — ``Strength::from(Hand)``, ``Isomorphism::from(Observation)``,
``IsomorphismIterator::from(street)``, ``Observation::equity`` and ``lloyd::Lookup`` (``lookup`` / ``projections``).

Card, hand and observation encodings are the reference's (crates/deuce/src/card.rs:16-20, hand.rs:7,
observation.rs:132-165).  Bulk arrays live in device memory as torch tensors (memory plumbing only); the kernels are
in robopoker_amd/csrc/deuce.hip.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch  # imported BEFORE the first HIP call of librp_mi355x.so: one HIP runtime per process

from . import _lib

RANKS = "23456789TJQKA"
SUITS = "cdhs"
STREETS = {"pref": 0, "flop": 1, "turn": 2, "rive": 3}
N_OBSERVED = (2, 5, 6, 7)                            # street.rs:59-66
N_ISOMORPHISMS = (169, 1_286_792, 13_960_050, 123_156_254)  # street.rs:120-127
N_POCKETS = 1326
RIVER_BUCKETS = 101                                  # pokerkit KMEANS_EQTY_CLUSTER_COUNT


def _street(street) -> int:
    return STREETS[street] if isinstance(street, str) else int(street)


def card(s: str) -> int:
    """Card::try_from(&str) (card.rs:59-71)."""
    return RANKS.index(s[0].upper()) * 4 + SUITS.index(s[1].lower())


def hand(s: str) -> int:
    """Hand::try_from(&str) (hand.rs:158-164)."""
    s = "".join(s.split())
    return sum(1 << card(s[i:i + 2]) for i in range(0, len(s), 2))


def observation(s: str) -> int:
    """Observation::try_from("AsKh~2c3d4h") as its i64 form (observation.rs:132-141,224-238)."""
    po, _, pu = s.partition("~")
    acc = 0
    for h in (hand(pu), hand(po)):
        for c in range(52):
            if h >> c & 1:
                acc = acc << 8 | (c + 1)
    return acc


def kernel_ms() -> float:
    ms = C.c_double()
    _lib.check(_lib.load().rp_deuce_kernel_ms(C.byref(ms)))
    return ms.value


def hand_strength(hands, device=0) -> np.ndarray:
    """Strength::from(Hand) as an order key (variant << 21 | rank1 << 17 | rank2 << 13 | kickers)."""
    h = np.ascontiguousarray(hands, dtype=np.uint64)
    out = np.zeros(h.size, dtype=np.uint32)
    _lib.check(_lib.load().rp_hand_strength(device, h.size, h.ctypes.data, out.ctypes.data))
    return out


def canonical(obs, device=0) -> np.ndarray:
    """i64::from(Isomorphism::from(Observation::from(obs)))."""
    o = np.ascontiguousarray(obs, dtype=np.int64)
    out = np.zeros(o.size, dtype=np.int64)
    _lib.check(_lib.load().rp_obs_canonical(device, o.size, o.ctypes.data, out.ctypes.data))
    return out


def count_isomorphisms(street, pocket_lo=0, pocket_hi=N_POCKETS, device=0) -> int:
    n = C.c_uint64()
    _lib.check(_lib.load().rp_isomorphisms(device, _street(street), pocket_lo, pocket_hi, None, 0, C.byref(n)))
    return n.value


def isomorphisms(street, pocket_lo=0, pocket_hi=N_POCKETS, device=0) -> torch.Tensor:
    """IsomorphismIterator::from(street) over the pockets [pocket_lo, pocket_hi): int64 tensor on the device."""
    n = count_isomorphisms(street, pocket_lo, pocket_hi, device)
    out = torch.empty(n, dtype=torch.int64, device=torch.device("cuda", device))
    torch.cuda.synchronize(device)
    m = C.c_uint64()
    _lib.check(_lib.load().rp_isomorphisms(device, _street(street), pocket_lo, pocket_hi, out.data_ptr() if n else None, n,
                                           C.byref(m)))
    assert m.value == n
    return out


def shard_pockets(rank: int, world: int) -> tuple[int, int]:
    """Pocket range of one rank: the iterator's outer loop splits with no exchange (independent units)."""
    return rank * N_POCKETS // world, (rank + 1) * N_POCKETS // world


def river_equity(obs: torch.Tensor):
    """Observation::equity and its river bucket for a device tensor of river observations -> (f32, u8) tensors."""
    assert obs.dtype == torch.int64 and obs.is_cuda and obs.is_contiguous()
    dev = obs.device
    e = torch.empty(obs.numel(), dtype=torch.float32, device=dev)
    b = torch.empty(obs.numel(), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    _lib.check(_lib.load().rp_river_equity(dev.index or 0, obs.numel(), obs.data_ptr() if obs.numel() else None,
                                           e.data_ptr() if obs.numel() else None, b.data_ptr() if obs.numel() else None))
    return e, b


class Lookup:
    """lloyd::Lookup (lookup.rs:9-45): isomorphisms of one street, in iterator order, with their abstractions."""

    def __init__(self, street, obs: torch.Tensor, abstraction: torch.Tensor):
        assert obs.dtype == torch.int64 and abstraction.dtype == torch.uint8 and obs.is_cuda and abstraction.is_cuda
        assert obs.numel() == abstraction.numel()
        self._lib = _lib.load()
        self.street, self.n, self.device = _street(street), obs.numel(), obs.device
        torch.cuda.synchronize(self.device)
        h = C.c_void_p()
        _lib.check(self._lib.rp_lookup_create(self.device.index or 0, self.street, self.n, obs.contiguous().data_ptr(),
                                              abstraction.contiguous().data_ptr(), C.byref(h)))
        self._h = h

    @classmethod
    def grow_river(cls, device=0, pocket_lo=0, pocket_hi=N_POCKETS):
        """Lookup::grow(Street::Rive) (lookup.rs:172-178): every river isomorphism with its quantised equity."""
        obs = isomorphisms("rive", pocket_lo, pocket_hi, device)
        _, bucket = river_equity(obs)
        return cls("rive", obs, bucket)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rp_lookup_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def lookup(self, obs: torch.Tensor) -> torch.Tensor:
        assert obs.dtype == torch.int64 and obs.is_cuda
        obs = obs.contiguous()
        out = torch.empty(obs.numel(), dtype=torch.uint8, device=self.device)
        torch.cuda.synchronize(self.device)
        if obs.numel():
            _lib.check(self._lib.rp_lookup_get(self._h, obs.numel(), obs.data_ptr(), out.data_ptr()))
        return out

    def projections(self, obs: torch.Tensor, bins: int) -> torch.Tensor:
        """Lookup::projections over the given previous-street observations -> u8[n][bins] on the device: the
        ``counts`` of ``rp_kmeans_create_device``."""
        assert obs.dtype == torch.int64 and obs.is_cuda
        obs = obs.contiguous()
        out = torch.empty((obs.numel(), bins), dtype=torch.uint8, device=self.device)
        torch.cuda.synchronize(self.device)
        if obs.numel():
            _lib.check(self._lib.rp_lookup_project(self._h, obs.numel(), obs.data_ptr(), bins, out.data_ptr()))
        return out


def bench_inputs(device=0) -> dict:
    """Full-size pass over the abstraction inputs on one GPU: every street's isomorphism list, the river table
    (123 156 254 equities), the turn layer's points from it, and the flop layer's points through a stand-in turn table.
    Device times come from HIP events inside the library (rp_deuce_kernel_ms)."""
    out = {}
    lists = {}
    for street in ("flop", "turn", "rive"):
        lists[street] = isomorphisms(street, device=device)
        out[f"isomorphisms_{street}"] = {"n": lists[street].numel(), "device_ms": round(kernel_ms(), 3)}
    river = lists["rive"]
    _, bucket = river_equity(river)
    ms = kernel_ms()
    # VALU issue model: one wave64 VALU instruction per SIMD per 4 cycles; 256 CUs x 4 SIMDs x 2.4 GHz
    out["river_equity"] = {"n": river.numel(), "device_ms": round(ms, 3), "showdowns_per_s": river.numel() * 990 / (ms * 1e-3),
                           "observations_per_s": river.numel() / (ms * 1e-3),
                           "simd_cycles_per_showdown_round": ms * 1e-3 * 256 * 4 * 2.4e9 / (river.numel() * 16)}
    table = Lookup("rive", river, bucket)
    pts = table.projections(lists["turn"], RIVER_BUCKETS)
    ms = kernel_ms()
    rows = int((pts.sum(dim=1, dtype=torch.int32) == 46).sum().item())
    out["project_turn"] = {"n": lists["turn"].numel(), "device_ms": round(ms, 3), "points_per_s": lists["turn"].numel() / (ms * 1e-3),
                           "lookups_per_s": lists["turn"].numel() * 46 / (ms * 1e-3), "rows_summing_to_46": rows,
                           "hbm_gb_per_s": (lists["turn"].numel() * (8 + 101)) / (ms * 1e-3) / 1e9}
    del pts
    table.close()
    stand_in = (((lists["turn"] * 2654435761) >> 20) % 200).to(torch.uint8)
    t2 = Lookup("turn", lists["turn"], stand_in)
    pts = t2.projections(lists["flop"], 200)
    ms = kernel_ms()
    out["project_flop"] = {"n": lists["flop"].numel(), "device_ms": round(ms, 3), "points_per_s": lists["flop"].numel() / (ms * 1e-3),
                           "table": "stand-in labels (the turn layer's clustering is the k-means path's output)"}
    t2.close()
    return out

"""Artifact files in the reference's PostgreSQL binary COPY row format (include/rp_mi355x.h "artifact files";
crates/daybook/src/traits/{streamable,row}.rs).  Host I/O only — no torch, no GPU."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

DTYPES = {"h": np.int16, "i": np.int32, "q": np.int64, "f": np.float32}
STREETS = {"pref": 0, "flop": 1, "turn": 2, "rive": 3}
# table and column names of the reference's schemas (lloyd/src/{lookup,metric,future}.rs copy())
TABLES = {"lookup": ("isomorphism", "obs, abs", "qh"), "metric": ("metric", "tri, dx", "if"),
          "transitions": ("transitions", "prev, next, dx", "hhf")}


def _street(street) -> int:
    return STREETS[street] if isinstance(street, str) else int(street)


def write_rows(path: str, types: str, columns) -> None:
    cols = [np.ascontiguousarray(c, dtype=DTYPES[t]) for t, c in zip(types, columns)]
    n = len(cols[0]) if cols else 0
    assert all(len(c) == n for c in cols)
    ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    _lib.check(_lib.load().rp_pgcopy_write(path.encode(), types.encode(), n, ptrs))


def read_rows(path: str, types: str):
    n = C.c_uint64()
    _lib.check(_lib.load().rp_pgcopy_read(path.encode(), types.encode(), 0, None, C.byref(n)))
    cols = [np.zeros(n.value, dtype=DTYPES[t]) for t in types]
    ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    m = C.c_uint64()
    _lib.check(_lib.load().rp_pgcopy_read(path.encode(), types.encode(), n.value, ptrs, C.byref(m)))
    assert m.value == n.value
    return cols


def write_lookup(path: str, street, obs, abs_index) -> None:
    obs = np.ascontiguousarray(obs, dtype=np.int64)
    a = np.ascontiguousarray(abs_index, dtype=np.uint8)
    assert obs.size == a.size
    _lib.check(_lib.load().rp_artifact_write_lookup(path.encode(), _street(street), obs.size, obs.ctypes.data, a.ctypes.data))


def write_metric(path: str, street, K: int, tri) -> None:
    tri = np.ascontiguousarray(tri, dtype=np.float32)
    assert tri.size == K * (K - 1) // 2
    _lib.check(_lib.load().rp_artifact_write_metric(path.encode(), _street(street), K, tri.ctypes.data))


def write_transitions(path: str, street, counts, weight) -> None:
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    weight = np.ascontiguousarray(weight, dtype=np.uint64)
    K, bins = counts.shape
    _lib.check(_lib.load().rp_artifact_write_transitions(path.encode(), _street(street), K, bins, counts.ctypes.data, weight.ctypes.data))


BLUEPRINT_TYPES = "qhqqfffi"  # past, present, choices, edge, weight, regret, payoff, visits (nlhe/src/profile.rs:20-31)
_ENC = np.dtype([("weight", "<f4"), ("regret", "<f4"), ("payoff", "<f4"), ("visits", "<u4")])


def write_blueprint(path: str, past, present, choices, enc, only_visited=True) -> int:
    """NlheProfile::rows as a COPY file from ``NlheSolver.export()``'s arrays; returns the number of rows written."""
    past, present, choices = (np.ascontiguousarray(past, np.uint64), np.ascontiguousarray(present, np.uint32),
                              np.ascontiguousarray(choices, np.uint64))
    enc = np.ascontiguousarray(enc, dtype=_ENC)
    assert enc.shape == (past.size, 9)
    n = C.c_uint64()
    _lib.check(_lib.load().rp_artifact_write_blueprint(path.encode(), past.size, past.ctypes.data, present.ctypes.data, choices.ctypes.data,
                                                       enc.ctypes.data, 1 if only_visited else 0, C.byref(n)))
    return n.value


def _edge_to_u64(code: int) -> int:
    opens, raises = [2, 3, 4, 5], [(1, 4), (1, 3), (1, 2), (2, 3), (3, 4), (1, 1), (5, 4), (3, 2), (2, 1), (3, 1)]
    if code < 6:
        return {1: 0, 2: 1, 3: 2, 4: 3, 5: 5}[code]
    if code < 10:
        return 6 | (opens[code - 6] << 3)
    return 4 | (raises[code - 10][0] << 3) | (raises[code - 10][1] << 11)


def read_blueprint(path: str):
    """the blueprint COPY file back into ``NlheSolver.load()``'s arrays (Hydrate, nlhe/src/profile.rs:90-141): rows grouped
    by infoset (in the order of their first row), each row's slot = the position of its edge in the infoset's choices.
    Vectorised: a blueprint of a few million infosets has tens of millions of rows."""
    past, present, choices, edge, weight, regret, payoff, visits = read_rows(path, BLUEPRINT_TYPES)
    past, choices = np.asarray(past).astype(np.int64).view(np.uint64), np.asarray(choices).astype(np.int64).view(np.uint64)
    present = (np.asarray(present).astype(np.int64) & 0xffff).astype(np.uint32)
    key = np.zeros(past.size, dtype=[("p", "<u8"), ("b", "<u4"), ("c", "<u8")])
    key["p"], key["b"], key["c"] = past, present, choices
    _, first, inverse = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")  # infosets in the order of their first row
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    group = rank[inverse]
    # the row's slot: the a with Edge::into_u64(choices[a]) == edge
    table = np.zeros(32, dtype=np.uint64)
    for code in range(1, 20):
        table[code] = _edge_to_u64(code)
    slot = np.full(past.size, -1, dtype=np.int64)
    e64 = np.asarray(edge).astype(np.int64).view(np.uint64)
    for a in range(9):
        code = (choices >> np.uint64(5 * a)) & np.uint64(0x1f)
        hit = (slot < 0) & (code != 0) & (table[code.astype(np.int64)] == e64)
        slot[hit] = a
    if (slot < 0).any():
        raise ValueError("read_blueprint: a row's edge is not among its infoset's choices")
    enc = np.zeros((order.size, 9), dtype=_ENC)
    enc["weight"][group, slot] = weight
    enc["regret"][group, slot] = regret
    enc["payoff"][group, slot] = payoff
    enc["visits"][group, slot] = np.asarray(visits).astype(np.uint32)
    f = first[order]
    return past[f], present[f], choices[f], enc


def save_artifacts(directory: str, art) -> dict:
    """Artifacts::stream (lloyd/src/artifacts.rs) to files: one lookup file per street plus metric / transitions where
    the street has them.  `art` is a robopoker_amd.pretraining.Artifacts.  Returns {kind: path}."""
    os.makedirs(directory, exist_ok=True)
    out = {}
    p = os.path.join(directory, f"{art.street}.isomorphism.pgcopy")
    write_lookup(p, art.street, art.obs.cpu().numpy(), art.abstraction.cpu().numpy())
    out["lookup"] = p
    if art.metric is not None:
        K = art.future.shape[0]
        p = os.path.join(directory, f"{art.street}.metric.pgcopy")
        write_metric(p, art.street, K, art.metric)
        out["metric"] = p
        p = os.path.join(directory, f"{art.street}.transitions.pgcopy")
        write_transitions(p, art.street, art.future, art.future_weight)
        out["transitions"] = p
    return out

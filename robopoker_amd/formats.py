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

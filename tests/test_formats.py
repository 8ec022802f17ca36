"""Artifact files (robopoker_amd/csrc/formats.cpp) against an independent restatement of PostgreSQL's binary COPY
format with `struct` — the byte stream tokio_postgres::binary_copy::BinaryCopyInWriter produces for the reference's
row shapes (crates/daybook/src/traits/row.rs:21-57).  Host code: runs without a GPU."""
import struct

import numpy as np
import pytest

from robopoker_amd import formats
from robopoker_amd._lib import RpError

SIG = b"PGCOPY\n\xff\r\n\x00"
PACK = {"h": ">h", "i": ">i", "q": ">q", "f": ">f"}


def pg_stream(types, rows):
    """The published format, written the slow way: header, (count, (len, value)*)*, trailer."""
    out = bytearray(SIG + struct.pack(">ii", 0, 0))
    for row in rows:
        out += struct.pack(">h", len(types))
        for t, v in zip(types, row):
            body = struct.pack(PACK[t], v)
            out += struct.pack(">i", len(body)) + body
    out += struct.pack(">h", -1)
    return bytes(out)


@pytest.mark.parametrize("types", ["qh", "if", "hhf", "qhqqfffi"])  # row.rs:21-57
def test_generic_rows_are_the_published_byte_stream(tmp_path, types):
    rng = np.random.default_rng(len(types))
    n = 257
    cols = []
    for t in types:
        if t == "f":
            cols.append(rng.standard_normal(n).astype(np.float32))
        else:
            info = np.iinfo(formats.DTYPES[t])
            cols.append(rng.integers(info.min, info.max, n, dtype=formats.DTYPES[t], endpoint=True))
    p = str(tmp_path / "rows.pgcopy")
    formats.write_rows(p, types, cols)
    want = pg_stream(types, [tuple(c[i].item() for c in cols) for i in range(n)])
    assert open(p, "rb").read() == want
    back = formats.read_rows(p, types)
    for a, b in zip(cols, back):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


def test_empty_table_and_bad_files(tmp_path):
    p = str(tmp_path / "empty.pgcopy")
    formats.write_rows(p, "qh", [np.zeros(0, np.int64), np.zeros(0, np.int16)])
    assert open(p, "rb").read() == SIG + b"\0" * 8 + b"\xff\xff"
    assert all(len(c) == 0 for c in formats.read_rows(p, "qh"))
    bad = tmp_path / "bad.pgcopy"
    bad.write_bytes(b"not a copy stream at all")
    with pytest.raises(RpError):
        formats.read_rows(str(bad), "qh")
    trunc = tmp_path / "trunc.pgcopy"
    trunc.write_bytes(pg_stream("qh", [(1, 2)])[:-2])  # no trailer
    with pytest.raises(RpError):
        formats.read_rows(str(trunc), "qh")
    with pytest.raises(RpError):
        formats.read_rows(p, "qx")
    good = tmp_path / "good.pgcopy"
    good.write_bytes(pg_stream("qh", [(1, 2)]))
    with pytest.raises(RpError):  # wrong shape for the file
        formats.read_rows(str(good), "if")


def test_lookup_rows(tmp_path):
    # lookup.rs:141-147: (i64::from(iso), i16::from(abs)); abs = street << 8 | index (abstraction.rs:65-71)
    obs = np.array([0x0102030405, 0x0A0B0C0D0E0F, 7], dtype=np.int64)
    idx = np.array([0, 255, 17], dtype=np.uint8)
    p = str(tmp_path / "turn.pgcopy")
    formats.write_lookup(p, "turn", obs, idx)
    rows = [(int(o), struct.unpack(">h", struct.pack(">H", 2 << 8 | int(i)))[0]) for o, i in zip(obs, idx)]
    assert open(p, "rb").read() == pg_stream("qh", rows)
    o2, a2 = formats.read_rows(p, "qh")
    assert np.array_equal(o2, obs) and np.array_equal(a2.view(np.uint16), (2 << 8) | idx.astype(np.uint16))
    # river abstractions carry street 3 in the high byte: 0x0364 = "R::64" (abstraction.rs:135-139)
    formats.write_lookup(p, "rive", obs[:1], np.array([100], dtype=np.uint8))
    assert formats.read_rows(p, "qh")[1][0] == 0x0364


def test_metric_rows(tmp_path):
    # metric.rs:219-226 over distances.rs:69-84: (street << 30 | t, dx) for t = 0 .. K(K-1)/2 - 1 (pair.rs:7-16)
    K = 6
    tri = np.linspace(0.0, 1.0, K * (K - 1) // 2, dtype=np.float32)
    p = str(tmp_path / "metric.pgcopy")
    formats.write_metric(p, "flop", K, tri)
    rows = [((1 << 30) | t, float(tri[t])) for t in range(len(tri))]
    assert open(p, "rb").read() == pg_stream("if", rows)
    formats.write_metric(p, "turn", K, tri)  # street 2 sets bit 31: the i32 is negative, as i32::from(Pair) (pair.rs:61-65)
    t2, d2 = formats.read_rows(p, "if")
    assert (t2 < 0).all() and np.array_equal(t2.view(np.uint32), (2 << 30) | np.arange(len(tri), dtype=np.uint32))
    assert np.array_equal(d2, tri)


def test_transition_rows(tmp_path):
    # future.rs:99-111 + bins.rs:113-117: density descending, ties in support (ascending) order
    counts = np.array([[0, 5, 0, 5, 10], [3, 0, 0, 0, 0], [0, 0, 0, 0, 0]], dtype=np.uint32)
    weight = counts.sum(axis=1).astype(np.uint64)
    p = str(tmp_path / "future.pgcopy")
    formats.write_transitions(p, "flop", counts, weight)
    f = np.float32
    rows = [(1 << 8 | 0, 2 << 8 | 4, float(f(10) / f(20))), (1 << 8 | 0, 2 << 8 | 1, float(f(5) / f(20))),
            (1 << 8 | 0, 2 << 8 | 3, float(f(5) / f(20))), (1 << 8 | 1, 2 << 8 | 0, 1.0)]  # the empty centroid has no rows
    assert open(p, "rb").read() == pg_stream("hhf", rows)


def test_blueprint_rows_roundtrip(tmp_path):
    # NlheProfile::rows (nlhe/src/profile.rs:144-163): (past, present, choices, edge, weight, regret, payoff, visits) per
    # (infoset, edge); checked against an independent `struct` reading of the PostgreSQL binary COPY stream
    import oracle_nlmc as M

    s = M.OracleNlhe(cap_log2=14, batch=24, seed=4)
    s.step()
    past, present, choices, enc = s.export()
    path = str(tmp_path / "blueprint.pgcopy")
    rows = formats.write_blueprint(path, past, present, choices, enc, only_visited=True)
    visited = enc["visits"][:, 0] > 0
    n_edges = lambda c: sum(1 for k in range(12) if (int(c) >> (5 * k)) & 0x1f)  # noqa: E731
    assert rows == sum(n_edges(c) for c in choices[visited]) and rows > 0
    raw = open(path, "rb").read()
    assert raw[:11] == b"PGCOPY\n\xff\r\n\0" and raw[-2:] == b"\xff\xff"
    off = 19
    nf, l0 = struct.unpack(">hi", raw[off:off + 6])
    assert nf == 8 and l0 == 8
    first = struct.unpack(">q", raw[off + 6:off + 14])[0]
    assert first == int(past[visited][0])
    kp, kb, kc, kenc = formats.read_blueprint(path)
    want = M.as_map(past[visited], present[visited], choices[visited], enc[visited])
    got = M.as_map(kp, kb, kc, kenc)
    assert got.keys() == want.keys()
    for k in want:
        n = n_edges(k[2])
        assert got[k][:n].tobytes() == want[k][:n].tobytes()

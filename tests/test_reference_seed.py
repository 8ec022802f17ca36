"""reference-seed mode (rp_rng_kind RP_RNG_REFERENCE): the sampled branches come from the reference's own chain,
DefaultHasher(t, info, tree id) -> SmallRng -> one draw (crates/mccfr/src/strategy/flow.rs:285-295, sample/external.rs:41-64,
sample/mod.rs:68-82, sample/pluribus.rs:91).  CPU part: the built-in games' Hash streams against an independent encoding of
the info names; the oracle's sampled branches against the Python restatement of tests/test_refrng.py; convergence."""
import ctypes as C

import numpy as np
import pytest

from robopoker_amd import _lib
from robopoker_amd.games import Game
import oracle
from test_refrng import py_draw_f32, py_draw_range, py_draw_weight, py_siphash


def isz(v):
    return int(v).to_bytes(8, "little")


def stream_bytes(hs) -> bytes:
    return bytes(hs.bytes[: hs.len])


# -- #[derive(Hash)] of the three info types, written from the Rust declarations, independent of games.cpp -------------------
HIST = {"": 0, "X": 1, "B": 2, "XB": 3}          # kuhn/src/info.rs:7-12 History::{Open, Check, Bet, CheckBet}
RANK = {"J": 0, "Q": 1, "K": 2}                   # card.rs Rank::{J, Q, K}
SPOT = {"": 0, "X": 1, "R": 2, "XR": 3}          # leduc/src/game.rs:7-12 Spot::{Open, Checked, Raised, CheckRaised}


def kuhn_stream(name: str) -> bytes:
    rank, hist = name.split("|")                  # Composite{public: KuhnPublic{acting: bool, node: History}, secret: Rank}
    return bytes([1]) + isz(HIST[hist]) + isz(RANK[rank])


def leduc_stream(name: str) -> bytes:
    parts = name.split("|")
    if len(parts) == 2:                           # round 1: "Q|XR"
        rank, h1 = parts
        return bytes([1]) + isz(0) + isz(SPOT[h1]) + isz(0) + isz(RANK[rank])
    rank, board, hist = parts                     # round 2: "Q|K|XRC" + round-2 edges
    # LeducPublic::subgame (info.rs:37-68): r1's edges, then the closing X / C, then r2's edges
    for h1 in ("XR", "X", "R", ""):
        closing = {"": None, "X": "X", "R": "C", "XR": "C"}[h1]
        if closing is None:
            continue
        if hist.startswith(h1 + closing) and hist[len(h1) + 1:] in SPOT:
            r1, r2 = SPOT[h1], SPOT[hist[len(h1) + 1:]]
            break
    else:
        raise AssertionError(name)
    return bytes([1]) + isz(1) + isz(RANK[board]) + isz(r1) + isz(1) + isz(r2) + isz(RANK[rank])


def names(game):
    lib = _lib.load()
    buf = C.create_string_buffer(64)
    out = []
    for i in range(game.n_infos):
        _lib.check(lib.rp_game_info_name(game._h, i, buf, 64))
        out.append(buf.value.decode())
    return out


def test_kuhn_streams_follow_derive_hash():
    g = Game("kuhn")
    st = g.hash_streams()
    assert st.n_infos == 12 and st.n_chance == 0
    for i, name in enumerate(names(g)):
        assert stream_bytes(st.infos[i]) == kuhn_stream(name), name
        assert st.infos[i].len == 17


def test_leduc_streams_follow_derive_hash():
    g = Game("leduc")
    st = g.hash_streams()
    assert st.n_infos == g.n_infos
    seen = set()
    for i, name in enumerate(names(g)):
        b = leduc_stream(name)
        assert stream_bytes(st.infos[i]) == b, name
        seen.add(b)
    assert len(seen) == g.n_infos  # distinct infosets hash distinct streams
    # the board-deal chance nodes: acting = false, board None, r1 in {Checked, Raised, CheckRaised}, r2 Some(Open), actor 0's rank
    want = {bytes([0]) + isz(0) + isz(r1) + isz(1) + isz(0) + isz(r) for r1 in (1, 2, 3) for r in (0, 1, 2)}
    assert {stream_bytes(st.chance[i]) for i in range(st.n_chance)} == want
    # every in-tree chance state names one of them, the two root-deal states none
    t = g.table
    roots = {t.train_root} | {t.children[t.states[t.train_root].offset + k] for k in range(t.states[t.train_root].n_children)}
    for s in range(t.n_states):
        stt = t.states[s]
        if stt.turn == 254:
            assert (stt.chance_info == 0) == (s in roots)
            assert stt.chance_info <= st.n_chance


def test_rps_streams():
    g = Game("rps")
    st = g.hash_streams()
    assert [stream_bytes(st.infos[i]) for i in range(2)] == [isz(0), isz(1)]  # RpsTurn::{P1, P2}


# -- the oracle's sampled branches against the Python restatement -------------------------------------------------------------
def node_seed(t, stream: bytes, tree):
    return py_siphash(0, 0, isz(t) + stream + isz(tree), 1, 3)


def test_leduc_board_deal_is_random_range_of_the_reference_chain():
    """external sampling, walker = player 0 at epoch 0: a tree's round-2 walker infosets reveal which board card the Deal node
    drew; it must be deals()[random_range(0..4)] with the generator seeded from DefaultHasher(t, chance info, tree)"""
    g = Game("leduc")
    st = g.hash_streams()
    o = oracle.OracleSolver(g, "linear", "linear", "external", batch=64, seed=5)
    o.set_rng("reference")
    nm = names(g)
    t = g.table
    checked = 0
    for epoch in range(4):
        dec = o.batch()
        by_tree = {}
        for d in dec:
            by_tree.setdefault(d["tree"], []).append(nm[d["info"]])
        for tree, infos in by_tree.items():
            r2 = [n for n in infos if n.count("|") == 2]
            r1 = [n for n in infos if n.count("|") == 1]
            for n2 in r2:
                rank, board, hist = n2.split("|")
                # recover r1 from the history and the chance stream of this Deal node; the walker's own rank = actor 0's rank
                # only when the walker is player 0
                if epoch % 2 != 0:
                    continue
                for h1, closing in (("XR", "C"), ("X", "X"), ("R", "C")):
                    if hist.startswith(h1 + closing) and hist[len(h1) + 1:] in SPOT:
                        r1s = SPOT[h1]
                        break
                stream = bytes([0]) + isz(0) + isz(r1s) + isz(1) + isz(0) + isz(RANK[rank])
                pick = py_draw_range(node_seed(epoch, stream, tree), 4)
                # deals(): Card::ALL minus the two hole cards; the drawn card's rank must be the infoset's board — the hole
                # cards are not visible here, so check the weaker, still discriminating statement: SOME deal consistent with
                # the draw exists (the rank of the pick-th remaining card for some opponent card)
                ok = False
                for c0 in (2 * RANK[rank], 2 * RANK[rank] + 1):
                    for c1 in range(6):
                        if c1 == c0:
                            continue
                        rest = [c for c in range(6) if c not in (c0, c1)]
                        ok = ok or rest[pick] // 2 == RANK[board]
                assert ok, (epoch, tree, n2, pick)
                checked += 1
        o.step()
    assert checked > 50


def test_reference_mode_changes_the_draws_and_is_deterministic():
    g = Game("leduc")
    a = oracle.OracleSolver(g, "linear", "linear", "external", batch=32, seed=3)
    b = oracle.OracleSolver(g, "linear", "linear", "external", batch=32, seed=3)
    c = oracle.OracleSolver(g, "linear", "linear", "external", batch=32, seed=3)
    b.set_rng("reference")
    c.set_rng("reference")
    for _ in range(5):
        a.step(), b.step(), c.step()
    assert np.array_equal(b.export(), c.export())
    assert not np.array_equal(a.export(), b.export())


def test_opponent_draw_is_weighted_index_over_the_sampling_distribution():
    """Kuhn, walker 0 at epoch 0, tables at their defaults: the opponent's sampling distribution is uniform over two actions, so
    WeightedIndex picks action 1 iff cum[0] <= x with x from the reference chain; the tree's later walker infoset (XB) exists
    only when the opponent bet after a check"""
    g = Game("kuhn")
    o = oracle.OracleSolver(g, "linear", "linear", "external", batch=256, seed=11)
    o.set_rng("reference")
    nm = names(g)
    dec = o.batch()
    by_tree = {}
    for d in dec:
        by_tree.setdefault(d["tree"], []).append(nm[d["info"]])
    hits = 0
    for tree, infos in by_tree.items():
        root = [n for n in infos if n.endswith("|")][0]
        # the opponent (player 1) after the walker's check holds some rank r1 != (same card); its info is "r1|X"
        saw_xb = any(n.endswith("|XB") for n in infos)
        bets = set()
        for r1 in "JQK":
            stream = kuhn_stream(r1 + "|X")
            # q = (0.5, 0.5) at default tables: weights max(q, EPS) -> cum[0] = 0.5, total = 1.0
            x = py_draw_weight(node_seed(0, stream, tree), 1.0)
            bets.add(bool(np.float32(0.5) <= x))
        if len(bets) == 1:  # the same answer whatever the opponent holds: the tree must agree
            assert saw_xb == bets.pop(), (tree, infos)
            hits += 1
    assert hits > 20


@pytest.mark.parametrize("sampling", ["external", "pluribus"])
def test_kuhn_converges_in_reference_mode(sampling):
    g = Game("kuhn")
    o = oracle.OracleSolver(g, "linear", "linear", sampling, batch=1, seed=1)
    o.set_rng("reference")
    o.solve(1 << 15)
    assert o.exploitability() < 0.05


def test_leduc_converges_in_reference_mode():
    g = Game("leduc")
    o = oracle.OracleSolver(g, "linear", "linear", "external", batch=16, seed=1)
    o.set_rng("reference")
    o.solve(1 << 16)
    assert o.exploitability() < 0.5


def test_pluribus_coin_is_random_f32():
    """the exploration coin at a walker node (pluribus.rs:91): random::<f32>() < explore keeps every branch.  Kuhn, walker =
    player 1 (epoch 1), a threshold no regret passes: at "r|X" the edge Check ends the hand (kept, pluribus.rs:95) and Bet does
    not (pruned) — so the Decision's expanded edges are {Check, Bet} iff the coin says explore, {Check} otherwise"""
    g = Game("kuhn")
    hp = oracle.default_hyper()
    hp.prune_warmup = 0
    hp.prune_explore = 0.5
    hp.prune_threshold = 1e9
    o = oracle.OracleSolver(g, "linear", "linear", "pluribus", batch=256, seed=2, hyper=hp)
    o.set_rng("reference")
    o.step()
    nm = names(g)
    seen = {True: 0, False: 0}
    for d in o.batch():
        name = nm[d["info"]]
        if not name.endswith("|X"):
            continue
        explore = bool(py_draw_f32(node_seed(1, kuhn_stream(name), d["tree"])) < np.float32(0.5))
        assert d["expanded"] == (3 if explore else 1), (name, d)
        seen[explore] += 1
    assert seen[True] > 20 and seen[False] > 20


# -- k-means++ (crates/lloyd/src/layer.rs:155-178): one SmallRng seeded from DefaultHasher(street), WeightedIndex<f32> per pick ----
def test_kmeanspp_picks_follow_weighted_index_f32():
    from lloyd_fixtures import turn_like_points
    from test_refrng import PyXoshiro

    N, K = 3000, 12
    pts = turn_like_points(N, bins=101, mass=46, seed=4)
    km = oracle.OracleKmeans(K, pts, "variation", seed=99)
    km.set_rng("reference", street=2)
    chosen = km.init_centroids()
    # independent restatement: potentials start at 1, f32 running sums, x = value0_1 * scale, partition_point; the potentials of
    # later rounds are min(d^2, old) with d = equity variation — recomputed here from the oracle's own distance entry point
    rng = PyXoshiro.seed_from_u64(py_siphash(0, 0, isz(2), 1, 3))
    pot = np.ones(N, dtype=np.float32)
    for k in range(K):
        cum = np.add.accumulate(pot, dtype=np.float32)  # sequential f32 running sums
        total = cum[-1]
        scale = total
        while np.float32(np.float32(scale * np.float32(1 - 2.0 ** -23)) + np.float32(0)) >= total:
            scale = np.nextafter(scale, np.float32(0), dtype=np.float32)
        v12 = np.frombuffer(np.uint32((rng.next_u32() >> 9) | 0x3F800000).tobytes(), np.float32)[0]
        x = np.float32(np.float32(v12 - np.float32(1)) * scale)
        pick = int(np.searchsorted(cum[:-1], x, side="right"))  # partition_point(w <= x)
        assert pick == int(chosen[k]), (k, pick, int(chosen[k]))
        pot[pick] = 0
        d = np.array([oracle.equity_variation(pts[pick], pts[i]) for i in range(N)], dtype=np.float32)
        pot = np.minimum(d * d, pot).astype(np.float32)
    assert len(set(int(c) for c in chosen)) == K
    # another street, another stream; the counter mode is something else again
    km2 = oracle.OracleKmeans(K, pts, "variation", seed=99)
    km2.set_rng("reference", street=1)
    assert not np.array_equal(km2.init_centroids(), chosen)

"""The built-in game tables (robopoker_amd/csrc/games.cpp, consumed by the device AND by the oracle) against an independent Python
model of the reference's rules, state by state: device <-> oracle parity must not rest on a shared table builder.

Written from crates/kuhn/src/{game.rs:28-168, info.rs:4-76, card.rs}, crates/leduc/src/{game.rs:7-245, info.rs:11-94, card.rs},
crates/roshambo/src/game.rs:7-78 — turn(), apply(), payoff(), choices(), deals(), the info names — not from games.cpp."""
import ctypes as C

import pytest

from robopoker_amd import Game, _lib

CHANCE, TERMINAL = 254, 255
RANKS = "JQK"


def rank(card):  # Card::ALL = J♠ J♥ Q♠ Q♥ K♠ K♥ (card.rs): rank = index / 2
    return card // 2


# ------------------------------------------------------------------------------------------------------------ Kuhn
class Kuhn:
    # Node::{Start, Dealt, Open, Check, Bet, CheckBet, Over} (game.rs:7-15)
    def __init__(self, hole=(None, None), node="Start", out=None):
        self.hole, self.node, self.out = hole, node, out

    def turn(self):  # game.rs:125-132
        if self.node in ("Start", "Dealt"):
            return CHANCE
        if self.node == "Over":
            return TERMINAL
        return 0 if self.node in ("Open", "CheckBet") else 1

    def children(self):
        if self.node == "Start":
            return [Kuhn((c, None), "Dealt") for c in range(6)]
        if self.node == "Dealt":
            return [Kuhn((self.hole[0], c), "Open") for c in range(6) if c != self.hole[0]]
        h = self.hole
        if self.node == "Open":  # choices [Check, Bet] (info.rs:36-42); apply game.rs:131-151
            return [Kuhn(h, "Check"), Kuhn(h, "Bet")]
        if self.node == "Check":
            return [Kuhn(h, "Over", ("show", False)), Kuhn(h, "CheckBet")]
        if self.node == "Bet":  # [Fold, Call]
            return [Kuhn(h, "Over", ("fold", 1)), Kuhn(h, "Over", ("show", True))]
        if self.node == "CheckBet":
            return [Kuhn(h, "Over", ("fold", 0)), Kuhn(h, "Over", ("show", True))]
        return []

    def payoff(self, p):  # Outcome::payoff (game.rs:35-64)
        kind, arg = self.out
        if kind == "fold":
            return -1.0 if arg == p else 1.0
        stake = 2.0 if arg else 1.0
        r = [rank(c) for c in self.hole]
        if r[0] == r[1]:
            return 0.0
        win = 0 if r[0] > r[1] else 1
        return stake if win == p else -stake

    def info_name(self):  # "{rank}|{history}" (info.rs:58-66)
        hist = {"Open": "", "Check": "X", "Bet": "B", "CheckBet": "XB"}[self.node]
        return RANKS[rank(self.hole[self.turn()])] + "|" + hist


# ------------------------------------------------------------------------------------------------------------ Leduc
RAISED = {"Raised", "CheckRaised"}
SP = {"Open": "", "Checked": "X", "Raised": "R", "CheckRaised": "XR"}


def spot_actor(s):  # Spot::actor (game.rs)
    return 0 if s in ("Open", "CheckRaised") else 1


class Leduc:
    def __init__(self, hole=(None, None), node=("Start",)):
        self.hole, self.node = hole, node

    def turn(self):  # game.rs:195-201
        k = self.node[0]
        if k in ("Start", "Dealt", "Deal"):
            return CHANCE
        if k == "Over":
            return TERMINAL
        return spot_actor(self.node[-1])

    def deals(self):  # game.rs:152-161
        taken = {c for c in self.hole if c is not None}
        return [c for c in range(6) if c not in taken]

    def children(self):
        k, h = self.node[0], self.hole
        if k == "Start":
            return [Leduc((c, None), ("Dealt",)) for c in range(6)]
        if k == "Dealt":
            return [Leduc((h[0], c), ("R1", "Open")) for c in self.deals()]
        if k == "Deal":
            return [Leduc(h, ("R2", c, self.node[1], "Open")) for c in self.deals()]
        if k == "R1":  # apply, game.rs:203-223
            s = self.node[1]
            return {
                "Open": [Leduc(h, ("R1", "Checked")), Leduc(h, ("R1", "Raised"))],
                "Checked": [Leduc(h, ("Deal", "Checked")), Leduc(h, ("R1", "CheckRaised"))],
                "Raised": [Leduc(h, ("Over", "FoldR1", 1)), Leduc(h, ("Deal", "Raised"))],
                "CheckRaised": [Leduc(h, ("Over", "FoldR1", 0)), Leduc(h, ("Deal", "CheckRaised"))],
            }[s]
        if k == "R2":
            _, c, r1, s = self.node
            return {
                "Open": [Leduc(h, ("R2", c, r1, "Checked")), Leduc(h, ("R2", c, r1, "Raised"))],
                "Checked": [Leduc(h, ("Over", "Showdown", c, r1, "Checked")), Leduc(h, ("R2", c, r1, "CheckRaised"))],
                "Raised": [Leduc(h, ("Over", "FoldR2", c, r1, 1)), Leduc(h, ("Over", "Showdown", c, r1, "Raised"))],
                "CheckRaised": [Leduc(h, ("Over", "FoldR2", c, r1, 0)), Leduc(h, ("Over", "Showdown", c, r1, "CheckRaised"))],
            }[s]
        return []

    def payoff(self, p):  # Outcome::{pot, payoff} (game.rs:57-110): ante 1, round-1 raise 2, round-2 raise 4
        o = self.node[1]
        if o == "FoldR1":
            who = self.node[2]
            pot = [1, 1]
            pot[1 - who] += 2
            return -float(pot[p]) if who == p else float(pot[who])
        if o == "FoldR2":
            _, _, c, r1, who = self.node
            base = 3 if r1 in RAISED else 1
            pot = [base, base]
            pot[1 - who] += 4
            return -float(pot[p]) if who == p else float(pot[who])
        _, _, c, r1, r2 = self.node
        each = (3 if r1 in RAISED else 1) + (4 if r2 in RAISED else 0)
        r = [rank(x) for x in self.hole]
        pair = [x == rank(c) for x in r]
        if pair[0] != pair[1]:
            win = 0 if pair[0] else 1
        elif r[0] == r[1]:
            return 0.0
        else:
            win = 0 if r[0] > r[1] else 1
        return float(each) if win == p else -float(each)

    def info_name(self):  # "{rank}|{board}|{subgame edges}" (info.rs:70-83, subgame :37-68)
        me = RANKS[rank(self.hole[self.turn()])]
        if self.node[0] == "R1":
            return me + "|" + SP[self.node[1]]
        _, c, r1, s = self.node
        closing = {"Checked": "X", "Raised": "C", "CheckRaised": "C"}[r1]
        return me + "|" + RANKS[rank(c)] + "|" + SP[r1] + closing + SP[s]


# ------------------------------------------------------------------------------------------------------------ RPS
class Rps:
    def __init__(self, moves=()):
        self.moves = moves

    def turn(self):
        return TERMINAL if len(self.moves) == 2 else len(self.moves)

    def children(self):
        return [] if len(self.moves) == 2 else [Rps(self.moves + (a,)) for a in range(3)]

    def payoff(self, p):  # game.rs:63-73: scissors wins count double
        a, b = self.moves
        pay = [[0, -1, 2], [1, 0, -2], [-2, 2, 0]][a][b]
        return float(pay if p == 0 else -pay)

    def info_name(self):
        return "P1" if len(self.moves) == 0 else "P2"


def info_names(game):
    lib = _lib.load()
    buf = C.create_string_buffer(64)
    out = []
    for i in range(game.n_infos):
        _lib.check(lib.rp_game_info_name(game._h, i, buf, 64))
        out.append(buf.value.decode())
    return out


@pytest.mark.parametrize("kind,model,n_infos", [("kuhn", Kuhn, 12), ("leduc", Leduc, 120), ("rps", Rps, 2)])
def test_table_equals_the_independent_model_state_by_state(kind, model, n_infos):
    g = Game(kind)
    t = g.table
    names = info_names(g)
    assert t.n_infos == n_infos and t.n_players == 2
    seen_states, info_of_name = set(), {}

    def walk(sid, m):
        assert sid not in seen_states, "the table is a tree: every state is reached once"
        seen_states.add(sid)
        st = t.states[sid]
        assert st.turn == m.turn(), (kind, sid)
        kids = m.children()
        assert st.n_children == len(kids), (kind, sid, m.__dict__)
        if st.turn == TERMINAL:
            for p in range(2):
                assert t.payoffs[st.offset * 2 + p] == m.payoff(p), (kind, sid, m.__dict__, p)
            assert t.payoffs[st.offset * 2] == -t.payoffs[st.offset * 2 + 1]
            return
        if st.turn != CHANCE:
            name = m.info_name()
            assert names[st.info] == name, (kind, sid, names[st.info], name)
            assert info_of_name.setdefault(name, st.info) == st.info
            assert t.info_actions[st.info] == len(kids) and t.info_player[st.info] == st.turn
        for k, c in enumerate(kids):  # children in choices() / deals() order
            walk(t.children[st.offset + k], c)

    walk(t.train_root, model())
    assert len(seen_states) == t.n_states
    assert len(info_of_name) == t.n_infos
    assert t.exploit_root == t.train_root


def test_leduc_infoset_count():
    # 3 ranks x 4 round-1 spots = 12 round-1 infosets, 3 ranks x 3 boards x 3 closed round-1 lines x 4 round-2 spots = 108
    g = Game("leduc")
    r1 = [n for n in info_names(g) if n.count("|") == 1]
    r2 = [n for n in info_names(g) if n.count("|") == 2]
    assert len(r1) == 12 and len(r2) == 108

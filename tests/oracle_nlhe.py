"""ctypes wrapper of the CPU oracle's no-limit hold'em rules engine (oracle/rp_oracle_nlhe.c).

TEST INFRASTRUCTURE ONLY.  Mirrors the surface of the reference's ``kicker::GameN<P>`` (crates/kicker/src/game.rs) so
the known-answer tests read like the reference's own.
"""
from __future__ import annotations

import ctypes as C
import random

import oracle

MAXP = 10
STACK, S_BLIND, B_BLIND = 200, 1, 2
DRAW, FOLD, CALL, CHECK, RAISE, SHOVE, BLIND = range(7)
TERMINAL, CHANCE = -2, -1
PRED = {n: i for i, n in enumerate(["must_stop", "must_deal", "must_post", "is_everyone_alright", "is_everyone_calling",
                                    "is_everyone_touched", "is_everyone_matched", "is_everyone_shoving", "is_everyone_folding",
                                    "may_fold", "may_call", "may_check", "may_raise", "may_shove", "is_showdown"])}
AMOUNT = {n: i for i, n in enumerate(["to_call", "to_post", "to_shove", "to_raise", "total", "effective"])}
BETTING, SHOVING, FOLDING = 0, 1, 2


class Seat(C.Structure):
    _fields_ = [("state", C.c_int32), ("stack", C.c_int16), ("stake", C.c_int16), ("spent", C.c_int16), ("cards", C.c_uint64)]


class GameStruct(C.Structure):
    _fields_ = [("n", C.c_int32), ("dealer", C.c_int32), ("ticker", C.c_int32), ("pot", C.c_int16), ("board", C.c_uint64),
                ("seats", Seat * MAXP)]


class ActionStruct(C.Structure):
    _fields_ = [("kind", C.c_int32), ("chips", C.c_int16), ("cards", C.c_uint64)]


_o = None


def lib():
    global _o
    if _o is None:
        o = oracle.load()
        G, A = C.POINTER(GameStruct), C.POINTER(ActionStruct)
        o.ora_nlhe_from_start.argtypes = [G, C.c_int, C.c_int, C.POINTER(C.c_int16), C.POINTER(C.c_uint64)]
        o.ora_nlhe_legal.restype = C.c_int
        o.ora_nlhe_legal.argtypes = [G, A]
        o.ora_nlhe_is_allowed.restype = C.c_int
        o.ora_nlhe_is_allowed.argtypes = [G, A]
        o.ora_nlhe_apply.restype = C.c_int
        o.ora_nlhe_apply.argtypes = [G, A]
        o.ora_nlhe_turn.restype = C.c_int
        o.ora_nlhe_turn.argtypes = [G]
        o.ora_nlhe_street.restype = C.c_int
        o.ora_nlhe_street.argtypes = [G]
        o.ora_nlhe_predicate.restype = C.c_int
        o.ora_nlhe_predicate.argtypes = [G, C.c_int]
        o.ora_nlhe_amount.restype = C.c_int
        o.ora_nlhe_amount.argtypes = [G, C.c_int]
        o.ora_nlhe_settlements.restype = C.c_int
        o.ora_nlhe_settlements.argtypes = [G, C.POINTER(C.c_int32)]
        o.ora_nlhe_continuation.restype = C.c_int
        o.ora_nlhe_continuation.argtypes = [G, C.POINTER(C.c_uint64)]
        o.ora_nlhe_snap.restype = ActionStruct
        o.ora_nlhe_snap.argtypes = [G, ActionStruct]
        o.ora_showdown_settle.argtypes = [C.c_int, C.POINTER(C.c_int16), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
        o.ora_edge_raises.restype = C.c_int
        o.ora_edge_raises.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint8)]
        o.ora_edge_into_chips.restype = C.c_int16
        o.ora_edge_into_chips.argtypes = [C.c_uint8, C.c_int16]
        o.ora_edge_to_u64.restype = C.c_uint64
        o.ora_edge_to_u64.argtypes = [C.c_uint8]
        o.ora_edge_from_u64.restype = C.c_uint8
        o.ora_edge_from_u64.argtypes = [C.c_uint64]
        o.ora_path_pack.restype = C.c_uint64
        o.ora_path_pack.argtypes = [C.POINTER(C.c_uint8), C.c_int]
        o.ora_path_unpack.restype = C.c_int
        o.ora_path_unpack.argtypes = [C.c_uint64, C.POINTER(C.c_uint8)]
        o.ora_path_length.restype = C.c_int
        o.ora_path_length.argtypes = [C.c_uint64]
        o.ora_path_aggression.restype = C.c_int
        o.ora_path_aggression.argtypes = [C.c_uint64]
        o.ora_nlhe_choices.restype = C.c_uint64
        o.ora_nlhe_choices.argtypes = [G, C.c_int]
        o.ora_nlhe_actionize.restype = ActionStruct
        o.ora_nlhe_actionize.argtypes = [G, C.c_uint8, C.c_uint64]
        o.ora_nlhe_edgify.restype = C.c_uint8
        o.ora_nlhe_edgify.argtypes = [G, A, C.c_int]
        o.ora_nlhe_apply_edge.restype = C.c_int
        o.ora_nlhe_apply_edge.argtypes = [G, C.c_uint8, C.POINTER(C.c_uint64)]
        o.ora_nlhe_payoff.restype = C.c_int
        o.ora_nlhe_payoff.argtypes = [G, C.c_int, C.POINTER(C.c_float)]
        o.ora_nlhe_info.argtypes = [G, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _o = o
    return _o


# ---- Edge codes (edge.rs:101-120) -----------------------------------------------------------------------------------
E_DRAW, E_FOLD, E_CHECK, E_CALL, E_SHOVE = 1, 2, 3, 4, 5
OPENS = [2, 3, 4, 5]
RAISES = [(1, 4), (1, 3), (1, 2), (2, 3), (3, 4), (1, 1), (5, 4), (3, 2), (2, 1), (3, 1)]


def Open(n):
    return 6 + OPENS.index(n)


def RaiseOdds(n, d):
    return 10 + RAISES.index((n, d))


def edge_raises(street, depth):
    out = (C.c_uint8 * 8)()
    k = lib().ora_edge_raises(street, depth, out)
    return list(out[:k])


def path_pack(edges):
    arr = (C.c_uint8 * max(len(edges), 1))(*edges)
    return lib().ora_path_pack(arr, len(edges))


def path_unpack(p):
    out = (C.c_uint8 * 13)()
    k = lib().ora_path_unpack(p, out)
    return list(out[:k])


def Action(kind, chips=0, cards=0):
    return (kind, int(chips), int(cards))


Fold, Check = Action(FOLD), Action(CHECK)
Call = lambda n: Action(CALL, n)  # noqa: E731
Raise = lambda n: Action(RAISE, n)  # noqa: E731
Shove = lambda n: Action(SHOVE, n)  # noqa: E731
Blind = lambda n: Action(BLIND, n)  # noqa: E731
Draw = lambda cards: Action(DRAW, 0, cards)  # noqa: E731


def _deal_holes(rng, n, taken=0):
    holes = []
    for _ in range(n):
        free = [c for c in range(52) if not taken >> c & 1]
        a, b = rng.sample(free, 2)
        holes.append(1 << a | 1 << b)
        taken |= holes[-1]
    return holes


class Game:
    """GameN<P> with value semantics: apply() returns a new game (game.rs:234-240)."""

    def __init__(self, s: GameStruct, rng):
        self._s, self._rng = s, rng

    @classmethod
    def root(cls, n=2, dealer=0, stacks=None, seed=0):
        rng = random.Random(seed)
        st = (C.c_int16 * n)(*(stacks or [STACK] * n))
        holes = (C.c_uint64 * n)(*_deal_holes(rng, n))
        s = GameStruct()
        lib().ora_nlhe_from_start(C.byref(s), n, dealer, st, holes)
        return cls(s, rng)

    def _copy(self):
        s = GameStruct()
        C.memmove(C.byref(s), C.byref(self._s), C.sizeof(GameStruct))
        return Game(s, self._rng)

    # state
    n = property(lambda self: self._s.n)
    dealer = property(lambda self: self._s.dealer)
    ticker = property(lambda self: self._s.ticker)
    pot = property(lambda self: self._s.pot)
    board = property(lambda self: self._s.board)
    street = property(lambda self: lib().ora_nlhe_street(C.byref(self._s)))
    turn = property(lambda self: lib().ora_nlhe_turn(C.byref(self._s)))
    actor_idx = property(lambda self: (self._s.dealer + self._s.ticker) % self._s.n)

    def seat(self, i):
        return self._s.seats[i]

    def __getattr__(self, name):
        if name in PRED:
            return bool(lib().ora_nlhe_predicate(C.byref(self._s), PRED[name]))
        if name in AMOUNT:
            return lib().ora_nlhe_amount(C.byref(self._s), AMOUNT[name])
        raise AttributeError(name)

    def deck(self):
        taken = self._s.board
        for i in range(self._s.n):
            taken |= self._s.seats[i].cards
        return [c for c in range(52) if not taken >> c & 1]

    def deal(self):
        """deck().deal(street): the next street's cards (random in the reference, seeded here)."""
        k = 3 if self.street == 0 else 1
        return sum(1 << c for c in self._rng.sample(self.deck(), k))

    def legal(self):
        out = (ActionStruct * 8)()
        k = lib().ora_nlhe_legal(C.byref(self._s), out)
        return [(out[i].kind, out[i].chips if out[i].kind not in (FOLD, CHECK) else 0, 0) for i in range(k)]

    def is_allowed(self, a):
        return bool(lib().ora_nlhe_is_allowed(C.byref(self._s), C.byref(ActionStruct(*a))))

    def apply(self, a):
        g = self._copy()
        if lib().ora_nlhe_apply(C.byref(g._s), C.byref(ActionStruct(*a))):
            raise ValueError(f"illegal action {a}")
        return g

    def settlements(self):
        r = (C.c_int32 * MAXP)()
        if lib().ora_nlhe_settlements(C.byref(self._s), r):
            raise ValueError("non terminal game state")
        return [(r[i], r[i] - self._s.seats[i].spent) for i in range(self._s.n)]  # (reward, won)

    def continuation(self):
        g = self._copy()
        holes = (C.c_uint64 * self._s.n)(*_deal_holes(self._rng, self._s.n))
        return g if lib().ora_nlhe_continuation(C.byref(g._s), holes) else None

    def snap(self, a):
        r = lib().ora_nlhe_snap(C.byref(self._s), ActionStruct(*a))
        return (r.kind, r.chips if r.kind not in (FOLD, CHECK) else 0, r.cards)

    def choices(self, depth):
        return path_unpack(lib().ora_nlhe_choices(C.byref(self._s), depth))

    def actionize(self, edge, cards=0):
        r = lib().ora_nlhe_actionize(C.byref(self._s), edge, cards)
        return (r.kind, r.chips if r.kind not in (FOLD, CHECK) else 0, r.cards)

    def edgify(self, a, depth):
        return lib().ora_nlhe_edgify(C.byref(self._s), C.byref(ActionStruct(*a)), depth)

    def apply_edge(self, edge):
        """NlheGame::apply (nlhe/src/game.rs:33-53): pending streets are dealt from the seeded deck."""
        g = self._copy()
        draws = (C.c_uint64 * 4)()
        probe = g._copy()
        for i in range(4):  # cards for every street that could be dealt on the way
            if probe.turn != CHANCE:
                break
            draws[i] = probe.deal()
            probe = probe.apply(Draw(draws[i]))
        if lib().ora_nlhe_apply_edge(C.byref(g._s), edge, draws) < 0:
            raise ValueError(f"edge {edge} not applicable")
        return g

    def payoff(self, seat):
        out = C.c_float()
        if lib().ora_nlhe_payoff(C.byref(self._s), seat, C.byref(out)):
            raise ValueError("non terminal game state")
        return out.value

    def info(self, history_edges):
        """(past, choices) of NlheInfo for the edges played since the root (nlhe/src/info.rs:72-103)."""
        past, choices = C.c_uint64(), C.c_uint64()
        lib().ora_nlhe_info(C.byref(self._s), path_pack(history_edges), C.byref(past), C.byref(choices))
        return path_unpack(past.value), path_unpack(choices.value)

    # the actions of game.rs:577-595
    raise_ = property(lambda self: Raise(self.to_raise))
    shove = property(lambda self: Shove(self.to_shove))
    calls = property(lambda self: Call(self.to_call))


def settle(rows):
    """Showdown::from(vec![(risked, state, strength key)]).settle() -> rewards (showdown.rs:36-52)."""
    n = len(rows)
    risked = (C.c_int16 * n)(*[r[0] for r in rows])
    status = (C.c_int32 * n)(*[r[1] for r in rows])
    strength = (C.c_uint32 * n)(*[r[2] for r in rows])
    out = (C.c_int32 * n)()
    lib().ora_showdown_settle(n, risked, status, strength, out)
    return list(out)

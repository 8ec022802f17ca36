"""Property tests (hypothesis) of the CPU oracles and the host-side formats: invariants the reference's design implies
but its unit tests only sample.  CPU only; bounded example counts keep the suite fast."""
import itertools
import os
import tempfile

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

import oracle_deuce as od
import oracle_nlhe as on
from robopoker_amd import formats

cards7 = st.lists(st.integers(0, 51), min_size=7, max_size=7, unique=True)
PERMS = list(itertools.permutations(range(4)))


def _mask(cs):
    return sum(1 << c for c in cs)


@settings(max_examples=60, deadline=None)
@given(cards7, st.integers(0, 23))
def test_strength_and_equity_do_not_depend_on_suit_names(cards, perm):
    # the evaluator and Observation::equity only ever ask "same suit?" (evaluator.rs:144-152): relabelling suits is free
    p = list(PERMS[perm])
    hand = _mask(cards)
    assert od.strength_key(hand) == od.strength_key(od.permute(p, hand))
    pocket, board = _mask(cards[:2]), _mask(cards[2:])
    assert od.river_equity(pocket, board) == od.river_equity(od.permute(p, pocket), od.permute(p, board))


@settings(max_examples=60, deadline=None)
@given(cards7)
def test_equity_counts_are_a_partition_of_the_990_holes(cards):
    pocket, board = _mask(cards[:2]), _mask(cards[2:])
    hero = od.strength_key(pocket | board)
    keys = [od.strength_key(h | board) for h in od.hand_iter(2, pocket | board)]
    assert len(keys) == 990
    e, won, total = od.river_equity(pocket, board)
    assert won == sum(k < hero for k in keys) and total - won == sum(k > hero for k in keys)
    assert e == (np.float32(0.5) if total == 0 else np.float32(won) / np.float32(total))


@settings(max_examples=40, deadline=None)
@given(st.lists(st.integers(0, 51), min_size=5, max_size=7, unique=True))
def test_adding_a_card_never_weakens_a_hand(cards):
    # a 6- or 7-card hand contains its 5-card sub-hands: its strength is at least theirs... except for the reference's
    # flush rule (top card only, evaluator.rs:109-115), under which it is still monotone: more cards only add candidates
    full = od.strength_key(_mask(cards))
    for drop in range(len(cards)):
        if len(cards) > 5:
            assert od.strength_key(_mask(cards[:drop] + cards[drop + 1:])) <= full


@settings(max_examples=40, deadline=None)
@given(st.integers(2, 6), st.integers(0, 2 ** 30), st.lists(st.integers(0, 1 << 30), min_size=1, max_size=60))
def test_nlhe_rules_keep_chips_and_offer_only_allowed_actions(n, seed, picks):
    g = on.Game.root(n, seed=seed)
    total = g.total
    for p in picks:
        if g.turn == on.TERMINAL:
            rewards = [r for r, _ in g.settlements()]
            assert sum(rewards) == g.pot and all(r >= 0 for r in rewards)
            break
        if g.turn == on.CHANCE:
            g = g.apply(on.Draw(g.deal()))
        else:
            opts = g.legal()
            assert opts and all(g.is_allowed(a) and g.snap(a) == a for a in opts)
            for e in g.choices(0):  # every abstract edge maps to an allowed concrete action (NlheGame::apply's path)
                assert g.is_allowed(g.snap(g.actionize(e)))
            g = g.apply(opts[p % len(opts)])
        assert g.total == total


@settings(max_examples=25, deadline=None)
@given(st.lists(st.tuples(st.integers(-2 ** 63, 2 ** 63 - 1), st.integers(-2 ** 15, 2 ** 15 - 1)), max_size=50))
def test_pgcopy_rows_round_trip(rows):
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "t.pgcopy")
        a = np.array([r[0] for r in rows], dtype=np.int64)
        b = np.array([r[1] for r in rows], dtype=np.int16)
        formats.write_rows(path, "qh", [a, b])
        a2, b2 = formats.read_rows(path, "qh")
        assert np.array_equal(a, a2) and np.array_equal(b, b2)
        assert os.path.getsize(path) == 19 + len(rows) * (2 + 4 + 8 + 4 + 2) + 2

"""Pins the NLHE rules oracle (oracle/rp_oracle_nlhe.c) to the reference's own unit tests: the action sequences and
expected values of crates/kicker/src/game.rs:960-1900 and crates/kicker/src/showdown.rs:96-262, cited per test.
Random deals of the reference (hole cards, boards) are seeded deals here; the assertions do not depend on the cards
unless stated.  CPU only."""
import pytest

import oracle_nlhe as on
from oracle_nlhe import BETTING, CHANCE, FOLDING, SHOVING, STACK, TERMINAL, Call, Check, Draw, Fold, Game, Raise, Shove

# strength keys in the order key: variant << 21 | rank << 17 (robopoker_amd ordering; showdown.rs:100-118)
ACE_HIGH, ONE_PAIR, TWO_PAIR, TRIPLETS, THE_NUTS = (0 << 21 | 12 << 17), (1 << 21 | 12 << 17), (2 << 21 | 12 << 17 | 11 << 13), \
    (3 << 21 | 12 << 17), (4 << 21 | 12 << 17)

SHOWDOWN_KATS = [  # showdown.rs:120-262: (risked, state, strength) rows -> rewards
    ([(100, BETTING, ACE_HIGH), (100, BETTING, ONE_PAIR)], [0, 200]),
    ([(50, FOLDING, THE_NUTS), (100, BETTING, TWO_PAIR), (75, FOLDING, THE_NUTS), (100, BETTING, ONE_PAIR)], [0, 325, 0, 0]),
    ([(100, BETTING, TWO_PAIR), (100, BETTING, TWO_PAIR), (100, BETTING, ONE_PAIR)], [150, 150, 0]),
    ([(200, BETTING, THE_NUTS), (150, SHOVING, TRIPLETS), (200, BETTING, TWO_PAIR), (100, SHOVING, ONE_PAIR), (50, FOLDING, THE_NUTS)],
     [700, 0, 0, 0, 0]),
    ([(150, SHOVING, THE_NUTS), (200, SHOVING, TRIPLETS), (350, SHOVING, ONE_PAIR), (50, SHOVING, ACE_HIGH)], [500, 100, 150, 0]),
    ([(50, SHOVING, THE_NUTS), (100, SHOVING, TRIPLETS), (150, BETTING, ONE_PAIR), (150, BETTING, ACE_HIGH)], [200, 150, 100, 0]),
    ([(50, SHOVING, TWO_PAIR), (100, BETTING, ONE_PAIR), (100, BETTING, ACE_HIGH)], [150, 100, 0]),
    ([(50, SHOVING, THE_NUTS), (100, BETTING, TWO_PAIR), (100, BETTING, TWO_PAIR)], [150, 50, 50]),
    ([(50, FOLDING, THE_NUTS), (100, BETTING, ACE_HIGH), (75, FOLDING, THE_NUTS), (25, FOLDING, THE_NUTS)], [0, 250, 0, 0]),
]


@pytest.mark.parametrize("rows,rewards", SHOWDOWN_KATS)
def test_showdown_known_answers(rows, rewards):
    assert on.settle(rows) == rewards


def next_hand(g):
    return g.apply(Fold).continuation()


def test_root():  # game.rs:964-971
    g = Game.root()
    assert g.street == 0 and g.seat(g.actor_idx).state == BETTING
    assert g.pot == 3 and g.turn == g.dealer


def test_everyone_folds_pref():  # game.rs:973-982
    g = Game.root().apply(Fold)
    assert g.is_everyone_folding and g.is_everyone_alright and not g.is_everyone_calling
    assert g.must_deal and g.must_stop  # "ambiguous" in the reference: both predicates hold


def test_everyone_folds_flop():  # game.rs:984-998
    g = Game.root().apply(Call(1)).apply(Check)
    g = g.apply(Draw(g.deal())).apply(Raise(10)).apply(Fold)
    assert g.is_everyone_folding and g.is_everyone_alright and not g.is_everyone_calling and g.must_deal and g.must_stop


def test_history_of_checks():  # game.rs:1000-1146: (street, pot, post, stop, deal, alright, calling, touched, matched)
    def state(g):
        return (g.street, g.pot, g.must_post, g.must_stop, g.must_deal, g.is_everyone_alright, g.is_everyone_calling,
                g.is_everyone_touched, g.is_everyone_matched)

    F, T = False, True
    g = Game.root()
    assert state(g) == (0, 3, F, F, F, F, F, F, F)
    g = g.apply(Call(1))
    assert state(g) == (0, 4, F, F, F, F, F, F, T)
    g = g.apply(Check)
    assert state(g) == (0, 4, F, F, T, T, T, T, T)
    g = g.apply(Draw(g.deal()))
    assert state(g) == (1, 4, F, F, F, F, F, F, T)
    g = g.apply(Check)
    assert state(g) == (1, 4, F, F, F, F, F, F, T)
    g = g.apply(Check)
    assert state(g) == (1, 4, F, F, T, T, T, T, T)
    g = g.apply(Draw(g.deal()))
    assert state(g) == (2, 4, F, F, F, F, F, F, T)
    g = g.apply(Check)
    assert state(g) == (2, 4, F, F, F, F, F, F, T)
    g = g.apply(Raise(4))
    assert state(g) == (2, 8, F, F, F, F, F, T, F)
    g = g.apply(Call(4))
    assert state(g) == (2, 12, F, F, T, T, T, T, T)
    g = g.apply(Draw(g.deal()))
    assert state(g) == (3, 12, F, F, F, F, F, F, T)
    g = g.apply(Check)
    assert state(g) == (3, 12, F, F, F, F, F, F, T)
    g = g.apply(Check)
    assert state(g) == (3, 12, F, T, F, T, T, T, T)


def test_next_after_fold():  # game.rs:1148-1160
    nxt = next_hand(Game.root())
    assert nxt.street == 0 and nxt.pot == 3 and nxt.board == 0 and nxt.dealer == 1 and nxt.turn == 1
    assert not nxt.is_everyone_touched


def test_dealer_rotation_and_ticker_reset():  # game.rs:1162-1186
    g0 = Game.root()
    g1 = next_hand(g0)
    g2 = next_hand(g1)
    g3 = next_hand(g2)
    assert [g.dealer for g in (g0, g1, g2, g3)] == [0, 1, 0, 1]
    assert g0.ticker == g1.ticker == g2.ticker == 2


def test_touched_with_rotated_dealer():  # game.rs:1188-1199
    g = next_hand(Game.root())
    assert g.dealer == 1 and not g.is_everyone_touched
    g = g.apply(Call(1))
    assert not g.is_everyone_touched
    g = g.apply(Check)
    assert g.is_everyone_touched and g.must_deal


def test_full_hand_rotated_dealer():  # game.rs:1202-1220
    g = next_hand(Game.root()).apply(Call(1)).apply(Check)
    assert g.must_deal
    g = g.apply(Draw(g.deal()))
    assert g.street == 1 and g.turn == 0 and not g.is_everyone_touched
    g = g.apply(Check).apply(Check)
    assert g.is_everyone_touched and g.must_deal


def test_hand_sequences():  # game.rs:1222-1234 five_hands_sequence, :1418-1426 ten_hands_alternation
    g = Game.root()
    for i in range(10):
        assert g.dealer == i % 2 and g.pot == 3 and g.street == 0 and not g.is_everyone_touched and g.turn == g.dealer
        g = next_hand(g)


def test_symmetric_preflop_action_and_flop_actor():  # game.rs:1236-1278
    for g, first_postflop in ((Game.root(), 1), (next_hand(Game.root()), 0)):
        g = g.apply(Call(1))
        assert not g.is_everyone_touched
        g = g.apply(Check)
        assert g.is_everyone_touched and g.must_deal
        assert g.apply(Draw(g.deal())).turn == first_postflop  # the non-dealer acts first after the flop


def test_allin_showdown_and_fold():  # game.rs:1280-1302
    g = Game.root()
    g = g.apply(Shove(g.to_shove))
    assert g.to_call == g.to_shove == STACK - 2  # "must use Shove not Call"
    g2 = g.apply(Shove(g.to_shove))
    assert g2.is_everyone_shoving and (g2.must_stop or g2.must_deal)
    g3 = g.apply(Fold)
    assert g3.must_stop and g3.is_everyone_folding


def test_raise_reraise():  # game.rs:1304-1316
    g = Game.root()
    g = g.apply(Raise(g.to_raise))
    g = g.apply(Raise(g.to_raise))
    assert not g.must_deal and not g.is_everyone_alright and g.turn == 0 and (g.may_raise or g.may_call)


def test_stacks_after_fold():  # game.rs:1318-1330
    s = Game.root().apply(Fold).settlements()
    assert s == [(0, -1), (3, 1)]  # (reward, won): the dealer loses the small blind


def test_stacks_after_flop_bet_fold():  # game.rs:1332-1349
    g = Game.root().apply(Call(1)).apply(Check)
    g = g.apply(Draw(g.deal()))
    g = g.apply(Raise(g.to_raise)).apply(Fold)
    assert g.must_stop
    s = g.settlements()
    assert s[0] == (0, -2) and s[1][0] > 0


def test_multi_hand_with_betting():  # game.rs:1351-1375
    g = Game.root().apply(Call(1)).apply(Check)
    g = g.apply(Draw(g.deal()))
    g = g.apply(Raise(g.to_raise)).apply(Fold).continuation()
    assert g.dealer == 1
    g = g.apply(Raise(g.to_raise))
    g = g.apply(Call(g.to_call))
    g = g.apply(Draw(g.deal()))
    g = g.apply(Raise(g.to_raise)).apply(Fold).continuation()
    assert g.dealer == 0 and g.pot == 3


def test_legal_options():  # game.rs:1377-1405
    g = Game.root()
    kinds = [a[0] for a in g.legal()]
    assert Fold in g.legal() and Call(1) in g.legal() and on.RAISE in kinds and on.SHOVE in kinds and Check not in g.legal()
    g = g.apply(Call(1))
    assert Check in g.legal() and Fold not in g.legal()
    g = g.apply(Check)
    g = g.apply(Draw(g.deal()))
    assert Check in g.legal() and on.RAISE in [a[0] for a in g.legal()] and Fold not in g.legal()
    # the order of the reference's options: raise, shove, call, fold, check (game.rs:253-287)
    assert [a[0] for a in Game.root().legal()] == [on.RAISE, on.SHOVE, on.CALL, on.FOLD]


def test_terminal_river_showdown():  # game.rs:1407-1420
    g = Game.root().apply(Call(1)).apply(Check)
    for _ in range(3):
        g = g.apply(Draw(g.deal())).apply(Check).apply(Check)
    assert g.street == 3 and g.must_stop and not g.must_deal and g.turn == TERMINAL


def test_min_raise_size_and_pot_tracking():  # game.rs:1432-1454
    g = Game.root()
    assert g.to_raise == 3
    assert g.apply(Raise(3)).to_raise == 4
    g = g.apply(Call(1))
    assert g.pot == 4
    g = g.apply(Raise(4))
    assert g.pot == 8
    assert g.apply(Call(4)).pot == 12


def test_bust_prevents_next():  # game.rs:1456-1481 (the reference's test goes on to check continuation() is None)
    g = Game.root()
    g = g.apply(Shove(g.to_shove))
    g = g.apply(Shove(g.to_shove))
    while not g.must_stop:
        assert g.turn == CHANCE
        g = g.apply(Draw(g.deal()))
    rewards = [r for r, _ in g.settlements()]
    assert sum(rewards) == 2 * STACK and (sorted(rewards) == [0, 2 * STACK] or rewards == [STACK, STACK])
    if 0 in rewards:
        assert g.continuation() is None


def test_actor_idx_wrapping():  # game.rs:1483-1493
    g = Game.root()
    assert g.actor_idx == 0
    g = g.apply(Call(1))
    assert g.actor_idx == 1
    g = g.apply(Check)
    assert (g.dealer + g.ticker) % g.n == 0


def test_snap():  # game.rs:1495-1546
    g = Game.root()
    for a in g.legal():
        assert g.snap(a) == a
    assert g.snap(Raise(32767)) == g.shove and g.snap(Raise(g.to_shove)) == g.shove
    assert g.snap(Raise(1)) == g.raise_ and g.snap(Raise(0)) == g.raise_
    limped = g.apply(Call(1))
    assert not limped.may_fold and limped.may_check and limped.snap(Fold) == Check
    assert not g.may_check and g.may_call and g.snap(Check) == g.calls


# ---- multiplayer (game.rs:1548-1900) -------------------------------------------------------------------------------
def test_multiplayer_roots():  # game.rs:1553-1571
    g3, g6 = Game.root(3), Game.root(6)
    assert g3.pot == 3 and g3.street == 0 and g3.n == 3 and g3.turn == g3.dealer
    assert g6.pot == 3 and g6.n == 6 and g6.turn == (g6.dealer + 3) % 6


def test_multiplayer_fold_to_terminal():  # game.rs:1573-1595
    g = Game.root(3).apply(Fold)
    assert not g.must_stop
    g = g.apply(Fold)
    assert g.must_stop and g.is_everyone_folding
    g = Game.root(6)
    for _ in range(5):
        assert not g.must_stop
        g = g.apply(Fold)
    assert g.must_stop and g.is_everyone_folding


def test_multiplayer_call_around():  # game.rs:1597-1633
    g = Game.root(3)
    g = g.apply(Call(g.to_call))
    assert not g.is_everyone_touched
    g = g.apply(Call(g.to_call))
    assert not g.is_everyone_touched
    g = g.apply(Check)
    assert g.is_everyone_touched and g.is_everyone_matched and g.must_deal and g.pot == 6
    g = Game.root(6)
    for _ in range(5):
        g = g.apply(Call(g.to_call))
    assert not g.is_everyone_touched
    g = g.apply(Check)
    assert g.is_everyone_touched and g.must_deal and g.pot == 12


def test_three_player_postflop_order_and_skip():  # game.rs:1635-1668
    g = Game.root(3)
    g = g.apply(Call(g.to_call))
    g = g.apply(Call(g.to_call)).apply(Check)
    g = g.apply(Draw(g.deal()))
    assert g.street == 1 and g.turn == (g.dealer + 1) % 3
    g = Game.root(3).apply(Fold)
    g = g.apply(Call(g.to_call)).apply(Check)
    assert g.must_deal
    g = g.apply(Draw(g.deal()))
    assert g.street == 1 and g.turn != g.dealer and g.seat(g.turn).state == BETTING


def test_multiplayer_dealer_rotation():  # game.rs:1670-1697
    g = Game.root(3)
    for want in (1, 2, 0):
        g = g.apply(Fold).apply(Fold).continuation()
        assert g.dealer == want
    g = Game.root(6)
    for i in range(6):
        assert g.dealer == i
        for _ in range(5):
            g = g.apply(Fold)
        g = g.continuation()
    assert g.dealer == 0


def test_three_player_raise_fold_full_hand_allin():  # game.rs:1699-1751
    g = Game.root(3)
    g = g.apply(Raise(g.to_raise)).apply(Fold)
    assert not g.must_stop
    g = g.apply(Call(g.to_call))
    assert g.is_everyone_touched and g.must_deal
    g = Game.root(3)
    g = g.apply(Call(g.to_call))
    g = g.apply(Call(g.to_call)).apply(Check)
    for _ in range(3):
        assert g.must_deal
        g = g.apply(Draw(g.deal())).apply(Check).apply(Check).apply(Check)
    assert g.street == 3 and g.must_stop
    g = Game.root(3)
    for _ in range(3):
        g = g.apply(Shove(g.to_shove))
    assert g.is_everyone_shoving and (g.must_stop or g.must_deal)


def test_three_player_chip_conservation():  # game.rs:1753-1775
    g = Game.root(3).apply(Fold).apply(Fold)
    assert g.must_stop and sum(r for r, _ in g.settlements()) == g.pot
    g = Game.root(3)
    initial = g.total
    g = g.apply(Call(g.to_call))
    assert g.total == initial
    g = g.apply(Call(g.to_call))
    assert g.total == initial
    assert g.apply(Raise(g.to_raise)).total == initial


def test_random_playouts_conserve_chips_and_stay_legal():
    # game.rs:1777-1800 three_player_legal_nonempty / six_player_multi_hand: every choice node has a legal action,
    # applying any of them keeps total chips constant, and a settled hand pays out exactly the pot
    import random
    rng = random.Random(5)
    for n in (2, 3, 6):
        g = Game.root(n, seed=n)
        total = g.total
        for _ in range(400):
            if g.turn == TERMINAL:
                assert sum(r for r, _ in g.settlements()) == g.pot
                nxt = g.continuation()
                if nxt is None:
                    g = Game.root(n, seed=rng.randrange(1 << 30))
                    total = g.total
                else:
                    g = nxt
                continue
            if g.turn == CHANCE:
                g = g.apply(Draw(g.deal()))
            else:
                opts = g.legal()
                assert opts and all(g.is_allowed(a) for a in opts)
                g = g.apply(rng.choice(opts))
            assert g.total == total


# ---- the action abstraction (edge.rs, size.rs, path.rs, game.rs:724-766) -------------------------------------------
def test_edge_codes_round_trip():  # edge.rs:284-302 bijective_u8 / bijective_u64, :325-332 backwards_compat_u64_bbs
    lib = on.lib()
    for e in range(1, 20):
        assert lib.ora_edge_from_u64(lib.ora_edge_to_u64(e)) == e
    assert lib.ora_edge_to_u64(on.Open(5)) == 6 | 5 << 3 and lib.ora_edge_to_u64(on.RaiseOdds(3, 2)) == 4 | 3 << 3 | 2 << 11
    assert lib.ora_edge_from_u64(4 | 1 << 19 | 5 << 3) == on.Open(5) and lib.ora_edge_from_u64(4 | 1 << 19 | 2 << 3) == on.Open(2)


def test_raise_grids():  # edge.rs:333-347, size.rs:113-138, pokerkit/src/lib.rs:138-151
    assert on.edge_raises(0, 0) == [on.Open(n) for n in (2, 3, 4, 5)]
    for street in (1, 2, 3):
        for depth in range(3):
            assert all(e >= 10 for e in on.edge_raises(street, depth))
    assert on.edge_raises(1, 0) == [on.RaiseOdds(*r) for r in ((1, 4), (1, 2), (3, 4), (1, 1), (2, 1))]
    assert on.edge_raises(0, 1) == [on.RaiseOdds(1, 1), on.RaiseOdds(2, 1)] and on.edge_raises(3, 7) == []
    assert on.edge_raises(2, 2) == on.edge_raises(2, 3) == [on.RaiseOdds(1, 1)]  # depth >= 2 collapses, > 3 is empty
    lib = on.lib()
    assert lib.ora_edge_into_chips(on.Open(3), 100) == 6  # big blinds, whatever the pot
    assert [lib.ora_edge_into_chips(on.RaiseOdds(n, d), 10) for n, d in on.RAISES] == [2, 3, 5, 6, 7, 10, 12, 15, 20, 30]


def test_path_packing():  # path.rs:178-262
    import random
    rng = random.Random(1)
    assert on.path_pack([]) == 0 and on.path_unpack(0) == []
    for n in range(13):
        edges = [rng.randrange(1, 20) for _ in range(n)]
        p = on.path_pack(edges)
        assert on.path_unpack(p) == edges and on.lib().ora_path_length(p) == n
    assert len(on.path_unpack(on.path_pack([2] * 20))) == 12  # MAX_PATH_EDGES
    D, R12, C_, X, R11, S, F = on.E_DRAW, on.RaiseOdds(1, 2), on.E_CALL, on.E_CHECK, on.RaiseOdds(1, 1), on.E_SHOVE, on.E_FOLD
    assert on.lib().ora_path_aggression(on.path_pack([D, R12, C_, C_, D, X, X, X, D, R11, S, F])) == 2  # path.rs:234-262
    assert on.lib().ora_path_aggression(on.path_pack([D, X, X, X])) == 0


def test_choices_actionize_edgify():  # game.rs:724-766,826-833 and the Snap cases of :1803-1812,1897-1906
    g = Game.root()
    # the dealer facing the big blind: four opens, then shove, call, fold (legal()'s order with the raise unfolded)
    assert g.choices(0) == [on.Open(2), on.Open(3), on.Open(4), on.Open(5), on.E_SHOVE, on.E_CALL, on.E_FOLD]
    assert g.choices(4) == [on.E_SHOVE, on.E_CALL, on.E_FOLD]  # past MAX_RAISE_REPEATS only the shove is aggressive
    assert g.actionize(on.Open(3)) == Raise(6) and g.actionize(on.E_CALL) == Call(1) and g.actionize(on.E_SHOVE) == Shove(STACK - 1)
    for chips, edge in ((4, on.Open(2)), (6, on.Open(3)), (8, on.Open(4)), (10, on.Open(5)), (16, on.Open(5)), (2, on.Open(2)),
                        (20, on.Open(5)), (7, on.Open(3))):  # 7 lies between two opens: the first of equal gaps
        assert g.edgify(Raise(chips), 0) == edge
    assert g.edgify(Fold, 0) == on.E_FOLD and g.edgify(Call(1), 0) == on.E_CALL and g.edgify(on.Blind(2), 0) == on.E_CALL
    assert g.edgify(Raise(50), 4) == on.E_SHOVE  # no raise grid at that depth (game.rs:829-832 unwrap_or(Shove))
    flop = g.apply(Call(1)).apply(Check)
    flop = flop.apply(Draw(flop.deal()))
    assert flop.choices(0) == [on.RaiseOdds(*r) for r in ((1, 4), (1, 2), (3, 4), (1, 1), (2, 1))] + [on.E_SHOVE, on.E_CHECK]
    assert flop.actionize(on.RaiseOdds(3, 4)) == Raise(3) and flop.edgify(Raise(4), 0) == on.RaiseOdds(1, 1)
    # every abstract choice is a legal concrete action after snapping to the rules (the trainer's apply path)
    for e in flop.choices(0):
        a = flop.snap(flop.actionize(e))
        assert flop.is_allowed(a)


# ---- the solver-facing NLHE types (nlhe/src/game.rs, info.rs) ------------------------------------------------------
def test_every_legal_raise_lands_on_a_trained_edge():  # nlhe/src/info.rs:153-198,250-263
    root = Game.root()
    flop = root.apply(Call(1)).apply(Check)
    flop = flop.apply(Draw(flop.deal()))
    turn = flop.apply(Check).apply(Check)
    turn = turn.apply(Draw(turn.deal()))
    for g in (root, flop, turn):
        trained = set(g.choices(0))
        for amount in range(g.to_raise, g.to_shove):
            assert g.edgify(Raise(amount), 0) in trained


def test_roundtrip_edgify_actionize():  # nlhe/src/info.rs:238-248
    g = Game.root()
    for a1 in g.legal():
        e1 = g.edgify(a1, 0)
        assert g.edgify(g.actionize(e1), 0) == e1


def test_aggression():  # nlhe/src/info.rs:200-224
    D, X, C_, S = on.E_DRAW, on.E_CHECK, on.E_CALL, on.E_SHOVE
    agg = lambda edges: on.lib().ora_path_aggression(on.path_pack(edges))  # noqa: E731
    assert agg([D, on.RaiseOdds(1, 1), C_, D, X, on.RaiseOdds(1, 2), S]) == 2
    assert agg([on.RaiseOdds(1, 1), on.RaiseOdds(1, 2), S]) == 3
    assert agg([X, C_, X]) == 0


def test_info_key_keeps_the_current_street_and_live_choices():  # nlhe/src/info.rs:72-103,265-297
    g = Game.root()
    history = []
    for e in (on.E_CALL, on.E_CHECK):
        g = g.apply_edge(e)
        history.append(e)
    assert g.turn == CHANCE
    g = g.apply_edge(on.RaiseOdds(1, 1))  # a choice edge at a chance node deals the flop first (game.rs:40-46)
    history += [on.E_DRAW, on.RaiseOdds(1, 1)]
    assert g.street == 1 and g.pot == 8
    past, choices = g.info(history)
    assert past == [on.RaiseOdds(1, 1)]  # only the choice edges since the last chance edge
    assert choices == g.choices(1)      # one raise so far on this street
    assert choices == [on.RaiseOdds(1, 2), on.RaiseOdds(1, 1), on.E_SHOVE, on.E_CALL, on.E_FOLD]
    # a Draw edge off a chance node changes nothing (game.rs:47-49); a terminal game ignores edges (game.rs:37-39)
    assert g.apply_edge(on.E_DRAW).pot == g.pot and g.apply_edge(on.E_DRAW).turn == g.turn
    done = g.apply_edge(on.E_FOLD)
    assert done.turn == TERMINAL and done.apply_edge(on.E_CALL).turn == TERMINAL
    assert done.payoff(0) == -done.payoff(1) and abs(done.payoff(0)) == 2.0  # the folder had put in two chips


def test_abstract_playouts_reach_terminal_states_that_settle():
    # the trainer's path end to end: choices -> apply_edge (deal / actionize / snap / apply) -> payoff; zero-sum heads up
    import random
    rng = random.Random(9)
    for trial in range(200):
        g, history = Game.root(seed=trial), []
        for _ in range(60):
            if g.turn == TERMINAL:
                break
            if g.turn == CHANCE:
                g = g.apply_edge(on.E_DRAW)
                history = (history + [on.E_DRAW])[-12:]
                continue
            past, choices = g.info(history)
            assert choices, "a choice node offers at least one edge"
            e = rng.choice(choices)
            g = g.apply_edge(e)
            history = (history + [e])[-12:]
        assert g.turn == TERMINAL
        assert g.payoff(0) + g.payoff(1) == 0.0

"""Pins the abstraction-input oracle (oracle/rp_oracle_deuce.c) to the reference's own known-answer tests.

Inputs and expected outputs below are the data of the reference's unit tests, cited by file:line; nothing here runs on
a GPU.
"""
import itertools
import random

import numpy as np
import pytest

import oracle_deuce as od

# deuce/src/evaluator.rs:176-372 — (hand, ranking variant, rank(s), kickers)
EVALUATOR_KATS = [
    ("As Kh Qd Jc 9s", "HighCard", "A", "", "KQJ9"),
    ("As Ah Kd Qc Js", "OnePair", "A", "", "KQJ"),
    ("As Ah Kd Kc Qs", "TwoPair", "A", "K", "Q"),
    ("As Ah Ad Kc Qs", "ThreeOAK", "A", "", "KQ"),
    ("Ts Jh Qd Kc As", "Straight", "A", "", ""),
    ("As Ks Qs Js 9s", "Flush", "A", "", ""),
    ("2s 2h 2d 3c 3s", "FullHouse", "2", "3", ""),
    ("As Ah Ad Ac Ks", "FourOAK", "A", "", "K"),
    ("Ts Js Qs Ks As", "StraightFlush", "A", "", ""),
    ("As 2h 3d 4c 5s", "Straight", "5", "", ""),
    ("As 2s 3s 4s 5s", "StraightFlush", "5", "", ""),
    ("As Ah Kd Kc Qs Jh 9d", "TwoPair", "A", "K", "Q"),
    ("4h 6h 7h 8h 9h Ts", "Flush", "9", "", ""),
    ("Kh Ah Ad As Ks Qs Js 9s", "FullHouse", "A", "K", ""),
    ("As Ah Ad Ac Ks Kh Qd", "FourOAK", "A", "", "K"),
    ("Ts Js Qs Ks As Ah Ad Ac", "StraightFlush", "A", "", ""),
    ("As 2s 3h 4d 5c 6s", "Straight", "6", "", ""),
    ("As Ah Kd Kc Qs Qh Jd", "TwoPair", "A", "K", "Q"),
    ("As Ah Ad Kc Ks Kh Qd", "FullHouse", "A", "K", ""),
]


@pytest.mark.parametrize("cards,variant,r1,r2,kicks", EVALUATOR_KATS)
def test_evaluator_known_answers(cards, variant, r1, r2, kicks):
    v, a, b, k = od.strength(od.hand(cards))
    assert v == variant
    assert a == od.RANKS.index(r1)
    assert b == (od.RANKS.index(r2) if r2 else 0)
    assert k == od.kick(kicks)


def test_strength_key_orders_like_the_derived_ord():
    # ranking.rs:17-29 (default build): variants compare in declaration order, then ranks, then kickers (strength.rs:6-10)
    order = ["As Kh Qd Jc 9s", "2s 2h Kd Qc Js", "2s 2h 3d 3c Qs", "2s 2h 2d Kc Qs", "As 2h 3d 4c 5s", "2s 2h 2d 3c 3s",
             "7s 5s 4s 3s 2s", "2s 2h 2d 2c 3s", "As 2s 3s 4s 5s"]
    keys = [od.strength_key(od.hand(h)) for h in order]
    assert keys == sorted(keys) and len(set(keys)) == len(keys)
    # kickers decide within a ranking
    assert od.strength_key(od.hand("As Ah Kd Qc Js")) > od.strength_key(od.hand("As Ah Kd Qc Ts"))
    # a flush is ranked by its top card only (evaluator.rs:109-115): these tie
    assert od.strength_key(od.hand("As Ks Qs Js 9s")) == od.strength_key(od.hand("As 8s 5s 3s 2s"))


def test_hand_iterator_known_answers():
    # hand_iter.rs:85-118
    assert len(list(od.hand_iter(0, 0))) == 0
    assert len(list(od.hand_iter(1, 0))) == 52
    assert len(list(od.hand_iter(2, 0))) == 1326
    assert len(list(od.hand_iter(0, 0xF))) == 0
    assert len(list(od.hand_iter(1, 0xF))) == 48
    assert len(list(od.hand_iter(2, 0xF))) == 1128
    # hand_iter.rs:120-133
    assert list(itertools.islice(od.hand_iter(3, 0), 10)) == [0b00111, 0b01011, 0b01101, 0b01110, 0b10011, 0b10101,
                                                              0b10110, 0b11001, 0b11010, 0b11100]
    # hand_iter.rs:135-150: three of the five cards {0, 3, 4, 5, 6}
    mask = ((1 << 52) - 1) ^ 0b1111001
    assert list(od.hand_iter(3, mask)) == [0b0011001, 0b0101001, 0b0110001, 0b0111000, 0b1001001, 0b1010001, 0b1011000,
                                           0b1100001, 0b1101000, 0b1110000]


def test_opponents_count():
    # observation.rs:289-296
    for n_board, count in [(5, 990), (4, 1035), (3, 1081), (0, 1225)]:
        taken = (1 << (2 + n_board)) - 1
        assert len(list(od.hand_iter(2, taken))) == count


def test_observation_i64_round_trip():
    # observation.rs:283-287
    rng = random.Random(7)
    for n_board in (0, 3, 4, 5):
        for _ in range(200):
            cards = rng.sample(range(52), 2 + n_board)
            pocket = sum(1 << c for c in cards[:2])
            public = sum(1 << c for c in cards[2:])
            v = od.obs_i64(pocket, public)
            assert od.obs_from_i64(v) == (pocket, public)
            assert v > 0


def test_permutation_known_answers():
    # permutation.rs:176-181 permute_simple: H -> S
    hearts, spades = 0x44444444, 0x88888888
    assert od.permute([2, 0, 3, 1], hearts) == spades
    # permutation.rs:196-202 permute_complex: [D, H, C, S]
    assert od.permute([1, 2, 0, 3], 0b1010_1010_1010_1010_0100_0100_0100_0100) == 0b1100_1100_1100_1100_0001_0001_0001_0001
    # permutation.rs:204-210 permute_rotation: [S, C, D, H]
    assert od.permute([3, 0, 1, 2], od.hand("Ac Kd Qh Js")) == od.hand("As Kc Qd Jh")
    # permutation.rs:212-219 permute_interior: [C, H, D, S]
    assert od.permute([0, 2, 1, 3], od.hand("2c 3d 4h 5s")) == od.hand("2c 3h 4d 5s")
    # permutation.rs:183-194 permute_unique: the 24 images of a four-suit hand are distinct
    images = {od.permute(list(p), od.hand("Ac Kd Qh Js")) for p in itertools.permutations(range(4))}
    assert len(images) == 24


ISOMORPHISM_KATS = [  # isomorphism.rs:83-222: pairs of observations in one class
    ("2s Ks~2d 5h 8c Tc Th", "2s Ks~2h 5c 8d Tc Td"),
    ("Ac Ad~Jc Ts 5s", "As Ah~Js Tc 5c"),
    ("Td As~Ts Ks Kh", "Tc Ad~Td Kd Kh"),
    ("As Jh~Ks Js 2d", "Ah Jd~Kh Jh 2c"),
    ("As Qh~Ks Js 2s", "Ad Qh~Kd Jd 2d"),
    ("Ad Kd~Qd Jd Td", "As Ks~Qs Js Ts"),
    ("Ac Kc~Qs Js Ts", "As Ks~Qh Jh Th"),
    ("Ac Ks~Qc Js Ts", "Ad Kh~Qd Jh Th"),
    ("Ac Kd~Qh Js 9c", "Ah Ks~Qc Jd 9h"),
]


@pytest.mark.parametrize("a,b", ISOMORPHISM_KATS)
def test_isomorphism_known_answers(a, b):
    ia, ib = od.isomorphism(*od.obs(a)), od.isomorphism(*od.obs(b))
    assert ia == ib
    assert od.is_canonical(*ia)


def test_isomorphism_is_invariant_under_every_suit_permutation():
    # isomorphism.rs:55-80 false_positives / false_negatives, on seeded observations of every street
    rng = random.Random(11)
    for n_board in (0, 3, 4, 5):
        for _ in range(100):
            cards = rng.sample(range(52), 2 + n_board)
            pocket = sum(1 << c for c in cards[:2])
            public = sum(1 << c for c in cards[2:])
            iso = od.isomorphism(pocket, public)
            images = [(od.permute(list(p), pocket), od.permute(list(p), public)) for p in itertools.permutations(range(4))]
            assert all(od.isomorphism(*im) == iso for im in images)
            assert any(tuple(od.permute(list(p), h) for h in iso) == (pocket, public) for p in itertools.permutations(range(4)))
            assert od.is_canonical(*iso)


def test_isomorphism_counts_preflop_and_flop():
    # street.rs:120-127: 169 and 1 286 792 (the turn and river counts are checked on the GPU: tests/test_gpu_deuce.py)
    assert od.isomorphisms("pref", count_only=True) == 169
    assert od.isomorphisms("flop", count_only=True) == 1_286_792


def test_isomorphism_iterator_order_and_range_split():
    whole = od.isomorphisms("flop", 0, 40)
    parts = np.concatenate([od.isomorphisms("flop", 0, 13), od.isomorphisms("flop", 13, 40)])
    assert np.array_equal(whole, parts)
    masks = [od.obs_from_i64(int(v)) for v in whole[:5000]]
    assert masks == sorted(masks)  # pockets ascending, then boards ascending (observation_iter.rs:43-53)
    assert masks[0] == (0b11, od.hand("2h 3c 3d")) or od.is_canonical(*masks[0])
    assert all(od.is_canonical(*m) for m in masks)


def test_river_equity_known_cases():
    # the nuts: a royal flush on board plus anything ties every opponent -> sum 0 -> 0.5 (observation.rs:59-62)
    e, won, total = od.river_equity(od.hand("2c 2d"), od.hand("Ts Js Qs Ks As"))
    assert (e, won, total) == (np.float32(0.5), 0, 0)
    # quad aces with the king kicker on board: only a tie or a loss to nothing
    e, won, total = od.river_equity(od.hand("Ah Ad"), od.hand("As Ac Kd 7h 2c"))
    assert won == total and e == np.float32(1.0)
    # the worst hand on a dry board loses to most holdings
    e, won, total = od.river_equity(od.hand("2c 3d"), od.hand("5h 7s 9c Jd Kh"))
    assert 0 <= won < total <= 990 and e == np.float32(won) / np.float32(total)
    # wins + losses + ties = 990 and suit symmetry
    a = od.river_equity(od.hand("As Kh"), od.hand("Qs Jd 4c 4h 9s"))
    b = od.river_equity(od.hand("Ah Ks"), od.hand("Qh Jc 4d 4s 9h"))
    assert a == b


def test_quantize_round_trip():
    # kicker/src/abstraction.rs:187-202: quantize(floatize(q)) == q for the 101 river buckets
    for q in range(101):
        assert od.quantize(np.float32(q) / np.float32(100)) == q
    assert od.quantize(0.005) == 1 and od.quantize(0.004) == 0 and od.quantize(1.0) == 100


def test_projection_matches_its_definition():
    # Lookup::future (lookup.rs:35-45): a turn observation's histogram over river buckets, through a table and directly
    rng = random.Random(3)
    turn = []
    for _ in range(3):
        cards = rng.sample(range(52), 6)
        turn.append(od.obs_i64(*od.isomorphism(sum(1 << c for c in cards[:2]), sum(1 << c for c in cards[2:]))))
    direct = od.project_river(turn)
    assert (direct.sum(axis=1) == 46).all()
    # a table holding just the children of these observations, in iterator order
    kids = set()
    for t in turn:
        po, pu = od.obs_from_i64(t)
        for r in od.hand_iter(1, po | pu):
            kids.add(od.isomorphism(po, pu | r))
    kids = sorted(kids)
    keys = np.array([od.obs_i64(*k) for k in kids], dtype=np.int64)
    abs_ = np.array([od.quantize(od.river_equity(*k)[0]) for k in kids], dtype=np.uint8)
    assert np.array_equal(od.project(turn, keys, abs_, 101), direct)

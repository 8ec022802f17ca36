"""MI355X parity of the device NLHE rules engine (robopoker_amd/csrc/nlhe.hip) against the CPU oracle's own rules
(oracle/rp_oracle_nlhe.c) on seeded random abstract hands: the digest folds EVERY intermediate state (pot, ticker,
dealer, board, every seat's stack / stake / spent / state), so one differing transition anywhere changes it."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle_nlhe as on  # noqa: E402


def _oracle(n, games, seed, max_steps):
    o = on.lib()
    o.ora_nlhe_playout.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint32)]
    pay = np.zeros((games, n), dtype=np.float32)
    dig = np.zeros(games, dtype=np.uint64)
    steps = np.zeros(games, dtype=np.uint32)
    for g in range(games):
        p, d, s = (C.c_float * n)(), C.c_uint64(), C.c_uint32()
        o.ora_nlhe_playout(n, g, seed, max_steps, p, C.byref(d), C.byref(s))
        pay[g], dig[g], steps[g] = list(p), d.value, s.value
    return pay, dig, steps


@pytest.mark.parametrize("n_players,games", [(2, 20000), (3, 8000), (6, 8000), (10, 4000)])
def test_device_playouts_match_the_oracle(n_players, games):
    from robopoker_amd import nlhe
    pay, dig, steps = nlhe.playouts(n_players, games, seed=1234)
    want_pay, want_dig, want_steps = _oracle(n_players, games, 1234, 200)
    assert np.array_equal(steps.cpu().numpy().view(np.uint32), want_steps)
    assert np.array_equal(dig.cpu().numpy().view(np.uint64), want_dig)
    assert np.array_equal(pay.cpu().numpy(), want_pay)
    assert (want_steps != 0xffffffff).all()                   # every hand reaches a terminal state
    assert (pay.sum(dim=1) == 0).all()                        # and settles zero-sum
    assert len(set(want_steps.tolist())) > 5                  # folds, showdowns, all-ins: many different lengths


def test_step_limit_and_bad_arguments():
    from robopoker_amd import _lib, nlhe
    _, _, steps = nlhe.playouts(6, 512, seed=5, max_steps=3)
    s = steps.cpu().numpy().view(np.uint32)
    assert (s == 0xffffffff).any() and (s[s != 0xffffffff] <= 3).all()
    with pytest.raises(_lib.RpError):
        nlhe.playouts(1, 8, seed=0)
    with pytest.raises(_lib.RpError):
        nlhe.playouts(11, 8, seed=0)

"""The NLHE rules engine on the device (include/rp_mi355x.h, rp_nlhe_playouts; robopoker_amd/csrc/nlhe.hip): random
abstract hands played by the device-side restatement of ``kicker::GameN`` / ``NlheGame::apply`` / ``Showdown::settle``."""
from __future__ import annotations

import torch  # before the first HIP call of librp_mi355x.so: one HIP runtime per process

from . import _lib


def playouts(n_players: int, n_games: int, seed: int, max_steps: int = 200, device: int = 0):
    """-> (payoffs float32[n_games][n_players], digests int64[n_games] (the u64 bit patterns), steps int32[n_games])."""
    dev = torch.device("cuda", device)
    pay = torch.empty((n_games, n_players), dtype=torch.float32, device=dev)
    dig = torch.empty(n_games, dtype=torch.int64, device=dev)
    steps = torch.empty(n_games, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    _lib.check(_lib.load().rp_nlhe_playouts(device, n_players, n_games, seed, max_steps, pay.data_ptr(), dig.data_ptr(), steps.data_ptr()))
    return pay, dig, steps

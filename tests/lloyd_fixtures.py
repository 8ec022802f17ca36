"""Seeded synthetic inputs of the lloyd path: the generators live in the package (robopoker_amd/fixtures.py, bench.py and
smoke() use them too); the tests import them under their historical name."""
from robopoker_amd.fixtures import *  # noqa: F401,F403
from robopoker_amd.fixtures import flop_hist, flop_like_points, flop_metric, random_metric, smooth_metric, tri_index, turn_like_points  # noqa: F401

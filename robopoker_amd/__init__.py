"""robopoker_amd — MI355X-native implementation of robopoker's two numeric hot paths.

* ``robopoker_amd.mccfr``  external-sampling MCCFR (crates/mccfr ``Solver`` surface)
* ``robopoker_amd.lloyd``  Elkan k-means over Sinkhorn EMD (crates/elkan + crates/lloyd surface)

Both are thin host mirrors over the C-ABI of ``librp_mi355x.so`` (include/rp_mi355x.h); the compute
runs in hand-written HIP kernels for gfx950.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .games import Game  # noqa: F401

__all__ = ["Game", "_lib"]
__version__ = "0.1.0"

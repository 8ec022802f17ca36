"""Swaps the handle of the ctypes binding for the emulated library — for THIS process, called from tests only
(tests/conftest.py under RP_EMUL=1, the rank processes of tests/test_emul_comm.py).  DESIGN.md §2b."""
from __future__ import annotations

import ctypes as C
import functools
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def load_emulated(cuda_means_host: bool = True, build: bool = True) -> str:
    """build=False: the library was built by the parent of this (rank) process; several ranks must not rebuild it at once"""
    import build as emul_build
    from robopoker_amd import _lib

    path = emul_build.build(jobs=os.cpu_count() or 4) if build else emul_build.LIB
    lib = C.CDLL(path)
    for name, (res, args) in _lib._SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib._lib = lib
    if cuda_means_host:
        _cuda_means_host()
    return path


def _cuda_means_host():
    """Under emulation the "device" is host memory: tensors asked for on "cuda" are made on the CPU (their data_ptr() is what the
    emulated library dereferences), .cuda() / .to("cuda") stay put, is_cuda answers True, synchronize does nothing."""
    import torch

    if getattr(torch, "_rp_emul_patched", False):
        return
    torch._rp_emul_patched = True

    def host(dev):
        if dev is None:
            return None
        d = torch.device(dev) if not isinstance(dev, torch.device) else dev
        return torch.device("cpu") if d.type == "cuda" else d

    def factory(fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            if "device" in k:
                k["device"] = host(k["device"])
            return fn(*a, **k)

        return wrapped

    for name in ("zeros", "empty", "ones", "full", "tensor", "as_tensor", "arange", "randint", "rand", "randn", "zeros_like",
                 "empty_like", "ones_like", "full_like", "frombuffer"):
        setattr(torch, name, factory(getattr(torch, name)))
    real_to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple(host(x) if isinstance(x, (str, torch.device)) else x for x in a)
        if "device" in k:
            k["device"] = host(k["device"])
        return real_to(self, *a, **k)

    torch.Tensor.to = to
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.is_available = lambda: True
    torch.cuda.current_device = lambda: 0
    torch.cuda.device_count = lambda: 1

    class _HostStream:  # torch.cuda.Stream(): the emulated library's streams are all the same synchronous one
        cuda_stream = 0

        def synchronize(self):
            pass

        def wait_stream(self, other):
            pass

    import contextlib

    torch.cuda.Stream = _HostStream
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.current_stream = lambda *a, **k: _HostStream()

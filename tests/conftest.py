import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU convergence runs")


def _build_once():
    """Build the product library and the oracle if their shared objects are missing (CPU-only build)."""
    lib = os.path.join(ROOT, "robopoker_amd", "librp_mi355x.so")
    ora = os.path.join(ROOT, "oracle", "_build", "librp_oracle.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "robopoker_amd", "csrc")])
    if not os.path.exists(ora):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


_build_once()


def _load_emulated():
    """RP_EMUL=1 (a developer switch, tests only): the `-m gpu` tests call the kernels' SOURCES compiled against the wave64
    execution model of tests/emul/ instead of the device library — a logic check for a machine without a GPU.  It replaces
    the handle the ctypes binding hands out, for this pytest process only; the product never does this."""
    import ctypes as C

    sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
    import build as emul_build
    from robopoker_amd import _lib

    lib = C.CDLL(emul_build.build())
    for name, (res, args) in _lib._SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib._lib = lib
    _cuda_means_host()


def _cuda_means_host():
    """Under RP_EMUL the "device" is host memory: tensors asked for on "cuda" are made on the CPU (their data_ptr() is what the
    emulated library dereferences), .cuda() / .to("cuda") stay put, is_cuda answers True, synchronize does nothing."""
    import functools

    import torch

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


if os.environ.get("RP_EMUL") == "1":
    _load_emulated()


def has_gpu() -> bool:
    from robopoker_amd import _lib

    return _lib.load().rp_device_count() > 0


@pytest.fixture(scope="session")
def gpu():
    if not has_gpu():
        pytest.fail("this test is marked gpu but no HIP device is visible (no CPU fallback exists)")
    return 0

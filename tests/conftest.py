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
    execution model of tests/emul/ instead of the device library — a logic check for a machine without a GPU.
    It replaces the handle the ctypes binding hands out, for this pytest process only; the product never does this."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
    import harness

    harness.load_emulated()


if os.environ.get("RP_EMUL") == "1":
    _load_emulated()


def pytest_terminal_summary(terminalreporter):
    """under RP_EMUL: how often the lanes of a wavefront were parked at more than one collective (the only situation in which the
    execution model has to decide an order)"""
    if os.environ.get("RP_EMUL") != "1":
        return
    import ctypes as C

    from robopoker_amd import _lib

    fn = _lib._lib.emu_split_rounds
    fn.restype = C.c_uint64
    terminalreporter.write_line(f"tests/emul: {fn()} rounds with lanes of one wavefront parked at different collectives")


def has_gpu() -> bool:
    from robopoker_amd import _lib

    return _lib.load().rp_device_count() > 0


@pytest.fixture(scope="session")
def gpu():
    if not has_gpu():
        pytest.fail("this test is marked gpu but no HIP device is visible (no CPU fallback exists)")
    return 0

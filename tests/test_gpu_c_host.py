"""GPU: a plain-C host program (tests/c/abi_smoke.c) solves Kuhn and trains the NLHE blueprint (the Flagship solver type through
rp_nlhe_train) through the C ABI — no Python between the caller and librp_mi355x.so, as a cgo / Rust-FFI binding would call it."""
import subprocess

import pytest

from test_abi import _build_c_host

pytestmark = pytest.mark.gpu


def test_c_host_solves_kuhn_on_the_gpu(gpu, tmp_path):
    exe = _build_c_host(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "exploitability" in r.stdout and "nlhe: epoch=6" in r.stdout and "training stopped" in r.stdout

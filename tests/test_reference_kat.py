"""tests/golden/reference_kat.json — known answers FOR the reference (INTEGRATION.md §4 holds the Rust test that checks them) — stays
equal to what the oracle computes, and the GPU path reproduces its Sinkhorn part in the glibc-arithmetic pass."""
import json
import os
import platform
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
KAT = os.path.join(ROOT, "tests", "golden", "reference_kat.json")


def test_the_committed_file_is_what_the_oracle_computes():
    import make_reference_kat

    assert json.load(open(KAT)) == json.loads(json.dumps(make_reference_kat.build()))


def test_the_fixture_keeps_the_tolerances_the_reference_asserts():
    # sinkhorn.rs:271-293: |S(h, h)| < 1e-4, |S(mu, nu) - S(nu, mu)| < 1e-3
    doc = json.load(open(KAT))
    f32 = lambda b: float(np.array([b], dtype=np.uint32).view(np.float32)[0])
    sk = doc["sinkhorn_fixture"]
    assert abs(f32(sk[0]["divergence_bits"])) < 1e-4
    assert abs(f32(sk[1]["divergence_bits"]) - f32(sk[2]["divergence_bits"])) < 1e-3 and f32(sk[1]["divergence_bits"]) > 0


@pytest.mark.skipif(platform.libc_ver()[0] != "glibc", reason="the vectors are those of a glibc host")
def test_the_sinkhorn_vectors_are_those_of_this_machines_libm():
    # mode 1 = the platform's own expf / logf: what a Rust build on this machine would call
    import ctypes as C

    import oracle
    from lloyd_fixtures import flop_hist, flop_metric

    o = oracle.load()
    o.ora_lloyd_set_libm.argtypes = [C.c_int]
    o.ora_lloyd_set_libm(1)
    try:
        tri = flop_metric()
        for case in json.load(open(KAT))["sinkhorn_fixture"]:
            a, b = flop_hist([tuple(e) for e in case["mu"]]), flop_hist([tuple(e) for e in case["nu"]])
            d = np.float32(oracle.sinkhorn_divergence(a, b, tri))
            c = np.float32(oracle.sinkhorn_cost(a, b, tri)[0])
            assert int(d.view(np.uint32)) == case["divergence_bits"] and int(c.view(np.uint32)) == case["cost_bits"]
    finally:
        o.ora_lloyd_set_libm(0)


@pytest.mark.gpu
def test_the_device_reproduces_the_sinkhorn_vectors_in_the_glibc_pass(gpu):
    from lloyd_fixtures import flop_hist, flop_metric
    from robopoker_amd import lloyd

    tri = flop_metric()
    cases = json.load(open(KAT))["sinkhorn_fixture"]
    mu = np.stack([flop_hist([tuple(e) for e in c["mu"]]) for c in cases])
    nu = np.stack([flop_hist([tuple(e) for e in c["nu"]]) for c in cases])
    lloyd.sinkhorn_set_libm("glibc")
    try:
        d = lloyd.sinkhorn_divergence(mu, nu, tri)
        c, it = lloyd.sinkhorn_cost(mu, nu, tri)
    finally:
        lloyd.sinkhorn_set_libm("contract")
    assert [int(x) for x in d.view(np.uint32)] == [k["divergence_bits"] for k in cases]
    assert [int(x) for x in c.view(np.uint32)] == [k["cost_bits"] for k in cases]
    assert [int(x) for x in it] == [k["iterations"] for k in cases]

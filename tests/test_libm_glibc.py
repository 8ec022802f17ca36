"""include/rp_libm_glibc.h: glibc's expf / logf (Arm Optimized Routines) restated, against THIS machine's libm over every float.

The reference's Sinkhorn calls f32::exp / f32::ln (crates/lloyd/src/sinkhorn.rs:115,120-127,136; phi.rs:36) = the platform's libm.
The restatement pins that boundary to the published algorithm the way tests/test_refrng.py pins the hash and the generator; the
oracle's mode 2 (ora_lloyd_set_libm) runs the Sinkhorn on it."""
import ctypes as C
import platform
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle

ROOT = Path(__file__).resolve().parent.parent


def _glibc_x86_fma() -> bool:
    if platform.machine() != "x86_64" or platform.libc_ver()[0] != "glibc":
        return False
    try:
        flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags"))
    except (OSError, StopIteration):
        return False
    return " fma " in flags + " "


needs_glibc = pytest.mark.skipif(not _glibc_x86_fma(), reason="the restatement is of glibc's x86-64 FMA variant")


def _ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def _sweep(lo, hi):
    o = oracle.load()
    f = o.ora_libm_glibc_sweep
    f.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    f.restype = None
    be, bl, first = C.c_uint64(), C.c_uint64(), (C.c_uint32 * 2)()
    f(lo, hi, be, bl, first)
    return be.value, bl.value, first[0], first[1]


@needs_glibc
def test_every_float_bit_pattern_equals_the_platform_libm():
    # all 2^32 inputs of both functions, NaNs, infinities, subnormals and the overflow / underflow edges included (~13 s on 8 cores)
    be, bl, fe, fl = _sweep(0, 1 << 32)
    assert (be, bl) == (0, 0), (be, bl, hex(fe), hex(fl))


def test_the_branch_free_forms_equal_the_ladder_forms_on_every_float():
    # rp_glibc_expf_tab / rp_glibc_exp_floor_tab / rp_glibc_logf_tab (what the device kernels evaluate, tables handed in) against
    # rp_glibc_expf / max(rp_glibc_expf, MIN_POSITIVE) / rp_glibc_logf over all 2^32 inputs (~40 s on 8 cores)
    o = oracle.load()
    f = o.ora_libm_glibc_tab_sweep
    f.argtypes, f.restype = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)], None
    bad, first = (C.c_uint64 * 3)(), (C.c_uint32 * 3)()
    f(0, 1 << 32, bad, first)
    assert list(bad) == [0, 0, 0], (list(bad), [hex(x) for x in first])


@needs_glibc
@pytest.mark.parametrize("y", [1.5, 0.5])
def test_powf_of_every_float_equals_the_platform_libm(y):
    # DiscountedRegret's exponent 1.5 (crates/mccfr/src/regret/discounted.rs:12,33) and 0.5 over EVERY x bit pattern: zeros (epoch 0:
    # powf(+0, 1.5) = +0), subnormals, the overflow range, infinities, NaNs, negative bases (invalid) — ~15 s each on 8 cores
    o = oracle.load()
    f = o.ora_libm_glibc_pow_sweep
    f.argtypes, f.restype = [C.c_uint64, C.c_uint64, C.c_float, C.POINTER(C.c_uint32)], C.c_uint64
    first = C.c_uint32()
    assert f(0, 1 << 32, y, first) == 0, hex(first.value)


def test_discounted_regret_at_epoch_zero_is_not_nan():
    # the first step runs at epoch 0: t^1.5 = powf(+0, 1.5) = +0, discount 0 / (0 + 1) = 0, so a positive accumulated regret (a default
    # bias, kicker/src/edge.rs:61-72) is forgotten: acc * 0 + imm.  (A restatement of powf without its zero case made this NaN.)
    o = oracle.load()
    f = o.ora_regret_accumulate
    f.argtypes, f.restype = [C.c_int, C.c_float, C.c_float, C.c_uint64], C.c_float
    assert f(2, 100.0, 0.5, 0) == 0.5 and f(2, -3.0, 0.5, 0) == 0.5 and f(2, 0.0, 0.5, 0) == 0.5
    g = o.ora_glibc_powf
    g.argtypes, g.restype = [C.c_float, C.c_float], C.c_float
    assert g(0.0, 1.5) == 0.0 and g(float("inf"), 1.5) == float("inf") and g(0.0, -1.5) == float("inf")
    assert g(-8.0, 3.0) == -512.0 and g(-8.0, 2.0) == 64.0 and np.isnan(g(-8.0, 0.5)) and g(-0.0, 3.0) == 0.0
    assert g(5.0, 0.0) == 1.0 and g(float("nan"), 0.0) == 1.0 and g(1.0, float("nan")) == 1.0 and np.isnan(g(2.0, float("nan")))


@needs_glibc
def test_powf_on_other_exponents_over_a_stride_of_the_floats():
    o = oracle.load()
    f = o.ora_glibc_powf
    f.argtypes, f.restype = [C.c_float, C.c_float], C.c_float
    libm = C.CDLL("libm.so.6")
    libm.powf.argtypes, libm.powf.restype = [C.c_float, C.c_float], C.c_float
    rng = np.random.default_rng(3)
    xs = rng.integers(0, 1 << 32, 30000, dtype=np.uint64).astype(np.uint32).view(np.float32)  # every kind of x, negative and NaN included
    ys = rng.uniform(-12, 12, 30000).astype(np.float32)
    ys[::7] = np.round(ys[::7])  # integer exponents: the sign of a negative base
    ys[::101] = rng.choice(np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, 2.0**24 + 2, 2.0**31], np.float32), len(ys[::101]))
    for x, y in zip(xs, ys):
        a, b = np.float32(f(float(x), float(y))), np.float32(libm.powf(float(x), float(y)))
        assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b)), (float(x), float(y))


def test_llvm_folds_pow_half_into_sqrt_and_leaves_pow_three_halves_a_libm_call():
    # f32::powf is the llvm.pow.f32 intrinsic (no errno); BETA = 0.5 and ALPHA = 1.5 are consts (discounted.rs:12-13) and the
    # workspace builds at opt-level 3 in dev and release: LLVM's libcall simplifier rewrites pow(x, 0.5) as fabs(sqrt(x)) (with
    # -inf -> +inf) WITHOUT any fast-math flag, and leaves pow(x, 1.5) to libm.  Checked with this image's LLVM on the same IR shape.
    clang = "/opt/rocm/lib/llvm/bin/clang"
    import os
    if not os.path.exists(clang):
        pytest.skip("no clang in this image")
    src = "float p05(float x){ return __builtin_powf(x, 0.5f); }\nfloat p15(float x){ return __builtin_powf(x, 1.5f); }\n"
    r = subprocess.run([clang, "--target=x86_64-unknown-linux-gnu", "-O2", "-fno-math-errno", "-S", "-x", "c", "-", "-o", "-"],
                       input=src, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    asm = r.stdout
    p05 = asm[asm.index("p05:") : asm.index("p15:")]
    p15 = asm[asm.index("p15:") :]
    assert "sqrtss" in p05 and "powf" not in p05
    assert "powf" in p15
    # ... and that is how the oracle and the library compute the two powers (include/rp_libm_glibc.h: rp_pow05 / rp_pow15)
    o = oracle.load()
    for f in (o.ora_pow05, o.ora_pow15, o.ora_glibc_powf):
        f.restype = C.c_float
    o.ora_pow05.argtypes = o.ora_pow15.argtypes = [C.c_float]
    o.ora_glibc_powf.argtypes = [C.c_float, C.c_float]
    t = np.arange(1, 1 << 14, dtype=np.float32)
    assert all(np.float32(o.ora_pow05(float(x))) == np.sqrt(x) for x in t)
    assert all(o.ora_pow15(float(x)) == o.ora_glibc_powf(float(x), 1.5) for x in t[::7])


def test_the_old_contract_differed_from_powf_on_a_quarter_of_the_epochs():
    # why t^1.5 moved from t * sqrt(t) to glibc's powf (round 4): over t = 1 .. 2^16 the two differ in the last bit for ~24 %
    # of the epochs (two roundings against one) — a Rust build on glibc follows powf there
    o = oracle.load()
    f = o.ora_glibc_powf
    f.argtypes, f.restype = [C.c_float, C.c_float], C.c_float
    t = np.arange(1, 1 << 16, dtype=np.float32)
    a = np.array([f(float(x), 1.5) for x in t], np.float32)
    b = t * np.sqrt(t)
    frac = float((a != b).mean())
    assert 0.1 < frac < 0.4, frac
    assert _ulps(a, b).max() <= 1
    # exact squares: t^1.5 is an integer and both agree with it while it fits 24 bits
    for q in (2, 3, 10, 100, 255):
        assert f(float(q * q), 1.5) == float(q**3) and f(float(q * q), 0.5) == float(q)


def test_known_values_and_edges():
    o = oracle.load()
    for f in (o.ora_glibc_expf, o.ora_glibc_logf):
        f.argtypes, f.restype = [C.c_float], C.c_float
    e, l = o.ora_glibc_expf, o.ora_glibc_logf
    assert e(0.0) == 1.0 and l(1.0) == 0.0
    assert e(float("-inf")) == 0.0 and e(float("inf")) == float("inf") and np.isnan(e(float("nan")))
    assert e(89.0) == float("inf") and e(-104.0) == 0.0
    assert l(0.0) == float("-inf") and l(float("inf")) == float("inf") and np.isnan(l(-1.0)) and np.isnan(l(float("nan")))
    # correctly rounded values computed in double (both functions are within 0.51 ulp by their error analysis; these are not ties)
    for x in (1.0, -1.0, 0.5, 10.0, -20.0, 1e-3, 87.0):
        assert abs(np.float32(e(x)) - np.float32(np.exp(np.float64(x)))) <= np.spacing(np.float32(np.exp(np.float64(x))))
    for x in (2.0, 0.5, 1e-30, 3e38, 1.0000001, 1e-40):
        assert abs(np.float32(l(x)) - np.float32(np.log(np.float64(np.float32(x))))) <= np.spacing(abs(np.float32(np.log(np.float64(np.float32(x))))))


def _both(xs):
    o = oracle.load()
    n = len(xs)
    fp = C.POINTER(C.c_float)
    xs = np.ascontiguousarray(xs, np.float32)
    e, l, sel = np.empty(n, np.float32), np.empty(n, np.float32), np.empty((6, n), np.float32)
    o.ora_glibc_vec.argtypes = [C.c_uint64, fp, fp, fp]
    o.ora_glibc_vec(n, xs.ctypes.data_as(fp), e.ctypes.data_as(fp), l.ctypes.data_as(fp))
    o.ora_math_selftest.argtypes = [C.c_uint64, fp, fp, fp]
    o.ora_math_selftest(n, xs.ctypes.data_as(fp), xs.ctypes.data_as(fp), sel.ctypes.data_as(fp))
    return e, l, sel[0], sel[1]  # glibc exp, glibc ln|x|, contract exp, contract ln|x|


def test_the_contract_functions_stay_within_one_ulp_of_the_restatement_where_sinkhorn_uses_them():
    # rp_expf / rp_logf (the build's contract, include/rp_math.h) against the restated glibc functions on the Sinkhorn's domain:
    # exp of [-40, 0] (kernel entries, potentials), ln of (1e-30, 4]
    rng = np.random.default_rng(5)
    ge, _, ce, _ = _both((-40.0 * rng.random(200000)).astype(np.float32))
    assert _ulps(ge, ce).max() <= 1
    xs = np.exp(rng.uniform(np.log(1e-30), np.log(4.0), 200000)).astype(np.float32)
    _, gl, _, cl = _both(xs)
    near_one = np.abs(xs - 1.0) < 0.05  # results near 0: an ulp of the result is far below what the input's own rounding is worth
    assert _ulps(gl, cl)[~near_one].max() <= 1
    assert np.abs(gl - cl)[near_one].max() <= 2.0**-23


@needs_glibc
def test_oracle_on_the_restatement_equals_the_oracle_on_the_platform_libm():
    from lloyd_fixtures import flop_like_points, smooth_metric

    o = oracle.load()
    o.ora_lloyd_set_libm.argtypes = [C.c_int]
    bins, n = 64, 60
    tri = smooth_metric(bins, 3)
    pts = flop_like_points(2 * n, bins=bins, mass=30, seed=11).astype(np.uint32)
    try:
        o.ora_lloyd_set_libm(1)
        a = [oracle.sinkhorn_trace(x, y, tri, bins=bins)[:2] for x, y in zip(pts[:n], pts[n:])]
        o.ora_lloyd_set_libm(2)
        b = [oracle.sinkhorn_trace(x, y, tri, bins=bins)[:2] for x, y in zip(pts[:n], pts[n:])]
    finally:
        o.ora_lloyd_set_libm(0)
    assert [(np.float32(c).tobytes(), i) for c, i in a] == [(np.float32(c).tobytes(), i) for c, i in b]


def test_tables_recompute():
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "glibc_tables.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_regret_schedules_against_a_float64_restatement():
    # RegretSchedule::accumulate (regret/{summed,linear,discounted,floored,asymmetric}.rs) written again in float64 from the Rust
    # sources, epochs 0 .. 2000 and the three signs of the accumulated regret: the oracle (f32, glibc's powf / sqrt) within 2e-6
    # relative — an error of kind (a NaN clamped to the floor, a wrong branch) would be orders of magnitude away
    o = oracle.load()
    f = o.ora_regret_accumulate
    f.argtypes, f.restype = [C.c_int, C.c_float, C.c_float, C.c_uint64], C.c_float
    SUMMED, LINEAR, DISCOUNTED, FLOORED, ASYMMETRIC = 0, 1, 2, 3, 4
    from robopoker_amd import _lib
    assert (_lib.REGRET["summed"], _lib.REGRET["linear"], _lib.REGRET["discounted"], _lib.REGRET["floored"], _lib.REGRET["asymmetric"]) == (
        SUMMED, LINEAR, DISCOUNTED, FLOORED, ASYMMETRIC)

    def ref(kind, acc, imm, t):
        t = float(t)
        if kind in (SUMMED, FLOORED):
            return acc + imm
        if kind == LINEAR:
            return acc * (t / (t + 1.0)) + imm
        if kind == ASYMMETRIC:
            return acc + imm if acc > 0 else acc * (t / (t + 1.0)) + imm
        x = t**1.5 if acc > 0 else (t**0.5 if acc < 0 else t)
        return acc * (x / (x + 1.0)) + imm

    for kind in (SUMMED, LINEAR, DISCOUNTED, FLOORED, ASYMMETRIC):
        for t in list(range(0, 40)) + [100, 999, 2000, 1 << 20]:
            for acc in (0.0, 1.0, -1.0, 250.0, -3.75e5):
                got, want = f(kind, acc, 0.625, t), ref(kind, acc, 0.625, t)
                assert np.isfinite(got) and abs(got - want) <= 2e-6 * max(1.0, abs(want)), (kind, t, acc, got, want)


def test_the_committed_checksums_are_the_hosts():
    # tests/golden/glibc_checksums.json is what the GPU's evaluation of the header is compared with (tests/test_gpu_z_glibc_mode.py)
    import json

    o = oracle.load()
    o.ora_libm_glibc_checksums.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    want = json.load(open(ROOT / "tests" / "golden" / "glibc_checksums.json"))
    host = (C.c_uint64 * 4)()
    o.ora_libm_glibc_checksums(want["range"][0], want["range"][1], host)
    assert [hex(x) for x in host] == want["sums"]

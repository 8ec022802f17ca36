#!/usr/bin/env python3
"""Known answers FOR the reference (tests/golden/reference_kat.json).

The reference holds no golden vectors for its float paths and cannot be built in this image (no Rust), so parity with it is argued
from published algorithms (include/rp_refrng.h, include/rp_libm_glibc.h).  This script turns the argument into something a maintainer
can falsify with `cargo test` on a Linux/glibc machine: bit patterns that a build of the reference must produce for inputs expressible
through its own public API and its own test fixtures —

  sinkhorn   the three solves of crates/lloyd/src/sinkhorn.rs's tests (flop_metric(), flop_hist(..), :240-293): Sinkhorn::divergence
             and the raw minimize().cost(), as f32 bit patterns (oracle on glibc's expf / logf = this machine's libm)
  dcfr       DiscountedRegret::accumulate(acc, imm, epoch) (crates/mccfr/src/regret/discounted.rs:27-45) at epochs where powf(t, 1.5)
             differs from t * sqrt(t) and where glibc's powf(t, 0.5) differs from sqrt(t)
  rng        DefaultHasher of a Street discriminant, SmallRng::seed_from_u64 of it, and what random::<f32>(), random_range(0..n) and
             WeightedIndex<f32>::sample draw from it (the chain of flow.rs:285-295 and layer.rs:155-178)

INTEGRATION.md §4 holds the Rust test that checks them.  tests/test_reference_kat.py keeps the file equal to the oracle."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from lloyd_fixtures import flop_hist, flop_metric  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "reference_kat.json")


def bits(x) -> int:
    return int(np.asarray(x, dtype=np.float32).view(np.uint32))


def sinkhorn_cases(o):
    o.ora_lloyd_set_libm.argtypes = [C.c_int]
    o.ora_lloyd_set_libm(2)  # glibc's expf / logf restated == the platform's on a glibc host (tests/test_libm_glibc.py)
    try:
        tri = flop_metric()
        named = {"h": [(0, 3), (5, 1), (12, 4), (24, 2)], "mu": [(0, 3), (5, 1), (12, 4)], "nu": [(2, 2), (8, 5), (20, 1), (24, 3)]}
        out = []
        for a, b in (("h", "h"), ("mu", "nu"), ("nu", "mu"), ("mu", "mu"), ("nu", "nu")):
            ha, hb = flop_hist(named[a]), flop_hist(named[b])
            cost, iters = oracle.sinkhorn_cost(ha, hb, tri)
            out.append({"mu": named[a], "nu": named[b], "divergence_bits": bits(oracle.sinkhorn_divergence(ha, hb, tri)),
                        "cost_bits": bits(cost), "iterations": int(iters)})
        return out
    finally:
        o.ora_lloyd_set_libm(0)


def dcfr_cases(o):
    f = o.ora_regret_accumulate
    f.argtypes, f.restype = [C.c_int, C.c_float, C.c_float, C.c_uint64], C.c_float
    g = o.ora_glibc_powf
    g.argtypes, g.restype = [C.c_float, C.c_float], C.c_float
    DISCOUNTED = 2
    epochs = [0, 1, 2, 3, 5, 7, 10, 100, 1000, 12345, 1 << 20]  # 0: the first step (powf(+0, 1.5) = +0: the discount is 0)
    # the first epochs where the three candidate arithmetics part: powf(t, 1.5) != t * sqrt(t), and glibc powf(t, 0.5) != sqrt(t)
    t, found15, found05 = 2, [], []
    while (len(found15) < 3 or len(found05) < 3) and t < 1 << 22:
        x = np.float32(t)
        if len(found15) < 3 and bits(g(float(x), 1.5)) != bits(x * np.sqrt(x)):
            found15.append(t)
        if len(found05) < 3 and bits(g(float(x), 0.5)) != bits(np.sqrt(x)):
            found05.append(t)
        t += 1
    out = []
    for e in sorted(set(epochs + found15 + found05)):
        for acc in (1.0, -1.0, 0.0, 3.25, -777.5):
            out.append({"acc_bits": bits(acc), "imm_bits": bits(0.5), "epoch": e, "out_bits": bits(f(DISCOUNTED, acc, 0.5, e))})
    return out, found15, found05


def rng_cases(o):
    o.ora_defaulthasher_ints.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.c_uint32]
    o.ora_defaulthasher_ints.restype = C.c_uint64
    o.ora_smallrng_seeded.argtypes = [C.c_uint64, C.POINTER(C.c_uint64), C.c_uint32]
    o.ora_ref_draw_f32.argtypes, o.ora_ref_draw_f32.restype = [C.c_uint64], C.c_float
    o.ora_ref_draw_range.argtypes, o.ora_ref_draw_range.restype = [C.c_uint64, C.c_uint32], C.c_uint32
    o.ora_ref_weighted_index.argtypes = [C.c_uint64, C.POINTER(C.c_float), C.c_uint32]
    o.ora_ref_weighted_index.restype = C.c_uint32
    out = []
    for street in range(4):  # #[derive(Hash)] on a fieldless enum writes the discriminant as isize: 8 bytes, little endian
        vals, widths = (C.c_uint64 * 1)(street), (C.c_uint8 * 1)(8)
        seed = int(o.ora_defaulthasher_ints(vals, widths, 1))
        words = (C.c_uint64 * 4)()
        o.ora_smallrng_seeded(seed, words, 4)
        w = np.array([1.0, 2.0, 3.0, 0.5, 0.0, 10.0, 0.25], dtype=np.float32)
        out.append({"street_discriminant": street, "defaulthasher_finish": seed, "smallrng_next_u64": [int(x) for x in words],
                    "first_random_f32_bits": bits(o.ora_ref_draw_f32(seed)),
                    "first_random_range_0_to_7": int(o.ora_ref_draw_range(seed, 7)),
                    "weights": [float(x) for x in w],
                    "first_weighted_index": int(o.ora_ref_weighted_index(seed, w.ctypes.data_as(C.POINTER(C.c_float)), len(w)))})
    return out


def build():
    o = oracle.load()
    dcfr, f15, f05 = dcfr_cases(o)
    return {"about": "bit patterns a Linux/glibc build of krukah/robopoker must reproduce; generated by scripts/make_reference_kat.py from "
                     "the CPU oracle; the Rust test that checks them is in INTEGRATION.md section 4",
            "sinkhorn_fixture": sinkhorn_cases(o),
            "discounted_regret": {"alpha": 1.5, "beta": 0.5, "first_epochs_where_powf15_differs_from_t_sqrt_t": f15,
                                  "first_epochs_where_glibc_powf05_differs_from_sqrt": f05, "cases": dcfr},
            "seed_chain": rng_cases(o)}


if __name__ == "__main__":
    doc = build()
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print(OUT, os.path.getsize(OUT), "bytes")

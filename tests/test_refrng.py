"""include/rp_refrng.h against the published algorithms it restates (reference-seed mode's building blocks).

The reference seeds one SmallRng per sampled node from DefaultHasher(t, info, tree id) (crates/mccfr/src/strategy/flow.rs:285-295)
and takes one draw (sample/external.rs:41-64, sample/mod.rs:68-82, sample/pluribus.rs:91); k-means++ the same per layer
(crates/lloyd/src/layer.rs:155-178).  The hash and the generator are third-party (rustc std, rand 0.9.2): checked here against
their published vectors and an independent pure-Python restatement written from the papers, not from the C header.
"""
import ctypes as C
import struct

import numpy as np
import pytest

import oracle

M64 = (1 << 64) - 1


# ------------------------------------------------------------------------------------------ independent Python restatements
def rotl(x, b):
    return ((x << b) | (x >> (64 - b))) & M64


def py_siphash(k0, k1, msg: bytes, c, d):
    v0 = k0 ^ 0x736F6D6570736575
    v1 = k1 ^ 0x646F72616E646F6D
    v2 = k0 ^ 0x6C7967656E657261
    v3 = k1 ^ 0x7465646279746573

    def rnd(v0, v1, v2, v3):
        v0 = (v0 + v1) & M64; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32)
        v2 = (v2 + v3) & M64; v3 = rotl(v3, 16); v3 ^= v2
        v0 = (v0 + v3) & M64; v3 = rotl(v3, 21); v3 ^= v0
        v2 = (v2 + v1) & M64; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32)
        return v0, v1, v2, v3

    n = len(msg)
    padded = msg[: n - n % 8]
    last = msg[n - n % 8:] + b"\0" * (7 - n % 8) + bytes([n & 0xFF])
    for i in range(0, len(padded) + 8, 8):
        m = struct.unpack("<Q", (padded + last)[i:i + 8])[0]
        v3 ^= m
        for _ in range(c):
            v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
        v0 ^= m
    v2 ^= 0xFF
    for _ in range(d):
        v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
    return v0 ^ v1 ^ v2 ^ v3


def py_splitmix(seed, n):
    out = []
    for _ in range(n):
        seed = (seed + 0x9E3779B97F4A7C15) & M64
        z = seed
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        out.append(z ^ (z >> 31))
    return out


class PyXoshiro:
    def __init__(self, s):
        self.s = list(s)

    @classmethod
    def seed_from_u64(cls, seed):
        return cls(py_splitmix(seed, 4))

    def next_u64(self):
        s = self.s
        result = (rotl((s[0] + s[3]) & M64, 23) + s[0]) & M64
        t = (s[1] << 17) & M64
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = rotl(s[3], 45)
        return result

    def next_u32(self):
        return self.next_u64() >> 32


def f32(x):
    return np.float32(x)


def py_draw_f32(seed):
    return f32(PyXoshiro.seed_from_u64(seed).next_u32() >> 8) * f32(2.0 ** -24)


def py_draw_range(seed, n):
    r = PyXoshiro.seed_from_u64(seed)
    wide = r.next_u32() * n
    result, lo = wide >> 32, wide & 0xFFFFFFFF
    if lo > ((1 << 32) - n) & 0xFFFFFFFF:
        hi = (r.next_u32() * n) >> 32
        result += (lo + hi) >> 32
    return result


def py_draw_weight(seed, total):
    total = f32(total)
    scale = total
    while f32(f32(scale * f32(1.0 - 2.0 ** -23)) + f32(0)) >= total:
        scale = np.frombuffer(struct.pack("<I", struct.unpack("<I", struct.pack("<f", scale))[0] - 1), np.float32)[0]
    v12 = np.frombuffer(struct.pack("<I", (PyXoshiro.seed_from_u64(seed).next_u32() >> 9) | 0x3F800000), np.float32)[0]
    return f32(f32(v12 - f32(1)) * scale) + f32(0)


# ------------------------------------------------------------------------------------------ the oracle's exports
@pytest.fixture(scope="module")
def ora():
    o = oracle.load()
    u64p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
    o.ora_siphash.restype = C.c_uint64
    o.ora_siphash.argtypes = [C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint32, C.c_int, C.c_int]
    o.ora_defaulthasher_ints.restype = C.c_uint64
    o.ora_defaulthasher_ints.argtypes = [u64p, u8p, C.c_uint32]
    o.ora_splitmix64.argtypes = [C.c_uint64, u64p, C.c_uint32]
    o.ora_xoshiro256pp.argtypes = [u64p, u64p, C.c_uint32]
    o.ora_smallrng_seeded.argtypes = [C.c_uint64, u64p, C.c_uint32]
    o.ora_ref_draw_f32.restype = C.c_float
    o.ora_ref_draw_f32.argtypes = [C.c_uint64]
    o.ora_ref_draw_range.restype = C.c_uint32
    o.ora_ref_draw_range.argtypes = [C.c_uint64, C.c_uint32]
    o.ora_ref_draw_weight.restype = C.c_float
    o.ora_ref_draw_weight.argtypes = [C.c_uint64, C.c_float]
    o.ora_uniform_f32_scale.restype = C.c_float
    o.ora_uniform_f32_scale.argtypes = [C.c_float]
    o.ora_ref_node_seed.restype = C.c_uint64
    o.ora_ref_node_seed.argtypes = [C.c_uint64, C.c_char_p, C.c_uint32, C.c_uint64]
    o.ora_ref_weighted_index.restype = C.c_uint32
    o.ora_ref_weighted_index.argtypes = [C.c_uint64, C.POINTER(C.c_float), C.c_uint32]
    return o


def u64arr(n):
    return (C.c_uint64 * n)()


KEY = (0x0706050403020100, 0x0F0E0D0C0B0A0908)


def test_siphash_2_4_paper_vector(ora):
    # Aumasson & Bernstein 2012, appendix A: key 00..0f, message 00..0e
    msg = bytes(range(15))
    assert ora.ora_siphash(*KEY, msg, 15, 2, 4) == 0xA129CA6149BE45E5
    assert py_siphash(*KEY, msg, 2, 4) == 0xA129CA6149BE45E5


def test_siphash_1_3_libcore_vector(ora):
    # rust-lang/rust library/core tests (hash/sip.rs, test_siphash_1_3): key 00..0f, empty message -> dc c4 0f 05 58 01 ac ab
    assert ora.ora_siphash(*KEY, b"", 0, 1, 3) == struct.unpack("<Q", bytes([0xDC, 0xC4, 0x0F, 0x05, 0x58, 0x01, 0xAC, 0xAB]))[0]


def test_siphash_equals_the_python_restatement(ora):
    rng = np.random.default_rng(1)
    for n in list(range(0, 70)) + [255, 256, 257, 300]:
        msg = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        k0, k1 = (int(x) for x in rng.integers(0, 1 << 63, 2))
        for c, d in ((1, 3), (2, 4)):
            assert ora.ora_siphash(k0, k1, msg, n, c, d) == py_siphash(k0, k1, msg, c, d), (n, c, d)


def test_defaulthasher_integer_writes_are_the_little_endian_byte_stream(ora):
    rng = np.random.default_rng(2)
    for _ in range(200):
        n = int(rng.integers(1, 12))
        widths = rng.choice([1, 2, 8], n).astype(np.uint8)
        vals = rng.integers(0, 1 << 63, n).astype(np.uint64)
        msg = b"".join(int(v).to_bytes(8, "little")[: int(w)] for v, w in zip(vals, widths))
        got = ora.ora_defaulthasher_ints(vals.ctypes.data_as(C.POINTER(C.c_uint64)), widths.ctypes.data_as(C.POINTER(C.c_uint8)), n)
        assert got == py_siphash(0, 0, msg, 1, 3)


def test_splitmix64_vigna_vector(ora):
    # splitmix64.c (Vigna), seed 1234567
    want = [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431, 16408922859458223821]
    out = u64arr(5)
    ora.ora_splitmix64(1234567, out, 5)
    assert list(out) == want
    assert py_splitmix(1234567, 5) == want


def test_xoshiro256pp_reference_vector(ora):
    # rand's xoshiro256plusplus.rs `reference` test = xoshiro256plusplus.c with state (1, 2, 3, 4)
    want = [41943041, 58720359, 3588806011781223, 3591011842654386, 9228616714210784205, 9973669472204895162,
            14011001112246962877, 12406186145184390807, 15849039046786891736, 10450023813501588000]
    out = u64arr(10)
    ora.ora_xoshiro256pp((C.c_uint64 * 4)(1, 2, 3, 4), out, 10)
    assert list(out) == want
    r = PyXoshiro([1, 2, 3, 4])
    assert [r.next_u64() for _ in range(10)] == want


def test_seed_from_u64_is_four_splitmix_outputs(ora):
    for seed in (0, 1, 42, M64, 0x123456789ABCDEF0):
        out = u64arr(6)
        ora.ora_smallrng_seeded(seed, out, 6)
        r = PyXoshiro.seed_from_u64(seed)
        assert list(out) == [r.next_u64() for _ in range(6)]
    # seed 0 never yields the all-zero state (the point of using SplitMix64)
    assert py_splitmix(0, 4)[0] == 0xE220A8397B1DCDAF


def test_the_three_draws_equal_the_python_restatement(ora):
    rng = np.random.default_rng(3)
    for seed in [int(x) for x in rng.integers(0, 1 << 63, 300)] + [0, M64]:
        assert ora.ora_ref_draw_f32(seed) == float(py_draw_f32(seed))
        for n in (1, 2, 3, 4, 5, 6, 47, 1326, 1 << 20, (1 << 32) - 1):
            assert ora.ora_ref_draw_range(seed, n) == py_draw_range(seed, n)
        for total in (1.0, 0.75, 2.0, 3.4e-38 * 2, 1.0000001, 123456.7):
            assert ora.ora_ref_draw_weight(seed, total) == float(py_draw_weight(seed, total)), (seed, total)


def test_range_draw_second_sample_branch(ora):
    # the widening multiply's low half exceeds 2^32 - n only about n / 2^32 of the time: search seeds with a large n so that the
    # "biased" variant's second draw is exercised
    n = (1 << 32) - 1
    hits = 0
    for seed in range(2000):
        r = PyXoshiro.seed_from_u64(seed)
        lo = (r.next_u32() * n) & 0xFFFFFFFF
        if lo > ((1 << 32) - n):
            hits += 1
        assert ora.ora_ref_draw_range(seed, n) == py_draw_range(seed, n)
    assert hits > 100


def test_uniform_scale_and_draws_stay_below_the_total(ora):
    for total in (1.0, 2.0, 0.5, 3.0, 1.1754944e-38 * 4, 16777216.0, 0.1, 1e30):
        scale = ora.ora_uniform_f32_scale(total)
        assert np.float32(scale) * np.float32(1 - 2.0 ** -23) < np.float32(total)
        assert np.float32(scale) <= np.float32(total)


def test_weighted_index_distribution_and_edges(ora):
    w = (C.c_float * 3)(0.2, 0.0, 0.8)  # a zero weight is never drawn (k-means++ zeroes the chosen point, layer.rs:168)
    counts = [0, 0, 0]
    for seed in range(20000):
        counts[ora.ora_ref_weighted_index(seed, w, 3)] += 1
    assert counts[1] == 0
    assert abs(counts[0] / 20000 - 0.2) < 0.01
    one = (C.c_float * 1)(5.0)
    assert ora.ora_ref_weighted_index(7, one, 1) == 0


def test_node_seed_is_defaulthasher_over_t_info_tree(ora):
    # flow.rs:290-294: t.hash(); info.hash(); node.seed().hash() — usize, the info's derive(Hash) stream, usize
    info = bytes([1]) + (2).to_bytes(8, "little") + (1).to_bytes(8, "little")  # e.g. KuhnInfo: acting, History::Bet, Rank::Q
    for t, tree in ((0, 0), (5, 3), (1 << 40, 255), (16384, 12345678901)):
        msg = t.to_bytes(8, "little") + info + tree.to_bytes(8, "little")
        assert ora.ora_ref_node_seed(t, info, len(info), tree) == py_siphash(0, 0, msg, 1, 3)

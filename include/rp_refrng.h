/* rp_refrng.h — the reference's own seed -> sample chain, restated from the published algorithms.
 *
 * The reference draws every sampled branch from a generator it rebuilds per node (crates/mccfr/src/strategy/flow.rs:285-295):
 *
 *     DefaultHasher::new()  ->  t.hash(), info.hash(), node.seed().hash()  ->  SmallRng::seed_from_u64(hasher.finish())
 *
 * and then takes exactly one draw from it: WeightedIndex<f32>::sample at an opponent node (sample/external.rs:41-64),
 * random_range(0..n) at a chance node (sample/mod.rs:68-82), random::<f32>() for Pluribus' exploration coin
 * (sample/pluribus.rs:91).  The k-means++ seeding does the same with one generator per layer (crates/lloyd/src/layer.rs:155-178).
 * None of the four building blocks lives under /root/reference; they are third-party code pinned by Cargo.lock
 * (rand 0.9.2, rand_core 0.9.5, rustc's libstd) and each is a published algorithm:
 *
 *   DefaultHasher          SipHash-1-3 (Aumasson & Bernstein, "SipHash: a fast short-input PRF", 2012; c = 1, d = 3), key (0, 0),
 *                          64-bit output; std::hash::Hash feeds integers as their native-endian (little-endian) bytes,
 *                          #[derive(Hash)] feeds a struct's fields in declaration order and an enum's discriminant as isize
 *                          (8 bytes) before the variant's fields, bool as one byte.
 *   SmallRng (64-bit)      xoshiro256++ (Blackman & Vigna, "Scrambled linear pseudorandom number generators", 2018);
 *                          seed_from_u64 fills the four state words with consecutive SplitMix64 outputs (Steele, Lea & Flood 2014;
 *                          rand's xoshiro256plusplus.rs overrides rand_core's default); next_u32 = next_u64 >> 32.
 *   random::<f32>()        StandardUniform: (next_u32 >> 8) * 2^-24                     (rand/src/distr/float.rs)
 *   random_range(0..n)     usize ranges that fit 32 bits sample as u32 (UniformUsize, rand 0.9): Lemire/Canon widening multiply,
 *                          one extra draw when the low half exceeds 2^32 - n, result incremented on carry (uniform_int.rs,
 *                          sample_single_inclusive, the default "biased" variant).
 *   WeightedIndex<f32>     cumulative f32 sums of all but the last weight, total = the running sum after the last;
 *                          x = Uniform::new(0, total).sample = ((next_u32 >> 9 | 0x3f800000 as f32) - 1) * scale + 0 with
 *                          scale = total (decreased one ulp at a time while scale * (1 - 2^-23) + 0 >= total);
 *                          index = partition_point(cum <= x)                             (weighted_index.rs, uniform_float.rs)
 *
 * This header is shared by the oracle (C) and the kernels (HIP): "reference-seed" mode (rp_rng_kind RP_RNG_REFERENCE) uses it,
 * the default mode keeps include/rp_math.h's counter hash.  tests/test_refrng.py checks the pieces against the published
 * vectors (the SipHash paper's 2-4 vector through the same round function, Rust libcore's 1-3 vector, Vigna's SplitMix64 and
 * xoshiro256++ outputs) and against an independent Python restatement.
 */
#ifndef RP_REFRNG_H
#define RP_REFRNG_H

#include <stdint.h>

#include "rp_math.h"

/* ------------------------------------------------------------------------------------------ SipHash-c-d, 64-bit output ---- */
typedef struct rp_sip {
    uint64_t v0, v1, v2, v3;
    uint64_t tail;  /* bytes not yet compressed, little-endian, low `ntail` bytes valid */
    uint32_t ntail; /* 0..7 */
    uint32_t len;   /* bytes written so far (only the low 8 bits enter the hash) */
} rp_sip;

RP_HD uint64_t rp_rotl64(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }

#define RP_SIPROUND(v0, v1, v2, v3) \
    do {                            \
        v0 += v1;                   \
        v1 = rp_rotl64(v1, 13);     \
        v1 ^= v0;                   \
        v0 = rp_rotl64(v0, 32);     \
        v2 += v3;                   \
        v3 = rp_rotl64(v3, 16);     \
        v3 ^= v2;                   \
        v0 += v3;                   \
        v3 = rp_rotl64(v3, 21);     \
        v3 ^= v0;                   \
        v2 += v1;                   \
        v1 = rp_rotl64(v1, 17);     \
        v1 ^= v2;                   \
        v2 = rp_rotl64(v2, 32);     \
    } while (0)

RP_HD void rp_sip_init(rp_sip* s, uint64_t k0, uint64_t k1) {
    s->v0 = k0 ^ 0x736f6d6570736575ull;
    s->v1 = k1 ^ 0x646f72616e646f6dull;
    s->v2 = k0 ^ 0x6c7967656e657261ull;
    s->v3 = k1 ^ 0x7465646279746573ull;
    s->tail = 0;
    s->ntail = 0;
    s->len = 0;
}
/* one message word, c compression rounds */
RP_HD void rp_sip_compress(rp_sip* s, uint64_t m, int c) {
    uint64_t v0 = s->v0, v1 = s->v1, v2 = s->v2, v3 = s->v3;
    v3 ^= m;
    for (int i = 0; i < c; ++i) RP_SIPROUND(v0, v1, v2, v3);
    v0 ^= m;
    s->v0 = v0;
    s->v1 = v1;
    s->v2 = v2;
    s->v3 = v3;
}
/* the low `nbytes` (1..8) bytes of x, little-endian — Hasher::write of an integer's to_ne_bytes() */
RP_HD void rp_sip_write_le(rp_sip* s, uint64_t x, uint32_t nbytes, int c) {
    if (nbytes < 8) x &= (1ull << (8 * nbytes)) - 1ull;
    s->len += nbytes;
    const uint32_t have = s->ntail;
    const uint64_t m = have ? (s->tail | (x << (8 * have))) : x;
    if (have + nbytes < 8) {
        s->tail = m;
        s->ntail = have + nbytes;
        return;
    }
    rp_sip_compress(s, m, c);
    const uint32_t used = 8 - have; /* bytes of x that went into m */
    s->ntail = nbytes - used;
    s->tail = used < 8 ? (x >> (8 * used)) : 0ull;
    if (s->ntail == 0) s->tail = 0;
}
RP_HD void rp_sip_write(rp_sip* s, const uint8_t* p, uint32_t n, int c) {
    for (uint32_t i = 0; i < n; ++i) rp_sip_write_le(s, p[i], 1, c);
}
RP_HD uint64_t rp_sip_finish(const rp_sip* s, int c, int d) {
    rp_sip t = *s;
    const uint64_t b = ((uint64_t)(t.len & 0xffu) << 56) | t.tail;
    rp_sip_compress(&t, b, c);
    uint64_t v0 = t.v0, v1 = t.v1, v2 = t.v2 ^ 0xffull, v3 = t.v3;
    for (int i = 0; i < d; ++i) RP_SIPROUND(v0, v1, v2, v3);
    return v0 ^ v1 ^ v2 ^ v3;
}
/* std::collections::hash_map::DefaultHasher */
RP_HD void rp_defaulthasher_new(rp_sip* s) { rp_sip_init(s, 0ull, 0ull); }
RP_HD void rp_defaulthasher_write_u64(rp_sip* s, uint64_t x) { rp_sip_write_le(s, x, 8, 1); } /* usize / isize / u64 */
RP_HD void rp_defaulthasher_write_u16(rp_sip* s, uint16_t x) { rp_sip_write_le(s, x, 2, 1); }
RP_HD void rp_defaulthasher_write_u8(rp_sip* s, uint8_t x) { rp_sip_write_le(s, x, 1, 1); }   /* u8 / bool */
RP_HD void rp_defaulthasher_write(rp_sip* s, const uint8_t* p, uint32_t n) { rp_sip_write(s, p, n, 1); }
RP_HD uint64_t rp_defaulthasher_finish(const rp_sip* s) { return rp_sip_finish(s, 1, 3); }

/* ------------------------------------------------------------------------------------------ SplitMix64, xoshiro256++ ------ */
RP_HD uint64_t rp_splitmix64_next(uint64_t* state) {
    uint64_t z = (*state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
typedef struct rp_smallrng {
    uint64_t s[4];
} rp_smallrng;
/* SmallRng::seed_from_u64 (Xoshiro256PlusPlus::seed_from_u64) */
RP_HD void rp_smallrng_seed(rp_smallrng* r, uint64_t seed) {
    for (int i = 0; i < 4; ++i) r->s[i] = rp_splitmix64_next(&seed);
}
RP_HD uint64_t rp_smallrng_next_u64(rp_smallrng* r) {
    const uint64_t result = rp_rotl64(r->s[0] + r->s[3], 23) + r->s[0];
    const uint64_t t = r->s[1] << 17;
    r->s[2] ^= r->s[0];
    r->s[3] ^= r->s[1];
    r->s[1] ^= r->s[2];
    r->s[0] ^= r->s[3];
    r->s[2] ^= t;
    r->s[3] = rp_rotl64(r->s[3], 45);
    return result;
}
RP_HD uint32_t rp_smallrng_next_u32(rp_smallrng* r) { return (uint32_t)(rp_smallrng_next_u64(r) >> 32); }

/* ------------------------------------------------------------------------------------------ the three draws ---------------- */
/* rng.random::<f32>() */
RP_HD float rp_rand_f32(rp_smallrng* r) { return (float)(rp_smallrng_next_u32(r) >> 8) * 5.9604644775390625e-8f; }
/* rng.random_range(0..n) for usize, 0 < n <= 2^32 - 1 */
RP_HD uint32_t rp_rand_range_u32(rp_smallrng* r, uint32_t n) {
    const uint64_t wide = (uint64_t)rp_smallrng_next_u32(r) * (uint64_t)n;
    uint32_t result = (uint32_t)(wide >> 32);
    const uint32_t lo_order = (uint32_t)wide;
    if (lo_order > (uint32_t)(0u - n)) {
        const uint32_t new_hi = (uint32_t)(((uint64_t)rp_smallrng_next_u32(r) * (uint64_t)n) >> 32);
        result += (uint32_t)(((uint64_t)lo_order + (uint64_t)new_hi) >> 32);
    }
    return result;
}
/* UniformFloat<f32>::new(0.0, total).scale */
RP_HD float rp_uniform_f32_scale(float total) {
    const float max_rand = 0.99999988079071044921875f; /* 1 - f32::EPSILON */
    float scale = total - 0.0f;
    while (scale * max_rand + 0.0f >= total) scale = rp_u2f(rp_f2u(scale) - 1u);
    return scale;
}
/* UniformFloat<f32>::sample with low = 0 */
RP_HD float rp_rand_uniform_f32(rp_smallrng* r, float scale) {
    const float value1_2 = rp_u2f((rp_smallrng_next_u32(r) >> 9) | 0x3f800000u);
    const float value0_1 = value1_2 - 1.0f;
    return value0_1 * scale + 0.0f;
}

/* The generator of one sampled node takes one draw; with the 64-bit seed in hand the draws are pure functions of it. */
RP_HD float rp_ref_draw_f32(uint64_t seed) {
    rp_smallrng r;
    rp_smallrng_seed(&r, seed);
    return rp_rand_f32(&r);
}
RP_HD uint32_t rp_ref_draw_range(uint64_t seed, uint32_t n) {
    rp_smallrng r;
    rp_smallrng_seed(&r, seed);
    return rp_rand_range_u32(&r, n);
}
/* the x that WeightedIndex compares its cumulative sums with; `total` > 0 */
RP_HD float rp_ref_draw_weight(uint64_t seed, float total) {
    rp_smallrng r;
    rp_smallrng_seed(&r, seed);
    return rp_rand_uniform_f32(&r, rp_uniform_f32_scale(total));
}

/* ------------------------------------------------------------------------------------------ the node's seed ---------------- */
/* DefaultHasher after t.hash() and info.hash(): every node of the step that shares the infoset continues from here */
typedef struct rp_sip_mid {
    uint64_t v0, v1, v2, v3;
    uint64_t tail;
    uint32_t ntail;
    uint32_t len;
} rp_sip_mid;
RP_HD void rp_ref_seed_prefix(rp_sip_mid* out, uint64_t t, const uint8_t* info_bytes, uint32_t n) {
    rp_sip s;
    rp_defaulthasher_new(&s);
    rp_defaulthasher_write_u64(&s, t);
    rp_defaulthasher_write(&s, info_bytes, n);
    out->v0 = s.v0;
    out->v1 = s.v1;
    out->v2 = s.v2;
    out->v3 = s.v3;
    out->tail = s.tail;
    out->ntail = s.ntail;
    out->len = s.len;
}
/* ... node.seed().hash(hasher); hasher.finish() */
RP_HD uint64_t rp_ref_seed_finish(const rp_sip_mid* mid, uint64_t tree_id) {
    rp_sip s;
    s.v0 = mid->v0;
    s.v1 = mid->v1;
    s.v2 = mid->v2;
    s.v3 = mid->v3;
    s.tail = mid->tail;
    s.ntail = mid->ntail;
    s.len = mid->len;
    rp_defaulthasher_write_u64(&s, tree_id);
    return rp_defaulthasher_finish(&s);
}
RP_HD uint64_t rp_ref_node_seed(uint64_t t, const uint8_t* info_bytes, uint32_t n, uint64_t tree_id) {
    rp_sip_mid mid;
    rp_ref_seed_prefix(&mid, t, info_bytes, n);
    return rp_ref_seed_finish(&mid, tree_id);
}

#endif /* RP_REFRNG_H */

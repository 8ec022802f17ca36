/* rp_oracle_refrng.c — TEST INFRASTRUCTURE: exports of include/rp_refrng.h's pieces (SipHash-c-d, SplitMix64, xoshiro256++,
 * rand 0.9.2's three draws) so that tests/test_refrng.py can check each against its published vectors and an independent
 * Python restatement.  The reference call sites they stand for: crates/mccfr/src/strategy/flow.rs:285-295,
 * sample/external.rs:41-64, sample/mod.rs:68-82, sample/pluribus.rs:91, crates/lloyd/src/layer.rs:155-178. */
#include <stddef.h>
#include <stdint.h>

#include "../include/rp_refrng.h"
#include "../include/rp_libm_glibc.h"

#define ORA_API __attribute__((visibility("default")))

ORA_API uint64_t ora_siphash(uint64_t k0, uint64_t k1, const uint8_t* msg, uint32_t n, int c, int d) {
    rp_sip s;
    rp_sip_init(&s, k0, k1);
    rp_sip_write(&s, msg, n, c);
    return rp_sip_finish(&s, c, d);
}
/* DefaultHasher over a sequence of integer writes: widths[i] in {1, 2, 8} bytes */
ORA_API uint64_t ora_defaulthasher_ints(const uint64_t* vals, const uint8_t* widths, uint32_t n) {
    rp_sip s;
    rp_defaulthasher_new(&s);
    for (uint32_t i = 0; i < n; ++i) rp_sip_write_le(&s, vals[i], widths[i], 1);
    return rp_defaulthasher_finish(&s);
}
ORA_API void ora_splitmix64(uint64_t seed, uint64_t* out, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) out[i] = rp_splitmix64_next(&seed);
}
ORA_API void ora_xoshiro256pp(const uint64_t* state, uint64_t* out, uint32_t n) {
    rp_smallrng r;
    for (int i = 0; i < 4; ++i) r.s[i] = state[i];
    for (uint32_t i = 0; i < n; ++i) out[i] = rp_smallrng_next_u64(&r);
}
ORA_API void ora_smallrng_seeded(uint64_t seed, uint64_t* out, uint32_t n) {
    rp_smallrng r;
    rp_smallrng_seed(&r, seed);
    for (uint32_t i = 0; i < n; ++i) out[i] = rp_smallrng_next_u64(&r);
}
ORA_API float ora_ref_draw_f32(uint64_t seed) { return rp_ref_draw_f32(seed); }
ORA_API uint32_t ora_ref_draw_range(uint64_t seed, uint32_t n) { return rp_ref_draw_range(seed, n); }
ORA_API float ora_ref_draw_weight(uint64_t seed, float total) { return rp_ref_draw_weight(seed, total); }
ORA_API float ora_uniform_f32_scale(float total) { return rp_uniform_f32_scale(total); }
/* a generator that keeps going (k-means++ takes K draws from one): state in/out */
ORA_API uint32_t ora_smallrng_range(uint64_t* state, uint32_t n) {
    rp_smallrng r;
    for (int i = 0; i < 4; ++i) r.s[i] = state[i];
    const uint32_t v = rp_rand_range_u32(&r, n);
    for (int i = 0; i < 4; ++i) state[i] = r.s[i];
    return v;
}
ORA_API uint64_t ora_ref_node_seed(uint64_t t, const uint8_t* info_bytes, uint32_t n, uint64_t tree_id) {
    return rp_ref_node_seed(t, info_bytes, n, tree_id);
}
/* WeightedIndex::<f32>::new(weights).sample(rng) with a fresh SmallRng::seed_from_u64(seed); cum is scratch of n floats */
ORA_API uint32_t ora_ref_weighted_index(uint64_t seed, const float* w, uint32_t n) {
    float total = w[0];
    uint32_t idx = 0;
    /* first pass: total; second: partition_point over the n - 1 cumulative sums */
    for (uint32_t i = 1; i < n; ++i) total += w[i];
    const float x = rp_ref_draw_weight(seed, total);
    float run = w[0];
    for (uint32_t i = 0; i + 1 < n; ++i) {
        if (run <= x) idx = i + 1;
        else break;
        run += w[i + 1];
    }
    return idx;
}

/* include/rp_libm_glibc.h against THIS machine's libm over every float bit pattern in [lo, hi): mismatches of expf and logf (both NaN
 * counts as equal); first[0..1] = the first mismatching bit pattern of each (~0 if none).  OpenMP: the whole 2^32 range takes ~10 s on
 * eight cores. */
ORA_API void ora_libm_glibc_sweep(uint64_t lo, uint64_t hi, uint64_t* bad_exp, uint64_t* bad_log, uint32_t* first) {
    uint64_t be = 0, bl = 0;
    uint32_t fe = 0xffffffffu, fl = 0xffffffffu;
#pragma omp parallel for reduction(+ : be, bl) reduction(min : fe, fl) schedule(static)
    for (long long b = (long long)lo; b < (long long)hi; ++b) {
        const float x = rp_u2f((uint32_t)b);
        const float a = rp_glibc_expf(x), c = expf(x);
        if (rp_f2u(a) != rp_f2u(c) && !(a != a && c != c)) {
            be += 1;
            if ((uint32_t)b < fe) fe = (uint32_t)b;
        }
        const float d = rp_glibc_logf(x), e = logf(x);
        if (rp_f2u(d) != rp_f2u(e) && !(d != d && e != e)) {
            bl += 1;
            if ((uint32_t)b < fl) fl = (uint32_t)b;
        }
    }
    *bad_exp = be;
    *bad_log = bl;
    first[0] = fe;
    first[1] = fl;
}
/* the branch-free, table-parameterised forms the device kernels evaluate (rp_glibc_expf_tab, rp_glibc_exp_floor_tab,
 * rp_glibc_logf_tab) against the forms above over every float bit pattern in [lo, hi): bad[0..2] mismatches (both NaN counts as
 * equal), first[0..2] the first mismatching pattern of each (~0 if none) */
ORA_API void ora_libm_glibc_tab_sweep(uint64_t lo, uint64_t hi, uint64_t* bad, uint32_t* first) {
    static const uint64_t T[32] = RP_GLIBC_EXP2F_TAB_INIT;
    static const double LT[16][2] = RP_GLIBC_LOGF_TAB_INIT;
    uint64_t b0 = 0, b1 = 0, b2 = 0;
    uint32_t f0 = 0xffffffffu, f1 = 0xffffffffu, f2 = 0xffffffffu;
#pragma omp parallel for reduction(+ : b0, b1, b2) reduction(min : f0, f1, f2) schedule(static)
    for (long long b = (long long)lo; b < (long long)hi; ++b) {
        const float x = rp_u2f((uint32_t)b);
        const float e = rp_glibc_expf(x);
        const float a = rp_glibc_expf_tab(x, T);
        if (rp_f2u(a) != rp_f2u(e) && !(a != a && e != e)) {
            b0 += 1;
            if ((uint32_t)b < f0) f0 = (uint32_t)b;
        }
        const float c = rp_glibc_exp_floor_tab(x, T), d = rp_maxf(e, RP_EPSILON);
        if (rp_f2u(c) != rp_f2u(d)) {
            b1 += 1;
            if ((uint32_t)b < f1) f1 = (uint32_t)b;
        }
        const float g = rp_glibc_logf_tab(x, LT), h = rp_glibc_logf(x);
        if (rp_f2u(g) != rp_f2u(h) && !(g != g && h != h)) {
            b2 += 1;
            if ((uint32_t)b < f2) f2 = (uint32_t)b;
        }
    }
    bad[0] = b0, bad[1] = b1, bad[2] = b2;
    first[0] = f0, first[1] = f1, first[2] = f2;
}
ORA_API float ora_glibc_expf(float x) { return rp_glibc_expf(x); }
ORA_API float ora_glibc_logf(float x) { return rp_glibc_logf(x); }
ORA_API void ora_glibc_vec(uint64_t n, const float* x, float* e, float* l) {
    for (uint64_t i = 0; i < n; ++i) {
        e[i] = rp_glibc_expf(x[i]);
        l[i] = rp_glibc_logf(x[i] < 0 ? -x[i] : x[i]);
    }
}
/* rp_glibc_powf(x, y) against this machine's powf for every float x in bit range [lo, hi) (both NaN counts as equal) */
ORA_API uint64_t ora_libm_glibc_pow_sweep(uint64_t lo, uint64_t hi, float y, uint32_t* first) {
    uint64_t bad = 0;
    uint32_t fb = 0xffffffffu;
#pragma omp parallel for reduction(+ : bad) reduction(min : fb) schedule(static)
    for (long long b = (long long)lo; b < (long long)hi; ++b) {
        const float x = rp_u2f((uint32_t)b);
        const float a = rp_glibc_powf(x, y), c = powf(x, y);
        if (rp_f2u(a) != rp_f2u(c) && !(a != a && c != c)) {
            bad += 1;
            if ((uint32_t)b < fb) fb = (uint32_t)b;
        }
    }
    *first = fb;
    return bad;
}
ORA_API float ora_glibc_powf(float x, float y) { return rp_glibc_powf(x, y); }
ORA_API float ora_pow15(float t) { return rp_pow15(t); }
ORA_API float ora_pow05(float t) { return rp_pow05(t); }
/* the four checksums of rp_libm_glibc_sweep (include/rp_mi355x_diag.h), computed on the host */
ORA_API void ora_libm_glibc_checksums(uint64_t lo, uint64_t hi, uint64_t* sums) {
    uint64_t se = 0, we = 0, sl = 0, wl = 0;
#pragma omp parallel for reduction(+ : se, we, sl, wl) schedule(static)
    for (long long b = (long long)lo; b < (long long)hi; ++b) {
        const uint32_t u = (uint32_t)b;
        const float x = rp_u2f(u);
        const float e = rp_glibc_expf(x), l = rp_glibc_logf(x);
        const uint64_t be = e != e ? 0x7fc00000u : rp_f2u(e), bl = l != l ? 0x7fc00000u : rp_f2u(l);
        const uint64_t odd = 2ull * u + 1ull;
        se += be;
        we += be * odd;
        sl += bl;
        wl += bl * odd;
    }
    sums[0] = se;
    sums[1] = we;
    sums[2] = sl;
    sums[3] = wl;
}

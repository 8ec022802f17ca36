/* rp_math.h — the arithmetic contract of the MI355X robopoker hot paths.
 *
 * Every f32 result the library produces (regret/strategy tables, Sinkhorn
 * potentials, EMD distances, bucket assignments) is defined by the functions
 * in this header evaluated in a documented order.  The same text is compiled
 * by hipcc for gfx950 (device), by g++ for the host side of the C-ABI library
 * and by gcc for the CPU oracle, always with floating-point contraction OFF,
 * so "identical seeds -> identical bits" holds between CPU and GPU.
 *
 * Why it exists (reference boundaries that are NOT reproducible):
 *   - f32::exp / f32::ln / f32::powf are platform libm in the reference
 *     (crates/lloyd/src/sinkhorn.rs:115,120-127,136; phi.rs:36;
 *      crates/mccfr/src/regret/discounted.rs:33,37)           -> rp_expf/rp_logf here; powf: rp_libm_glibc.h
 *   - DefaultHasher(SipHash-1-3) + SmallRng::seed_from_u64 + WeightedIndex /
 *     random_range / random::<f32>() from rand 0.9.2
 *     (crates/mccfr/src/strategy/flow.rs:285-295, sample/external.rs:57-63,
 *      sample/mod.rs:76-81, sample/pluribus.rs:91, crates/lloyd/src/layer.rs:155-165)
 *                                                               -> rp_node_hash / rp_u01 / rp_pick_* (the default,
 *     cheap on the device) or, in reference-seed mode, include/rp_refrng.h's restatement of those published algorithms
 * Only IEEE-754 correctly rounded primitives are used: + - * / sqrt, fma,
 * int<->float conversion and integer bit operations.
 */
#ifndef RP_MATH_H
#define RP_MATH_H

#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RP_HD __host__ __device__ __forceinline__
#else
#define RP_HD static inline
#endif

/* pokerkit::EPSILON = f32::MIN_POSITIVE (crates/pokerkit/src/lib.rs:204) */
#define RP_EPSILON 1.17549435e-38f
#define RP_F32_MAX 3.40282347e+38f

RP_HD float rp_u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
RP_HD uint32_t rp_f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
/* f32::max / f32::min: a NaN operand yields the other operand (NaN never occurs on the hot path: the reference
 * debug_asserts it, flow.rs:81-82; it does in a degenerate layer whose empty clusters have 0/0 densities); otherwise
 * IEEE 754-2019 maximum / minimum: +0 is greater
 * than -0.  On gfx950 this is exactly one v_max_f32 / v_min_f32 (CDNA ISA: max(+0,-0) = +0,
 * min(+0,-0) = -0); the host spells the same function with compares. */
#if defined(__HIP_DEVICE_COMPILE__)
RP_HD float rp_maxf(float a, float b) { return __builtin_fmaxf(a, b); }
RP_HD float rp_minf(float a, float b) { return __builtin_fminf(a, b); }
#else
RP_HD float rp_maxf(float a, float b) {
    if (a != a) return b; /* f32::max / v_max_f32: a NaN operand yields the other one (a degenerate layer's empty cluster) */
    if (b != b) return a;
    if (a == b) return (rp_f2u(a) & 0x80000000u) ? b : a; /* equal: prefer the one without a sign bit */
    return a > b ? a : b;
}
RP_HD float rp_minf(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (rp_f2u(a) & 0x80000000u) ? a : b; /* equal: prefer the one with a sign bit */
    return a < b ? a : b;
}
#endif
RP_HD float rp_absf(float a) { return rp_u2f(rp_f2u(a) & 0x7fffffffu); }

/* e^x, ~1 ulp.  Cephes-style: k = rint(x*log2e), r = x - k*ln2 (two-step),
 * e^r = 1 + r + r^2 * P(r), result scaled by 2^k in two exact steps.
 * x < -87.34 flushes to +0 (the Sinkhorn softmin clamps every term at
 * MIN_POSITIVE afterwards, sinkhorn.rs:124-126); x > 88.72 -> +inf. */
RP_HD float rp_expf_spec(float x) {
    /* branch free: evaluate on the clamped argument, patch the special cases with selects at the end */
    float xc = x > 88.72283f ? 88.72283f : x;
    xc = xc < -87.33654f ? -87.33654f : xc;
    xc = (x == x) ? xc : 0.0f;
    const float MAGIC = 12582912.0f; /* 1.5 * 2^23: round-to-nearest-even via add/sub */
    float kf = (xc * 1.44269504088896341f + MAGIC) - MAGIC;
    float r = fmaf(kf, -0.693359375f, xc);
    r = fmaf(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    int k = (int)kf;
    int k1 = k >> 1;
    int k2 = k - k1;
    float s1 = rp_u2f((uint32_t)(k1 + 127) << 23);
    float s2 = rp_u2f((uint32_t)(k2 + 127) << 23);
    float res = (y * s1) * s2;
    res = x > 88.72283f ? rp_u2f(0x7f800000u) : res;
    res = x < -87.33654f ? 0.0f : res;
    return (x == x) ? res : x;
}


#if defined(__HIPCC__) || defined(__HIP__)
#define RP_D __device__ __forceinline__
/* gfx950 spellings of rp_expf_spec.  Same value for EVERY input bit pattern (rp_math_selftest sweeps all 2^32 on
 * the device), fewer VALU instructions: v_max/v_min clamp, v_rndne instead of the magic add/sub (same round-to-nearest-
 * even for |t| < 2^22), one v_ldexp instead of the two exact scalings (both round once), packed v_pk_fma/mul/add
 * for two arguments at a time.  The Sinkhorn softmin evaluates max(exp(x), MIN_POSITIVE) (sinkhorn.rs:124-126):
 * there the underflow / NaN patches are subsumed by the final max (a subnormal, 0 or NaN all give MIN_POSITIVE),
 * and the overflow patch by clamping above the overflow threshold so that ldexp saturates to +inf by itself. */
typedef float rp_f2 __attribute__((ext_vector_type(2)));

RP_D float rp_exp_core(float xc) { /* xc already clamped, not NaN */
    const float kf = __builtin_rintf(xc * 1.44269504088896341f);
    float r = __builtin_fmaf(kf, -0.693359375f, xc);
    r = __builtin_fmaf(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float y = __builtin_fmaf(p, r2, r) + 1.0f;
    return __builtin_ldexpf(y, (int)kf);
}
RP_D float rp_expf_dev(float x) {
    float res = rp_exp_core(__builtin_amdgcn_fmed3f(x, -87.33654f, 88.72283f));
    res = x > 88.72283f ? rp_u2f(0x7f800000u) : res;
    res = x < -87.33654f ? 0.0f : res;
    return (x == x) ? res : x;
}
/* == rp_maxf(rp_expf_spec(x), RP_EPSILON) */
RP_D float rp_exp_floor(float x) {
    return __builtin_fmaxf(rp_exp_core(__builtin_fminf(__builtin_fmaxf(x, -104.0f), 89.5f)), RP_EPSILON);
}
/* two arguments at a time on the packed-f32 pipe */
RP_D rp_f2 rp_exp_floor2(rp_f2 x) {
    rp_f2 xc;
    xc.x = __builtin_fminf(__builtin_fmaxf(x.x, -104.0f), 89.5f); /* max first: a NaN becomes -104 -> MIN_POSITIVE */
    xc.y = __builtin_fminf(__builtin_fmaxf(x.y, -104.0f), 89.5f);
    const rp_f2 t = xc * 1.44269504088896341f;
    rp_f2 kf;
    kf.x = __builtin_rintf(t.x);
    kf.y = __builtin_rintf(t.y);
    rp_f2 r = __builtin_elementwise_fma(kf, (rp_f2)(-0.693359375f), xc);
    r = __builtin_elementwise_fma(kf, (rp_f2)(2.12194440e-4f), r);
    rp_f2 p = __builtin_elementwise_fma((rp_f2)(1.9875691500e-4f), r, (rp_f2)(1.3981999507e-3f));
    p = __builtin_elementwise_fma(p, r, (rp_f2)(8.3334519073e-3f));
    p = __builtin_elementwise_fma(p, r, (rp_f2)(4.1665795894e-2f));
    p = __builtin_elementwise_fma(p, r, (rp_f2)(1.6666665459e-1f));
    p = __builtin_elementwise_fma(p, r, (rp_f2)(5.0000001201e-1f));
    const rp_f2 r2 = r * r;
    const rp_f2 y = __builtin_elementwise_fma(p, r2, r) + 1.0f;
    rp_f2 res;
    res.x = __builtin_fmaxf(__builtin_ldexpf(y.x, (int)kf.x), RP_EPSILON);
    res.y = __builtin_fmaxf(__builtin_ldexpf(y.y, (int)kf.y), RP_EPSILON);
    return res;
}
#endif
RP_HD float rp_expf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return rp_expf_dev(x);
#else
    return rp_expf_spec(x);
#endif
}

/* ln x, ~1 ulp.  Cephes-style: x = m * 2^e with m in [sqrt(1/2), sqrt(2)),
 * polynomial in (m - 1); no division.  x == 0 -> -inf, x < 0 -> NaN.
 * Subnormal inputs are pre-scaled by 2^23. */
RP_HD float rp_logf(float x) {
    if (!(x == x)) return x;
    if (x < 0.0f) return rp_u2f(0x7fc00000u);
    if (x == 0.0f) return rp_u2f(0xff800000u);
    if (x == rp_u2f(0x7f800000u)) return x;
    uint32_t u = rp_f2u(x);
    int e = 0;
    if (u < 0x00800000u) {
        x = x * 8388608.0f;
        u = rp_f2u(x);
        e = -23;
    }
    e += (int)(u >> 23) - 126;
    float m = rp_u2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = fmaf(p, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float y = (m * z) * p;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    float res = m + y;
    res = fmaf(fe, 0.693359375f, res);
    return res;
}

/* a / b for a correctly rounded reciprocal r = 1.0f / b, without a division on the dependent path
 * (Markstein's sequence: two fma corrections; the first makes the quotient faithful, the second
 * correctly rounded).  Bit-identical to IEEE a / b whenever no intermediate under/overflows:
 * callers guarantee 1 <= b <= 2^32 and (a == 0 or |a| >= 2^-60), see rp_div_by_recip_ok.
 * Used only to shorten the serial Welford chain of the MCCFR update; the oracle divides plainly. */
RP_HD float rp_div_by_recip(float a, float b, float r) {
    float q0 = a * r;
    float e0 = fmaf(-b, q0, a);
    float q1 = fmaf(e0, r, q0);
    float e1 = fmaf(-b, q1, a);
    return fmaf(e1, r, q1);
}
/* One correction only, plus an EXACT proof of the result that stays off the dependent path: with the exact
 * residual r1 = a - b*q1, q1 is the correctly rounded quotient iff |a/b - q1| < half the f32 spacing next to
 * q1, i.e. 2|r1| < |b| * spacing (the smaller spacing is used when |q1| is a power of two; ties count as
 * "not proven").  *proven = 0 happens essentially never; the caller then divides plainly. */
RP_HD float rp_div_by_recip1(float a, float b, float r, int* proven) {
    float q0 = a * r;
    float e0 = fmaf(-b, q0, a);
    float q1 = fmaf(e0, r, q0);
    float r1 = fmaf(-b, q1, a);
    uint32_t uq = rp_f2u(q1) & 0x7fffffffu;
    uint32_t ex = (uq >> 23) - 23u - ((uq & 0x007fffffu) == 0u ? 1u : 0u);
    float spacing = rp_u2f(ex << 23);
    *proven = (a == 0.0f) || ((uq >> 23) >= 30u && 2.0f * rp_absf(r1) < rp_absf(b) * spacing);
    return q1;
}

/* the same quotient through one f64 multiply: rd = 1.0 / (double)b.  a * rd carries a relative error
 * <= 2^-52 while an f32 quotient of f32 operands stays >= 2^-49 (relative) away from every rounding
 * midpoint, so rounding the product to f32 gives RN(a / b).  Same operand guarantees as above. */
RP_HD float rp_div_by_recip64(float a, double rd) { return (float)((double)a * rd); }
RP_HD int rp_div_by_recip_ok(float a) { return a == 0.0f || rp_absf(a) >= 8.6736174e-19f; /* 2^-60 */ }

/* powf(t, 1.5) and powf(t, 0.5) of DiscountedRegret's ALPHA / BETA (crates/mccfr/src/regret/discounted.rs:12-13,33,37) are
 * per-epoch scalars: the HOST computes them as a build of the reference does (include/rp_libm_glibc.h: rp_pow15 = glibc's powf
 * restated, rp_pow05 = sqrtf, which is what LLVM makes of pow(x, 0.5)) and hands them to the kernels as parameters; there is no
 * device powf. */

/* ---------------------------------------------------------------- RNG ----
 * The reference builds a fresh SmallRng per sampled node from
 * SipHash(epoch, info, tree-id) (flow.rs:285-295).  Same structure here with
 * a documented 64-bit mixer: one 64-bit hash per (seed, epoch, tree, key).
 * key = infoset id at player nodes, 0x80000000|state id at chance nodes. */
RP_HD uint64_t rp_mix64(uint64_t z) { /* SplitMix64 finalizer */
    z ^= z >> 30;
    z *= 0xbf58476d1ce4e5b9ull;
    z ^= z >> 27;
    z *= 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z;
}
/* rp_node_hash in two halves: the (seed, epoch) half is the same for every draw of a step (a host computes it once per
 * launch), the (tree, key) half is per draw.  rp_node_hash is their composition by definition. */
RP_HD uint64_t rp_node_hash_step(uint64_t seed, uint64_t epoch) {
    uint64_t h = rp_mix64(seed + 0x9e3779b97f4a7c15ull);
    return rp_mix64(h ^ (epoch * 0xd1342543de82ef95ull + 0x632be59bd9b4e019ull));
}
/* ... and the per-draw half in two again: a kernel that draws several times for one tree hashes the tree once */
RP_HD uint64_t rp_node_hash_tree(uint64_t step_hash, uint64_t tree) {
    return rp_mix64(step_hash ^ (tree * 0xaf251af3b0f025b5ull + 0x2545f4914f6cdd1dull));
}
/* the per-DRAW half is 32-bit arithmetic (round 4): a draw reads only the top 32 bits of its hash (rp_u01: 24, rp_pick_uniform:
 * 32), and on gfx950 an integer multiply issues at a quarter of the VALU rate — the 64-bit mixer's eight multiplies per draw were a
 * fifth of the headline traversal's issue time.  Murmur3's 32-bit finaliser (full avalanche, a bijection of its 32-bit argument) over
 * the tree hash's low word and the key FOLDED to 32 bits (low word ^ high word), the high word multiplied in afterwards: three
 * multiplies.  Distinct keys of one tree never collide as long as they are distinct after the fold — always for the dense solvers,
 * whose keys (infoset << 8 | salt) fit 32 bits; the NLHE traversal hands in 64-bit infoset hashes, where two nodes of one tree share
 * a draw with probability 2^-32 per pair (both draws stay valid samples of their distributions; the oracle folds the same way). */
RP_HD uint32_t rp_fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
RP_HD uint64_t rp_node_hash_key(uint64_t tree_hash, uint64_t key) {
    const uint32_t lo = (uint32_t)tree_hash, hi = (uint32_t)(tree_hash >> 32);
    const uint32_t k = (uint32_t)key ^ (uint32_t)(key >> 32);
    uint32_t x = rp_fmix32(lo ^ k);
    x = (x ^ hi) * 0x9e3779b1u;
    x ^= x >> 15;
    return (uint64_t)x << 32;
}
RP_HD uint64_t rp_node_hash_draw(uint64_t step_hash, uint64_t tree, uint64_t key) {
    return rp_node_hash_key(rp_node_hash_tree(step_hash, tree), key);
}
RP_HD uint64_t rp_node_hash(uint64_t seed, uint64_t epoch, uint64_t tree, uint64_t key) {
    return rp_node_hash_draw(rp_node_hash_step(seed, epoch), tree, key);
}
/* uniform f32 in [0,1): top 24 bits (rand's random::<f32>() shape) */
RP_HD float rp_u01(uint64_t h) { return (float)(uint32_t)(h >> 40) * 5.9604644775390625e-8f; }
/* uniform index in 0..n (rand's random_range shape, multiply-shift without rejection) */
RP_HD uint32_t rp_pick_uniform(uint64_t h, uint32_t n) {
    return (uint32_t)(((h >> 32) * (uint64_t)n) >> 32);
}
/* xoshiro-free stream for the k-means++ seeding loop: counter-based */
RP_HD uint64_t rp_stream(uint64_t seed, uint64_t counter) {
    return rp_mix64(rp_mix64(seed + 0x9e3779b97f4a7c15ull) ^ (counter * 0xd1342543de82ef95ull + 1ull));
}
/* k-means++ weighted draw (crates/lloyd/src/layer.rs:160-178 uses WeightedIndex<f32>,
 * whose f32 cumulative sums are order dependent).  Here each potential is quantised
 * to a 2^-36 fixed-point integer so prefix sums are exact and order independent
 * (a parallel scan on the GPU and a serial loop on the CPU agree bit for bit);
 * the draw is r = floor(h * total / 2^64), winner = first i with prefix_incl(i) > r. */
RP_HD uint64_t rp_kpp_quant(float p) {
    if (!(p > 0.0f)) return 0ull;
    return (uint64_t)(p * 68719476736.0f);
}
RP_HD uint64_t rp_mulhi64(uint64_t a, uint64_t b) {
    uint64_t a0 = a & 0xffffffffull, a1 = a >> 32;
    uint64_t b0 = b & 0xffffffffull, b1 = b >> 32;
    uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
    uint64_t mid = (p00 >> 32) + (p01 & 0xffffffffull) + (p10 & 0xffffffffull);
    return p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
}

#endif /* RP_MATH_H */

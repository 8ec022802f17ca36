/* rp_libm_glibc.h — the platform libm's expf / logf / powf, restated.
 *
 * The reference's Sinkhorn calls f32::exp and f32::ln (crates/lloyd/src/sinkhorn.rs:115,120-127,136; phi.rs:36), which Rust's std
 * forwards to the platform's libm.  On Linux that is glibc, whose expf / logf have been, since 2.27 / 2.28, the algorithms of Arm's
 * Optimized Routines (Szabolcs Nagy, 2017; sysdeps/ieee754/flt-32/{e_expf.c, e_exp2f_data.c, e_logf.c, e_logf_data.c}): double-precision
 * arithmetic around a small table, the result rounded to float once.
 *   expf   x N / ln2 = k + r, |r| <= 1/2 (N = 32); 2^(k/N) from a 32-entry table of correctly rounded 2^(i/32) (computed here, not
 *          copied: scripts/glibc_tables.py), times a cubic in r.
 *   logf   x = 2^k z, z in [OFF, 2 OFF); log(z) = log1p(z / c - 1) + log(c) with (1/c, log c) from a 16-entry table, a cubic in r.
 *          The sixteen 1/c are the authors' choice (data); log c = RN(-ln(1/c)) is recomputed by scripts/glibc_tables.py.
 *   powf   2^(y log2 x): log2 x like logf (same 1/c, log2 c = RN(-log2(1/c)) recomputed likewise, a quartic), then expf's table.
 * x86-64 glibc selects, on every CPU with FMA, the variant compiled with -mfma, where the multiply-adds below are fused; they are
 * spelled fma() here, so the functions do not depend on the compiler's contraction.  tests/test_libm_glibc.py sweeps ALL 2^32 float
 * bit patterns of expf and logf against the machine's own libm, and powf over every positive float for DiscountedRegret's two
 * exponents: zero mismatches on glibc 2.35 / x86-64 with FMA (the one fusion that matters is r = x N / ln2 - k in expf: unfused, two
 * of the 2^32 inputs differ in the last bit — x = 0x1.04845ep+5 and x = -0x1.f8cbb2p+5, which is what a pre-FMA x86 host would return
 * there; logf matches either way).
 *
 * What this is for: it pins the last third-party boundary (exp / ln of the lloyd path, powf of DiscountedRegret) to a published
 * algorithm, like include/rp_refrng.h does for the hash and the generator.
 *   powf   IS the build's contract: t^1.5 and t^0.5 are per-epoch scalars, computed on the host (rp_pow15 = rp_glibc_powf(t, 1.5),
 *          rp_pow05 = sqrtf(t): what LLVM makes of the two calls, see below) and handed to the kernels as parameters — DCFR's
 *          discounts are those of a Rust build on glibc, bit for bit (the earlier t * sqrt(t) differed from powf(t, 1.5) in the last
 *          bit on 24 % of the epochs).
 *   exp/ln the DEFAULT device arithmetic stays include/rp_math.h's rp_expf / rp_logf (f32 only, <= 1 ulp from these:
 *          tests/test_libm_glibc.py; cheaper in the softmin loops); every lloyd kernel is also compiled on THESE functions
 *          (csrc/lloyd_kernels.hpp, namespace lm_glibc; rp_kmeans_set_libm / rp_sinkhorn_set_libm), and the oracle runs on them under
 *          ora_lloyd_set_libm(2) (equal to the platform's libm = mode 1): the checker of that pass, and the measure of what the
 *          default's <= 1 ulp is worth.
 */
#ifndef RP_LIBM_GLIBC_H
#define RP_LIBM_GLIBC_H

#include <math.h>
#include <stdint.h>

#include "rp_math.h"

RP_HD uint64_t rp_d2u(double d) {
    uint64_t u;
    memcpy(&u, &d, 8);
    return u;
}
RP_HD double rp_u2d(uint64_t u) {
    double d;
    memcpy(&d, &u, 8);
    return d;
}

/* T[i] = bits(RN(2^(i/32))) - (i << 47): the exponent field is added back from k (e_exp2f_data.c).  The initialiser is a macro so that
 * the device kernels can keep the same 32 numbers in LDS (robopoker_amd/csrc/lm_glibc_dev.hpp). */
#define RP_GLIBC_EXP2F_TAB_INIT                                                                                                              \
    {0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, \
     0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, \
     0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, \
     0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, \
     0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, \
     0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull}
RP_HD uint64_t rp_glibc_exp2f_tab(uint32_t i) {
    const uint64_t T[32] = RP_GLIBC_EXP2F_TAB_INIT;
    return T[i & 31u];
}
RP_HD float rp_glibc_expf(float x) {
    const double N = 32.0, InvLn2N = 0x1.71547652b82fep+0 * N, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
    const uint32_t ux = rp_f2u(x), abstop = (ux >> 20) & 0x7ffu;
    if (abstop >= (0x42b00000u >> 20)) { /* |x| >= 88 or NaN */
        if (ux == 0xff800000u) return 0.0f;
        if (abstop >= (0x7f800000u >> 20)) return x + x;
        if (x > 0x1.62e42ep6f) return rp_u2f(0x7f800000u); /* x > log(0x1p128): overflow */
        if (x < -0x1.9fe368p6f) return 0.0f;               /* x < log(0x1p-150): underflow */
    }
    const double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + SHIFT; /* round to nearest integer, ties to even */
    const uint64_t ki = rp_d2u(kd);
    kd -= SHIFT;
    const double r = fma(InvLn2N, xd, -kd); /* the -mfma build fuses the product into this subtraction (found by the 2^32 sweep) */
    uint64_t t = rp_glibc_exp2f_tab((uint32_t)ki);
    t += ki << (52 - 5);
    const double s = rp_u2d(t);
    z = fma(C0, r, C1);
    const double r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(z, r2, y);
    y = y * s;
    return (float)y;
}
/* logf's table: sixteen (1/c, log c) pairs (e_logf_data.c) */
#define RP_GLIBC_LOGF_TAB_INIT                                                                                                                \
    {{0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2}, \
     {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},   \
     {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, \
     {0x1p+0, 0x0p+0},                              {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},   \
     {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},   {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  \
     {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}}
RP_HD float rp_glibc_logf(float x) {
    const double LT[16][2] = RP_GLIBC_LOGF_TAB_INIT;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2, Ln2 = 0x1.62e42fefa39efp-1;
    uint32_t ix = rp_f2u(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) { /* x < 0x1p-126, inf or NaN */
        if (ix * 2u == 0u) return rp_u2f(0xff800000u); /* log(0) = -inf */
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return rp_u2f(0x7fc00000u); /* negative or NaN */
        ix = rp_f2u(x * 0x1p23f); /* subnormal: normalise */
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> (23 - 4)) & 15u;
    const int32_t k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = LT[i][0], logc = LT[i][1], z = (double)rp_u2f(iz);
    const double r = fma(z, invc, -1.0);
    const double y0 = fma((double)k, Ln2, logc);
    const double r2 = r * r;
    double y = fma(A1, r, A2);
    y = fma(A0, r2, y);
    y = fma(y, r2, y0 + r);
    return (float)y;
}

/* ---- The same two functions shaped for a wavefront (round 6).
 * rp_glibc_expf above is glibc's control flow: a ladder of special cases in front of the arithmetic, which a GPU compiles into ~25
 * exec-mask instructions per call, and a table the compiler leaves in global memory.  The functions below return THE SAME BITS for
 * every input (swept over all 2^32 patterns against the ones above: tests/test_libm_glibc.py on the host, rp_libm_glibc_sweep on the
 * device) without a branch: the argument is clamped to a range on which the arithmetic alone gives the ladder's answers —
 *   x >= 89.5 and +inf  : y = e^89.5 > 2^128, and the double -> float conversion overflows to +inf like the ladder's `return inf`
 *   x <= -104 and -inf  : y < 2^-150, the conversion rounds to +0 like `return 0`
 *   NaN                 : fmaxf drops it (-104); the one compare + select at the end puts x + x back
 * and the table pointer is a parameter, so that a kernel can hand in a copy it keeps in LDS (csrc/lm_glibc_dev.hpp).
 * kd = rint(z) is (z + 0x1.8p52) - 0x1.8p52 for |z| < 2^51 (here |z| < 4200) and k its integer value, so ki's low 17 bits are k's.
 * The _floor form is max(expf(x), f32::MIN_POSITIVE) (sinkhorn.rs:119-128): below -88 expf is under MIN_POSITIVE whatever its bits,
 * and a NaN term yields MIN_POSITIVE (f32::max), which is what the clamp to -88 gives. */
RP_HD double rp_glibc_exp_core(float xc, const uint64_t* T) { /* xc in [-104, 89.5] */
    const double N = 32.0, InvLn2N = 0x1.71547652b82fep+0 * N;
    const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
    const double xd = (double)xc;
    const double z = InvLn2N * xd;
#if defined(__HIP_DEVICE_COMPILE__)
    const double kd = __builtin_rint(z); /* v_rndne_f64 */
#else
    const double kd = rint(z);
#endif
    const int32_t k = (int32_t)kd;
    const double r = fma(InvLn2N, xd, -kd);
    const uint64_t t0 = T[(uint32_t)k & 31u];
    const uint32_t hi = (uint32_t)(t0 >> 32) + ((uint32_t)k << 15); /* t += ki << 47: only the high word moves */
    const double s = rp_u2d(((uint64_t)hi << 32) | (uint32_t)t0);
    const double p = fma(C0, r, C1);
    const double r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(p, r2, y);
    return y * s;
}
RP_HD float rp_glibc_expf_tab(float x, const uint64_t* T) { /* == rp_glibc_expf(x) */
#if defined(__HIP_DEVICE_COMPILE__)
    const float xc = __builtin_fminf(__builtin_fmaxf(x, -104.0f), 89.5f);
#else
    const float xc = x != x ? -104.0f : (x < -104.0f ? -104.0f : (x > 89.5f ? 89.5f : x));
#endif
    const float y = (float)rp_glibc_exp_core(xc, T);
    return x != x ? x + x : y;
}
RP_HD float rp_glibc_exp_floor_tab(float x, const uint64_t* T) { /* == rp_maxf(rp_glibc_expf(x), RP_EPSILON) */
#if defined(__HIP_DEVICE_COMPILE__)
    const float xc = __builtin_fminf(__builtin_fmaxf(x, -88.0f), 89.5f);
#else
    const float xc = x != x ? -88.0f : (x < -88.0f ? -88.0f : (x > 89.5f ? 89.5f : x));
#endif
    return rp_maxf((float)rp_glibc_exp_core(xc, T), RP_EPSILON);
}
/* logf: one test in front (zero, subnormal, negative, inf, NaN: never on the Sinkhorn's path, whose arguments are sums of terms
 * >= MIN_POSITIVE) instead of the ladder; log(1) = +0 comes out of the arithmetic (r = 0, k = 0, log c = 0). */
RP_HD float rp_glibc_logf_tab(float x, const double (*LT)[2]) { /* == rp_glibc_logf(x) */
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2, Ln2 = 0x1.62e42fefa39efp-1;
    const uint32_t ix = rp_f2u(x);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) return rp_glibc_logf(x);
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> (23 - 4)) & 15u;
    const int32_t k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = LT[i][0], logc = LT[i][1], z = (double)rp_u2f(iz);
    const double r = fma(z, invc, -1.0);
    const double y0 = fma((double)k, Ln2, logc);
    const double r2 = r * r;
    double y = fma(A1, r, A2);
    y = fma(A0, r2, y);
    y = fma(y, r2, y0 + r);
    return (float)y;
}

/* powf(x, y), all of e_powf.c: log2(x) in double from a 16-entry table (the same 1/c as logf's, with log2 c = RN(-log2(1/c)): recomputed
 * by scripts/glibc_tables.py) and a quartic, times y, then 2^(.) through expf's table; in front of it the special-case ladder (zeros,
 * infinities, NaNs, negative bases with integer exponents).  DiscountedRegret uses t^1.5 with t = epoch as f32 (discounted.rs:33) —
 * epoch 0 included: powf(+0, 1.5) = +0, a discount of 0. */
RP_HD int rp_glibc_checkint(uint32_t iy) { /* 0: not an integer, 1: odd, 2: even */
    const int e = (int)((iy >> 23) & 0xffu);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
RP_HD float rp_glibc_powf(float x, float y) {
    const double LT[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2}, {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},
        {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2}, {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4}, {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5},
        {0x1p+0, 0x0p+0},                              {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
        {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},  {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},
        {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1, SHIFT = 0x1.8p+52 / 32.0;
    uint32_t ix = rp_f2u(x);
    const uint32_t iy = rp_f2u(y);
    uint64_t sign_bias = 0;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || 2u * iy - 1u >= 2u * 0x7f800000u - 1u) {
        /* x < 0x1p-126, inf or NaN; or y is 0, inf or NaN */
        if (2u * iy - 1u >= 2u * 0x7f800000u - 1u) {
            if (2u * iy == 0u) return 2u * (ix ^ 0x00400000u) > 2u * 0x7fc00000u ? x + y : 1.0f; /* x^0 = 1 unless x is a signalling NaN */
            if (ix == 0x3f800000u) return 2u * (iy ^ 0x00400000u) > 2u * 0x7fc00000u ? x + y : 1.0f; /* 1^y = 1, likewise */
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y; /* NaN in, NaN out */
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;                               /* (-1)^(+-inf) */
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;       /* |x| < 1, y = inf or |x| > 1, y = -inf */
            return y * y;
        }
        if (2u * ix - 1u >= 2u * 0x7f800000u - 1u) { /* x is 0, inf or NaN */
            float x2 = x * x;
            if ((ix & 0x80000000u) && rp_glibc_checkint(iy) == 1) x2 = -x2;
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        if (ix & 0x80000000u) { /* finite x < 0 */
            const int yint = rp_glibc_checkint(iy);
            if (yint == 0) return rp_u2f(0x7fc00000u) /* invalid */;
            if (yint == 1) sign_bias = 1ull << (5 + 11);
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) { /* subnormal x: normalise so that the exponent goes negative */
            ix = rp_f2u(rp_u2f(ix) * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    /* log2(x) = log1p(z / c - 1) / ln2 + log2(c) + k */
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> (23 - 4)) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int32_t k = (int32_t)top >> 23;
    const double invc = LT[i][0], logc = LT[i][1], z = (double)rp_u2f(iz);
    const double r = fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double yy = fma(A0, r, A1);
    const double p = fma(A2, r, A3);
    const double r4 = r2 * r2;
    double q = fma(A4, r, y0);
    q = fma(p, r2, q);
    yy = fma(yy, r4, q);
    const double ylogx = (double)y * yy; /* cannot overflow: y is single precision */
    if (((rp_d2u(ylogx) >> 47) & 0xffffu) >= (rp_d2u(126.0) >> 47)) { /* |y log2 x| >= 126 */
        if (ylogx > 0x1.fffffffd1d571p+6) return rp_u2f(sign_bias ? 0xff800000u : 0x7f800000u); /* overflow */
        if (ylogx <= -150.0) return rp_u2f(sign_bias ? 0x80000000u : 0u);                          /* underflow */
    }
    /* 2^(y log2 x): x = k/N + r, |r| <= 1/(2N) */
    double kd = ylogx + SHIFT;
    const uint64_t ki = rp_d2u(kd);
    kd -= SHIFT;
    const double rr = ylogx - kd;
    uint64_t t = rp_glibc_exp2f_tab((uint32_t)ki);
    t += (ki + sign_bias) << (52 - 5);
    const double s = rp_u2d(t);
    const double zz = fma(C0, rr, C1);
    const double rr2 = rr * rr;
    double e = fma(C2, rr, 1.0);
    e = fma(zz, rr2, e);
    e = e * s;
    return (float)e;
}

/* DiscountedRegret's two powers (discounted.rs:33,37) of t = epoch as f32, t >= 1, as a build of the reference computes them.  Rust's
 * f32::powf is the llvm.pow.f32 intrinsic, and the exponents are associated consts (discounted.rs:12-13), so at the workspace's
 * opt-level = 3 (Cargo.toml: dev AND release) LLVM's libcall simplifier sees pow(x, 0.5) and pow(x, 1.5):
 *   pow(x, 0.5) -> fabs(sqrt(x)), with -inf -> +inf   (replacePowWithSqrt: no fast-math flag needed on the errno-free intrinsic;
 *                                                      checked with this image's LLVM: `sqrtss`, no call)  = sqrtf(t) for t >= 1
 *   pow(x, 1.5) -> stays a call to libm's powf        (the n + 0.5 expansion needs `afn`)                   = glibc's powf
 * glibc's powf(t, 0.5) would differ from sqrtf(t) in the last bit on 11 294 of the first 2^24 epochs. */
RP_HD float rp_pow15(float t) { return rp_glibc_powf(t, 1.5f); }
RP_HD float rp_pow05(float t) { return sqrtf(t); }

#endif /* RP_LIBM_GLIBC_H */

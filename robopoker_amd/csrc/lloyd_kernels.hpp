// lloyd_kernels.hpp — every device function and kernel of the lloyd path.  NO include guard: lloyd.hip includes this file twice, inside
//   namespace rp::lm_contract  (LM_GLIBC 0)  exp / ln = include/rp_math.h's rp_expf / rp_logf / rp_exp_floor2: the build's f32 contract
//   namespace rp::lm_glibc     (LM_GLIBC 1)  exp / ln = glibc's expf / logf evaluated in double (include/rp_libm_glibc.h, equal to glibc
//                                            2.35's on all 2^32 inputs): f32::exp / f32::ln of a Rust build on Linux, so a layer clustered
//                                            in this pass is the reference's bit for bit (rp_kmeans_set_libm / rp_sinkhorn_set_libm)
// The two passes are the same source, term for term and in the same order; only the spellings below differ.  The glibc pass evaluates
// the header's branch-free forms on tables it keeps in LDS (lm_glibc_dev.hpp): every kernel that reaches an exponential or a logarithm
// starts with LM_TABLES().
#if LM_GLIBC
#define LM_EXPF(x) rp_glibc_expf_tab(x, lmg::lds_exp)
#define LM_LOGF(x) rp_glibc_logf_tab(x, lmg::lds_log)
#define LM_EXP_FLOOR2(v) lm_exp_floor2_glibc(v)
#define LM_TABLES() lmg::tables_init()
// max(exp(x), MIN_POSITIVE) of two terms (sinkhorn.rs:119-128), as the oracle spells it on glibc's expf
__device__ __forceinline__ rp_f2 lm_exp_floor2_glibc(rp_f2 x) {
    rp_f2 r;
    r.x = rp_glibc_exp_floor_tab(x.x, lmg::lds_exp);
    r.y = rp_glibc_exp_floor_tab(x.y, lmg::lds_exp);
    return r;
}
#else
#define LM_EXPF(x) rp_expf(x)
#define LM_LOGF(x) rp_logf(x)
#define LM_EXP_FLOOR2(v) rp_exp_floor2(v)
#define LM_TABLES() ((void)0)
#endif




__device__ __forceinline__ void kpp_note(const Metric& M, uint64_t i, uint32_t k, float d) {
    if (M.kpp_d && d < M.kpp_d[i]) {  // strict: the first minimum in centroid order stays (a NaN never enters, -1 never leaves)
        M.kpp_d[i] = d;
        M.kpp_j[i] = (uint8_t)k;
    }
    if (M.kpp_claim) {  // a sampled point the interval filter dropped in this round: its bound against the distance just solved
        const float c = M.kpp_claim[i];
        if (c > 0.0f) {
            if (d < c) atomicAdd(M.kpp_bad, 1ull);
            M.kpp_claim[i] = 0.0f;
        }
    }
}
__device__ __forceinline__ unsigned long long* STAT(const Metric& M, uint32_t k) {
    return M.stats + (size_t)(blockIdx.x % M.stat_stripes) * STAT_STRIDE + k;
}




struct __attribute__((aligned(16))) WaveLds {
    uint16_t supA[MAXB];
    uint16_t supB[MAXB];
    float lnA[MAXB];
    float lnB[MAXB];
    float f[MAXB];
    float g[MAXB];
    float tmp[MAXB];
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
#ifndef SPLIT_ROWS
#define SPLIT_ROWS 1  // 0: every softmin row on one lane (the shape the split is checked against)
#endif

#if !LM_GLIBC  // the scaling-domain filters have no exponential of their own: one copy (the contract namespace) serves both passes,
               // their margins audited at full size in both arithmetics (profiles/r05_glibc_audit.json, rp_kmeans_set_prune)
#include "sinkhorn_bound.hpp"
#include "kpp_bound.hpp"
#include "kpp_refpick.hpp"
#endif

// Bins::support + Bins::density (bins.rs:58-60,84-88) of a dense histogram into LDS; returns the support size
template <typename CT>
__device__ uint32_t wave_load_hist(const CT* counts, uint32_t weight, uint32_t bins, uint16_t* sup, float* lnd) {
    const uint32_t lane = lane_id();
    const float fw = (float)weight;
    uint32_t base = 0;
    for (uint32_t q = 0; q * 64 < bins; ++q) {
        const uint32_t b = q * 64 + lane;
        const uint32_t c = b < bins ? (uint32_t)counts[b] : 0u;
        const bool has = c > 0;
        const unsigned long long mask = __ballot(has);
        if (has) {
            const uint32_t r = base + __popcll(mask & ((1ull << lane) - 1ull));
            sup[r] = (uint16_t)b;
            lnd[r] = LM_LOGF((float)c / fw);
        }
        base += __popcll(mask);
    }
    __syncthreads();
    return base;
}

__device__ uint32_t wave_load_centroid(const CentroidSet& cs, uint32_t k, uint16_t* sup, float* lnd) {
    const uint32_t n = cs.n[k];
    for (uint32_t i = lane_id(); i < n; i += 64) {
        sup[i] = cs.sup[(size_t)k * MAXB + i];
        lnd[i] = cs.lnd[(size_t)k * MAXB + i];
    }
    __syncthreads();
    return n;
}

// Sinkhorn::from(mu, nu, metric).minimize().cost() (sinkhorn.rs:77-92,194-230).  A = mu, B = nu, supports and
// log-densities already in LDS.  All 64 lanes return the same value.
// sum_j max(exp(pot[j] - Rt[sup[j]][x]), MIN_POSITIVE), j ascending: the reference's left fold (sinkhorn.rs:119-128).
// `sup`/`pot` are the OTHER side's support and potential (LDS, wave uniform), `xi` = this lane's bin.
// The row base of Rt is uniform, so it is formed on the scalar unit (readfirstlane of two packed u16 bins -> SALU
// shifts/adds) and the load is `global_load saddr + voffset`: no per-term VALU address arithmetic.  Exponentials go
// through the packed-f32 pipe two at a time (rp_exp_floor2); the adds stay sequential.
// C/T through a buffer descriptor: `buffer_load_dword v, voffset, rsrc, soffset` takes the (uniform) row offset
// from an SGPR and the lane's column offset from a loop-invariant VGPR, so a term needs no VALU address arithmetic.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rt_resource(const Metric& M) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(M.Rt), 0, (int)(M.bins * M.bins * 4u), 0x00020000);
}
__device__ __forceinline__ float rt_load(__amdgpu_buffer_rsrc_t rt, uint32_t col_bytes, uint32_t row_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, (int)col_bytes, (int)row_bytes, 0));
}
// left fold t[0] + t[1] + ... + t[n-1] from 0.0f in index order, reading four floats per LDS access (t 16-B aligned;
// the array extends to a multiple of four)
__device__ __forceinline__ float lds_sum_in_order(const float* t, uint32_t n) {
    float e = 0.0f;
    uint32_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const float4 v = *reinterpret_cast<const float4*>(t + i);
        e += v.x; e += v.y; e += v.z; e += v.w;
    }
    if (i < n) {
        const float4 v = *reinterpret_cast<const float4*>(t + i);
        e += v.x;
        if (i + 1 < n) e += v.y;
        if (i + 2 < n) e += v.z;
    }
    return e;
}
struct SoftminGroup {  // 8 consecutive terms: potentials + the C/T entries of this lane's column
    float4 p0, p1;
    float r[8];
};
__device__ __forceinline__ void softmin_fetch(SoftminGroup& gq, const uint16_t* sup, const float* pot, uint32_t j,
                                              __amdgpu_buffer_rsrc_t rt, uint32_t rowb, uint32_t xoff) {
    const uint4 sp = *reinterpret_cast<const uint4*>(sup + j);  // 8 bins (LDS arrays are 16-B aligned, j % 8 == 0)
    gq.p0 = *reinterpret_cast<const float4*>(pot + j);
    gq.p1 = *reinterpret_cast<const float4*>(pot + j + 4);
    const uint32_t w0 = __builtin_amdgcn_readfirstlane(sp.x), w1 = __builtin_amdgcn_readfirstlane(sp.y);
    const uint32_t w2 = __builtin_amdgcn_readfirstlane(sp.z), w3 = __builtin_amdgcn_readfirstlane(sp.w);
    gq.r[0] = rt_load(rt, xoff, (w0 & 0xffffu) * rowb);
    gq.r[1] = rt_load(rt, xoff, (w0 >> 16) * rowb);
    gq.r[2] = rt_load(rt, xoff, (w1 & 0xffffu) * rowb);
    gq.r[3] = rt_load(rt, xoff, (w1 >> 16) * rowb);
    gq.r[4] = rt_load(rt, xoff, (w2 & 0xffffu) * rowb);
    gq.r[5] = rt_load(rt, xoff, (w2 >> 16) * rowb);
    gq.r[6] = rt_load(rt, xoff, (w3 & 0xffffu) * rowb);
    gq.r[7] = rt_load(rt, xoff, (w3 >> 16) * rowb);
}
__device__ __forceinline__ float softmin_fold(float s, const SoftminGroup& gq) {
    rp_f2 e0, e1, e2, e3;
    e0.x = gq.p0.x - gq.r[0]; e0.y = gq.p0.y - gq.r[1];
    e1.x = gq.p0.z - gq.r[2]; e1.y = gq.p0.w - gq.r[3];
    e2.x = gq.p1.x - gq.r[4]; e2.y = gq.p1.y - gq.r[5];
    e3.x = gq.p1.z - gq.r[6]; e3.y = gq.p1.w - gq.r[7];
    e0 = LM_EXP_FLOOR2(e0);
    e1 = LM_EXP_FLOOR2(e1);
    e2 = LM_EXP_FLOOR2(e2);
    e3 = LM_EXP_FLOOR2(e3);
    s += e0.x; s += e0.y; s += e1.x; s += e1.y; s += e2.x; s += e2.y; s += e3.x; s += e3.y;
    return s;
}
// the first r (1..7, wave uniform) terms of a group
__device__ __forceinline__ float softmin_fold_first(float s, const SoftminGroup& gq, uint32_t r) {
    rp_f2 e0, e1, e2, e3;
    e0.x = gq.p0.x - gq.r[0]; e0.y = gq.p0.y - gq.r[1];
    e1.x = gq.p0.z - gq.r[2]; e1.y = gq.p0.w - gq.r[3];
    e2.x = gq.p1.x - gq.r[4]; e2.y = gq.p1.y - gq.r[5];
    e3.x = gq.p1.z - gq.r[6]; e3.y = gq.p1.w - gq.r[7];
    e0 = LM_EXP_FLOOR2(e0);
    s += e0.x;
    if (r > 1) s += e0.y;
    if (r > 2) {
        e1 = LM_EXP_FLOOR2(e1);
        s += e1.x;
        if (r > 3) s += e1.y;
    }
    if (r > 4) {
        e2 = LM_EXP_FLOOR2(e2);
        s += e2.x;
        if (r > 5) s += e2.y;
    }
    if (r > 6) {
        e3 = LM_EXP_FLOOR2(e3);
        s += e3.x;
    }
    return s;
}
// softmin_sum with a PER-LANE walk: each lane folds the cnt terms of ITS OWN (sup, pot) arrays (lane groups of a
// wavefront working on different solves), so the C/T row offset is a VGPR.  Same terms, same order as softmin_sum.
__device__ __forceinline__ float softmin_sum_lane(const uint16_t* sup, const float* pot, uint32_t cnt,
                                                   __amdgpu_buffer_rsrc_t rt, uint32_t bins, uint32_t xi) {
    const uint32_t rowb = bins * 4u, xoff = xi * 4u;
    float s = 0.0f;
    for (uint32_t j = 0; j < cnt; j += 8) {
        const uint4 sp = *reinterpret_cast<const uint4*>(sup + j);
        SoftminGroup gq;
        gq.p0 = *reinterpret_cast<const float4*>(pot + j);
        gq.p1 = *reinterpret_cast<const float4*>(pot + j + 4);
        gq.r[0] = rt_load(rt, (sp.x & 0xffffu) * rowb + xoff, 0);
        gq.r[1] = rt_load(rt, (sp.x >> 16) * rowb + xoff, 0);
        gq.r[2] = rt_load(rt, (sp.y & 0xffffu) * rowb + xoff, 0);
        gq.r[3] = rt_load(rt, (sp.y >> 16) * rowb + xoff, 0);
        gq.r[4] = rt_load(rt, (sp.z & 0xffffu) * rowb + xoff, 0);
        gq.r[5] = rt_load(rt, (sp.z >> 16) * rowb + xoff, 0);
        gq.r[6] = rt_load(rt, (sp.w & 0xffffu) * rowb + xoff, 0);
        gq.r[7] = rt_load(rt, (sp.w >> 16) * rowb + xoff, 0);
        rp_f2 e0, e1, e2, e3;
        e0.x = gq.p0.x - gq.r[0]; e0.y = gq.p0.y - gq.r[1];
        e1.x = gq.p0.z - gq.r[2]; e1.y = gq.p0.w - gq.r[3];
        e2.x = gq.p1.x - gq.r[4]; e2.y = gq.p1.y - gq.r[5];
        e3.x = gq.p1.z - gq.r[6]; e3.y = gq.p1.w - gq.r[7];
        e0 = LM_EXP_FLOOR2(e0);
        e1 = LM_EXP_FLOOR2(e1);
        e2 = LM_EXP_FLOOR2(e2);
        e3 = LM_EXP_FLOOR2(e3);
        const uint32_t r = cnt - j;  // this lane's terms left (>= 1)
        s += e0.x;
        s = r > 1 ? s + e0.y : s;
        s = r > 2 ? s + e1.x : s;
        s = r > 3 ? s + e1.y : s;
        s = r > 4 ? s + e2.x : s;
        s = r > 5 ? s + e2.y : s;
        s = r > 6 ? s + e3.x : s;
        s = r > 7 ? s + e3.y : s;
    }
    return s;
}
// the fold continues from (s, j): j a multiple of 8, s the left fold of the terms before j
template <bool PIPE = true>
__device__ __forceinline__ float softmin_sum_from(float s, uint32_t j, const uint16_t* sup, const float* pot, uint32_t cnt,
                                                   __amdgpu_buffer_rsrc_t rt, uint32_t bins, uint32_t xi) {
    const uint32_t rowb = bins * 4u, xoff = xi * 4u;
    if (!PIPE) {  // one group in flight: 16 fewer VGPRs (the two-point kernels keep 7 waves per SIMD with it)
        for (; j + 8 <= cnt; j += 8) {
            SoftminGroup cur;
            softmin_fetch(cur, sup, pot, j, rt, rowb, xoff);
            s = softmin_fold(s, cur);
        }
    } else if (j + 8 <= cnt) {  // software pipeline: the loads of group j+8 are in flight while group j is exponentiated
        SoftminGroup cur, nxt;
        softmin_fetch(cur, sup, pot, j, rt, rowb, xoff);
        for (j += 8; j + 8 <= cnt; j += 8) {
            softmin_fetch(nxt, sup, pot, j, rt, rowb, xoff);
            s = softmin_fold(s, cur);
            cur = nxt;
        }
        s = softmin_fold(s, cur);
    }
    if (j < cnt) {
        // 1..7 terms left: one more group with its eight loads in flight together, of which only the first cnt - j are
        // added (the others read whatever follows in the LDS arrays: a bin past the table is an out-of-range buffer
        // load, which returns 0).  rp_exp_floor2 == rp_exp_floor on every input (rp_math_exp_sweep), so a term's value
        // is the scalar path's.
        SoftminGroup last;
        softmin_fetch(last, sup, pot, j, rt, rowb, xoff);
        s = softmin_fold_first(s, last, cnt - j);
    }
    return s;
}
template <bool PIPE = true>
__device__ __forceinline__ float softmin_sum(const uint16_t* sup, const float* pot, uint32_t cnt,
                                              __amdgpu_buffer_rsrc_t rt, uint32_t bins, uint32_t xi) {
    return softmin_sum_from<PIPE>(0.0f, 0u, sup, pot, cnt, rt, bins, xi);
}

// ------------------------------------------------------------------------------------------------
// A support with <= 32 rows leaves half the wavefront without a row while it walks the other support (the point side of a
// point-against-centroid solve: up to 256 columns).  Lanes l and l ^ 32 then take the SAME row and alternate over the column groups:
// of every 16 columns the lower half exponentiates the first 8, the upper half the next 8 — the exponentials are the cost of a
// term — and the left fold stays the reference's: the running sum visits the lower half's eight terms, crosses to the upper half
// (v_permlane32_swap), visits its eight, and crosses back.  Same terms, same order, same additions as softmin_sum.
// ------------------------------------------------------------------------------------------------
template <uint32_t SRC>
__device__ __forceinline__ float half_take(float x) {  // the value lane (l & 31) + 32 SRC holds, in every lane l
#if defined(RP_EMUL)
    return __shfl(x, (int)((lane_id() & 31u) + 32u * SRC), 64);
#else
    const uint32_t w = __builtin_bit_cast(uint32_t, x);
    const auto r = __builtin_amdgcn_permlane32_swap(w, w, false, false);  // r[0]: the lower half's values in all lanes, r[1]: the upper's
    const uint32_t lo = r[0], hi = r[1];
    return __builtin_bit_cast(float, SRC ? hi : lo);
#endif
}
__device__ __forceinline__ float softmin_sum_split(const uint16_t* sup, const float* pot, uint32_t cnt, __amdgpu_buffer_rsrc_t rt,
                                                   uint32_t bins, uint32_t xi) {
    const uint32_t rowb = bins * 4u, xoff = xi * 4u, up = lane_id() >> 5;
    float s = 0.0f;
    uint32_t j = 0;
    for (; j + 16 <= cnt; j += 16) {
        const uint32_t jj = j + 8u * up;
        const uint4 sp = *reinterpret_cast<const uint4*>(sup + jj);
        const float4 p0 = *reinterpret_cast<const float4*>(pot + jj), p1 = *reinterpret_cast<const float4*>(pot + jj + 4);
        float r[8];
        r[0] = rt_load(rt, (sp.x & 0xffffu) * rowb + xoff, 0);
        r[1] = rt_load(rt, (sp.x >> 16) * rowb + xoff, 0);
        r[2] = rt_load(rt, (sp.y & 0xffffu) * rowb + xoff, 0);
        r[3] = rt_load(rt, (sp.y >> 16) * rowb + xoff, 0);
        r[4] = rt_load(rt, (sp.z & 0xffffu) * rowb + xoff, 0);
        r[5] = rt_load(rt, (sp.z >> 16) * rowb + xoff, 0);
        r[6] = rt_load(rt, (sp.w & 0xffffu) * rowb + xoff, 0);
        r[7] = rt_load(rt, (sp.w >> 16) * rowb + xoff, 0);
        rp_f2 e0, e1, e2, e3;
        e0.x = p0.x - r[0]; e0.y = p0.y - r[1];
        e1.x = p0.z - r[2]; e1.y = p0.w - r[3];
        e2.x = p1.x - r[4]; e2.y = p1.y - r[5];
        e3.x = p1.z - r[6]; e3.y = p1.w - r[7];
        e0 = LM_EXP_FLOOR2(e0);
        e1 = LM_EXP_FLOOR2(e1);
        e2 = LM_EXP_FLOOR2(e2);
        e3 = LM_EXP_FLOOR2(e3);
        float a = s;  // meaningful in the lower half: the fold over columns j .. j+7
        a += e0.x; a += e0.y; a += e1.x; a += e1.y; a += e2.x; a += e2.y; a += e3.x; a += e3.y;
        float b = half_take<0>(a);  // meaningful in the upper half: it goes on over columns j+8 .. j+15
        b += e0.x; b += e0.y; b += e1.x; b += e1.y; b += e2.x; b += e2.y; b += e3.x; b += e3.y;
        s = half_take<1>(b);
    }
    // fewer than 16 columns left: both halves fold them alike
    return softmin_sum_from<false>(s, j, sup, pot, cnt, rt, bins, xi);
}

// NTHR = 64: one wavefront per solve.  NTHR = 256: the rows of every half-iteration spread over the four wavefronts of a workgroup (the K
// centroid-against-centroid solves of Elkan::drift and the centroids' self costs: 256 solves of 256 x 256 supports, one wavefront each
// would leave three quarters of the chip idle and every one of them latency bound) — the same operations on the same rows, the folds
// (error sums, cost) evaluated by every work-item from LDS as before.
template <uint32_t NTHR = 64u>
__device__ __forceinline__ float wave_sinkhorn_cost(WaveLds& w, uint32_t m, uint32_t n, const Metric& M) {
    const uint32_t lane = NTHR == 64u ? lane_id() : threadIdx.x;
    if (m == 0 || n == 0) return -0.0f;  // empty support: the cost sum is empty, and f32's Sum folds from -0.0 (libcore since 1.83)
    const uint32_t bins = M.bins;
    const __amdgpu_buffer_rsrc_t rt = rt_resource(M);
    const float lu = LM_LOGF(1.0f / (float)m), ru = LM_LOGF(1.0f / (float)n);  // Potential::uniform (phi.rs:34-39)
    for (uint32_t i = lane; i < m; i += NTHR) w.f[i] = lu;
    for (uint32_t j = lane; j < n; j += NTHR) w.g[j] = ru;
    __syncthreads();
    // a side with <= 32 rows against >= 32 columns: two lanes per row (softmin_sum_split)
    const bool split_a = SPLIT_ROWS && NTHR == 64u && m <= 32u && n >= 32u, split_b = SPLIT_ROWS && NTHR == 64u && n <= 32u && m >= 32u;
    uint32_t t = 0;
    for (; t < M.iters; ++t) {
        // lhs(): f(x) <- ln mu(x) - ln sum_y max(exp(g(y) - C(x,y)/T), MIN_POSITIVE)   (sinkhorn.rs:94-102,119-128)
        if (split_a) {
            const uint32_t i = lane & 31u;
            const bool act = i < m;
            const uint32_t x = act ? w.supA[i] : w.supA[0];
            const float s = softmin_sum_split(w.supB, w.g, n, rt, bins, x);
            if (act && lane < 32u) {
                const float nf = w.lnA[i] - LM_LOGF(s);
                w.tmp[i] = rp_absf(LM_EXPF(nf) - LM_EXPF(w.f[i]));  // delta term (sinkhorn.rs:134-139)
                w.f[i] = nf;
            }
        } else
        for (uint32_t i0 = 0; i0 < m; i0 += NTHR) {
            if (NTHR > 64u && i0 + (lane & ~63u) >= m) break;  // this wavefront has no row in the pass (wave uniform)
            const uint32_t i = i0 + lane;
            const bool act = i < m;
            const uint32_t x = act ? w.supA[i] : w.supA[0];
            const float s = softmin_sum(w.supB, w.g, n, rt, bins, x);
            if (act) {
                const float nf = w.lnA[i] - LM_LOGF(s);
                w.tmp[i] = rp_absf(LM_EXPF(nf) - LM_EXPF(w.f[i]));  // delta term (sinkhorn.rs:134-139)
                w.f[i] = nf;
            }
        }
        __syncthreads();
        const float lhs_err = lds_sum_in_order(w.tmp, m);
        __syncthreads();
        // rhs(): sees the fresh lhs (Gauss-Seidel, sinkhorn.rs:80-87)
        if (split_b) {
            const uint32_t j = lane & 31u;
            const bool act = j < n;
            const uint32_t y = act ? w.supB[j] : w.supB[0];
            const float s = softmin_sum_split(w.supA, w.f, m, rt, bins, y);
            if (act && lane < 32u) {
                const float ng = w.lnB[j] - LM_LOGF(s);
                w.tmp[j] = rp_absf(LM_EXPF(ng) - LM_EXPF(w.g[j]));
                w.g[j] = ng;
            }
        } else
        for (uint32_t j0 = 0; j0 < n; j0 += NTHR) {
            if (NTHR > 64u && j0 + (lane & ~63u) >= n) break;
            const uint32_t j = j0 + lane;
            const bool act = j < n;
            const uint32_t y = act ? w.supB[j] : w.supB[0];
            const float s = softmin_sum(w.supA, w.f, m, rt, bins, y);
            if (act) {
                const float ng = w.lnB[j] - LM_LOGF(s);
                w.tmp[j] = rp_absf(LM_EXPF(ng) - LM_EXPF(w.g[j]));
                w.g[j] = ng;
            }
        }
        __syncthreads();
        const float rhs_err = lds_sum_in_order(w.tmp, n);
        __syncthreads();
        if (lhs_err + rhs_err < M.tol) {
            t += 1;
            break;
        }
    }
    if (lane == 0) {
        atomicAdd(STAT(M, 1), (unsigned long long)t);
        atomicAdd(STAT(M, 2), (unsigned long long)(2 * t + 1) * m * n);
    }
    // cost(): x-major left fold of coupling * distance (sinkhorn.rs:206-217)
    float cost = -0.0f;  // .sum::<Energy>() folds from -0.0 (sinkhorn.rs:216; same result unless the sum is empty)
    for (uint32_t i = 0; i < m; ++i) {
        const uint32_t x = w.supA[i];
        const float fi = w.f[i];
        for (uint32_t j = lane; j < n; j += NTHR) {
            const uint32_t y = w.supB[j];
            const float c = M.Cm[x * bins + y];
            w.tmp[j] = LM_EXPF(fi + w.g[j] - M.Rt[x * bins + y]) * c;
        }
        __syncthreads();
        for (uint32_t j = 0; j < n; ++j) cost += w.tmp[j];
        __syncthreads();
    }
    return cost;
}

// Sinkhorn::divergence (sinkhorn.rs:166-171) with memoised self terms
template <uint32_t NTHR = 64u>
__device__ __forceinline__ float wave_divergence(WaveLds& w, uint32_t m, uint32_t n, float selfA, float selfB,
                                                 const Metric& M) {
    const float xy = wave_sinkhorn_cost<NTHR>(w, m, n, M);
    if ((NTHR == 64u ? lane_id() : threadIdx.x) == 0) atomicAdd(STAT(M, 0), 1ull);
    return rp_maxf(xy - 0.5f * selfA - 0.5f * selfB, 0.0f);
}

// ------------------------------------------------------------------------------------------------
// G = 2 or 4 points against ONE centroid in one wavefront.
//
// A point has few support bins (synthetic flop-like points: <= 47, 28 on average; the REAL flop points: 11 on average,
// 27 at most): in the half-iteration whose rows are the point's bins a lane-per-row mapping leaves most of the wave
// idle while it walks the centroid's (up to 256) bins.  When G points with <= 64 / G bins each meet the SAME centroid,
// lane group g takes the rows of point g: the column walk (row offsets of C/T: wave uniform) is shared, only the
// potential a group reads differs.  The other half-iteration (rows = centroid bins) runs once per point as before.
// Each solve keeps its own iteration count: a converged solve is frozen while the others finish.  Every float
// operation of a solve is the one wave_sinkhorn_cost performs, in the same order.
// ------------------------------------------------------------------------------------------------
template <uint32_t G>
struct __attribute__((aligned(16))) GroupLds {
    static constexpr uint32_t ROWS = 64u / G;
    uint16_t supC[MAXB];       // centroid support
    float lnC[MAXB];
    float potC[G][MAXB];       // centroid-side potential of each solve
    float tmpC[G][MAXB];
    uint16_t supP[G][ROWS];    // the G points
    float lnP[G][ROWS];
    float potP[G][ROWS];
    float tmpP[G][ROWS];
};
// v[g] for a lane-varying g (small arrays stay in registers)
template <uint32_t G, typename T>
__device__ __forceinline__ T pick(const T (&v)[G], uint32_t g) {
    T r = v[0];
#pragma unroll
    for (uint32_t h = 1; h < G; ++h) r = g == h ? v[h] : r;
    return r;
}

// cost[h] = OT(centroid, point h) if centroid_is_A else OT(point h, centroid); all lanes return all G values
template <uint32_t G>
__device__ __forceinline__ void wave_sinkhorn_costG(GroupLds<G>& w, uint32_t m, const uint32_t (&n)[G], const Metric& M,
                                                    bool centroid_is_A, float (&cost_out)[G]) {
    constexpr uint32_t ROWS = 64u / G;
    const uint32_t lane = lane_id(), grp = lane / ROWS, r = lane % ROWS;
    const uint32_t bins = M.bins;
    const __amdgpu_buffer_rsrc_t rt = rt_resource(M);
#pragma unroll
    for (uint32_t h = 0; h < G; ++h) cost_out[h] = -0.0f;
    if (m == 0) return;
    bool active[G];
    uint32_t iters_done[G];
    const float lc = LM_LOGF(1.0f / (float)m);
#pragma unroll
    for (uint32_t h = 0; h < G; ++h) {
        active[h] = n[h] > 0;
        iters_done[h] = 0;
        for (uint32_t i = lane; i < m; i += 64) w.potC[h][i] = lc;
        if (n[h] > 0 && lane < n[h]) w.potP[h][lane] = LM_LOGF(1.0f / (float)n[h]);
    }
    __syncthreads();
    const uint32_t nh = pick<G>(n, grp);
    uint32_t nmax = 0;
#pragma unroll
    for (uint32_t h = 0; h < G; ++h) nmax = max(nmax, n[h]);
    // rows = centroid bins, one solve at a time (columns = that point's bins)
    auto centroid_rows = [&](uint32_t h) {
        for (uint32_t i0 = 0; i0 < m; i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool act = i < m;
            const uint32_t x = act ? w.supC[i] : w.supC[0];
            const float s = softmin_sum<false>(w.supP[h], w.potP[h], n[h], rt, bins, x);
            if (act) {
                const float nv = w.lnC[i] - LM_LOGF(s);
                w.tmpC[h][i] = rp_absf(LM_EXPF(nv) - LM_EXPF(w.potC[h][i]));
                w.potC[h][i] = nv;
            }
        }
    };
    // the same for a centroid with <= ROWS bins: all solves at once, lane (g, i) = row i of the centroid in solve g,
    // walking point g's bins (per-lane C/T row offsets)
    const bool small_centroid = m <= ROWS;
    auto centroid_rows_all = [&]() {
        const bool valid = r < m && pick<G>(active, grp);
        const uint32_t x = w.supC[r < m ? r : 0u];
        const float s = softmin_sum_lane(w.supP[grp], w.potP[grp], valid ? nh : 0u, rt, bins, x);
        if (valid) {
            const float nv = w.lnC[r] - LM_LOGF(s);
            w.tmpC[grp][r] = rp_absf(LM_EXPF(nv) - LM_EXPF(w.potC[grp][r]));
            w.potC[grp][r] = nv;
        }
    };
    // rows = point bins, all solves at once (columns = the centroid's bins, potential per lane group)
    auto point_rows = [&]() {
        const bool valid = r < nh && pick<G>(active, grp);
        const uint32_t y = w.supP[grp][r < nh ? r : 0u];
        const float s = softmin_sum<false>(w.supC, w.potC[grp], m, rt, bins, y);
        if (valid) {
            const float nv = w.lnP[grp][r] - LM_LOGF(s);
            w.tmpP[grp][r] = rp_absf(LM_EXPF(nv) - LM_EXPF(w.potP[grp][r]));
            w.potP[grp][r] = nv;
        }
    };
    auto err_centroid = [&]() -> float {  // lanes of group g: sum over the centroid rows of solve g
        return lds_sum_in_order(w.tmpC[grp], m);
    };
    auto err_point = [&]() -> float { return lds_sum_in_order(w.tmpP[grp], nh); };
    for (uint32_t t = 0; t < M.iters; ++t) {
        float lhs_err, rhs_err;
        if (centroid_is_A) {  // lhs updates the centroid side, rhs the point side (Gauss-Seidel, sinkhorn.rs:80-87)
            if (small_centroid) centroid_rows_all();
            else
#pragma unroll
                for (uint32_t h = 0; h < G; ++h)
                    if (active[h]) centroid_rows(h);
            __syncthreads();
            lhs_err = err_centroid();
            __syncthreads();
            point_rows();
            __syncthreads();
            rhs_err = err_point();
            __syncthreads();
        } else {
            point_rows();
            __syncthreads();
            lhs_err = err_point();
            __syncthreads();
            if (small_centroid) centroid_rows_all();
            else
#pragma unroll
                for (uint32_t h = 0; h < G; ++h)
                    if (active[h]) centroid_rows(h);
            __syncthreads();
            rhs_err = err_centroid();
            __syncthreads();
        }
        const float tot = lhs_err + rhs_err;
        bool any = false;
#pragma unroll
        for (uint32_t h = 0; h < G; ++h) {
            const float th = __shfl(tot, (int)(h * ROWS), 64);
            if (active[h] && th < M.tol) {
                active[h] = false;
                iters_done[h] = t + 1;
            }
            any = any || active[h];
        }
        if (!any) break;
    }
    unsigned long long its = 0, exps = 0;
#pragma unroll
    for (uint32_t h = 0; h < G; ++h) {
        if (active[h]) iters_done[h] = M.iters;
        its += (unsigned long long)iters_done[h] * (n[h] > 0);
        exps += (unsigned long long)(2 * iters_done[h] + 1) * m * n[h];
    }
    if (lane == 0) {
        atomicAdd(STAT(M, 1), its);
        atomicAdd(STAT(M, 2), exps);
    }
    // cost(): A-major left fold of coupling * distance (sinkhorn.rs:206-217), one solve per lane group
    float cost = -0.0f;  // .sum::<Energy>() folds from -0.0 (sinkhorn.rs:216; same result unless the sum is empty)
    if (centroid_is_A) {
        for (uint32_t i = 0; i < m; ++i) {
            const uint32_t x = w.supC[i];
            const float fi = w.potC[grp][i];
            if (r < nh) {
                const uint32_t y = w.supP[grp][r];
                w.tmpP[grp][r] = LM_EXPF(fi + w.potP[grp][r] - M.Rt[x * bins + y]) * M.Cm[x * bins + y];
            }
            __syncthreads();
            for (uint32_t j = 0; j < nh; ++j) cost += w.tmpP[grp][j];
            __syncthreads();
        }
    } else {
        for (uint32_t i = 0; i < nmax; ++i) {
#pragma unroll
            for (uint32_t h = 0; h < G; ++h) {
                if (i >= n[h]) continue;
                const uint32_t x = w.supP[h][i];
                const float fi = w.potP[h][i];
                for (uint32_t j = lane; j < m; j += 64) {
                    const uint32_t y = w.supC[j];
                    w.tmpC[h][j] = LM_EXPF(fi + w.potC[h][j] - M.Rt[x * bins + y]) * M.Cm[x * bins + y];
                }
            }
            __syncthreads();
            if (i < nh) {
                const float* t = w.tmpC[grp];
                for (uint32_t j = 0; j < m; ++j) cost += t[j];
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (uint32_t h = 0; h < G; ++h) cost_out[h] = __shfl(cost, (int)(h * ROWS), 64);
}

// support of a dense histogram into a group slot (at most `rows` bins, guaranteed by the grouping lists)
template <typename CT>
__device__ uint32_t pair_load_hist(const CT* counts, uint32_t weight, uint32_t bins, uint16_t* sup, float* lnd, uint32_t rows) {
    const uint32_t lane = lane_id();
    const float fw = (float)weight;
    uint32_t base = 0;
    for (uint32_t q = 0; q * 64 < bins; ++q) {
        const uint32_t b = q * 64 + lane;
        const uint32_t c = b < bins ? (uint32_t)counts[b] : 0u;
        const bool has = c > 0;
        const unsigned long long mask = __ballot(has);
        if (has) {
            const uint32_t rr = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (rr < rows) {
                sup[rr] = (uint16_t)b;
                lnd[rr] = LM_LOGF((float)c / fw);
            }
        }
        base += __popcll(mask);
    }
    __syncthreads();
    return base;
}

__global__ __launch_bounds__(64) void k_point_support(Points P, uint32_t bins, uint8_t* nsup) {
    const uint64_t i = blockIdx.x;
    uint32_t c = 0;
    for (uint32_t b = lane_id(); b < bins; b += 64) c += P.counts[i * P.stride + b] > 0;
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
    if (lane_id() == 0) nsup[i] = (uint8_t)min(c, 255u);
}

// ------------------------------------------------------------------------------------------------
// histogram preparation
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_point_weights(const uint8_t* counts, uint32_t stride, uint32_t bins, uint64_t N,
                                                      uint32_t* weight) {
    const uint64_t i = blockIdx.x;
    if (i >= N) return;
    uint32_t s = 0;
    for (uint32_t b = lane_id(); b < bins; b += 64) s += counts[i * stride + b];
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane_id() == 0) weight[i] = s;
}

__global__ __launch_bounds__(64) void k_point_self(Points P, Metric M, float* self_out) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint64_t i = blockIdx.x;
    const uint32_t m = wave_load_hist(P.counts + i * P.stride, P.weight[i], M.bins, w.supA, w.lnA);
    for (uint32_t k = lane_id(); k < m; k += 64) {
        w.supB[k] = w.supA[k];
        w.lnB[k] = w.lnA[k];
    }
    __syncthreads();
    const float c = wave_sinkhorn_cost(w, m, m, M);
    if (lane_id() == 0) self_out[i] = c;
}

// derive support / ln-density / transposed density tables of a centroid set, and OT(c,c) for Sinkhorn layers
// self_from: OT(c, c) is already known (the centroid is a copy of a point whose memoised OT(p, p) this is: the same histogram, the same
// solve, the same bits) — the k-means++ rounds install one such centroid each and would otherwise wait for a one-wavefront solve
__global__ __launch_bounds__(64) void k_prepare_centroids(CentroidSet cs, uint32_t K, Metric M, int kind, uint32_t k0, const float* self_from) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint32_t k = k0 + blockIdx.x;
    const uint32_t bins = M.bins;
    const uint32_t wt = cs.weight[k];
    const uint32_t m = wave_load_hist(cs.counts + (size_t)k * bins, wt, bins, w.supA, w.lnA);
    for (uint32_t i = lane_id(); i < m; i += 64) {
        cs.sup[(size_t)k * MAXB + i] = w.supA[i];
        cs.lnd[(size_t)k * MAXB + i] = w.lnA[i];
        w.supB[i] = w.supA[i];
        w.lnB[i] = w.lnA[i];
    }
    for (uint32_t b = lane_id(); b < bins; b += 64)
        cs.dens[(size_t)b * K + k] = (float)cs.counts[(size_t)k * bins + b] / (float)wt;  // NaN for an empty cluster, as in the reference
    for (uint32_t b = lane_id(); b < bins; b += 64)
        cs.densR[(size_t)k * MAXB + b] = wt ? (float)cs.counts[(size_t)k * bins + b] / (float)wt : 0.0f;
    if (kind == RP_METRIC_SINKHORN)
        for (uint32_t y = lane_id(); y < bins; y += 64) {  // the column-marginal bound's table (sinkhorn_bound.hpp)
            float mn = m ? rp_u2f(0x7f800000u) : 0.0f;
            for (uint32_t i = 0; i < m; ++i) mn = fminf(mn, M.Cm[(size_t)w.supA[i] * bins + y]);
            cs.mincT[(size_t)y * MAXB + k] = mn;
        }
    if (lane_id() == 0) cs.n[k] = m;
    __syncthreads();
    float self = 0.0f;
    if (kind == RP_METRIC_SINKHORN && self_from == cs.self) return;  // the caller runs k_self_block on the prepared tables next
    if (kind == RP_METRIC_SINKHORN) self = self_from ? *self_from : wave_sinkhorn_cost(w, m, m, M);
    if (lane_id() == 0) cs.self[k] = self;
}

// centroid k <- copy of point idx (Layer::init_centroids pushes points, layer.rs:166-168)
__global__ void k_centroid_from_point(CentroidSet cs, uint32_t k, Points P, uint64_t idx, uint32_t bins) {
    for (uint32_t b = threadIdx.x; b < bins; b += blockDim.x) cs.counts[(size_t)k * bins + b] = P.counts[idx * P.stride + b];
    if (threadIdx.x == 0) cs.weight[k] = P.weight[idx];
}

__global__ void k_centroid_from_hist(CentroidSet cs, uint32_t k, const uint32_t* hist, uint32_t bins) {
    __shared__ uint32_t wsum;
    if (threadIdx.x == 0) wsum = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t b = threadIdx.x; b < bins; b += blockDim.x) {
        cs.counts[(size_t)k * bins + b] = hist[b];
        mine += hist[b];
    }
    atomicAdd(&wsum, mine);
    __syncthreads();
    if (threadIdx.x == 0) cs.weight[k] = wsum;
}

// ------------------------------------------------------------------------------------------------
// Equity::variation (equity.rs:41-53): one LANE per centroid, points' densities broadcast from LDS
// ------------------------------------------------------------------------------------------------
__device__ void wave_point_density(const Points& P, uint64_t i, uint32_t bins, float* pd) {
    const float fw = (float)P.weight[i];
    for (uint32_t b = lane_id(); b < bins; b += 64) pd[b] = (float)P.counts[i * P.stride + b] / fw;
    __syncthreads();
}
// d[q] = variation(point, centroid q*64+lane) for q < 4
__device__ void wave_variation_all(const float* pd, const CentroidSet& cs, uint32_t K, uint32_t bins, float d[4],
                                   const Metric& M, bool evaluated_all = true) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t k = q * 64 + lane;
        float cx = 0.0f, cy = 0.0f, s = 0.0f;
        if (k < K) {
            for (uint32_t b = 0; b < bins; ++b) {
                cx += pd[b];
                cy += cs.dens[(size_t)b * K + k];
                s += rp_absf(cx - cy);
            }
            s = s / (float)bins;
        }
        d[q] = s;
    }
    // Elkan::neighbor evaluates all K; k_elkan_step computes all K and counts the ones its rule evaluates itself (STAT 4: computed)
    if (lane == 0) atomicAdd(STAT(M, evaluated_all ? 0 : 4), (unsigned long long)K);
}

// ------------------------------------------------------------------------------------------------
// Elkan::neighbor for every point (elkan.rs:68-77): init_bounds / Layer::lookup / step_naive
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool memo_valid(const Bounds& B, uint64_t i, uint32_t j) {
    return B.memo_ver && B.memo_j[i] == (uint8_t)j && B.memo_ver[i] == B.cver[j];
}
__device__ __forceinline__ void memo_store(const Bounds& B, uint64_t i, uint32_t j, float d) {
    if (!B.memo_ver) return;
    B.memo_d[i] = d;
    B.memo_j[i] = (uint8_t)j;
    B.memo_ver[i] = B.cver[j];
}
#if !LM_GLIBC  // like the other scaling-domain filters: one copy, in the contract namespace, for both passes
#include "refresh_bound.hpp"
#endif

__global__ __launch_bounds__(64) void k_neighbor(Points P, CentroidSet cs, uint32_t K, Metric M, int kind,
                                                 uint8_t* out_j, float* out_d, Bounds init, const uint32_t* only) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint64_t i = only ? only[blockIdx.x] : blockIdx.x;  // `only`: the points the grouped kernels do not take
    const uint32_t lane = lane_id();
    uint32_t bj = 0;
    float bd = 0.0f;
    if (kind == RP_METRIC_SINKHORN) {
        const uint32_t n = wave_load_hist(P.counts + i * P.stride, P.weight[i], M.bins, w.supB, w.lnB);
        const float sp = P.self[i];
        for (uint32_t k = 0; k < K; ++k) {
            const uint32_t m = wave_load_centroid(cs, k, w.supA, w.lnA);
            const float d = wave_divergence(w, m, n, cs.self[k], sp, M);  // distance(centroid, point)
            if (k == 0 || d < bd) {
                bj = k;
                bd = d;
            }
            __syncthreads();
        }
    } else {
        wave_point_density(P, i, M.bins, w.f);
        float d[4];
        wave_variation_all(w.f, cs, K, M.bins, d, M);
        // first minimum in ascending k: per-lane scan over q then wave argmin with index tie-break
        float best = 0.0f;
        uint32_t bk = 0xffffffffu;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t k = q * 64 + lane;
            if (k < K && (bk == 0xffffffffu || d[q] < best)) {
                best = d[q];
                bk = k;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const uint32_t ok = __shfl_xor(bk, o, 64);
            if (ok != 0xffffffffu && (bk == 0xffffffffu || ob < best || (ob == best && ok < bk))) {
                best = ob;
                bk = ok;
            }
        }
        bj = bk;
        bd = best;
    }
    if (lane == 0) {
        if (out_j) out_j[i] = (uint8_t)bj;
        if (out_d) out_d[i] = bd;
        if (init.j) {  // Bounds::from((j, upper)) (bounds.rs:111-120)
            init.j[i] = (uint8_t)bj;
            init.u[i] = bd;
            init.stale[i] = 0;
        }
    }
    if (init.lower)
        for (uint32_t k = lane; k < K; k += 64) init.lower[i * K + k] = 0.0f;
}

// Elkan::neighbor for a SHORT list of points (the sampled self-check of the pruned passes, the points k-means++ leaves to init_bounds):
// a few thousand wavefronts that each walk all K centroids leave most of the chip idle, so the K centroids of a point are split over
// NB_CHUNKS wavefronts; k_neighbor_merge takes the first minimum over the chunks in ascending order (strict <), which is the first
// minimum over all K (elkan.rs:68-77: min_by keeps the first).  Sinkhorn only.
#define NB_CHUNKS 8u
__global__ __launch_bounds__(64) void k_neighbor_chunk(Points P, CentroidSet cs, uint32_t K, Metric M, const uint32_t* list, uint8_t* tmp_j,
                                                       float* tmp_d) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint32_t e = blockIdx.x / NB_CHUNKS, q = blockIdx.x % NB_CHUNKS;
    const uint64_t i = list[e];
    const uint32_t per = (K + NB_CHUNKS - 1u) / NB_CHUNKS, k0 = q * per, k1 = min(K, k0 + per);
    const uint32_t n = wave_load_hist(P.counts + i * P.stride, P.weight[i], M.bins, w.supB, w.lnB);
    const float sp = P.self[i];
    // the sequential rule (k_neighbor: the first centroid is taken whatever its distance, a later one only if strictly smaller — a NaN
    // never wins) holds across chunks only if a chunk other than the first starts "unset": +inf loses to every finite distance here
    // and wins against nothing in k_neighbor_merge
    uint32_t bj = 0xffu;
    float bd = __builtin_inff();
    for (uint32_t k = k0; k < k1; ++k) {
        const uint32_t m = wave_load_centroid(cs, k, w.supA, w.lnA);
        const float d = wave_divergence(w, m, n, cs.self[k], sp, M);  // distance(centroid, point)
        if (k == 0 || d < bd) {
            bj = k;
            bd = d;
        }
        __syncthreads();
    }
    if (lane_id() == 0) {
        tmp_j[(size_t)e * NB_CHUNKS + q] = (uint8_t)bj;
        tmp_d[(size_t)e * NB_CHUNKS + q] = bd;
    }
}
__global__ __launch_bounds__(256) void k_neighbor_merge(const uint32_t* list, uint32_t n, uint32_t K, const uint8_t* tmp_j, const float* tmp_d,
                                                        uint8_t* out_j, float* out_d, Bounds init) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const uint32_t per = (K + NB_CHUNKS - 1u) / NB_CHUNKS;
    uint32_t bj = tmp_j[(size_t)e * NB_CHUNKS];
    float bd = tmp_d[(size_t)e * NB_CHUNKS];
    for (uint32_t q = 1; q < NB_CHUNKS && q * per < K; ++q) {
        const float d = tmp_d[(size_t)e * NB_CHUNKS + q];
        if (d < bd) {
            bd = d;
            bj = tmp_j[(size_t)e * NB_CHUNKS + q];
        }
    }
    const uint64_t i = list[e];
    if (out_j) out_j[i] = (uint8_t)bj;
    if (out_d) out_d[i] = bd;
    if (init.j) {  // Bounds::from((j, upper)) (bounds.rs:111-120)
        init.j[i] = (uint8_t)bj;
        init.u[i] = bd;
        init.stale[i] = 0;
    }
    if (init.lower)
        for (uint32_t k = 0; k < K; ++k) init.lower[i * K + k] = 0.0f;
}

// Elkan::neighbor over the survivors of the MFMA bound (sinkhorn_bound.hpp): the centroids whose bit is set in the
// point's 256-bit mask, in ascending index, first minimum wins (elkan.rs:68-77: min_by keeps the first) — the unpruned
// loop's result bit for bit as long as every minimiser survives.  `audit_*`: RP_LLOYD_AUDIT compares with the unpruned
// pass instead of writing.
__global__ __launch_bounds__(64) void k_neighbor_masked(Points P, CentroidSet cs, uint32_t K, Metric M, const unsigned long long* mask,
                                                        uint8_t* out_j, float* out_d, Bounds init, const uint8_t* hint_j,
                                                        const float* hint_d) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint64_t i = blockIdx.x;
    const uint32_t lane = lane_id();
    const uint32_t n = wave_load_hist(P.counts + i * P.stride, P.weight[i], M.bins, w.supB, w.lnB);
    const float sp = P.self[i];
    uint32_t bj = 0;
    float bd = 0.0f;
    bool first = true;
    // Two guards so that a bound that missed can cost time, never a bucket: the centroid whose exact distance seeded the bound
    // (the hint) is always a candidate, and a mask without any centroid below K means "search them all".
    unsigned long long mq[4];
    bool any = false;
    for (uint32_t q = 0; q < 4; ++q) {
        mq[q] = mask[i * 4 + q];
        if (hint_j && (hint_j[i] >> 6) == q) mq[q] |= 1ull << (hint_j[i] & 63u);
        const uint32_t below = K > q * 64u ? min(K - q * 64u, 64u) : 0u;
        any = any || (mq[q] & (below == 64u ? ~0ull : ((1ull << below) - 1ull))) != 0ull;
    }
    for (uint32_t q = 0; q < 4; ++q) {
        unsigned long long bits = any ? mq[q] : ~0ull;
        while (bits) {
            const uint32_t k = q * 64 + (uint32_t)__builtin_ctzll(bits);
            bits &= bits - 1;
            if (k >= K) break;
            float d;
            if (hint_j && hint_j[i] == k) {
                d = hint_d[i];  // this very solve was done for the upper bound handed to the MFMA bound
            } else {
                const uint32_t m = wave_load_centroid(cs, k, w.supA, w.lnA);
                d = wave_divergence(w, m, n, cs.self[k], sp, M);  // distance(centroid, point)
            }
            if (first || d < bd) {
                bj = k;
                bd = d;
                first = false;
            }
            __syncthreads();
        }
    }
    if (lane == 0) {
        if (out_j) out_j[i] = (uint8_t)bj;
        if (out_d) out_d[i] = bd;
        if (init.j) {
            init.j[i] = (uint8_t)bj;
            init.u[i] = bd;
            init.stale[i] = 0;
        }
    }
    if (init.lower)
        for (uint32_t k = lane; k < K; k += 64) init.lower[i * K + k] = 0.0f;
}

// init_bounds right after k-means++: Elkan::neighbor of a point = the nearest centroid k-means++ noted, when that is known to
// be the minimum over ALL K centroids.  A pair k-means++ did not solve was skipped because its rigorous lower bound (with the
// margins of k_kpp_filter) squared was >= the potential at that time, which is >= the final potential; so once the noted
// distance is what the final potential stands for — d*d below the initial potential 1 — every unsolved pair is farther, and
// among the solved ones the note is the first minimum.  Everything else (the K picked points, points that ended at
// potential 1, NaNs) goes on `todo` for the exact search.
__global__ __launch_bounds__(256) void k_init_from_kpp(Metric M, uint64_t N, uint32_t K, uint8_t* out_j, Bounds init, uint32_t* todo,
                                                       unsigned int* n_todo) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float d = M.kpp_d[i];
    if (d >= 0.0f && d * d < 1.0f) {
        const uint8_t j = M.kpp_j[i];
        if (out_j) out_j[i] = j;
        init.j[i] = j;  // Bounds::from((j, upper)) (bounds.rs:111-120)
        init.u[i] = d;
        init.stale[i] = 0;
    } else {
        todo[atomicAdd(n_todo, 1u)] = (uint32_t)i;
    }
}
__global__ __launch_bounds__(256) void k_zero_f32(float* p, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) p[i] = 0.0f;
}

// upper bounds handed to the MFMA bound before it starts
__global__ __launch_bounds__(256) void k_hint_masks(const uint8_t* j, uint64_t N, uint32_t K, unsigned long long* mask) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t k = min((uint32_t)j[i], K - 1);
    for (uint32_t q = 0; q < 4; ++q) mask[i * 4 + q] = (k >> 6) == q ? 1ull << (k & 63u) : 0ull;
}
// k-means++ left potentials = min_k d(c_k, x)^2 (layer.rs:170-178, the same centroid-first distance): sqrt, two ulps up
__global__ __launch_bounds__(256) void k_ub_from_pot(const float* pot, uint64_t N, float* ub) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float p = pot[i];
    ub[i] = p >= 0.0f ? sqrtf(p) * 1.0000003f + 1e-30f : rp_u2f(0x7f800000u);
}
__global__ __launch_bounds__(256) void k_mask_all(const uint32_t* list, uint32_t n, unsigned long long* mask) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    for (int q = 0; q < 4; ++q) mask[(size_t)list[e] * 4 + q] = ~0ull;
}

// the production sample check: the same comparison over a list of points
__global__ __launch_bounds__(256) void k_audit_compare_list(const uint8_t* ja, const float* da, const uint8_t* jb, const float* db,
                                                            const uint32_t* list, uint32_t n, unsigned long long* bad) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const uint32_t i = list[e];
    if (ja[i] != jb[i] || __float_as_uint(da[i]) != __float_as_uint(db[i])) atomicAdd(bad, 1ull);
}
// RP_LLOYD_AUDIT: count the points on which two neighbor passes disagree (bucket or distance bits)
__global__ __launch_bounds__(256) void k_audit_compare(const uint8_t* ja, const float* da, const uint8_t* jb, const float* db, uint64_t N,
                                                       unsigned long long* bad) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    if (ja[i] != jb[i] || __float_as_uint(da[i]) != __float_as_uint(db[i])) atomicAdd(bad, 1ull);
}

// Elkan::neighbor for G points per wavefront (each with <= 64 / G support bins), Sinkhorn metric
template <uint32_t G>
__global__ __launch_bounds__(64) void k_neighborG(Points P, CentroidSet cs, uint32_t K, Metric M, const uint32_t* groups,
                                                  uint8_t* out_j, float* out_d, Bounds init) {
    LM_TABLES();
    __shared__ GroupLds<G> w;
    const uint32_t lane = lane_id();
    uint64_t ip[G];
    uint32_t n[G];
    float sp[G];
#pragma unroll
    for (uint32_t h = 0; h < G; ++h) {
        ip[h] = groups[G * blockIdx.x + h];
        n[h] = pair_load_hist(P.counts + ip[h] * P.stride, P.weight[ip[h]], M.bins, w.supP[h], w.lnP[h], GroupLds<G>::ROWS);
        sp[h] = P.self[ip[h]];
    }
    uint32_t bj[G];
    float bd[G];
#pragma unroll
    for (uint32_t h = 0; h < G; ++h) bj[h] = 0, bd[h] = 0.0f;
    for (uint32_t k = 0; k < K; ++k) {
        const uint32_t m = wave_load_centroid(cs, k, w.supC, w.lnC);
        float xy[G];
        wave_sinkhorn_costG<G>(w, m, n, M, true, xy);  // distance(centroid, point)
        const float sc = cs.self[k];
#pragma unroll
        for (uint32_t h = 0; h < G; ++h) {
            const float d = rp_maxf(xy[h] - 0.5f * sc - 0.5f * sp[h], 0.0f);
            if (k == 0 || d < bd[h]) {
                bj[h] = k;
                bd[h] = d;
            }
        }
        __syncthreads();
    }
    if (lane == 0) atomicAdd(STAT(M, 0), (unsigned long long)G * K);
    if (lane < G) {
        const uint64_t i = pick<G>(ip, lane);
        const uint32_t j = pick<G>(bj, lane);
        const float d = pick<G>(bd, lane);
        if (out_j) out_j[i] = (uint8_t)j;
        if (out_d) out_d[i] = d;
        if (init.j) {
            init.j[i] = (uint8_t)j;
            init.u[i] = d;
            init.stale[i] = 0;
        }
    }
    if (init.lower)
#pragma unroll
        for (uint32_t h = 0; h < G; ++h)
            for (uint32_t k = lane; k < K; k += 64) init.lower[ip[h] * K + k] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// The column-marginal bound (rigorous): after the rhs update that ends every Sinkhorn iteration the coupling
// pi(x, y) = exp(f(x) + g(y) - C/T) has column sums nu(y) — by construction of g, whatever the iteration count — so
//     cost = sum_y sum_x pi(x, y) C(x, y)  >=  sum_y nu(y) min_{x in supp mu} C(x, y).
// In f32 the column sums hold to ~1e-5 (exp/ln rounding at arguments up to C/T) and the x-major cost sum to ~1e-4
// relative in the worst case: the bound is used with the factor KPP_LB_SAFETY and only when max C / T <= 64 (no term near
// the MIN_POSITIVE clamp).  k-means++ (layer.rs:170-178) updates potentials <- min(potentials, d^2): a point whose bound
// already gives d^2 >= potential keeps its potential without the solve.
// ------------------------------------------------------------------------------------------------
#define KPP_LB_SAFETY 0.999f
#define KPP_LB_SLACK 1e-6f
__global__ __launch_bounds__(256) void k_minc(CentroidSet cs, uint32_t k, Metric M, float* minc) {
    const uint32_t y = threadIdx.x, n = cs.n[k];
    if (y >= M.bins) return;
    float m = n ? rp_u2f(0x7f800000u) : 0.0f;
    for (uint32_t i = 0; i < n; ++i) m = fminf(m, M.Cm[(size_t)cs.sup[(size_t)k * MAXB + i] * M.bins + y]);
    minc[y] = m;
}
// 16 lanes per point: the bound, the test against the potential, the point's place in its class list
__global__ __launch_bounds__(1024) void k_kpp_filter(Points P, CentroidSet cs, uint32_t k, Metric M, const float* minc, const float* pot,
                                                     const uint8_t* nsup, KppLists out, uint32_t quad_rows, uint32_t pair_rows) {
    __shared__ unsigned int cnt[3], base[3];
    __shared__ float mc[MAXB];
    const uint32_t tid = threadIdx.x, sub = tid & 15u;
    if (tid < 3) cnt[tid] = 0;
    for (uint32_t b = tid; b < M.bins; b += 1024) mc[b] = minc[b];
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * 64 + (tid >> 4);
    const bool real = i < P.N;
    float acc = 0.0f;
    if (real) {
        const uint8_t* row = P.counts + i * P.stride;
        for (uint32_t b = sub; b < M.bins; b += 16) acc += (float)row[b] * mc[b];
    }
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    uint32_t cls = 3, slot = 0;
    if (real && sub == 0) {
        const float lb = (acc / (float)P.weight[i]) * KPP_LB_SAFETY - KPP_LB_SLACK;
        const float d = rp_maxf(lb - 0.5f * cs.self[k] - 0.5f * P.self[i], 0.0f);
        if (!(d * d >= pot[i])) {  // the solve may lower the potential (NaN counts as "may")
            const uint32_t ns = nsup[i];
            cls = (quad_rows && ns <= quad_rows) ? 0u : ((pair_rows && ns <= pair_rows) ? 1u : 2u);
            slot = atomicAdd(&cnt[cls], 1u);
        }
    }
    __syncthreads();
    if (tid < 3) base[tid] = cnt[tid] ? atomicAdd(&out.count[tid], cnt[tid]) : 0u;
    __syncthreads();
    if (cls < 3) out.list[cls][base[cls] + slot] = (uint32_t)i;
}

// k-means++ potentials for G points per wavefront: potentials <- min(potentials, d(new centroid, point)^2).
// `count`: number of valid entries of `groups` (the filtered lists of k_kpp_filter), NULL = every group is full.
template <uint32_t G>
__global__ __launch_bounds__(64) void k_kpp_updateG(Points P, CentroidSet cs, uint32_t k, Metric M, const uint32_t* groups,
                                                    const unsigned int* count, float* pot) {
    LM_TABLES();
    __shared__ GroupLds<G> w;
    const uint32_t have = count ? *count : 0xffffffffu;
    if (G * blockIdx.x >= have) return;
    uint64_t ip[G];
    uint32_t n[G];
    bool real[G];
    const uint32_t m = wave_load_centroid(cs, k, w.supC, w.lnC);
#pragma unroll
    for (uint32_t h = 0; h < G; ++h) {
        real[h] = G * blockIdx.x + h < have;
        ip[h] = real[h] ? groups[G * blockIdx.x + h] : groups[G * blockIdx.x];
        const uint32_t got = pair_load_hist(P.counts + ip[h] * P.stride, P.weight[ip[h]], M.bins, w.supP[h], w.lnP[h], GroupLds<G>::ROWS);
        n[h] = real[h] ? got : 0u;
    }
    float xy[G];
    wave_sinkhorn_costG<G>(w, m, n, M, true, xy);
    const uint32_t lane = lane_id();
    uint32_t nreal = 0;
#pragma unroll
    for (uint32_t h = 0; h < G; ++h) nreal += real[h];
    if (lane == 0) atomicAdd(STAT(M, 0), (unsigned long long)nreal);
    if (lane < G && pick<G>(real, lane)) {
        const uint64_t i = pick<G>(ip, lane);
        const float d = rp_maxf(pick<G>(xy, lane) - 0.5f * cs.self[k] - 0.5f * P.self[i], 0.0f);
        pot[i] = rp_minf(d * d, pot[i]);
        kpp_note(M, i, k, d);
    }
}

// Elkan::pairwises for the variation metric: one LANE per ordered pair (a, b), b fastest so the transposed density
// table is read coalesced; same left folds as equity.rs:41-53
__global__ __launch_bounds__(256) void k_pairwise_var(CentroidSet cs, uint32_t K, Metric M, float* pairw) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= K * K) return;
    const uint32_t a = e / K, b = e % K;
    float d = 0.0f;
    if (a != b) {
        float cx = 0.0f, cy = 0.0f, s = 0.0f;
        for (uint32_t t = 0; t < M.bins; ++t) {
            cx += cs.dens[(size_t)t * K + a];
            cy += cs.dens[(size_t)t * K + b];
            s += rp_absf(cx - cy);
        }
        d = s / (float)M.bins;
    }
    pairw[e] = d;
    if (e == 0) atomicAdd(STAT(M, 0), (unsigned long long)K * (K - 1));
}

// Elkan::pairwises (elkan.rs:80-93): both orders; one wave per ordered pair
// pver[2e], pver[2e+1]: the centroid versions pairw[e] was computed from (0 = never): an entry whose two centroids have not
// changed keeps its value (distance(a, b) is a pure function of the two)
__global__ __launch_bounds__(64) void k_pairwise(CentroidSet cs, uint32_t K, Metric M, int kind, float* pairw, const uint32_t* cver,
                                                 uint32_t* pver) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint32_t a = blockIdx.x / K, b = blockIdx.x % K;
    if (pver) {
        const uint32_t va = cver[a], vb = cver[b];
        // one decision for the wavefront, taken before lane 0 overwrites what it was taken from
        const bool same = pver[2 * blockIdx.x] == va && pver[2 * blockIdx.x + 1] == vb;
        if (__builtin_amdgcn_readfirstlane((uint32_t)same)) {
            if (a != b && lane_id() == 0) atomicAdd(STAT(M, 3), 1ull);  // remembered
            return;
        }
        if (lane_id() == 0) {
            pver[2 * blockIdx.x] = va;
            pver[2 * blockIdx.x + 1] = vb;
        }
    }
    float d = 0.0f;
    if (a != b) {
        if (kind == RP_METRIC_SINKHORN) {
            const uint32_t m = wave_load_centroid(cs, a, w.supA, w.lnA);
            const uint32_t n = wave_load_centroid(cs, b, w.supB, w.lnB);
            d = wave_divergence(w, m, n, cs.self[a], cs.self[b], M);
        } else {
            float cx = 0.0f, cy = 0.0f, s = 0.0f;
            for (uint32_t t = 0; t < M.bins; ++t) {
                cx += cs.dens[(size_t)t * K + a];
                cy += cs.dens[(size_t)t * K + b];
                s += rp_absf(cx - cy);
            }
            d = s / (float)M.bins;
            if (lane_id() == 0) atomicAdd(STAT(M, 0), 1ull);
        }
    }
    if (lane_id() == 0) pairw[(size_t)a * K + b] = d;
}

// Elkan::midpoints (elkan.rs:96-105)
__global__ void k_midpoints(const float* pairw, uint32_t K, float* mid) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    float r = RP_F32_MAX;
    for (uint32_t j = 0; j < K; ++j)
        if (j != i) r = rp_minf(r, pairw[(size_t)i * K + j] * 0.5f);
    mid[i] = r;
}

// ------------------------------------------------------------------------------------------------
// Elkan::step_elkan's bound refresh (elkan.rs:153-168 with refresh :113-117, rebound :119-123,
// Bounds::{has_shifted :57-61, witness :85-91, refresh :79-83})
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_elkan_step(Points P, CentroidSet cs, uint32_t K, Metric M, int kind, Bounds B,
                                                   const float* pairw, const float* mid) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint64_t i = blockIdx.x;
    const uint32_t lane = lane_id();
    uint32_t j = B.j[i];
    float u = B.u[i];
    if (!(u > mid[j])) return;  // filter(|b| b.u() > midpoints[b.j()])
    float dv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t m = 0;
    float sp = 0.0f;
    if (kind == RP_METRIC_SINKHORN) {
        m = wave_load_hist(P.counts + i * P.stride, P.weight[i], M.bins, w.supA, w.lnA);
        sp = P.self[i];
    } else {
        wave_point_density(P, i, M.bins, w.f);
        wave_variation_all(w.f, cs, K, M.bins, dv, M, false);  // distances to every centroid; the replay below uses (and
                                                               // counts) only the ones the sequential rule evaluates
    }
    unsigned long long var_evaluated = 0;
    auto distance_to = [&](uint32_t k) -> float {  // distance(point, centroid k)
        if (kind == RP_METRIC_SINKHORN) {
            const uint32_t n = wave_load_centroid(cs, k, w.supB, w.lnB);
            const float d = wave_divergence(w, m, n, sp, cs.self[k], M);
            __syncthreads();
            return d;
        }
        const uint32_t q = k >> 6, src = k & 63u;
        float mine = q == 0 ? dv[0] : (q == 1 ? dv[1] : (q == 2 ? dv[2] : dv[3]));
        var_evaluated += 1;  // one atomic per wavefront at the end, not one per distance
        return __shfl(mine, (int)src, 64);
    };
    float lw[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t k = q * 64 + lane;
        lw[q] = k < K ? B.lower[i * K + k] : 0.0f;
    }
    auto set_lower = [&](uint32_t k, float d) {
        const uint32_t q = k >> 6;
        if (lane == (k & 63u)) {
            if (q == 0) lw[0] = d;
            else if (q == 1) lw[1] = d;
            else if (q == 2) lw[2] = d;
            else lw[3] = d;
        }
    };
    bool exact = false;  // u is an exact distance to the CURRENT centroid j, measured in this call
    if (B.stale[i]) {
        const bool remembered = memo_valid(B, i, j);
        if (remembered && lane == 0) atomicAdd(STAT(M, 3), 1ull);  // a distance the reference evaluates here and this pass remembers
        const float d = remembered ? B.memo_d[i] : distance_to(j);
        set_lower(j, d);
        u = d;
        exact = true;
    }
    uint32_t start = 0;
    for (;;) {
        uint32_t found = K;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t k = q * 64 + lane;
            const bool hit = k < K && k >= start && k != j && u > lw[q] && u > 0.5f * pairw[(size_t)j * K + k];
            const unsigned long long mask = __ballot(hit);
            if (mask && found == K) found = q * 64 + (uint32_t)__ffsll((long long)mask) - 1u;
        }
        if (found == K) break;
        const float d = distance_to(found);
        set_lower(found, d);
        if (d < u) {
            j = found;
            u = d;
            exact = true;
        }
        start = found + 1;
    }
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t k = q * 64 + lane;
        if (k < K) B.lower[i * K + k] = lw[q];
    }
    if (lane == 0) {
        B.j[i] = (uint8_t)j;
        B.u[i] = u;
        B.stale[i] = 0;
        if (exact) memo_store(B, i, j, u);
        if (exact && B.uiv) B.uiv[i] = 0;
        if (var_evaluated) atomicAdd(STAT(M, 0), var_evaluated);
    }
}

// ------------------------------------------------------------------------------------------------
// The stale-bound refresh of step_elkan (elkan.rs:113-117: u = l[j] = distance(x, c_j)) as its own, grouped pass.
// Every point that passes the filter with a stale bound needs exactly this one distance before its candidate loop, and
// all the points of a cluster need it against the SAME centroid: they are bucketed by assignment and solved two per
// wavefront (wave_sinkhorn_costG<2>, point first).  k_elkan_step then finds them fresh.  A point whose refreshed bound
// no longer passes the filter has no candidates either (mid[j] = min_k P[j][k] / 2), so skipping its loop changes
// nothing.  Which two points share a wavefront is decided by atomics and does not matter: a solve's operations do not
// depend on its partner.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool needs_refresh(const Bounds& B, const Refresh& R, const float* mid, uint64_t i) {
    if (R.code) return R.code[i] == R.want;  // interval mode: k_rb_prepare / k_refresh_interval / the exactify pass decided
    return B.stale[i] && B.u[i] > mid[B.j[i]] && R.nsup[i] <= PAIR_ROWS;
}
// stale bounds whose refresh is remembered: Bounds::refresh without the solve
__global__ __launch_bounds__(256) void k_refresh_memo(Bounds B, const float* mid, uint64_t N, uint32_t K, Metric M) {
    unsigned long long hits = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (uint64_t)gridDim.x * 256) {
        const uint32_t j = B.j[i];
        if (B.stale[i] && B.u[i] > mid[j] && memo_valid(B, i, j)) {
            const float d = B.memo_d[i];
            B.u[i] = d;
            B.lower[i * K + j] = d;
            B.stale[i] = 0;
            hits += 1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) hits += __shfl_xor(hits, o, 64);
    if ((threadIdx.x & 63u) == 0 && hits) atomicAdd(STAT(M, 3), hits);  // distances the reference evaluates and this pass remembers
}
__global__ __launch_bounds__(256) void k_refresh_count(Bounds B, Refresh R, const float* mid, uint64_t N, uint32_t K) {
    __shared__ uint32_t c[MAXB];
    for (uint32_t k = threadIdx.x; k < K; k += 256) c[k] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (uint64_t)gridDim.x * 256)
        if (needs_refresh(B, R, mid, i)) atomicAdd(&c[B.j[i]], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < K; k += 256)
        if (c[k]) atomicAdd(&R.count[k], c[k]);
}
__global__ void k_refresh_offsets(Refresh R, uint32_t K) {  // one thread: K <= 256
    uint32_t at = 0;
    for (uint32_t k = 0; k < K; ++k) {
        R.offset[k] = at;
        at += (R.count[k] + 1u) & ~1u;
        R.count[k] = 0;  // becomes the fill cursor
    }
    R.offset[K] = at;
}
__global__ __launch_bounds__(256) void k_refresh_fill(Bounds B, Refresh R, const float* mid, uint64_t N) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (uint64_t)gridDim.x * 256)
        if (needs_refresh(B, R, mid, i)) {
            const uint32_t j = B.j[i];
            R.list[R.offset[j] + atomicAdd(&R.count[j], 1u)] = (uint32_t)i;
        }
}
// exactify (interval mode, refresh_bound.hpp): `cs` is the PREVIOUS centroid set; the solve returns the value the last step's refresh
// stood for, Bounds::update's additions of the last drift are replayed on it, and the reference's filter then sees an exact u
// (code_out: 1 = it passes, the refresh against the current centroids follows; 0 = skipped like the reference skips it).
__global__ __launch_bounds__(64) void k_refresh_pairs(Points P, CentroidSet cs, uint32_t K, Metric M, Bounds B, Refresh R, int exactify,
                                                      const float* drift_prev, const float* mid, uint8_t* code_out) {
    LM_TABLES();
    __shared__ GroupLds<2> w;
    const uint32_t e0 = 2u * blockIdx.x;
    if (e0 >= R.offset[K]) return;
    uint32_t lo = 0, hi = K;  // the cluster whose bucket holds entry e0: last k with offset[k] <= e0
    while (hi - lo > 1) {
        const uint32_t mid_k = (lo + hi) / 2;
        if (R.offset[mid_k] <= e0) lo = mid_k;
        else hi = mid_k;
    }
    const uint32_t j = lo;
    uint32_t ip[2], n[2];
    float sp[2];
    const uint32_t m = wave_load_centroid(cs, j, w.supC, w.lnC);
#pragma unroll
    for (uint32_t h = 0; h < 2; ++h) {
        ip[h] = R.list[e0 + h];
        const bool real = ip[h] != 0xffffffffu;
        const uint64_t i = real ? ip[h] : 0;
        const uint32_t got = pair_load_hist(P.counts + i * P.stride, P.weight[i], M.bins, w.supP[h], w.lnP[h], PAIR_ROWS);
        n[h] = real ? got : 0u;
        sp[h] = P.self[i];
    }
    float xy[2];
    wave_sinkhorn_costG<2>(w, m, n, M, false, xy);  // distance(point, centroid)
    const uint32_t lane = lane_id();
    const float sc = cs.self[j];
    if (lane < 2) {
        const uint32_t i = lane ? ip[1] : ip[0];
        if (i != 0xffffffffu) {
            const float d = rp_maxf((lane ? xy[1] : xy[0]) - 0.5f * (lane ? sp[1] : sp[0]) - 0.5f * sc, 0.0f);
            if (exactify) {
                const float dp = drift_prev[j], u = d + dp;  // Bounds::update: error += movement, lower = (lower - movement).max(0)
                B.u[i] = u;
                B.lower[(uint64_t)i * K + j] = rp_maxf(d - dp, 0.0f);
                B.uiv[i] = 0;
                code_out[i] = u > mid[j] ? 1 : 0;
                atomicAdd(STAT(M, 6), 1ull);  // a solve the reference does not repeat: counted apart
            } else {
                B.u[i] = d;
                B.lower[(uint64_t)i * K + j] = d;
                B.stale[i] = 0;
                if (B.uiv) B.uiv[i] = 0;
                memo_store(B, i, j, d);
                atomicAdd(STAT(M, 0), 1ull);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Equity::variation against ALL K centroids with the centroid CDFs in REGISTERS (turn layer: bins = 101).
//
// Workgroup = 4 waves x VB points.  Wave q owns centroids q*64 + lane: its CDF column (BINS running sums of the
// transposed density table, the reference's left fold) is loaded once per workgroup into BINS VGPRs.  The points'
// CDFs are built once in LDS (division in parallel, the sequential prefix by one lane per point) and broadcast to the
// waves as float4.  A (point, 64 centroids) step is then 2 VALU instructions per bin: t = cx_b - CY_b, s += |t|
// — the same operations, in the same order, as equity.rs:41-53.
// ------------------------------------------------------------------------------------------------

template <int BINS>
struct VarLds {
    static constexpr int ROW = (BINS + 3) & ~3;
    float cx[VB][ROW];       // point CDFs
    uint32_t active[VB];     // compacted list of points that need distances
    uint32_t n_active;
};

// phase A: CDFs of the workgroup's points into LDS
template <int BINS>
__device__ __forceinline__ void var_point_cdfs(VarLds<BINS>& L, const Points& P, uint64_t i0, uint32_t np) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t e = tid; e < np * BINS; e += 256) {
        const uint32_t pl = e / BINS, b = e % BINS;
        const uint64_t i = i0 + pl;
        L.cx[pl][b] = (float)P.counts[i * P.stride + b] / (float)P.weight[i];
    }
    __syncthreads();
    if (tid < np) {
        float acc = 0.0f;
        for (int b = 0; b < BINS; ++b) {
            acc += L.cx[tid][b];
            L.cx[tid][b] = acc;
        }
    }
    __syncthreads();
}
// the CDF column of centroid k into registers
template <int BINS>
__device__ __forceinline__ void var_centroid_cdf(float (&CY)[BINS], const CentroidSet& cs, uint32_t K, uint32_t k) {
    float acc = 0.0f;
#pragma unroll
    for (int b = 0; b < BINS; ++b) {
        acc += k < K ? cs.dens[(size_t)b * K + k] : 0.0f;
        CY[b] = acc;
    }
}
template <int BINS>
__device__ __forceinline__ float var_distance(const float* cxrow, const float (&CY)[BINS]) {
    float s = 0.0f;
#pragma unroll
    for (int b = 0; b < BINS; b += 4) {
        const float4 c = *reinterpret_cast<const float4*>(cxrow + b);
        s += rp_absf(c.x - CY[b]);
        if (b + 1 < BINS) s += rp_absf(c.y - CY[b + 1]);
        if (b + 2 < BINS) s += rp_absf(c.z - CY[b + 2]);
        if (b + 3 < BINS) s += rp_absf(c.w - CY[b + 3]);
    }
    return s / (float)BINS;
}

// Elkan::neighbor for every point (init_bounds / lookup / step_naive), variation metric
template <int BINS>
__global__ __launch_bounds__(256) void k_neighbor_var(Points P, CentroidSet cs, uint32_t K, Metric M, uint8_t* out_j,
                                                      float* out_d, Bounds init) {
    __shared__ __attribute__((aligned(16))) VarLds<BINS> L;
    __shared__ float wbest[VB][4];
    __shared__ uint32_t wbk[VB][4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, q = tid >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * VB;
    const uint32_t np = (uint32_t)min((uint64_t)VB, P.N - i0);
    var_point_cdfs<BINS>(L, P, i0, np);
    const uint32_t k = q * 64 + lane;
    float CY[BINS];
    var_centroid_cdf<BINS>(CY, cs, K, k);
    for (uint32_t pl = 0; pl < np; ++pl) {
        float best = var_distance<BINS>(L.cx[pl], CY);
        uint32_t bk = k < K ? k : 0xffffffffu;
        // first minimum in ascending k within the wave
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const uint32_t ok = __shfl_xor(bk, o, 64);
            if (ok != 0xffffffffu && (bk == 0xffffffffu || ob < best || (ob == best && ok < bk))) {
                best = ob;
                bk = ok;
            }
        }
        if (lane == 0) {
            wbest[pl][q] = best;
            wbk[pl][q] = bk;
        }
    }
    __syncthreads();
    if (tid < np) {
        float best = wbest[tid][0];
        uint32_t bk = wbk[tid][0];
        for (uint32_t w = 1; w < 4; ++w)
            if (wbk[tid][w] != 0xffffffffu && wbest[tid][w] < best) {  // strict: ties keep the lower index
                best = wbest[tid][w];
                bk = wbk[tid][w];
            }
        const uint64_t i = i0 + tid;
        if (out_j) out_j[i] = (uint8_t)bk;
        if (out_d) out_d[i] = best;
        if (init.j) {
            init.j[i] = (uint8_t)bk;
            init.u[i] = best;
            init.stale[i] = 0;
        }
        atomicAdd(STAT(M, 0), (unsigned long long)K);
    }
    if (init.lower)
        for (uint64_t e = tid; e < (uint64_t)np * K; e += 256) init.lower[i0 * K + e] = 0.0f;
}

// Elkan::step_elkan's per-point part (elkan.rs:144-168), variation metric: distances of the unfiltered points of
// the workgroup to every centroid (phase B), then the sequential candidate rule replayed per point (phase C)
template <int BINS>
__global__ __launch_bounds__(256) void k_elkan_step_var(Points P, CentroidSet cs, uint32_t K, Metric M, Bounds B,
                                                        const float* pairw, const float* mid) {
    __shared__ __attribute__((aligned(16))) VarLds<BINS> L;
    __shared__ float dist[VB][MAXB];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, q = tid >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * VB;
    const uint32_t np = (uint32_t)min((uint64_t)VB, P.N - i0);
    if (tid < 64) {  // filter(|b| b.u() > midpoints[b.j()]), compacted in point order
        bool need = false;
        if (tid < np) need = B.u[i0 + tid] > mid[B.j[i0 + tid]];
        const unsigned long long mask = __ballot(need);
        if (need) L.active[__popcll(mask & ((1ull << tid) - 1ull))] = tid;
        if (tid == 0) L.n_active = (uint32_t)__popcll(mask);
    }
    __syncthreads();
    const uint32_t na = L.n_active;
    if (na == 0) return;
    var_point_cdfs<BINS>(L, P, i0, np);
    {
        const uint32_t k = q * 64 + lane;
        float CY[BINS];
        var_centroid_cdf<BINS>(CY, cs, K, k);
        for (uint32_t a = 0; a < na; ++a) {
            const uint32_t pl = L.active[a];
            const float d = var_distance<BINS>(L.cx[pl], CY);
            if (k < K) dist[a][k] = d;
        }
    }
    __syncthreads();
    // (skipping a wave's 64 centroids when none of them can become a candidate was measured: candidates are spread
    // over all four waves, the test costs more than it saves)
    // STAT 0 counts the distances Elkan's rule EVALUATES (the reference's count: the replay below), STAT 4 the ones computed to feed it
    if (tid == 0) atomicAdd(STAT(M, 4), (unsigned long long)K * na);
    unsigned long long evaluated = 0;
    for (uint32_t a = q; a < na; a += 4) {  // one wave per point, as k_elkan_step
        const uint64_t i = i0 + L.active[a];
        uint32_t j = B.j[i];
        float u = B.u[i];
        float lw[4];
#pragma unroll
        for (uint32_t qq = 0; qq < 4; ++qq) {
            const uint32_t k = qq * 64 + lane;
            lw[qq] = k < K ? B.lower[i * K + k] : 0.0f;
        }
        auto set_lower = [&](uint32_t k, float d) {
            const uint32_t qq = k >> 6;
            if (lane == (k & 63u)) {
                if (qq == 0) lw[0] = d;
                else if (qq == 1) lw[1] = d;
                else if (qq == 2) lw[2] = d;
                else lw[3] = d;
            }
        };
        if (B.stale[i]) {
            const float d = dist[a][j];
            set_lower(j, d);
            u = d;
            evaluated += 1;
        }
        uint32_t start = 0;
        for (;;) {
            uint32_t found = K;
#pragma unroll
            for (uint32_t qq = 0; qq < 4; ++qq) {
                const uint32_t k = qq * 64 + lane;
                const bool hit = k < K && k >= start && k != j && u > lw[qq] && u > 0.5f * pairw[(size_t)j * K + k];
                const unsigned long long mask = __ballot(hit);
                if (mask && found == K) found = qq * 64 + (uint32_t)__ffsll((long long)mask) - 1u;
            }
            if (found == K) break;
            evaluated += 1;
            const float d = dist[a][found];
            set_lower(found, d);
            if (d < u) {
                j = found;
                u = d;
            }
            start = found + 1;
        }
#pragma unroll
        for (uint32_t qq = 0; qq < 4; ++qq) {
            const uint32_t k = qq * 64 + lane;
            if (k < K) B.lower[i * K + k] = lw[qq];
        }
        if (lane == 0) {
            B.j[i] = (uint8_t)j;
            B.u[i] = u;
            B.stale[i] = 0;
        }
    }
    if (lane == 0 && evaluated) atomicAdd(STAT(M, 0), evaluated);
}

// ------------------------------------------------------------------------------------------------
// Elkan::recompute (elkan.rs:128-142): centroid[k] = integer sum of member histograms (Bins::merge)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_recompute(Points P, const uint8_t* assign, uint32_t bins, uint32_t* counts_out,
                                                   uint32_t* weight_out, unsigned long long* sizes_out) {
    __shared__ uint32_t hist[MAXB];
    __shared__ unsigned long long members;
    __shared__ uint16_t queue[RC_CHUNK];
    __shared__ uint32_t qn;
    const uint32_t k = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t b = tid; b < MAXB; b += 256) hist[b] = 0;
    if (tid == 0) members = 0;
    __syncthreads();
    // Rounds of RC_CHUNK points: (1) every thread scans 64 assignment bytes and queues the members of centroid k
    // in LDS, (2) the waves drain the queue four member rows at a time (loads of independent rows in flight
    // together).  Integer sums: the order of members is free.
    uint32_t acc[4] = {0, 0, 0, 0};  // lane owns bins lane, lane+64, lane+128, lane+192
    unsigned long long mine = 0;
    // gridDim.y workgroups share a centroid (chunks interleaved) and ADD their sums to outputs the host has zeroed: one workgroup per
    // centroid was one per CU, 1.3 ms per turn-layer iteration for 176 MB of rows; integer sums, any order
    for (uint64_t cbase = (uint64_t)blockIdx.y * RC_CHUNK; cbase < P.N; cbase += (uint64_t)gridDim.y * RC_CHUNK) {
        if (tid == 0) qn = 0;
        __syncthreads();
        const uint64_t t0 = cbase + (uint64_t)tid * 64;
        if (t0 < P.N) {
            const uint32_t cnt = (uint32_t)min((uint64_t)64, P.N - t0);
            if (cnt == 64 && ((uintptr_t)(assign + t0) & 15u) == 0) {
                const uint4* v = reinterpret_cast<const uint4*>(assign + t0);
#pragma unroll
                for (uint32_t g = 0; g < 4; ++g) {
                    const uint4 w4 = v[g];
                    const uint32_t ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (uint32_t e = 0; e < 16; ++e)
                        if (((ws[e >> 2] >> ((e & 3u) * 8)) & 0xffu) == k) queue[atomicAdd(&qn, 1u)] = (uint16_t)(tid * 64 + g * 16 + e);
                }
            } else {
                for (uint32_t e = 0; e < cnt; ++e)
                    if (assign[t0 + e] == k) queue[atomicAdd(&qn, 1u)] = (uint16_t)(tid * 64 + e);
            }
        }
        __syncthreads();
        const uint32_t n = qn;
        if (wave == 0 && lane == 0) mine += n;
        for (uint32_t m0 = wave * 4; m0 < n; m0 += 16) {
            uint32_t v[4][4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const bool on = m0 + u < n;
                const uint8_t* row = P.counts + (cbase + (on ? queue[m0 + u] : 0u)) * P.stride;
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
                    const uint32_t bb = q * 64 + lane;
                    v[u][q] = (on && bb < bins) ? (uint32_t)row[bb] : 0u;
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u)
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) acc[q] += v[u][q];
        }
        __syncthreads();
    }
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q)
        if (acc[q]) atomicAdd(&hist[q * 64 + lane], acc[q]);
    if (lane == 0) atomicAdd(&members, mine);
    __syncthreads();
    uint32_t wsum = 0;
    for (uint32_t b = tid; b < bins; b += 256) {
        if (gridDim.y == 1) counts_out[(size_t)k * bins + b] = hist[b];
        else if (hist[b]) atomicAdd(&counts_out[(size_t)k * bins + b], hist[b]);
        wsum += hist[b];
    }
    for (int d = 32; d > 0; d >>= 1) wsum += __shfl_xor(wsum, d, 64);
    __syncthreads();
    if (tid == 0) hist[0] = 0;
    __syncthreads();
    if (lane == 0) atomicAdd(&hist[0], wsum);
    __syncthreads();
    if (tid == 0) {
        if (gridDim.y == 1) {
            weight_out[k] = hist[0];
            sizes_out[k] = members;
        } else {
            if (hist[0]) atomicAdd(&weight_out[k], hist[0]);
            if (members) atomicAdd(&sizes_out[k], members);
        }
    }
}

// Elkan::drift (elkan.rs:108-110): distance(new_k, old_k)
__global__ __launch_bounds__(64) void k_drift(CentroidSet nw, CentroidSet old, uint32_t K, Metric M, int kind, float* drift) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint32_t k = blockIdx.x;
    float d;
    if (kind == RP_METRIC_SINKHORN) {
        const uint32_t m = wave_load_centroid(nw, k, w.supA, w.lnA);
        const uint32_t n = wave_load_centroid(old, k, w.supB, w.lnB);
        d = wave_divergence(w, m, n, nw.self[k], old.self[k], M);
    } else {
        float cx = 0.0f, cy = 0.0f, s = 0.0f;
        for (uint32_t t = 0; t < M.bins; ++t) {
            cx += nw.dens[(size_t)t * K + k];
            cy += old.dens[(size_t)t * K + k];
            s += rp_absf(cx - cy);
        }
        d = s / (float)M.bins;
        if (lane_id() == 0) atomicAdd(STAT(M, 0), 1ull);
    }
    if (lane_id() == 0) drift[k] = d;
}

// Elkan::drift and the centroids' OT(c, c) with four wavefronts per solve (wave_sinkhorn_cost<256>): Sinkhorn layers
__global__ __launch_bounds__(256) void k_drift_block(CentroidSet nw, CentroidSet old, uint32_t K, Metric M, float* drift) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint32_t k = blockIdx.x;
    const uint32_t m = wave_load_centroid(nw, k, w.supA, w.lnA);  // every wavefront stores the same values
    const uint32_t n = wave_load_centroid(old, k, w.supB, w.lnB);
    const float d = wave_divergence<256u>(w, m, n, nw.self[k], old.self[k], M);
    if (threadIdx.x == 0) drift[k] = d;
}
__global__ __launch_bounds__(256) void k_self_block(CentroidSet cs, uint32_t K, Metric M) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint32_t k = blockIdx.x;
    const uint32_t m = wave_load_centroid(cs, k, w.supA, w.lnA);
    (void)wave_load_centroid(cs, k, w.supB, w.lnB);
    const float c = wave_sinkhorn_cost<256u>(w, m, m, M);
    if (threadIdx.x == 0) cs.self[k] = c;
}

// Bounds::update (bounds.rs:69-77): the HBM-streaming part of an iteration (N*K lower bounds read + written)
__global__ __launch_bounds__(256) void k_bounds_update(Bounds B, uint64_t N, uint32_t K, const float* drift) {
    __shared__ float dr[MAXB];
    for (uint32_t k = threadIdx.x; k < K; k += 256) dr[k] = drift[k];
    __syncthreads();
    const uint64_t total = N * K;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
        const uint32_t k = (uint32_t)(e % K);
        B.lower[e] = rp_maxf(B.lower[e] - dr[k], 0.0f);
    }
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (uint64_t)gridDim.x * 256) {
        B.u[i] = B.u[i] + dr[B.j[i]];
        if (B.ulo) B.ulo[i] = B.ulo[i] + dr[B.j[i]];  // the same addition on the interval's lower end: rounding is monotone
        B.stale[i] = 1;
    }
}

// Prior::tally's reassignment count (prior.rs:35-47); sizes come from k_recompute
__global__ void k_tally(const uint8_t* j, uint8_t* prior, uint64_t N, unsigned long long* moved) {
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (uint64_t)gridDim.x * blockDim.x) {
        if (j[i] != prior[i]) {
            mine += 1;
            prior[i] = j[i];
        }
    }
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(moved, mine);
}

// distance(point, centroid[j]) for Elkan::rms_with (elkan.rs:191-200)
__global__ __launch_bounds__(64) void k_point_dist(Points P, CentroidSet cs, uint32_t K, Metric M, int kind, const uint8_t* j,
                                                   float* out) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint64_t i = blockIdx.x;
    const uint32_t k = j[i];
    float d;
    if (kind == RP_METRIC_SINKHORN) {
        const uint32_t m = wave_load_hist(P.counts + i * P.stride, P.weight[i], M.bins, w.supA, w.lnA);
        const uint32_t n = wave_load_centroid(cs, k, w.supB, w.lnB);
        d = wave_divergence(w, m, n, P.self[i], cs.self[k], M);
    } else {
        wave_point_density(P, i, M.bins, w.f);
        float cx = 0.0f, cy = 0.0f, s = 0.0f;
        for (uint32_t t = 0; t < M.bins; ++t) {
            cx += w.f[t];
            cy += cs.dens[(size_t)t * K + k];
            s += rp_absf(cx - cy);
        }
        d = s / (float)M.bins;
        if (lane_id() == 0) atomicAdd(STAT(M, 0), 1ull);
    }
    if (lane_id() == 0) out[i] = d;
}

// ------------------------------------------------------------------------------------------------
// variation(point i, ONE centroid) with a LANE per point (k-means++ rounds, rms): the centroid's density column is
// wave uniform, the point's row is read by its own lane.  Same folds as equity.rs:41-53.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_variation(const Points& P, uint64_t i, const CentroidSet& cs, uint32_t K, uint32_t k,
                                                uint32_t bins) {
    const uint8_t* row = P.counts + i * P.stride;
    const float fw = (float)P.weight[i];
    float cx = 0.0f, cy = 0.0f, s = 0.0f;
    if ((P.stride & 15u) == 0u) {
        // rows padded to 16 bytes (every layer created from host counts): the lane's row as 16-byte loads, the next one in flight while
        // this one is folded.  A byte load per bin made every wave instruction touch 64 cache lines for 64 bytes, and with 6.6 KB of rows
        // per wavefront the lines were evicted between two of a lane's bytes: 0.7 ms per k-means++ round of the turn layer for 176 MB.
        // Same folds in the same order (equity.rs:41-53).
        const uint4* row4 = reinterpret_cast<const uint4*>(row);
        uint4 q = row4[0];
        for (uint32_t t0 = 0; t0 < bins; t0 += 16u) {
            const uint4 cur = q;
            if (t0 + 16u < bins) q = row4[(t0 >> 4) + 1u];
            const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (uint32_t b = 0; b < 16u; ++b) {
                const uint32_t t = t0 + b;
                if (t < bins) {
                    cx += (float)((w[b >> 2] >> (8u * (b & 3u))) & 255u) / fw;
                    cy += cs.dens[(size_t)t * K + k];
                    s += rp_absf(cx - cy);
                }
            }
        }
        return s / (float)bins;
    }
    for (uint32_t t = 0; t < bins; ++t) {
        cx += (float)row[t] / fw;
        cy += cs.dens[(size_t)t * K + k];
        s += rp_absf(cx - cy);
    }
    return s / (float)bins;
}
__global__ __launch_bounds__(256) void k_kpp_update_var(Points P, CentroidSet cs, uint32_t k, uint32_t K, Metric M, float* pot) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) atomicAdd(STAT(M, 0), (unsigned long long)P.N);
    if (i >= P.N) return;
    const float d = lane_variation(P, i, cs, K, k, M.bins);
    pot[i] = rp_minf(d * d, pot[i]);
    kpp_note(M, i, k, d);
}
__global__ __launch_bounds__(256) void k_point_dist_var(Points P, CentroidSet cs, uint32_t K, Metric M, const uint8_t* j, float* out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) atomicAdd(STAT(M, 0), (unsigned long long)P.N);
    if (i >= P.N) return;
    out[i] = lane_variation(P, i, cs, K, j[i], M.bins);
}

// ------------------------------------------------------------------------------------------------
// k-means++ (layer.rs:140-181) with the fixed-point weighted draw of rp_math.h
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_kpp_blocksum(const float* pot, uint64_t N, unsigned long long* bsum) {
    __shared__ unsigned long long part[4];
    const uint64_t base = (uint64_t)blockIdx.x * KPP_BLOCK;
    unsigned long long s = 0;
    for (uint32_t t = threadIdx.x; t < KPP_BLOCK; t += 256) {
        const uint64_t i = base + t;
        if (i < N) s += rp_kpp_quant(pot[i]);
    }
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// block sums -> one u64 (single block)
__global__ __launch_bounds__(1024) void k_kpp_total(const unsigned long long* bsum, uint32_t nblocks, unsigned long long* total) {
    __shared__ unsigned long long part[16];
    unsigned long long s = 0;
    for (uint32_t b = threadIdx.x; b < nblocks; b += 1024) s += bsum[b];
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; ++w) t += part[w];
        total[0] = t;
    }
}
// single block: winner = first i whose inclusive quantised prefix exceeds r (r < this shard's total)
__global__ __launch_bounds__(1024) void k_kpp_pick(float* pot, float* kpp_d, uint64_t N, const unsigned long long* bsum, uint32_t nblocks,
                                                   unsigned long long r, unsigned long long* picked) {
    __shared__ unsigned long long strip[1024];
    __shared__ unsigned long long sh_before;
    __shared__ uint32_t sh_block;
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (nblocks + 1023) / 1024;
    unsigned long long s = 0;
    for (uint32_t b = tid * per; b < (tid + 1) * per && b < nblocks; ++b) s += bsum[b];
    strip[tid] = s;
    __syncthreads();
    if (tid == 0) {
        unsigned long long acc = 0;
        uint32_t t = 0;
        while (t < 1023 && acc + strip[t] <= r) acc += strip[t++];
        uint32_t b = t * per;
        while (b + 1 < nblocks && acc + bsum[b] <= r) acc += bsum[b++];
        sh_block = b;
        sh_before = acc;
    }
    __syncthreads();
    const uint64_t i = (uint64_t)sh_block * KPP_BLOCK + tid;
    strip[tid] = i < N ? rp_kpp_quant(pot[i]) : 0ull;
    __syncthreads();
    if (tid == 0) {
        unsigned long long acc = sh_before;
        uint32_t t = 0;
        for (; t + 1 < KPP_BLOCK; ++t) {
            acc += strip[t];
            if (acc > r) break;
        }
        const uint64_t win = (uint64_t)sh_block * KPP_BLOCK + t;
        picked[0] = win;
        pot[win] = 0.0f;  // potentials[i] = 0 (layer.rs:169)
        if (kpp_d) kpp_d[win] = -1.0f;  // no solve stands behind that 0: its neighbor is found the long way
    }
}
// reference-seed mode (rp_kmeans_set_rng RP_RNG_REFERENCE): WeightedIndex::<f32>::new(potentials).sample(rng) (layer.rs:164-166;
// rand 0.9.2 weighted_index.rs).  Its cumulative weights are f32 running sums in index order — f32 addition does not re-associate,
// so the chain is sequential by definition: ONE wavefront walks it, 1024 potentials at a time through LDS (coalesced loads, broadcast
// ds_read_b128), every lane running the same chain of dependent adds: one add per term is all the chain costs (~3 ms per pick at the
// flop layer's 1.3 M points).  Only the running sum at the END of every 1024-chunk is kept (`cum`, N / 1024 floats): the sums are
// non-decreasing (potentials >= 0), so the sample's partition point lies in the first chunk whose end exceeds the draw, and that one
// chunk is walked again.
// v01 = the generator's draw as UniformFloat<f32> maps it to [0, 1) (the host owns the SmallRng: one next_u32 per pick).
__global__ __launch_bounds__(64) void k_kpp_ref_pick(float* pot, float* kpp_d, uint64_t N, float* cum, float v01, unsigned long long* picked,
                                                    unsigned long long* picked_total) {
    __shared__ __attribute__((aligned(16))) float buf[KR_CHUNK];
    const uint32_t ln = threadIdx.x;
    const uint64_t nchunks = (N + KR_CHUNK - 1) / KR_CHUNK;
    auto load_chunk = [&](uint64_t base) {
#pragma unroll
        for (uint32_t q = 0; q < KR_CHUNK / 64u; ++q) {
            const uint64_t i = base + q * 64u + ln;
            buf[q * 64u + ln] = i < N ? pot[i] : 0.0f;  // + 0 past the end leaves the sum as it is
        }
    };
    float run = 0.0f;  // total_weight (0 + w0 = w0 exactly)
    for (uint64_t c = 0; c < nchunks; ++c) {
        load_chunk(c * KR_CHUNK);
        __syncthreads();
#pragma unroll 8
        for (uint32_t j = 0; j < KR_CHUNK; j += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(buf + j);
            run += w4.x; run += w4.y; run += w4.z; run += w4.w;
        }
        if (ln == 0) cum[c] = run;  // w_0 + ... + w_(1024 c + 1023)
        __syncthreads();
    }
    const float total = run;
    uint64_t win = N;  // invalid weights (total == 0): the reference panics ("valid weights array"); the host falls back
    if (total > 0.0f) {  // wave uniform
        const float x = v01 * rp_uniform_f32_scale(total) + 0.0f;  // UniformFloat::sample: value0_1 * scale + low
        // partition_point(|w| w <= x) over cum[0 .. N-1): the first index whose running sum exceeds x, N - 1 if none does
        __threadfence();
        uint64_t lo = 0, hi = nchunks;  // first chunk whose END sum exceeds x
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            // (an agent-scope load: the sums were written by lane 0 of this wavefront a moment ago)
            if (rp_u2f(__hip_atomic_load(reinterpret_cast<const uint32_t*>(cum) + mid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) <= x) lo = mid + 1;
            else hi = mid;
        }
        win = N - 1;
        if (lo < nchunks) {
            float r2 = lo ? rp_u2f(__hip_atomic_load(reinterpret_cast<const uint32_t*>(cum) + lo - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0f;
            load_chunk(lo * KR_CHUNK);
            __syncthreads();
            uint32_t at = KR_CHUNK;
            for (uint32_t j = 0; j < KR_CHUNK; ++j) {  // the same adds as above, from the same start: the same sums
                r2 += buf[j];
                if (at == KR_CHUNK && r2 > x) at = j;
            }
            if (at < KR_CHUNK) win = min(lo * KR_CHUNK + at, N - 1);
        }
        if (ln == 0) {
            pot[win] = 0.0f;                // potentials[i] = 0 (layer.rs:168)
            if (kpp_d) kpp_d[win] = -1.0f;  // no solve stands behind that 0
        }
    }
    if (ln == 0) picked[0] = win;
    if (ln == 0 && picked_total) *picked_total = rp_f2u(total);  // the probe compares the totals (rp_weighted_index_probe)
}
// potentials <- min(potentials, d(new centroid, point)^2) (layer.rs:170-178): distance(&x, h), centroid first
__global__ __launch_bounds__(64) void k_kpp_update(Points P, CentroidSet cs, uint32_t k, uint32_t K, Metric M, int kind,
                                                   float* pot, const uint32_t* only, const unsigned int* count) {
    LM_TABLES();
    __shared__ WaveLds w;
    if (count && blockIdx.x >= *count) return;
    const uint64_t i = only ? only[blockIdx.x] : blockIdx.x;
    float d;
    if (kind == RP_METRIC_SINKHORN) {
        const uint32_t m = wave_load_centroid(cs, k, w.supA, w.lnA);
        const uint32_t n = wave_load_hist(P.counts + i * P.stride, P.weight[i], M.bins, w.supB, w.lnB);
        d = wave_divergence(w, m, n, cs.self[k], P.self[i], M);
    } else {
        wave_point_density(P, i, M.bins, w.f);
        float cx = 0.0f, cy = 0.0f, s = 0.0f;
        for (uint32_t t = 0; t < M.bins; ++t) {
            cx += cs.dens[(size_t)t * K + k];
            cy += w.f[t];
            s += rp_absf(cx - cy);
        }
        d = s / (float)M.bins;
        if (lane_id() == 0) atomicAdd(STAT(M, 0), 1ull);
    }
    if (lane_id() == 0) {
        pot[i] = rp_minf(d * d, pot[i]);
        kpp_note(M, i, k, d);
    }
}
__global__ void k_fill(float* p, uint64_t n, float v) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// stand-alone batched entry points: one wave per pair of u32 histograms
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_pair_sinkhorn(const uint32_t* mu, const uint32_t* nu, Metric M, int divergence,
                                                      float* out, uint32_t* iters_out) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint64_t p = blockIdx.x;
    const uint32_t bins = M.bins, lane = lane_id();
    uint32_t wa = 0, wb = 0;
    for (uint32_t b = lane; b < bins; b += 64) {
        wa += mu[p * bins + b];
        wb += nu[p * bins + b];
    }
    for (int d = 32; d > 0; d >>= 1) {
        wa += __shfl_xor(wa, d, 64);
        wb += __shfl_xor(wb, d, 64);
    }
    float xx = 0.0f, yy = 0.0f;
    if (divergence) {
        uint32_t m = wave_load_hist(mu + p * bins, wa, bins, w.supA, w.lnA);
        for (uint32_t k = lane; k < m; k += 64) { w.supB[k] = w.supA[k]; w.lnB[k] = w.lnA[k]; }
        __syncthreads();
        xx = wave_sinkhorn_cost(w, m, m, M);
        __syncthreads();
        m = wave_load_hist(nu + p * bins, wb, bins, w.supA, w.lnA);
        for (uint32_t k = lane; k < m; k += 64) { w.supB[k] = w.supA[k]; w.lnB[k] = w.lnA[k]; }
        __syncthreads();
        yy = wave_sinkhorn_cost(w, m, m, M);
        __syncthreads();
    }
    const uint32_t m = wave_load_hist(mu + p * bins, wa, bins, w.supA, w.lnA);
    const uint32_t n = wave_load_hist(nu + p * bins, wb, bins, w.supB, w.lnB);
    unsigned long long before = 0;
    if (iters_out && lane == 0) before = M.stats[1];
    const float xy = wave_sinkhorn_cost(w, m, n, M);
    float r = xy;
    if (divergence) r = rp_maxf(xy - 0.5f * xx - 0.5f * yy, 0.0f);
    if (lane == 0) out[p] = r;
    (void)before;
}
// Coupling::flow (monge/src/coupling.rs:23-51 as Sinkhorn implements it, sinkhorn.rs:114-116,202-204) of ONE minimised pair:
// coupling(x, y) = exp(lhs(x) + rhs(y) - C/T) and flow = coupling * C on supp(mu) x supp(nu), 0 elsewhere
__global__ __launch_bounds__(64) void k_pair_flow(const uint32_t* mu, const uint32_t* nu, Metric M, float* flow, float* coupling) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint32_t bins = M.bins, lane = lane_id();
    uint32_t wa = 0, wb = 0;
    for (uint32_t b = lane; b < bins; b += 64) {
        wa += mu[b];
        wb += nu[b];
    }
    for (int d = 32; d > 0; d >>= 1) {
        wa += __shfl_xor(wa, d, 64);
        wb += __shfl_xor(wb, d, 64);
    }
    const uint32_t m = wave_load_hist(mu, wa, bins, w.supA, w.lnA);
    const uint32_t n = wave_load_hist(nu, wb, bins, w.supB, w.lnB);
    (void)wave_sinkhorn_cost(w, m, n, M);  // leaves the minimised potentials in w.f / w.g
    for (uint32_t i = 0; i < m; ++i) {
        const uint32_t x = w.supA[i];
        for (uint32_t j = lane; j < n; j += 64) {
            const uint32_t y = w.supB[j];
            const float pi = LM_EXPF(w.f[i] + w.g[j] - M.Rt[x * bins + y]);
            if (coupling) coupling[x * bins + y] = pi;
            flow[x * bins + y] = pi * M.Cm[x * bins + y];
        }
    }
}
// iteration counts need a private counter per pair: a second tiny variant keeps the hot kernel lean
__global__ __launch_bounds__(64) void k_pair_iters(const uint32_t* mu, const uint32_t* nu, Metric M, unsigned long long* scratch,
                                                   uint32_t* iters_out) {
    LM_TABLES();
    __shared__ WaveLds w;
    const uint64_t p = blockIdx.x;
    const uint32_t bins = M.bins, lane = lane_id();
    uint32_t wa = 0, wb = 0;
    for (uint32_t b = lane; b < bins; b += 64) {
        wa += mu[p * bins + b];
        wb += nu[p * bins + b];
    }
    for (int d = 32; d > 0; d >>= 1) {
        wa += __shfl_xor(wa, d, 64);
        wb += __shfl_xor(wb, d, 64);
    }
    Metric mine = M;
    mine.stats = scratch + 4 * p;
    const uint32_t m = wave_load_hist(mu + p * bins, wa, bins, w.supA, w.lnA);
    const uint32_t n = wave_load_hist(nu + p * bins, wb, bins, w.supB, w.lnB);
    (void)wave_sinkhorn_cost(w, m, n, mine);
    __syncthreads();
    if (lane == 0) iters_out[p] = (uint32_t)scratch[4 * p + 1];
}
__global__ void k_pair_variation(const uint32_t* x, const uint32_t* y, uint32_t bins, uint64_t pairs, float* out) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pairs) return;
    uint32_t wx = 0, wy = 0;
    for (uint32_t b = 0; b < bins; ++b) {
        wx += x[p * bins + b];
        wy += y[p * bins + b];
    }
    const float fx = (float)wx, fy = (float)wy;
    float cx = 0.0f, cy = 0.0f, s = 0.0f;
    for (uint32_t b = 0; b < bins; ++b) {
        cx += (float)x[p * bins + b] / fx;
        cy += (float)y[p * bins + b] / fy;
        s += rp_absf(cx - cy);
    }
    out[p] = s / (float)bins;
}

#undef LM_EXPF
#undef LM_LOGF
#undef LM_EXP_FLOOR2
#undef LM_TABLES

// lloyd.hip — Elkan k-means over Sinkhorn EMD / equity variation on MI355X (gfx950): kernels + rp_kmeans_* ABI.
//
// Reference path: crates/elkan (Elkan<K,N>::{init_bounds, neighbor, pairwises, step_elkan, step_naive}),
// crates/lloyd (Layer, Kmeans, Sinkhorn, Metric::emd, Equity::variation).  MI355X mapping:
//
//   one WAVEFRONT per (point) work item.  A Sinkhorn solve (sinkhorn.rs:77-92) is wave-cooperative:
//   lane i owns support row i and walks the other support sequentially, so every softmin sum is the
//   reference's left fold in ascending bin order — bit-exact with the CPU oracle — and needs no cross-lane
//   reduction.  Potentials, supports and log-densities live in 6 KB of LDS per wave; the ground cost C and
//   C/T (bins x bins f32, 2 x 256 KB) are L2-resident and read row-wise (coalesced or in-row gathers).
//   Elkan's candidate loop (elkan.rs:153-168) keeps its sequential semantics per point: the wave evaluates
//   has_shifted() for all K centroids at once (ballot), solves the first hit, re-tests with the updated (j,u).
//   Centroids are exact integer sums (bins.rs:75-82): order-free, so recompute() is a parallel reduction.
//
// Everything f32 is spelled with the primitives of include/rp_math.h and compiled -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <atomic>
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/rp_math.h"
#include "../../include/rp_refrng.h"
#include "../../include/rp_libm_glibc.h"
#include "rp_internal.h"

namespace rp {

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define MAXB 256  // bins and K are both <= 256 (Abstraction index is 8 bits, kicker/src/abstraction.rs:22-23)

struct Metric {
    const float* Cm;  // [bins][bins] raw_distance (0 on the diagonal)          metric.rs:41-55
    const float* Rt;  // [bins][bins] raw_distance / temperature                sinkhorn.rs:129-131
    uint32_t bins;
    uint32_t iters;
    float tol;
    unsigned long long* stats;  // [0] distances, [1] sinkhorn iterations, [2] exp evaluations of the softmin/cost loops
    uint32_t stat_stripes;      // the counters are striped over this many 128-byte lines (STAT): every wavefront adds to
                                // them once per solve, and adds to ONE address serialise at its L2 channel (~9 ns each)
    // k-means++ computes distance(centroid_k, point) — the very value Elkan::neighbor needs (elkan.rs:68-77: the same
    // centroid-first call, first minimum wins).  Every solved distance is noted: nearest centroid so far and its distance.
    // -1 = not known (the picked point, whose potential is set to 0 without a solve).  See k_init_from_kpp.
    float* kpp_d;               // [N] or NULL
    uint8_t* kpp_j;             // [N]
};
__device__ __forceinline__ void kpp_note(const Metric& M, uint64_t i, uint32_t k, float d) {
    if (M.kpp_d && d < M.kpp_d[i]) {  // strict: the first minimum in centroid order stays (a NaN never enters, -1 never leaves)
        M.kpp_d[i] = d;
        M.kpp_j[i] = (uint8_t)k;
    }
}
#define STAT_STRIDE 16u  // u64 per stripe
#define KM_STAT_STRIPES 256u
__device__ __forceinline__ unsigned long long* STAT(const Metric& M, uint32_t k) {
    return M.stats + (size_t)(blockIdx.x % M.stat_stripes) * STAT_STRIDE + k;
}


// one prepared centroid set: integer sums + the derived support / log-density tables
struct CentroidSet {
    uint32_t* counts;  // [K][bins]
    uint32_t* weight;  // [K]
    uint32_t* n;       // [K]  support size
    uint16_t* sup;     // [K][MAXB] support bins ascending
    float* lnd;        // [K][MAXB] ln(density) on the support
    float* dens;       // [bins][K] density, transposed (variation path)
    float* densR;      // [K][256] density by centroid, 0 off the support and past `bins` (the MFMA bound's mu operand)
    float* mincT;      // [256][256] mincT[y][k] = min over x in supp(centroid k) of C(x, y): the column-marginal bound
    float* self;       // [K] OT(c,c)
};

struct Points {
    const uint8_t* counts;  // [N][stride]
    const uint32_t* weight; // [N]
    const float* self;      // [N] OT(p,p)
    uint32_t stride;
    uint64_t N;
};

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

#include "sinkhorn_bound.hpp"

// Bins::support + Bins::density (bins.rs:58-60,84-88) of a dense histogram into LDS; returns the support size
template <typename CT, int LIBM = 0>  // LIBM 1: ln as glibc computes it (rp_sinkhorn_set_libm; the stand-alone operators only)
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
            if constexpr (LIBM) lnd[r] = rp_glibc_logf((float)c / fw);
            else lnd[r] = rp_logf((float)c / fw);
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
    e0 = rp_exp_floor2(e0);
    e1 = rp_exp_floor2(e1);
    e2 = rp_exp_floor2(e2);
    e3 = rp_exp_floor2(e3);
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
    e0 = rp_exp_floor2(e0);
    s += e0.x;
    if (r > 1) s += e0.y;
    if (r > 2) {
        e1 = rp_exp_floor2(e1);
        s += e1.x;
        if (r > 3) s += e1.y;
    }
    if (r > 4) {
        e2 = rp_exp_floor2(e2);
        s += e2.x;
        if (r > 5) s += e2.y;
    }
    if (r > 6) {
        e3 = rp_exp_floor2(e3);
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
        e0 = rp_exp_floor2(e0);
        e1 = rp_exp_floor2(e1);
        e2 = rp_exp_floor2(e2);
        e3 = rp_exp_floor2(e3);
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
template <bool PIPE = true>
__device__ __forceinline__ float softmin_sum(const uint16_t* sup, const float* pot, uint32_t cnt,
                                              __amdgpu_buffer_rsrc_t rt, uint32_t bins, uint32_t xi) {
    const uint32_t rowb = bins * 4u, xoff = xi * 4u;
    float s = 0.0f;
    uint32_t j = 0;
    if (!PIPE) {  // one group in flight: 16 fewer VGPRs (the two-point kernels keep 7 waves per SIMD with it)
        for (; j + 8 <= cnt; j += 8) {
            SoftminGroup cur;
            softmin_fetch(cur, sup, pot, j, rt, rowb, xoff);
            s = softmin_fold(s, cur);
        }
    } else if (cnt >= 8) {  // software pipeline: the loads of group j+8 are in flight while group j is exponentiated
        SoftminGroup cur, nxt;
        softmin_fetch(cur, sup, pot, 0, rt, rowb, xoff);
        for (j = 8; j + 8 <= cnt; j += 8) {
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

__device__ __forceinline__ float wave_sinkhorn_cost(WaveLds& w, uint32_t m, uint32_t n, const Metric& M) {
    const uint32_t lane = lane_id();
    if (m == 0 || n == 0) return 0.0f;  // empty support: the cost sum is empty
    const uint32_t bins = M.bins;
    const __amdgpu_buffer_rsrc_t rt = rt_resource(M);
    const float lu = rp_logf(1.0f / (float)m), ru = rp_logf(1.0f / (float)n);  // Potential::uniform (phi.rs:34-39)
    for (uint32_t i = lane; i < m; i += 64) w.f[i] = lu;
    for (uint32_t j = lane; j < n; j += 64) w.g[j] = ru;
    __syncthreads();
    uint32_t t = 0;
    for (; t < M.iters; ++t) {
        // lhs(): f(x) <- ln mu(x) - ln sum_y max(exp(g(y) - C(x,y)/T), MIN_POSITIVE)   (sinkhorn.rs:94-102,119-128)
        for (uint32_t i0 = 0; i0 < m; i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool act = i < m;
            const uint32_t x = act ? w.supA[i] : w.supA[0];
            const float s = softmin_sum(w.supB, w.g, n, rt, bins, x);
            if (act) {
                const float nf = w.lnA[i] - rp_logf(s);
                w.tmp[i] = rp_absf(rp_expf(nf) - rp_expf(w.f[i]));  // delta term (sinkhorn.rs:134-139)
                w.f[i] = nf;
            }
        }
        __syncthreads();
        const float lhs_err = lds_sum_in_order(w.tmp, m);
        __syncthreads();
        // rhs(): sees the fresh lhs (Gauss-Seidel, sinkhorn.rs:80-87)
        for (uint32_t j0 = 0; j0 < n; j0 += 64) {
            const uint32_t j = j0 + lane;
            const bool act = j < n;
            const uint32_t y = act ? w.supB[j] : w.supB[0];
            const float s = softmin_sum(w.supA, w.f, m, rt, bins, y);
            if (act) {
                const float ng = w.lnB[j] - rp_logf(s);
                w.tmp[j] = rp_absf(rp_expf(ng) - rp_expf(w.g[j]));
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
    float cost = 0.0f;
    for (uint32_t i = 0; i < m; ++i) {
        const uint32_t x = w.supA[i];
        const float fi = w.f[i];
        for (uint32_t j = lane; j < n; j += 64) {
            const uint32_t y = w.supB[j];
            const float c = M.Cm[x * bins + y];
            w.tmp[j] = rp_expf(fi + w.g[j] - M.Rt[x * bins + y]) * c;
        }
        __syncthreads();
        for (uint32_t j = 0; j < n; ++j) cost += w.tmp[j];
        __syncthreads();
    }
    return cost;
}

// The same solve in the REFERENCE'S OWN libm arithmetic on a glibc host (rp_sinkhorn_set_libm(RP_LIBM_GLIBC)): f32::exp / f32::ln =
// glibc's expf / logf, which compute in double around small tables (include/rp_libm_glibc.h, equal to the platform's functions on all
// 2^32 inputs).  Same loops, same order, one term at a time: lane i owns row i and folds the other support ascending.  This is the
// stand-alone operators' path (rp_sinkhorn_cost / _divergence / _flow), where "identical to a Rust build" can be asked of one
// solve; the clustering keeps the f32 contract (rp_expf / rp_logf, <= 1 ulp away, packed and pipelined above).
__device__ float wave_sinkhorn_cost_glibc(WaveLds& w, uint32_t m, uint32_t n, const Metric& M) {
    const uint32_t lane = lane_id();
    if (m == 0 || n == 0) return 0.0f;
    const uint32_t bins = M.bins;
    const float lu = rp_glibc_logf(1.0f / (float)m), ru = rp_glibc_logf(1.0f / (float)n);
    for (uint32_t i = lane; i < m; i += 64) w.f[i] = lu;
    for (uint32_t j = lane; j < n; j += 64) w.g[j] = ru;
    __syncthreads();
    uint32_t t = 0;
    for (; t < M.iters; ++t) {
        for (uint32_t i = lane; i < m; i += 64) {
            const uint32_t x = w.supA[i];
            float s = 0.0f;
            for (uint32_t j = 0; j < n; ++j) s += rp_maxf(rp_glibc_expf(w.g[j] - M.Rt[x * bins + w.supB[j]]), RP_EPSILON);
            const float nf = w.lnA[i] - rp_glibc_logf(s);
            w.tmp[i] = rp_absf(rp_glibc_expf(nf) - rp_glibc_expf(w.f[i]));
            w.f[i] = nf;
        }
        __syncthreads();
        const float lhs_err = lds_sum_in_order(w.tmp, m);
        __syncthreads();
        for (uint32_t j = lane; j < n; j += 64) {
            const uint32_t y = w.supB[j];
            float s = 0.0f;
            for (uint32_t i = 0; i < m; ++i) s += rp_maxf(rp_glibc_expf(w.f[i] - M.Rt[y * bins + w.supA[i]]), RP_EPSILON);
            const float ng = w.lnB[j] - rp_glibc_logf(s);
            w.tmp[j] = rp_absf(rp_glibc_expf(ng) - rp_glibc_expf(w.g[j]));
            w.g[j] = ng;
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
    float cost = 0.0f;
    for (uint32_t i = 0; i < m; ++i) {
        const uint32_t x = w.supA[i];
        const float fi = w.f[i];
        for (uint32_t j = lane; j < n; j += 64) {
            const uint32_t y = w.supB[j];
            w.tmp[j] = rp_glibc_expf(fi + w.g[j] - M.Rt[x * bins + y]) * M.Cm[x * bins + y];
        }
        __syncthreads();
        for (uint32_t j = 0; j < n; ++j) cost += w.tmp[j];
        __syncthreads();
    }
    return cost;
}
template <int LIBM>
__device__ __forceinline__ float pair_cost(WaveLds& w, uint32_t m, uint32_t n, const Metric& M) {
    if constexpr (LIBM) return wave_sinkhorn_cost_glibc(w, m, n, M);
    else return wave_sinkhorn_cost(w, m, n, M);
}

// Sinkhorn::divergence (sinkhorn.rs:166-171) with memoised self terms
__device__ __forceinline__ float wave_divergence(WaveLds& w, uint32_t m, uint32_t n, float selfA, float selfB,
                                                 const Metric& M) {
    const float xy = wave_sinkhorn_cost(w, m, n, M);
    if (lane_id() == 0) atomicAdd(STAT(M, 0), 1ull);
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
#define PAIR_ROWS 32u
#define QUAD_ROWS 16u
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
    for (uint32_t h = 0; h < G; ++h) cost_out[h] = 0.0f;
    if (m == 0) return;
    bool active[G];
    uint32_t iters_done[G];
    const float lc = rp_logf(1.0f / (float)m);
#pragma unroll
    for (uint32_t h = 0; h < G; ++h) {
        active[h] = n[h] > 0;
        iters_done[h] = 0;
        for (uint32_t i = lane; i < m; i += 64) w.potC[h][i] = lc;
        if (n[h] > 0 && lane < n[h]) w.potP[h][lane] = rp_logf(1.0f / (float)n[h]);
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
                const float nv = w.lnC[i] - rp_logf(s);
                w.tmpC[h][i] = rp_absf(rp_expf(nv) - rp_expf(w.potC[h][i]));
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
            const float nv = w.lnC[r] - rp_logf(s);
            w.tmpC[grp][r] = rp_absf(rp_expf(nv) - rp_expf(w.potC[grp][r]));
            w.potC[grp][r] = nv;
        }
    };
    // rows = point bins, all solves at once (columns = the centroid's bins, potential per lane group)
    auto point_rows = [&]() {
        const bool valid = r < nh && pick<G>(active, grp);
        const uint32_t y = w.supP[grp][r < nh ? r : 0u];
        const float s = softmin_sum<false>(w.supC, w.potC[grp], m, rt, bins, y);
        if (valid) {
            const float nv = w.lnP[grp][r] - rp_logf(s);
            w.tmpP[grp][r] = rp_absf(rp_expf(nv) - rp_expf(w.potP[grp][r]));
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
    float cost = 0.0f;
    if (centroid_is_A) {
        for (uint32_t i = 0; i < m; ++i) {
            const uint32_t x = w.supC[i];
            const float fi = w.potC[grp][i];
            if (r < nh) {
                const uint32_t y = w.supP[grp][r];
                w.tmpP[grp][r] = rp_expf(fi + w.potP[grp][r] - M.Rt[x * bins + y]) * M.Cm[x * bins + y];
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
                    w.tmpC[h][j] = rp_expf(fi + w.potC[h][j] - M.Rt[x * bins + y]) * M.Cm[x * bins + y];
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
                lnd[rr] = rp_logf((float)c / fw);
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
__global__ __launch_bounds__(64) void k_prepare_centroids(CentroidSet cs, uint32_t K, Metric M, int kind, uint32_t k0) {
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
    if (kind == RP_METRIC_SINKHORN) self = wave_sinkhorn_cost(w, m, m, M);
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
                                   const Metric& M) {
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
    if (lane == 0) atomicAdd(STAT(M, 0), (unsigned long long)K);
}

// ------------------------------------------------------------------------------------------------
// Elkan::neighbor for every point (elkan.rs:68-77): init_bounds / Layer::lookup / step_naive
// ------------------------------------------------------------------------------------------------
struct Bounds {
    uint8_t* j;      // [N]
    float* u;        // [N]   Bounds::error
    uint8_t* stale;  // [N]
    float* lower;    // [N][K]
    // the last EXACT distance(point, its centroid) and what it was measured against.  Bounds::refresh (bounds.rs:79-83)
    // recomputes distance(point, centroid j) whenever the bound is stale; the distance is a pure function of the two
    // histograms, so while centroid j has not changed (cver[j], bumped when its integer sums change) and the point still
    // belongs to it the refresh would return memo_d bit for bit — the solve is skipped.  Late iterations move a few dozen
    // points: most centroids, hence most refreshes, repeat.
    float* memo_d;          // [N]
    uint32_t* memo_ver;     // [N]  cver[memo_j] at the time; 0 = nothing remembered
    uint8_t* memo_j;        // [N]
    const uint32_t* cver;   // [K]  content version of the centroids in use (starts at 1)
};
__device__ __forceinline__ bool memo_valid(const Bounds& B, uint64_t i, uint32_t j) {
    return B.memo_ver && B.memo_j[i] == (uint8_t)j && B.memo_ver[i] == B.cver[j];
}
__device__ __forceinline__ void memo_store(const Bounds& B, uint64_t i, uint32_t j, float d) {
    if (!B.memo_ver) return;
    B.memo_d[i] = d;
    B.memo_j[i] = (uint8_t)j;
    B.memo_ver[i] = B.cver[j];
}

__global__ __launch_bounds__(64) void k_neighbor(Points P, CentroidSet cs, uint32_t K, Metric M, int kind,
                                                 uint8_t* out_j, float* out_d, Bounds init, const uint32_t* only) {
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

// Elkan::neighbor over the survivors of the MFMA bound (sinkhorn_bound.hpp): the centroids whose bit is set in the
// point's 256-bit mask, in ascending index, first minimum wins (elkan.rs:68-77: min_by keeps the first) — the unpruned
// loop's result bit for bit as long as every minimiser survives.  `audit_*`: RP_LLOYD_AUDIT compares with the unpruned
// pass instead of writing.
__global__ __launch_bounds__(64) void k_neighbor_masked(Points P, CentroidSet cs, uint32_t K, Metric M, const unsigned long long* mask,
                                                        uint8_t* out_j, float* out_d, Bounds init, const uint8_t* hint_j,
                                                        const float* hint_d) {
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
struct KppLists {
    uint32_t* list[3];     // active points that are solved four / two / one per wavefront
    unsigned int* count;   // [3]
};
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
    __shared__ WaveLds w;
    const uint32_t a = blockIdx.x / K, b = blockIdx.x % K;
    if (pver) {
        const uint32_t va = cver[a], vb = cver[b];
        // one decision for the wavefront, taken before lane 0 overwrites what it was taken from
        const bool same = pver[2 * blockIdx.x] == va && pver[2 * blockIdx.x + 1] == vb;
        if (__builtin_amdgcn_readfirstlane((uint32_t)same)) return;
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
        wave_variation_all(w.f, cs, K, M.bins, dv, M);  // distances to every centroid; the replay below uses only
                                                        // the ones the sequential rule would have evaluated
    }
    auto distance_to = [&](uint32_t k) -> float {  // distance(point, centroid k)
        if (kind == RP_METRIC_SINKHORN) {
            const uint32_t n = wave_load_centroid(cs, k, w.supB, w.lnB);
            const float d = wave_divergence(w, m, n, sp, cs.self[k], M);
            __syncthreads();
            return d;
        }
        const uint32_t q = k >> 6, src = k & 63u;
        float mine = q == 0 ? dv[0] : (q == 1 ? dv[1] : (q == 2 ? dv[2] : dv[3]));
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
        const float d = memo_valid(B, i, j) ? B.memo_d[i] : distance_to(j);
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
struct Refresh {
    const uint8_t* nsup;  // [N] support sizes
    uint32_t* count;      // [K]   points needing a refresh per cluster, then the fill cursor
    uint32_t* offset;     // [K+1] start of each cluster's (even-padded) bucket; offset[K] = entries in the list
    uint32_t* list;       // [N + 2K] point indices, 0xffffffff = padding
};
__device__ __forceinline__ bool needs_refresh(const Bounds& B, const Refresh& R, const float* mid, uint64_t i) {
    return B.stale[i] && B.u[i] > mid[B.j[i]] && R.nsup[i] <= PAIR_ROWS;
}
// stale bounds whose refresh is remembered: Bounds::refresh without the solve
__global__ __launch_bounds__(256) void k_refresh_memo(Bounds B, const float* mid, uint64_t N, uint32_t K) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (uint64_t)gridDim.x * 256) {
        const uint32_t j = B.j[i];
        if (B.stale[i] && B.u[i] > mid[j] && memo_valid(B, i, j)) {
            const float d = B.memo_d[i];
            B.u[i] = d;
            B.lower[i * K + j] = d;
            B.stale[i] = 0;
        }
    }
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
__global__ __launch_bounds__(64) void k_refresh_pairs(Points P, CentroidSet cs, uint32_t K, Metric M, Bounds B, Refresh R) {
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
            B.u[i] = d;
            B.lower[(uint64_t)i * K + j] = d;
            B.stale[i] = 0;
            memo_store(B, i, j, d);
            atomicAdd(STAT(M, 0), 1ull);
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
#define VB 32  // points per workgroup

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
    if (tid == 0) atomicAdd(STAT(M, 0), (unsigned long long)K * na);
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
}

// ------------------------------------------------------------------------------------------------
// Elkan::recompute (elkan.rs:128-142): centroid[k] = integer sum of member histograms (Bins::merge)
// ------------------------------------------------------------------------------------------------
#define RC_CHUNK 16384  // points per round (256 threads x 64 assignment bytes)
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
    for (uint64_t cbase = 0; cbase < P.N; cbase += RC_CHUNK) {
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
        counts_out[(size_t)k * bins + b] = hist[b];
        wsum += hist[b];
    }
    for (int d = 32; d > 0; d >>= 1) wsum += __shfl_xor(wsum, d, 64);
    __syncthreads();
    if (tid == 0) hist[0] = 0;
    __syncthreads();
    if (lane == 0) atomicAdd(&hist[0], wsum);
    __syncthreads();
    if (tid == 0) {
        weight_out[k] = hist[0];
        sizes_out[k] = members;
    }
}

// Elkan::drift (elkan.rs:108-110): distance(new_k, old_k)
__global__ __launch_bounds__(64) void k_drift(CentroidSet nw, CentroidSet old, uint32_t K, Metric M, int kind, float* drift) {
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
#define KPP_BLOCK 1024
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
// so the chain is sequential by definition: ONE wavefront walks it, 1024 potentials at a time through LDS; every lane runs the
// same chain off broadcast LDS reads and keeps the sums of its own positions, so loads and stores stay coalesced.  ~4 ns per point:
// 5 ms per pick at the flop layer's 1.3 M points, a few per cent of the round's Sinkhorn solves.
// v01 = the generator's draw as UniformFloat<f32> maps it to [0, 1) (the host owns the SmallRng: one next_u32 per pick).
#define KR_CHUNK 1024u
__global__ __launch_bounds__(64) void k_kpp_ref_pick(float* pot, float* kpp_d, uint64_t N, float* cum, float v01, unsigned long long* picked) {
    __shared__ float buf[KR_CHUNK];
    const uint32_t ln = threadIdx.x;
    float run = 0.0f;  // total_weight (0 + w0 = w0 exactly)
    for (uint64_t base = 0; base < N; base += KR_CHUNK) {
#pragma unroll
        for (uint32_t q = 0; q < KR_CHUNK / 64u; ++q) {
            const uint64_t i = base + q * 64u + ln;
            buf[q * 64u + ln] = i < N ? pot[i] : 0.0f;  // + 0 past the end leaves the sum as it is
        }
        __syncthreads();
        float mine[KR_CHUNK / 64u];
#pragma unroll
        for (uint32_t q = 0; q < KR_CHUNK / 64u; ++q) {
            mine[q] = 0.0f;
#pragma unroll
            for (uint32_t j = 0; j < 64u; ++j) {
                run += buf[q * 64u + j];
                mine[q] = j == ln ? run : mine[q];
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < KR_CHUNK / 64u; ++q) {
            const uint64_t i = base + q * 64u + ln;
            if (i < N) cum[i] = mine[q];  // cum[i] = w_0 + ... + w_i; WeightedIndex keeps i < N - 1, the last one is the total
        }
        __syncthreads();
    }
    __threadfence();
    __syncthreads();
    if (ln != 0) return;
    const float total = run;
    uint64_t win = N;  // invalid weights (total == 0): the reference panics ("valid weights array"); the host falls back
    if (total > 0.0f) {
        const float x = v01 * rp_uniform_f32_scale(total) + 0.0f;  // UniformFloat::sample: value0_1 * scale + low
        uint64_t lo = 0, hi = N - 1;                               // partition_point(|w| w <= x) over cum[0 .. N-1)
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            // (an agent-scope load: the sums were written by the other lanes of this wavefront a moment ago)
            if (rp_u2f(__hip_atomic_load(reinterpret_cast<const uint32_t*>(cum) + mid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) <= x) lo = mid + 1;
            else hi = mid;
        }
        win = lo;
        pot[win] = 0.0f;                // potentials[i] = 0 (layer.rs:168)
        if (kpp_d) kpp_d[win] = -1.0f;  // no solve stands behind that 0
    }
    picked[0] = win;
}
// potentials <- min(potentials, d(new centroid, point)^2) (layer.rs:170-178): distance(&x, h), centroid first
__global__ __launch_bounds__(64) void k_kpp_update(Points P, CentroidSet cs, uint32_t k, uint32_t K, Metric M, int kind,
                                                   float* pot, const uint32_t* only, const unsigned int* count) {
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
template <int LIBM>
__global__ __launch_bounds__(64) void k_pair_sinkhorn(const uint32_t* mu, const uint32_t* nu, Metric M, int divergence,
                                                      float* out, uint32_t* iters_out) {
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
        uint32_t m = wave_load_hist<uint32_t, LIBM>(mu + p * bins, wa, bins, w.supA, w.lnA);
        for (uint32_t k = lane; k < m; k += 64) { w.supB[k] = w.supA[k]; w.lnB[k] = w.lnA[k]; }
        __syncthreads();
        xx = pair_cost<LIBM>(w, m, m, M);
        __syncthreads();
        m = wave_load_hist<uint32_t, LIBM>(nu + p * bins, wb, bins, w.supA, w.lnA);
        for (uint32_t k = lane; k < m; k += 64) { w.supB[k] = w.supA[k]; w.lnB[k] = w.lnA[k]; }
        __syncthreads();
        yy = pair_cost<LIBM>(w, m, m, M);
        __syncthreads();
    }
    const uint32_t m = wave_load_hist<uint32_t, LIBM>(mu + p * bins, wa, bins, w.supA, w.lnA);
    const uint32_t n = wave_load_hist<uint32_t, LIBM>(nu + p * bins, wb, bins, w.supB, w.lnB);
    unsigned long long before = 0;
    if (iters_out && lane == 0) before = M.stats[1];
    const float xy = pair_cost<LIBM>(w, m, n, M);
    float r = xy;
    if (divergence) r = rp_maxf(xy - 0.5f * xx - 0.5f * yy, 0.0f);
    if (lane == 0) out[p] = r;
    (void)before;
}
// Coupling::flow (monge/src/coupling.rs:23-51 as Sinkhorn implements it, sinkhorn.rs:114-116,202-204) of ONE minimised pair:
// coupling(x, y) = exp(lhs(x) + rhs(y) - C/T) and flow = coupling * C on supp(mu) x supp(nu), 0 elsewhere
template <int LIBM>
__global__ __launch_bounds__(64) void k_pair_flow(const uint32_t* mu, const uint32_t* nu, Metric M, float* flow, float* coupling) {
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
    const uint32_t m = wave_load_hist<uint32_t, LIBM>(mu, wa, bins, w.supA, w.lnA);
    const uint32_t n = wave_load_hist<uint32_t, LIBM>(nu, wb, bins, w.supB, w.lnB);
    (void)pair_cost<LIBM>(w, m, n, M);  // leaves the minimised potentials in w.f / w.g
    for (uint32_t i = 0; i < m; ++i) {
        const uint32_t x = w.supA[i];
        for (uint32_t j = lane; j < n; j += 64) {
            const uint32_t y = w.supB[j];
            const float e = w.f[i] + w.g[j] - M.Rt[x * bins + y];
            float pi;
            if constexpr (LIBM) pi = rp_glibc_expf(e);
            else pi = rp_expf(e);
            if (coupling) coupling[x * bins + y] = pi;
            flow[x * bins + y] = pi * M.Cm[x * bins + y];
        }
    }
}
// iteration counts need a private counter per pair: a second tiny variant keeps the hot kernel lean
template <int LIBM>
__global__ __launch_bounds__(64) void k_pair_iters(const uint32_t* mu, const uint32_t* nu, Metric M, unsigned long long* scratch,
                                                   uint32_t* iters_out) {
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
    const uint32_t m = wave_load_hist<uint32_t, LIBM>(mu + p * bins, wa, bins, w.supA, w.lnA);
    const uint32_t n = wave_load_hist<uint32_t, LIBM>(nu + p * bins, wb, bins, w.supB, w.lnB);
    (void)pair_cost<LIBM>(w, m, n, mine);
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

}  // namespace rp

// =================================================================================================
// host side
// =================================================================================================
using namespace rp;

namespace {
struct Clock {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    uint64_t launches = 0;
};
const char* const CLOCK_NAMES[] = {"pairwise", "step", "recompute", "bounds", "neighbor", "selfcost", "kpp", "drift", "mfma_bound"};
enum { CK_PAIRWISE, CK_STEP, CK_RECOMPUTE, CK_BOUNDS, CK_NEIGHBOR, CK_SELF, CK_KPP, CK_DRIFT, CK_BOUND, CK_COUNT };
}  // namespace

#define SB_SAMPLE_STRIDE 521u  // the production self-check of the MFMA prune looks at every 521st point (0.2 % more exact solves)
struct rp_kmeans {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t K = 0, bins = 0, stride = 0;
    uint64_t N = 0;
    int kind = 0;
    rp_sinkhorn_hp hp{};
    uint64_t seed = 0;
    rp_rng_kind rng = RP_RNG_COUNTER;  // rp_kmeans_set_rng: RP_RNG_REFERENCE = the reference's generator and WeightedIndex<f32>
    int street = 0;                    // Street discriminant hashed into that generator's seed (layer.rs:156-158)
    float* kpp_cum = nullptr;          // [N] running sums of the potentials (reference-seed mode)
    std::vector<void*> allocs;
    bool owns_counts = true;
    Points P{};
    Metric M{};
    CentroidSet cs[2]{};
    int cur = 0;
    Bounds B{};
    uint8_t* prior = nullptr;
    float* pairw = nullptr;
    uint32_t* cver = nullptr;   // [K] content versions of the centroids (Bounds::cver)
    uint32_t* pver = nullptr;   // [K*K][2] versions pairw was computed from
    uint32_t* kpp_todo = nullptr;      // init_bounds after k-means++: the points whose neighbor the notes do not settle
    unsigned int* kpp_ntodo = nullptr;
    uint32_t memo_epoch = 0;
    bool memo_dirty = true;     // centroids were installed outside an Elkan step: forget everything at the next one
    bool memo_on = true;        // RP_LLOYD_NO_MEMO=1 switches the remembered refreshes off
    float* mid = nullptr;
    float* drift = nullptr;
    float* pot = nullptr;
    float* pdist = nullptr;
    uint32_t* quads = nullptr;    // [n_quads][4] points with <= QUAD_ROWS support bins, four per wavefront (Sinkhorn)
    uint32_t* pairs = nullptr;    // [n_pairs][2] points with <= PAIR_ROWS support bins, two per wavefront
    uint32_t* singles = nullptr;  // [n_singles] the other points
    uint64_t n_quads = 0, n_pairs = 0, n_singles = 0;
    Refresh refresh{};            // grouped stale-bound refresh (Sinkhorn; null nsup = off)
    // k-means++ with the column-marginal bound (k_kpp_filter): per round, only the points whose potential can still drop
    unsigned char* comm_partial = nullptr;  // rp_kmeans_step_comm's exchange buffer
    bool kpp_lb = false;
    uint8_t* d_nsup = nullptr;
    float* minc = nullptr;        // [bins]
    KppLists kpp{};
    uint64_t kpp_cap[3] = {0, 0, 0};  // points per support class (<= QUAD_ROWS, <= PAIR_ROWS, more): grid bounds
    // the MFMA bound in front of the neighbor passes (sinkhorn_bound.hpp)
    bool sb_on = false, sb_audit = false;
    SbParams sb{};
    uint32_t* sb_list[4] = {nullptr, nullptr, nullptr, nullptr};  // points by ceil(support / 16) = 1..4
    uint32_t sb_count[4] = {0, 0, 0, 0};
    uint32_t* sb_big = nullptr;   // points with more than SB_MAXROWS bins: never pruned
    uint32_t sb_nbig = 0;
    unsigned int* sb_cursor = nullptr;      // [4]
    unsigned long long* sb_mask = nullptr;  // [N][4]
    unsigned long long* sb_stats = nullptr; // striped: survivors, points, column-block iterations, cost passes
    unsigned long long* sb_bad = nullptr;   // [0] audit disagreements, [1] disagreements of the production sample check
    uint32_t* sb_sample = nullptr;          // the sampled points (every SB_SAMPLE_STRIDE-th)
    uint32_t sb_nsample = 0;
    uint64_t sb_sampled = 0;
    bool sb_check_pending = false;
    uint8_t* audit_j = nullptr;
    float* audit_d = nullptr;
    float* sb_d = nullptr;
    float* sb_ub0 = nullptr;      // [N] upper bound handed to the bound kernel
    uint8_t* sb_hint_j = nullptr; // [N]
    unsigned long long* sb_hint_mask = nullptr;  // [N][4]
    bool pot_is_min_d2 = false;   // the k-means++ potentials describe the CURRENT centroids
    uint64_t sb_audited = 0;
    unsigned long long* bsum = nullptr;
    unsigned long long* scal = nullptr;  // [0] picked, [1] moved
    unsigned long long* sizes = nullptr; // [K]
    unsigned long long* stats = nullptr; // [2]
    uint8_t* tmp_j = nullptr;
    uint32_t* hist_stage = nullptr;
    bool bounds_ready = false, centroids_ready = false;
    bool profiling = false;
    Clock clk[CK_COUNT];
};

namespace {

template <typename T>
int dev_alloc(rp_kmeans* h, T** out, size_t count) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    h->allocs.push_back(p);
    *out = reinterpret_cast<T*>(p);
    return RP_OK;
}

void ck_begin(rp_kmeans* h, int id) {
    if (!h->profiling) return;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, h->stream);
    h->clk[id].pending.emplace_back(a, b);
}
void ck_end(rp_kmeans* h, int id) {
    if (!h->profiling) return;
    (void)hipEventRecord(h->clk[id].pending.back().second, h->stream);
    h->clk[id].launches += 1;
}
void ck_drain(rp_kmeans* h) {
    for (auto& c : h->clk) {
        for (auto& pr : c.pending) {
            float ms = 0.0f;
            (void)hipEventSynchronize(pr.second);
            (void)hipEventElapsedTime(&ms, pr.first, pr.second);
            c.total_ms += ms;
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        c.pending.clear();
    }
}

int alloc_centroid_set(rp_kmeans* h, CentroidSet* cs) {
    int rc;
    if ((rc = dev_alloc(h, &cs->counts, (size_t)h->K * h->bins))) return rc;
    if ((rc = dev_alloc(h, &cs->weight, h->K))) return rc;
    if ((rc = dev_alloc(h, &cs->n, h->K))) return rc;
    if ((rc = dev_alloc(h, &cs->sup, (size_t)h->K * MAXB))) return rc;
    if ((rc = dev_alloc(h, &cs->lnd, (size_t)h->K * MAXB))) return rc;
    if ((rc = dev_alloc(h, &cs->dens, (size_t)h->K * h->bins))) return rc;
    if ((rc = dev_alloc(h, &cs->densR, (size_t)h->K * MAXB))) return rc;
    HIP_TRY(hipMemset(cs->densR, 0, (size_t)h->K * MAXB * 4));
    if ((rc = dev_alloc(h, &cs->mincT, (size_t)MAXB * MAXB))) return rc;
    HIP_TRY(hipMemset(cs->mincT, 0, (size_t)MAXB * MAXB * 4));
    if ((rc = dev_alloc(h, &cs->self, h->K))) return rc;
    return RP_OK;
}

__global__ void k_fill_u32(uint32_t* p, uint32_t n, uint32_t v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// content versions of the centroids (Bounds::cver).  against = the set being replaced: centroid k keeps its version iff its
// integer sums and weight are unchanged; against = none: every centroid gets a new version
__global__ __launch_bounds__(64) void k_centroid_versions(CentroidSet now, CentroidSet was, bool compare, uint32_t bins, uint32_t* cver) {
    const uint32_t k = blockIdx.x, lane = lane_id();
    bool differs = !compare;
    if (compare) {
        for (uint32_t t = lane; t < bins; t += 64) differs |= now.counts[(size_t)k * bins + t] != was.counts[(size_t)k * bins + t];
        differs |= now.weight[k] != was.weight[k];
    }
    if (__ballot(differs) && lane == 0) cver[k] += 1u;
}

int prepare_centroids(rp_kmeans* h, int set, bool replaces_other = false) {
    // an Elkan step replaces the other set: compare contents.  Any other way of installing centroids (k-means++, set_*,
    // step_naive) invalidates everything remembered, lazily at the next Elkan step (memo_reset).
    if (h->cver && replaces_other)
        hipLaunchKernelGGL(k_centroid_versions, dim3(h->K), dim3(64), 0, h->stream, h->cs[set], h->cs[set ^ 1], true, h->bins, h->cver);
    else
        h->memo_dirty = true;
    ck_begin(h, CK_SELF);
    hipLaunchKernelGGL(k_prepare_centroids, dim3(h->K), dim3(64), 0, h->stream, h->cs[set], h->K, h->M, h->kind, 0u);
    ck_end(h, CK_SELF);
    HIP_TRY(hipGetLastError());
    return RP_OK;
}

int create_common(uint32_t K, uint64_t N, uint32_t bins, const void* counts, bool counts_on_device, rp_metric_kind kind,
                  const float* tri_metric, const rp_sinkhorn_hp* hp, uint64_t seed, int device, rp_kmeans** out) {
    if (!out || !counts) return rp::fail(RP_ERR_INVALID, "rp_kmeans_create: NULL argument");
    if (K == 0 || K > MAXB || bins == 0 || bins > MAXB || N == 0)
        return rp::fail(RP_ERR_INVALID, "rp_kmeans_create: need 1 <= K <= 256, 1 <= bins <= 256, N >= 1");
    if (kind != RP_METRIC_SINKHORN && kind != RP_METRIC_VARIATION) return rp::fail(RP_ERR_INVALID, "rp_kmeans_create: unknown metric kind");
    if (kind == RP_METRIC_SINKHORN && !tri_metric && bins > 1) return rp::fail(RP_ERR_INVALID, "rp_kmeans_create: Sinkhorn needs tri_metric");
    if (rp_device_count() <= 0)
        return rp::fail(RP_ERR_NO_DEVICE, "rp_kmeans_create: no HIP device visible; the MI355X path has no CPU fallback");
    rp_kmeans* h = new rp_kmeans();
    h->device = device;
    h->K = K; h->N = N; h->bins = bins; h->kind = kind; h->seed = seed;
    if (hp) h->hp = *hp; else rp_sinkhorn_hp_default(&h->hp);
    h->stride = counts_on_device ? bins : ((bins + 15u) & ~15u);
#define KM_TRY(expr)                                                                          \
    do {                                                                                      \
        int _rc = (expr);                                                                     \
        if (_rc) { rp_kmeans_destroy(h); return _rc; }                                        \
    } while (0)
#define KM_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            int _rc = rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));    \
            rp_kmeans_destroy(h);                                                             \
            return _rc;                                                                       \
        }                                                                                     \
    } while (0)
    KM_HIP(hipSetDevice(device));
    KM_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    uint8_t* d_counts = nullptr;
    if (counts_on_device) {
        d_counts = const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(counts));
        h->owns_counts = false;
    } else {
        KM_TRY(dev_alloc(h, &d_counts, (size_t)N * h->stride));
        if (h->stride == bins) {
            KM_HIP(hipMemcpy(d_counts, counts, (size_t)N * bins, hipMemcpyHostToDevice));
        } else {  // pad rows to 16 B on the host: one large copy instead of N pitched ones
            std::vector<uint8_t> padded((size_t)N * h->stride, 0);
            const uint8_t* src = reinterpret_cast<const uint8_t*>(counts);
            for (uint64_t i = 0; i < N; ++i) memcpy(&padded[i * h->stride], src + i * bins, bins);
            KM_HIP(hipMemcpy(d_counts, padded.data(), padded.size(), hipMemcpyHostToDevice));
        }
    }
    uint32_t* d_w = nullptr;
    float* d_self = nullptr;
    KM_TRY(dev_alloc(h, &d_w, N));
    KM_TRY(dev_alloc(h, &d_self, N));
    h->P = Points{d_counts, d_w, d_self, h->stride, N};
    // metric matrices
    float *d_C = nullptr, *d_R = nullptr;
    KM_TRY(dev_alloc(h, &d_C, (size_t)bins * bins));
    KM_TRY(dev_alloc(h, &d_R, (size_t)bins * bins));
    {
        std::vector<float> C((size_t)bins * bins, 0.0f), R((size_t)bins * bins, 0.0f);
        if (kind == RP_METRIC_SINKHORN) {
            for (uint32_t x = 0; x < bins; ++x)
                for (uint32_t y = 0; y < bins; ++y) {
                    const float c = x == y ? 0.0f : tri_metric[rp_tri_index(x, y)];
                    C[(size_t)x * bins + y] = c;
                    R[(size_t)x * bins + y] = c / h->hp.temperature;  // regularization (sinkhorn.rs:129-131)
                }
        }
        KM_HIP(hipMemcpy(d_C, C.data(), C.size() * 4, hipMemcpyHostToDevice));
        KM_HIP(hipMemcpy(d_R, R.data(), R.size() * 4, hipMemcpyHostToDevice));
    }
    KM_TRY(dev_alloc(h, &h->stats, (size_t)KM_STAT_STRIPES * STAT_STRIDE));
    KM_HIP(hipMemset(h->stats, 0, (size_t)KM_STAT_STRIPES * STAT_STRIDE * 8));
    h->M = Metric{d_C, d_R, bins, h->hp.iterations, h->hp.tolerance, h->stats, KM_STAT_STRIPES, nullptr, nullptr};
    {
        KM_TRY(dev_alloc(h, &h->M.kpp_d, N));
        KM_TRY(dev_alloc(h, &h->M.kpp_j, N));
        KM_TRY(dev_alloc(h, &h->kpp_todo, N));
        KM_TRY(dev_alloc(h, &h->kpp_ntodo, 1));
    }
    KM_TRY(alloc_centroid_set(h, &h->cs[0]));
    KM_TRY(alloc_centroid_set(h, &h->cs[1]));
    KM_TRY(dev_alloc(h, &h->B.j, N));
    KM_TRY(dev_alloc(h, &h->B.u, N));
    KM_TRY(dev_alloc(h, &h->B.stale, N));
    KM_TRY(dev_alloc(h, &h->B.lower, (size_t)N * K));
    KM_TRY(dev_alloc(h, &h->prior, N));
    KM_TRY(dev_alloc(h, &h->tmp_j, N));
    KM_TRY(dev_alloc(h, &h->pairw, (size_t)K * K));
    h->memo_on = kind == RP_METRIC_SINKHORN;
    if (h->memo_on) {
        KM_TRY(dev_alloc(h, &h->cver, (size_t)K));
        KM_TRY(dev_alloc(h, &h->pver, (size_t)K * K * 2));
        KM_TRY(dev_alloc(h, &h->B.memo_d, N));
        KM_TRY(dev_alloc(h, &h->B.memo_ver, N));
        KM_TRY(dev_alloc(h, &h->B.memo_j, N));
        h->B.cver = h->cver;
    }
    KM_TRY(dev_alloc(h, &h->mid, K));
    KM_TRY(dev_alloc(h, &h->drift, K));
    KM_TRY(dev_alloc(h, &h->pot, N));
    KM_TRY(dev_alloc(h, &h->pdist, N));
    KM_TRY(dev_alloc(h, &h->bsum, (N + KPP_BLOCK - 1) / KPP_BLOCK));
    KM_TRY(dev_alloc(h, &h->scal, 2));
    KM_TRY(dev_alloc(h, &h->sizes, K));
    KM_TRY(dev_alloc(h, &h->hist_stage, MAXB));
    // point masses and memoised self costs OT(p,p) (sinkhorn.rs:175-191)
    hipLaunchKernelGGL(k_point_weights, dim3((unsigned)N), dim3(64), 0, h->stream, d_counts, h->stride, bins, N, d_w);
    KM_HIP(hipGetLastError());
    if (kind == RP_METRIC_SINKHORN) {
        hipLaunchKernelGGL(k_point_self, dim3((unsigned)N), dim3(64), 0, h->stream, h->P, h->M, d_self);
        KM_HIP(hipGetLastError());
    } else {
        KM_HIP(hipMemsetAsync(d_self, 0, N * 4, h->stream));
    }
    KM_HIP(hipStreamSynchronize(h->stream));
    std::vector<uint8_t> ns;
    uint8_t* d_ns = nullptr;
    if (kind == RP_METRIC_SINKHORN) {
        KM_TRY(dev_alloc(h, &d_ns, N));
        hipLaunchKernelGGL(k_point_support, dim3((unsigned)N), dim3(64), 0, h->stream, h->P, bins, d_ns);
        KM_HIP(hipGetLastError());
        ns.resize(N);
        KM_HIP(hipMemcpyAsync(ns.data(), d_ns, N, hipMemcpyDeviceToHost, h->stream));  // same (non-blocking) stream as the kernel
        KM_HIP(hipStreamSynchronize(h->stream));
    }
    if (kind == RP_METRIC_SINKHORN && !getenv("RP_LLOYD_NO_KPP_BOUND")) {
        float cmax = 0.0f;
        for (size_t t = 0; t < (size_t)bins * (bins - 1) / 2; ++t) cmax = std::max(cmax, tri_metric[t]);
        if (cmax / h->hp.temperature <= 64.0f) {  // no softmin term near the MIN_POSITIVE clamp: column sums are nu to ~1e-5
            h->d_nsup = d_ns;
            KM_TRY(dev_alloc(h, &h->minc, MAXB));
            for (int c = 0; c < 3; ++c) KM_TRY(dev_alloc(h, &h->kpp.list[c], (size_t)N + 4));
            KM_TRY(dev_alloc(h, &h->kpp.count, 4));
            for (uint64_t i = 0; i < N; ++i) h->kpp_cap[ns[i] <= QUAD_ROWS ? 0 : (ns[i] <= PAIR_ROWS ? 1 : 2)] += 1;
            h->kpp_lb = true;
        }
    }
    if (kind == RP_METRIC_SINKHORN && !getenv("RP_LLOYD_NO_MFMA_BOUND")) {
        // the MFMA bound (sinkhorn_bound.hpp): K = exp(-C/T) padded to 256 x 256, point lists by ceil(support / 16)
        std::vector<float> Km((size_t)MAXB * MAXB, 1.0f);
        for (uint32_t x = 0; x < bins; ++x)
            for (uint32_t y = 0; y < bins; ++y) {
                const float c = x == y ? 0.0f : tri_metric[rp_tri_index(x, y)];
                Km[(size_t)x * MAXB + y] = (float)std::exp(-(double)c / (double)h->hp.temperature);
            }
        float* d_K = nullptr;
        KM_TRY(dev_alloc(h, &d_K, Km.size()));
        KM_HIP(hipMemcpy(d_K, Km.data(), Km.size() * 4, hipMemcpyHostToDevice));
        auto envf = [](const char* name, float dflt) {
            const char* v = getenv(name);
            return v ? (float)atof(v) : dflt;
        };
        h->sb.Kmat = d_K;
        h->sb.neg_t_ln2 = -h->hp.temperature * 0.6931472f;
        h->sb.tol = h->hp.tolerance;
        h->sb.iters = h->hp.iterations;
        // the margins can be WIDENED through the environment (more survivors, same results); values on the unsafe side of the
        // validated defaults (profiles/r03_mfma_audit.json) are clamped back to them
        h->sb.kappa = std::max(envf("RP_SB_KAPPA", 2.0f), 2.0f);
        h->sb.rho = std::max(envf("RP_SB_RHO", 1.25f), 1.25f);
        h->sb.dc_abs = std::max(envf("RP_SB_DC_ABS", 4e-6f), 4e-6f);
        h->sb.dc_rel = std::max(envf("RP_SB_DC_REL", 4e-5f), 4e-5f);
        h->sb.flat = std::min(envf("RP_SB_FLAT", 4.0f), 4.0f);
        {
            float cmax = 0.0f;
            for (size_t t = 0; t < (size_t)bins * (bins - 1) / 2; ++t) cmax = std::max(cmax, tri_metric[t]);
            h->sb.use_lb0 = (cmax / h->hp.temperature <= 64.0f && !getenv("RP_SB_NO_LB0")) ? 1 : 0;
        }
        std::vector<uint32_t> cls[4], big;
        for (uint64_t i = 0; i < N; ++i) {
            // k_point_support saturates at 255 bins; 0 (an empty histogram) is left to the exact kernel as well
            const uint32_t t = ns[i] == 0 || ns[i] > SB_MAXROWS ? 4u : (uint32_t)(ns[i] - 1) / 16u;
            (t < 4 ? cls[t] : big).push_back((uint32_t)i);
        }
        for (int t = 0; t < 4; ++t) {
            h->sb_count[t] = (uint32_t)cls[t].size();
            KM_TRY(dev_alloc(h, &h->sb_list[t], cls[t].size()));
            if (!cls[t].empty()) KM_HIP(hipMemcpy(h->sb_list[t], cls[t].data(), cls[t].size() * 4, hipMemcpyHostToDevice));
        }
        h->sb_nbig = (uint32_t)big.size();
        KM_TRY(dev_alloc(h, &h->sb_big, big.size()));
        if (!big.empty()) KM_HIP(hipMemcpy(h->sb_big, big.data(), big.size() * 4, hipMemcpyHostToDevice));
        KM_TRY(dev_alloc(h, &h->sb_cursor, 4));
        KM_TRY(dev_alloc(h, &h->sb_mask, (size_t)N * 4));
        KM_TRY(dev_alloc(h, &h->sb_stats, (size_t)KM_STAT_STRIPES * STAT_STRIDE));
        KM_HIP(hipMemset(h->sb_stats, 0, (size_t)KM_STAT_STRIPES * STAT_STRIDE * 8));
        KM_TRY(dev_alloc(h, &h->sb_bad, 2));
        KM_HIP(hipMemset(h->sb_bad, 0, 16));
        KM_TRY(dev_alloc(h, &h->sb_d, N));
        KM_TRY(dev_alloc(h, &h->sb_ub0, N));
        KM_TRY(dev_alloc(h, &h->sb_hint_j, N));
        KM_TRY(dev_alloc(h, &h->sb_hint_mask, (size_t)N * 4));
        h->sb_audit = getenv("RP_LLOYD_AUDIT") != nullptr;
        {
            // production self-check: every SB_SAMPLE_STRIDE-th point goes through the unpruned search after every pruned pass;
            // a disagreement fails the next call that hands results out (never a silently different bucket)
            std::vector<uint32_t> smp;
            for (uint64_t i = 0; i < N; i += SB_SAMPLE_STRIDE) smp.push_back((uint32_t)i);
            h->sb_nsample = (uint32_t)smp.size();
            KM_TRY(dev_alloc(h, &h->sb_sample, smp.size()));
            if (!smp.empty()) KM_HIP(hipMemcpy(h->sb_sample, smp.data(), smp.size() * 4, hipMemcpyHostToDevice));
            KM_TRY(dev_alloc(h, &h->audit_j, N));
            KM_TRY(dev_alloc(h, &h->audit_d, N));
        }
        h->sb_on = true;
    }
    // RP_LLOYD_GROUPING (tests): "none" = one point per wavefront everywhere, "pairs" = no groups of four, "norefresh" = no regrouped
    // refresh pass; every grouping performs the same float operations per solve (tests/test_gpu_lloyd.py)
    const std::string grouping = getenv("RP_LLOYD_GROUPING") ? getenv("RP_LLOYD_GROUPING") : "";
    if (kind == RP_METRIC_SINKHORN && grouping != "none") {
        // grouping lists: <= QUAD_ROWS bins -> four per wavefront, <= PAIR_ROWS -> two, the others one
        std::vector<uint32_t> tiny, small, rest;
        const bool no_quads = grouping == "pairs";
        if (grouping != "norefresh") {
            h->refresh.nsup = d_ns;
            KM_TRY(dev_alloc(h, &h->refresh.count, (size_t)K));
            KM_TRY(dev_alloc(h, &h->refresh.offset, (size_t)K + 1));
            KM_TRY(dev_alloc(h, &h->refresh.list, (size_t)N + 2 * (size_t)K));
        }
        for (uint64_t i = 0; i < N; ++i) (ns[i] <= QUAD_ROWS && !no_quads ? tiny : (ns[i] <= PAIR_ROWS ? small : rest)).push_back((uint32_t)i);
        while (tiny.size() & 3u) {
            small.push_back(tiny.back());
            tiny.pop_back();
        }
        if (small.size() & 1u) {
            rest.push_back(small.back());
            small.pop_back();
        }
        h->n_quads = tiny.size() / 4;
        h->n_pairs = small.size() / 2;
        h->n_singles = rest.size();
        KM_TRY(dev_alloc(h, &h->quads, tiny.size()));
        KM_TRY(dev_alloc(h, &h->pairs, small.size()));
        KM_TRY(dev_alloc(h, &h->singles, rest.size()));
        if (!tiny.empty()) KM_HIP(hipMemcpy(h->quads, tiny.data(), tiny.size() * 4, hipMemcpyHostToDevice));
        if (!small.empty()) KM_HIP(hipMemcpy(h->pairs, small.data(), small.size() * 4, hipMemcpyHostToDevice));
        if (!rest.empty()) KM_HIP(hipMemcpy(h->singles, rest.data(), rest.size() * 4, hipMemcpyHostToDevice));
    }
#undef KM_TRY
#undef KM_HIP
    // hipMemset on device memory returns before it has run, and a non-blocking stream does not wait for the null stream: the
    // first launch on this handle's stream could otherwise overtake the initialisation above and be overwritten by it
    (void)hipDeviceSynchronize();
    *out = h;
    return RP_OK;
}

size_t partial_sizes_offset(const rp_kmeans* h) { return ((((size_t)h->K * h->bins + h->K) * 4) + 7) & ~(size_t)7; }

int need_centroids(const rp_kmeans* h, const char* who) {
    if (!h->centroids_ready) return rp::fail(RP_ERR_INVALID, "%s: centroids not initialised (init_centroids / set_centroids)", who);
    return RP_OK;
}
int need_bounds(const rp_kmeans* h, const char* who) {
    if (!h->bounds_ready) return rp::fail(RP_ERR_INVALID, "%s: bounds not initialised (init_bounds)", who);
    return RP_OK;
}

// every (point, centroid) distance through the bit-faithful kernels: the reference's loop as it stands
// the production self-check of the pruned passes (launch_neighbor, the init_bounds shortcut): a sampled point whose pruned result
// differs from the unpruned search is a library bug that would silently move a bucket — the call fails instead
int prune_check(rp_kmeans* h, const char* who) {
    if (!h->sb_on || !h->sb_check_pending) return RP_OK;
    unsigned long long bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, h->sb_bad + 1, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->sb_check_pending = false;
    if (bad)
        return rp::fail(RP_ERR_INTERNAL, "%s: the pruned nearest-centroid search disagrees with the unpruned one on %llu of the sampled points "
                                         "(set RP_LLOYD_NO_MFMA_BOUND=1 and report)", who, bad);
    return RP_OK;
}
int launch_neighbor_full(rp_kmeans* h, uint8_t* out_j, float* out_d, Bounds init) {
    ck_begin(h, CK_NEIGHBOR);
    if (h->kind == RP_METRIC_VARIATION && h->bins == 101)  // turn layer: register-resident centroid CDFs
        hipLaunchKernelGGL(k_neighbor_var<101>, dim3((unsigned)((h->N + VB - 1) / VB)), dim3(256), 0, h->stream, h->P,
                           h->cs[h->cur], h->K, h->M, out_j, out_d, init);
    else if (h->kind == RP_METRIC_SINKHORN && (h->n_pairs || h->n_quads)) {
        if (h->n_quads)
            hipLaunchKernelGGL(k_neighborG<4>, dim3((unsigned)h->n_quads), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M,
                               h->quads, out_j, out_d, init);
        if (h->n_pairs)
            hipLaunchKernelGGL(k_neighborG<2>, dim3((unsigned)h->n_pairs), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M,
                               h->pairs, out_j, out_d, init);
        if (h->n_singles)
            hipLaunchKernelGGL(k_neighbor, dim3((unsigned)h->n_singles), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M,
                               h->kind, out_j, out_d, init, h->singles);
    } else
        hipLaunchKernelGGL(k_neighbor, dim3((unsigned)h->N), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M, h->kind, out_j,
                           out_d, init, (const uint32_t*)nullptr);
    ck_end(h, CK_NEIGHBOR);
    HIP_TRY(hipGetLastError());
    return RP_OK;
}

// the MFMA bound over every point (one launch per support class), leaving the survivor masks in h->sb_mask
int launch_bound(rp_kmeans* h, float* dbg_lo, float* dbg_hi, const float* ub0 = nullptr) {
    HIP_TRY(hipMemsetAsync(h->sb_cursor, 0, 16, h->stream));
    const CentroidSet& cs = h->cs[h->cur];
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
    ck_begin(h, CK_BOUND);
    for (int t = 3; t >= 0; --t) {  // the longest-running class first
        const uint32_t n = h->sb_count[t];
        if (!n) continue;
        const dim3 grid(std::min<uint32_t>(n, (uint32_t)cus)), block(SB_THREADS);
#define SB_LAUNCH(NT)                                                                                                      \
    hipLaunchKernelGGL(k_sinkhorn_bound<NT>, grid, block, 0, h->stream, h->P, cs, h->K, h->bins, h->sb, h->sb_list[t], n, \
                       h->sb_cursor + t, h->sb_mask, dbg_lo, dbg_hi, h->sb_stats, ub0)
        if (t == 0) SB_LAUNCH(1);
        else if (t == 1) SB_LAUNCH(2);
        else if (t == 2) SB_LAUNCH(3);
        else SB_LAUNCH(4);
#undef SB_LAUNCH
    }
    if (h->sb_nbig) hipLaunchKernelGGL(k_mask_all, dim3((h->sb_nbig + 255) / 256), dim3(256), 0, h->stream, h->sb_big, h->sb_nbig, h->sb_mask);
    ck_end(h, CK_BOUND);
    HIP_TRY(hipGetLastError());
    return RP_OK;
}

// Elkan::neighbor for every point (init_bounds / Layer::lookup / step_naive).  Sinkhorn layers: the MFMA bound discards
// the centroids that cannot be the argmin, the bit-faithful kernel runs on the survivors (same bits).
enum { NB_INIT_BOUNDS, NB_LOOKUP, NB_NAIVE };
int launch_neighbor(rp_kmeans* h, uint8_t* out_j, float* out_d, Bounds init, int pass = NB_NAIVE) {
    if (!h->sb_on) return launch_neighbor_full(h, out_j, out_d, init);
    const unsigned nblk = (unsigned)((h->N + 255) / 256);
    const float* ub0 = nullptr;
    const uint8_t* hint_j = nullptr;
    if (h->sb.use_lb0) {
        if (pass == NB_INIT_BOUNDS && h->pot_is_min_d2) {
            // right after k-means++: potentials = min_k d(c_k, x)^2 for exactly these centroids
            hipLaunchKernelGGL(k_ub_from_pot, dim3(nblk), dim3(256), 0, h->stream, h->pot, h->N, h->sb_ub0);
            ub0 = h->sb_ub0;
        } else if (pass == NB_LOOKUP && h->bounds_ready) {
            // after the Elkan iterations: the exact distance to the point's assigned centroid (one solve per point, reused below)
            HIP_TRY(hipMemcpyAsync(h->sb_hint_j, h->B.j, h->N, hipMemcpyDeviceToDevice, h->stream));
            hipLaunchKernelGGL(k_hint_masks, dim3(nblk), dim3(256), 0, h->stream, h->sb_hint_j, h->N, h->K, h->sb_hint_mask);
            ck_begin(h, CK_NEIGHBOR);
            Bounds none{};
            hipLaunchKernelGGL(k_neighbor_masked, dim3((unsigned)h->N), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M,
                               h->sb_hint_mask, (uint8_t*)nullptr, h->sb_ub0, none, (const uint8_t*)nullptr, (const float*)nullptr);
            ck_end(h, CK_NEIGHBOR);
            ub0 = h->sb_ub0;
            hint_j = h->sb_hint_j;
        }
    }
    int rc = launch_bound(h, nullptr, nullptr, ub0);
    if (rc) return rc;
    float* dd = out_d ? out_d : h->sb_d;
    ck_begin(h, CK_NEIGHBOR);
    hipLaunchKernelGGL(k_neighbor_masked, dim3((unsigned)h->N), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M, h->sb_mask,
                       out_j, dd, init, hint_j, hint_j ? ub0 : (const float*)nullptr);
    ck_end(h, CK_NEIGHBOR);
    HIP_TRY(hipGetLastError());
    if (!h->sb_audit && out_j && h->sb_nsample) {  // the sampled points once more, unpruned (k_neighbor takes a point list)
        Bounds none{};
        ck_begin(h, CK_NEIGHBOR);
        hipLaunchKernelGGL(k_neighbor, dim3(h->sb_nsample), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M, h->kind, h->audit_j,
                           h->audit_d, none, (const uint32_t*)h->sb_sample);
        ck_end(h, CK_NEIGHBOR);
        hipLaunchKernelGGL(k_audit_compare_list, dim3((h->sb_nsample + 255) / 256), dim3(256), 0, h->stream, out_j, dd, h->audit_j,
                           h->audit_d, h->sb_sample, h->sb_nsample, h->sb_bad + 1);
        HIP_TRY(hipGetLastError());
        h->sb_sampled += h->sb_nsample;
        h->sb_check_pending = true;
    }
    if (h->sb_audit && out_j) {  // RP_LLOYD_AUDIT: the unpruned pass next to it; disagreements are counted, never corrected
        Bounds none{};
        if ((rc = launch_neighbor_full(h, h->audit_j, h->audit_d, none))) return rc;
        hipLaunchKernelGGL(k_audit_compare, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, out_j, dd, h->audit_j,
                           h->audit_d, h->N, h->sb_bad);
        HIP_TRY(hipGetLastError());
        h->sb_audited += h->N;
    }
    return RP_OK;
}

int launch_recompute(rp_kmeans* h, const uint8_t* assign, int set) {
    ck_begin(h, CK_RECOMPUTE);
    hipLaunchKernelGGL(k_recompute, dim3(h->K), dim3(256), 0, h->stream, h->P, assign, h->bins, h->cs[set].counts,
                       h->cs[set].weight, h->sizes);
    ck_end(h, CK_RECOMPUTE);
    HIP_TRY(hipGetLastError());
    return RP_OK;
}

// the part of step_elkan before the centroid exchange: pairwise, midpoints, bound refresh, partial sums
int step_front(rp_kmeans* h) {
    const int cur = h->cur;
    if (h->memo_on && h->memo_dirty) {  // centroids installed outside an Elkan step: nothing remembered is valid
        h->memo_epoch += 1;
        hipLaunchKernelGGL(k_fill_u32, dim3((h->K + 255) / 256), dim3(256), 0, h->stream, h->cver, h->K, h->memo_epoch << 16);
        HIP_TRY(hipMemsetAsync(h->pver, 0, (size_t)h->K * h->K * 8, h->stream));
        HIP_TRY(hipMemsetAsync(h->B.memo_ver, 0, (size_t)h->N * 4, h->stream));
        h->memo_dirty = false;
    }
    ck_begin(h, CK_PAIRWISE);
    if (h->kind == RP_METRIC_VARIATION)
        hipLaunchKernelGGL(k_pairwise_var, dim3((h->K * h->K + 255) / 256), dim3(256), 0, h->stream, h->cs[cur], h->K, h->M, h->pairw);
    else
        hipLaunchKernelGGL(k_pairwise, dim3(h->K * h->K), dim3(64), 0, h->stream, h->cs[cur], h->K, h->M, h->kind, h->pairw, h->cver, h->pver);
    hipLaunchKernelGGL(k_midpoints, dim3((h->K + 63) / 64), dim3(64), 0, h->stream, h->pairw, h->K, h->mid);
    ck_end(h, CK_PAIRWISE);
    ck_begin(h, CK_STEP);
    if (h->kind == RP_METRIC_VARIATION && h->bins == 101)
        hipLaunchKernelGGL(k_elkan_step_var<101>, dim3((unsigned)((h->N + VB - 1) / VB)), dim3(256), 0, h->stream, h->P, h->cs[cur],
                           h->K, h->M, h->B, h->pairw, h->mid);
    else {
        if (h->kind == RP_METRIC_SINKHORN && h->refresh.nsup) {
            const size_t entries = (size_t)h->N + 2 * (size_t)h->K;
            HIP_TRY(hipMemsetAsync(h->refresh.count, 0, (size_t)h->K * 4, h->stream));
            HIP_TRY(hipMemsetAsync(h->refresh.list, 0xff, entries * 4, h->stream));
            if (h->B.memo_ver) hipLaunchKernelGGL(k_refresh_memo, dim3(1024), dim3(256), 0, h->stream, h->B, h->mid, h->N, h->K);
            hipLaunchKernelGGL(k_refresh_count, dim3(1024), dim3(256), 0, h->stream, h->B, h->refresh, h->mid, h->N, h->K);
            hipLaunchKernelGGL(k_refresh_offsets, dim3(1), dim3(1), 0, h->stream, h->refresh, h->K);
            hipLaunchKernelGGL(k_refresh_fill, dim3(1024), dim3(256), 0, h->stream, h->B, h->refresh, h->mid, h->N);
            hipLaunchKernelGGL(k_refresh_pairs, dim3((unsigned)(entries / 2 + 1)), dim3(64), 0, h->stream, h->P, h->cs[cur], h->K, h->M,
                               h->B, h->refresh);
        }
        hipLaunchKernelGGL(k_elkan_step, dim3((unsigned)h->N), dim3(64), 0, h->stream, h->P, h->cs[cur], h->K, h->M, h->kind, h->B,
                           h->pairw, h->mid);
    }
    ck_end(h, CK_STEP);
    HIP_TRY(hipGetLastError());
    return launch_recompute(h, h->B.j, cur ^ 1);
}

// after the (optional) all-reduce of the integer sums in cs[cur^1]: drift, bounds update, install, tally
int step_back(rp_kmeans* h, float* drift, uint64_t* sizes, double* reassigned) {
    const int cur = h->cur, nxt = cur ^ 1;
    int rc = prepare_centroids(h, nxt, true);
    if (rc) return rc;
    ck_begin(h, CK_DRIFT);
    hipLaunchKernelGGL(k_drift, dim3(h->K), dim3(64), 0, h->stream, h->cs[nxt], h->cs[cur], h->K, h->M, h->kind, h->drift);
    ck_end(h, CK_DRIFT);
    ck_begin(h, CK_BOUNDS);
    hipLaunchKernelGGL(k_bounds_update, dim3(2048), dim3(256), 0, h->stream, h->B, h->N, h->K, h->drift);
    ck_end(h, CK_BOUNDS);
    HIP_TRY(hipMemsetAsync(h->scal + 1, 0, 8, h->stream));
    hipLaunchKernelGGL(k_tally, dim3(1024), dim3(256), 0, h->stream, h->B.j, h->prior, h->N, h->scal + 1);
    HIP_TRY(hipGetLastError());
    h->cur = nxt;  // Kmeans::next installs the new centroids (kmeans.rs:88)
    h->pot_is_min_d2 = false;
    std::vector<unsigned long long> sz(h->K);
    unsigned long long moved = 0;
    if (drift) HIP_TRY(hipMemcpyAsync(drift, h->drift, h->K * 4, hipMemcpyDeviceToHost, h->stream));
    if (sizes) HIP_TRY(hipMemcpyAsync(sz.data(), h->sizes, h->K * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(&moved, h->scal + 1, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (sizes) for (uint32_t k = 0; k < h->K; ++k) sizes[k] = sz[k];
    if (reassigned) *reassigned = (double)moved / (double)h->N;
    ck_drain(h);
    return RP_OK;
}

}  // namespace

extern "C" {

int rp_kmeans_create(uint32_t K, uint64_t N, uint32_t bins, const uint8_t* counts, rp_metric_kind kind, const float* tri_metric,
                     const rp_sinkhorn_hp* hp, uint64_t seed, int device, rp_kmeans** out) {
    return create_common(K, N, bins, counts, false, kind, tri_metric, hp, seed, device, out);
}
int rp_kmeans_create_device(uint32_t K, uint64_t N, uint32_t bins, const void* counts_dev, rp_metric_kind kind,
                            const float* tri_metric, const rp_sinkhorn_hp* hp, uint64_t seed, int device, rp_kmeans** out) {
    return create_common(K, N, bins, counts_dev, true, kind, tri_metric, hp, seed, device, out);
}

int rp_kmeans_destroy(rp_kmeans* h) {
    if (!h) return RP_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    ck_drain(h);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return RP_OK;
}

int rp_kmeans_set_centroids(rp_kmeans* h, const uint64_t* point_index) {
    if (!h || !point_index) return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_centroids: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    for (uint32_t k = 0; k < h->K; ++k) {
        if (point_index[k] >= h->N) return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_centroids: point index out of range");
        hipLaunchKernelGGL(k_centroid_from_point, dim3(1), dim3(256), 0, h->stream, h->cs[h->cur], k, h->P, point_index[k], h->bins);
    }
    HIP_TRY(hipGetLastError());
    int rc = prepare_centroids(h, h->cur);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->centroids_ready = true;
    h->bounds_ready = false;
    h->pot_is_min_d2 = false;
    return RP_OK;
}

int rp_kmeans_set_rng(rp_kmeans* h, rp_rng_kind kind, int street) {
    if (!h || (kind != RP_RNG_COUNTER && kind != RP_RNG_REFERENCE) || street < 0 || street > 3)
        return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_rng: bad argument");
    h->rng = kind;
    h->street = street;
    return RP_OK;
}

int rp_kmeans_kpp_begin(rp_kmeans* h) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_kpp_begin: NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    CentroidSet& cs = h->cs[h->cur];
    // all centroids start empty so every derived table is well defined for the not-yet-chosen ones
    HIP_TRY(hipMemsetAsync(cs.counts, 0, (size_t)h->K * h->bins * 4, h->stream));
    HIP_TRY(hipMemsetAsync(cs.weight, 0, h->K * 4, h->stream));
    hipLaunchKernelGGL(k_prepare_centroids, dim3(h->K), dim3(64), 0, h->stream, cs, h->K, h->M, h->kind, 0u);
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, h->stream, h->pot, h->N, 1.0f);  // potentials = 1 (layer.rs:161)
    if (h->M.kpp_d) hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, h->stream, h->M.kpp_d, h->N, rp_u2f(0x7f800000u));
    HIP_TRY(hipGetLastError());
    h->centroids_ready = false;
    h->bounds_ready = false;
    h->pot_is_min_d2 = false;
    return RP_OK;
}

int rp_kmeans_kpp_total(rp_kmeans* h, uint64_t* total) {
    if (!h || !total) return rp::fail(RP_ERR_INVALID, "rp_kmeans_kpp_total: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    const uint32_t nblocks = (uint32_t)((h->N + KPP_BLOCK - 1) / KPP_BLOCK);
    ck_begin(h, CK_KPP);
    hipLaunchKernelGGL(k_kpp_blocksum, dim3(nblocks), dim3(256), 0, h->stream, h->pot, h->N, h->bsum);
    hipLaunchKernelGGL(k_kpp_total, dim3(1), dim3(1024), 0, h->stream, h->bsum, nblocks, h->scal);
    ck_end(h, CK_KPP);
    HIP_TRY(hipGetLastError());
    unsigned long long t = 0;
    HIP_TRY(hipMemcpyAsync(&t, h->scal, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *total = t;
    return RP_OK;
}

// requires a preceding rp_kmeans_kpp_total (block sums) with unchanged potentials, and r < that total
int rp_kmeans_kpp_pick(rp_kmeans* h, uint64_t r, uint64_t* index) {
    if (!h || !index) return rp::fail(RP_ERR_INVALID, "rp_kmeans_kpp_pick: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    const uint32_t nblocks = (uint32_t)((h->N + KPP_BLOCK - 1) / KPP_BLOCK);
    hipLaunchKernelGGL(k_kpp_pick, dim3(1), dim3(1024), 0, h->stream, h->pot, h->M.kpp_d, h->N, h->bsum, nblocks, (unsigned long long)r, h->scal);
    HIP_TRY(hipGetLastError());
    unsigned long long p = 0;
    HIP_TRY(hipMemcpyAsync(&p, h->scal, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *index = p;
    return RP_OK;
}

int rp_kmeans_kpp_update(rp_kmeans* h, uint32_t k) {
    if (!h || k >= h->K) return rp::fail(RP_ERR_INVALID, "rp_kmeans_kpp_update: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    ck_begin(h, CK_KPP);
    if (h->kind == RP_METRIC_VARIATION)
        hipLaunchKernelGGL(k_kpp_update_var, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, h->P, h->cs[h->cur], k, h->K,
                           h->M, h->pot);
    else if (h->kpp_lb) {
        // the column-marginal bound first: only points whose potential can still drop are solved, regrouped by support class
        const bool grouped = h->n_pairs || h->n_quads;
        const uint32_t qrows = grouped && h->n_quads ? QUAD_ROWS : 0u, prows = grouped ? PAIR_ROWS : 0u;
        HIP_TRY(hipMemsetAsync(h->kpp.count, 0, 16, h->stream));
        hipLaunchKernelGGL(k_minc, dim3(1), dim3(256), 0, h->stream, h->cs[h->cur], k, h->M, h->minc);
        hipLaunchKernelGGL(k_kpp_filter, dim3((unsigned)((h->N + 63) / 64)), dim3(1024), 0, h->stream, h->P, h->cs[h->cur], k, h->M,
                           h->minc, h->pot, h->d_nsup, h->kpp, qrows, prows);
        // grids sized for the worst case (every eligible point active); surplus wavefronts leave at once
        const uint64_t cap4 = qrows ? h->kpp_cap[0] : 0, cap2 = prows ? h->kpp_cap[1] + (qrows ? 0 : h->kpp_cap[0]) : 0,
                       cap1 = h->N - cap4 - cap2;
        if (cap4)
            hipLaunchKernelGGL(k_kpp_updateG<4>, dim3((unsigned)((cap4 + 3) / 4)), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->M,
                               h->kpp.list[0], h->kpp.count + 0, h->pot);
        if (cap2)
            hipLaunchKernelGGL(k_kpp_updateG<2>, dim3((unsigned)((cap2 + 1) / 2)), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->M,
                               h->kpp.list[1], h->kpp.count + 1, h->pot);
        if (cap1)
            hipLaunchKernelGGL(k_kpp_update, dim3((unsigned)cap1), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->K, h->M, h->kind,
                               h->pot, h->kpp.list[2], h->kpp.count + 2);
    } else {
        if (h->n_pairs || h->n_quads) {
            if (h->n_quads)
                hipLaunchKernelGGL(k_kpp_updateG<4>, dim3((unsigned)h->n_quads), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->M,
                                   h->quads, (const unsigned int*)nullptr, h->pot);
            if (h->n_pairs)
                hipLaunchKernelGGL(k_kpp_updateG<2>, dim3((unsigned)h->n_pairs), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->M,
                                   h->pairs, (const unsigned int*)nullptr, h->pot);
            if (h->n_singles)
                hipLaunchKernelGGL(k_kpp_update, dim3((unsigned)h->n_singles), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->K,
                                   h->M, h->kind, h->pot, h->singles, (const unsigned int*)nullptr);
        } else {
            hipLaunchKernelGGL(k_kpp_update, dim3((unsigned)h->N), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->K, h->M, h->kind,
                               h->pot, (const uint32_t*)nullptr, (const unsigned int*)nullptr);
        }
    }
    ck_end(h, CK_KPP);
    HIP_TRY(hipGetLastError());
    if (k + 1 == h->K) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->centroids_ready = true;
        h->pot_is_min_d2 = true;  // potentials = min_k d(c_k, x)^2 over the K centroids now installed
        ck_drain(h);
    }
    return RP_OK;
}

int rp_kmeans_get_point(rp_kmeans* h, uint64_t index, uint32_t* counts) {
    if (!h || !counts || index >= h->N) return rp::fail(RP_ERR_INVALID, "rp_kmeans_get_point: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    std::vector<uint8_t> row(h->bins);
    HIP_TRY(hipMemcpyAsync(row.data(), h->P.counts + index * h->P.stride, h->bins, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (uint32_t b = 0; b < h->bins; ++b) counts[b] = row[b];
    return RP_OK;
}

int rp_kmeans_set_centroid(rp_kmeans* h, uint32_t k, const uint32_t* counts) {
    if (!h || !counts || k >= h->K) return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_centroid: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->hist_stage, counts, (size_t)h->bins * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_centroid_from_hist, dim3(1), dim3(256), 0, h->stream, h->cs[h->cur], k, h->hist_stage, h->bins);
    hipLaunchKernelGGL(k_prepare_centroids, dim3(1), dim3(64), 0, h->stream, h->cs[h->cur], h->K, h->M, h->kind, k);
    h->memo_dirty = true;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));  // `counts` may be a temporary on the caller's side
    h->bounds_ready = false;
    h->pot_is_min_d2 = false;
    return RP_OK;
}

// Layer::init_centroids (layer.rs:140-181) composed from the primitives above
int rp_kmeans_init_centroids(rp_kmeans* h, uint64_t* chosen) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_init_centroids: NULL handle");
    int rc = rp_kmeans_kpp_begin(h);
    if (rc) return rc;
    std::vector<uint32_t> hist(h->bins);
    rp_smallrng rng;
    if (h->rng == RP_RNG_REFERENCE) {  // DefaultHasher::default(); self.street().hash(hasher); SmallRng::seed_from_u64(hasher.finish())
        rp_sip sh;
        rp_defaulthasher_new(&sh);
        rp_defaulthasher_write_u64(&sh, (uint64_t)(int64_t)h->street);  // a fieldless enum hashes its discriminant as isize
        rp_smallrng_seed(&rng, rp_defaulthasher_finish(&sh));
        if (!h->kpp_cum && (rc = dev_alloc(h, &h->kpp_cum, h->N))) return rc;
    }
    for (uint32_t k = 0; k < h->K; ++k) {
        uint64_t total = 0, pick = 0;
        if (h->rng == RP_RNG_REFERENCE) {
            const float v01 = rp_u2f((rp_smallrng_next_u32(&rng) >> 9) | 0x3f800000u) - 1.0f;  // UniformFloat<f32>: [1, 2) - 1
            ck_begin(h, CK_KPP);
            hipLaunchKernelGGL(k_kpp_ref_pick, dim3(1), dim3(64), 0, h->stream, h->pot, h->M.kpp_d, h->N, h->kpp_cum, v01, h->scal);
            ck_end(h, CK_KPP);
            HIP_TRY(hipGetLastError());
            unsigned long long pk = 0;
            HIP_TRY(hipMemcpyAsync(&pk, h->scal, 8, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            if (pk >= h->N) return rp::fail(RP_ERR_INVALID, "rp_kmeans_init_centroids: every potential is zero after %u picks (fewer distinct points than K; the reference panics here)", k);
            pick = pk;
        } else {
        if ((rc = rp_kmeans_kpp_total(h, &total))) return rc;
        const uint64_t hsh = rp_stream(h->seed, k);
        if (total == 0) {
            pick = rp_mulhi64(hsh, h->N);
        } else if ((rc = rp_kmeans_kpp_pick(h, rp_mulhi64(hsh, total), &pick))) {
            return rc;
        }
        }
        if (chosen) chosen[k] = pick;
        hipLaunchKernelGGL(k_centroid_from_point, dim3(1), dim3(256), 0, h->stream, h->cs[h->cur], k, h->P, pick, h->bins);
        hipLaunchKernelGGL(k_prepare_centroids, dim3(1), dim3(64), 0, h->stream, h->cs[h->cur], h->K, h->M, h->kind, k);
        h->memo_dirty = true;
        HIP_TRY(hipGetLastError());
        if ((rc = rp_kmeans_kpp_update(h, k))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->centroids_ready = true;
    h->bounds_ready = false;
    ck_drain(h);
    return RP_OK;
}

int rp_kmeans_init_bounds(rp_kmeans* h) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_init_bounds: NULL handle");
    int rc = need_centroids(h, "rp_kmeans_init_bounds");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if (h->M.kpp_d && h->pot_is_min_d2) {
        // right after k-means++ on these very centroids: the nearest centroid of almost every point is already known
        // (k_init_from_kpp); the exact search runs on the rest
        unsigned int n_todo = 0;
        HIP_TRY(hipMemsetAsync(h->kpp_ntodo, 0, 4, h->stream));
        ck_begin(h, CK_NEIGHBOR);
        hipLaunchKernelGGL(k_zero_f32, dim3(2048), dim3(256), 0, h->stream, h->B.lower, (uint64_t)h->N * h->K);
        hipLaunchKernelGGL(k_init_from_kpp, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, h->M, h->N, h->K, h->prior, h->B,
                           h->kpp_todo, h->kpp_ntodo);
        HIP_TRY(hipMemcpyAsync(&n_todo, h->kpp_ntodo, 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (n_todo)
            hipLaunchKernelGGL(k_neighbor, dim3(n_todo), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M, h->kind, h->prior,
                               (float*)nullptr, h->B, h->kpp_todo);
        ck_end(h, CK_NEIGHBOR);
        HIP_TRY(hipGetLastError());
        // the shortcut trusts the k-means++ column-marginal filter: checked like the MFMA prune — on the sample in production,
        // on every point under RP_LLOYD_AUDIT — against the unpruned search (bucket and upper bound, bit for bit)
        if (h->sb_on && (h->sb_audit || h->sb_nsample)) {
            Bounds none{};
            if (h->sb_audit) {
                if ((rc = launch_neighbor_full(h, h->audit_j, h->audit_d, none))) return rc;
                hipLaunchKernelGGL(k_audit_compare, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, h->B.j, h->B.u, h->audit_j,
                                   h->audit_d, h->N, h->sb_bad);
                h->sb_audited += h->N;
            } else {
                hipLaunchKernelGGL(k_neighbor, dim3(h->sb_nsample), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M, h->kind,
                                   h->audit_j, h->audit_d, none, (const uint32_t*)h->sb_sample);
                hipLaunchKernelGGL(k_audit_compare_list, dim3((h->sb_nsample + 255) / 256), dim3(256), 0, h->stream, h->B.j, h->B.u,
                                   h->audit_j, h->audit_d, h->sb_sample, h->sb_nsample, h->sb_bad + 1);
                h->sb_sampled += h->sb_nsample;
                h->sb_check_pending = true;
            }
            HIP_TRY(hipGetLastError());
        }
    } else if ((rc = launch_neighbor(h, h->prior, nullptr, h->B, NB_INIT_BOUNDS))) {  // Prior::from_bounds (prior.rs:23-32)
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->bounds_ready = true;
    ck_drain(h);
    return prune_check(h, "rp_kmeans_init_bounds");
}

int rp_kmeans_step(rp_kmeans* h, float* drift, uint64_t* sizes, double* reassigned) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_step: NULL handle");
    int rc = need_centroids(h, "rp_kmeans_step");
    if (rc || (rc = need_bounds(h, "rp_kmeans_step"))) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if ((rc = step_front(h))) return rc;
    return step_back(h, drift, sizes, reassigned);
}

int rp_kmeans_partial_bytes(rp_kmeans* h, size_t* bytes) {
    if (!h || !bytes) return rp::fail(RP_ERR_INVALID, "rp_kmeans_partial_bytes: NULL argument");
    *bytes = partial_sizes_offset(h) + (size_t)h->K * 8;
    return RP_OK;
}

// partial layout: [K*bins u32 counts][K u32 weights][pad to 8][K u64 sizes] — exact integers, all-reduce(sum)-able
int rp_kmeans_step_local(rp_kmeans* h, void* partial_dev) {
    if (!h || !partial_dev) return rp::fail(RP_ERR_INVALID, "rp_kmeans_step_local: NULL argument");
    int rc = need_centroids(h, "rp_kmeans_step_local");
    if (rc || (rc = need_bounds(h, "rp_kmeans_step_local"))) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if ((rc = step_front(h))) return rc;
    const int nxt = h->cur ^ 1;
    unsigned char* out = reinterpret_cast<unsigned char*>(partial_dev);
    const size_t cb = (size_t)h->K * h->bins * 4, wb = (size_t)h->K * 4;
    HIP_TRY(hipMemcpyAsync(out, h->cs[nxt].counts, cb, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(out + cb, h->cs[nxt].weight, wb, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(out + cb + wb, 0, partial_sizes_offset(h) - (cb + wb), h->stream));
    HIP_TRY(hipMemcpyAsync(out + partial_sizes_offset(h), h->sizes, (size_t)h->K * 8, hipMemcpyDeviceToDevice, h->stream));
    return RP_OK;
}

int rp_kmeans_step_finish(rp_kmeans* h, const void* reduced_dev, float* drift, uint64_t* sizes, double* reassigned) {
    if (!h || !reduced_dev) return rp::fail(RP_ERR_INVALID, "rp_kmeans_step_finish: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    const int nxt = h->cur ^ 1;
    const unsigned char* in = reinterpret_cast<const unsigned char*>(reduced_dev);
    const size_t cb = (size_t)h->K * h->bins * 4, wb = (size_t)h->K * 4;
    HIP_TRY(hipMemcpyAsync(h->cs[nxt].counts, in, cb, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->cs[nxt].weight, in + cb, wb, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->sizes, in + partial_sizes_offset(h), (size_t)h->K * 8, hipMemcpyDeviceToDevice, h->stream));
    return step_back(h, drift, sizes, reassigned);
}

int rp_kmeans_step_comm(rp_kmeans* h, rp_comm* c, float* drift, uint64_t* sizes, double* reassigned) {
    if (!h || !c) return rp::fail(RP_ERR_INVALID, "rp_kmeans_step_comm: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t nb = partial_sizes_offset(h) + (size_t)h->K * 8;
    if (!h->comm_partial) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, nb));
        h->allocs.push_back(p);
        h->comm_partial = reinterpret_cast<unsigned char*>(p);
    }
    int rc = rp_kmeans_step_local(h, h->comm_partial);
    if (rc) return rc;
    // exact integers, order free: the u32 block (counts, weights) and the u64 block (sizes) of the partial
    if ((rc = rp::comm_all_reduce_sum(c, h->comm_partial, (size_t)h->K * h->bins + h->K, 0, h->stream))) return rc;
    if ((rc = rp::comm_all_reduce_sum(c, h->comm_partial + partial_sizes_offset(h), h->K, 1, h->stream))) return rc;
    return rp_kmeans_step_finish(h, h->comm_partial, drift, sizes, reassigned);
}

int rp_kmeans_step_naive(rp_kmeans* h) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_step_naive: NULL handle");
    int rc = need_centroids(h, "rp_kmeans_step_naive");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    Bounds none{};
    if ((rc = launch_neighbor(h, h->tmp_j, nullptr, none))) return rc;
    const int nxt = h->cur ^ 1;
    if ((rc = launch_recompute(h, h->tmp_j, nxt))) return rc;
    if ((rc = prepare_centroids(h, nxt))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->cur = nxt;
    h->pot_is_min_d2 = false;
    ck_drain(h);
    return prune_check(h, "rp_kmeans_step_naive");
}

int rp_kmeans_assign(rp_kmeans* h, uint8_t* bucket, float* distance) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_assign: NULL handle");
    int rc = need_centroids(h, "rp_kmeans_assign");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    Bounds none{};
    if ((rc = launch_neighbor(h, h->tmp_j, h->pdist, none, NB_LOOKUP))) return rc;
    if (bucket) HIP_TRY(hipMemcpyAsync(bucket, h->tmp_j, h->N, hipMemcpyDeviceToHost, h->stream));
    if (distance) HIP_TRY(hipMemcpyAsync(distance, h->pdist, h->N * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    ck_drain(h);
    return prune_check(h, "rp_kmeans_assign");
}

int rp_kmeans_bounds(rp_kmeans* h, uint8_t* j, float* upper, float* lower) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_bounds: NULL handle");
    int rc = need_bounds(h, "rp_kmeans_bounds");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if (j) HIP_TRY(hipMemcpyAsync(j, h->B.j, h->N, hipMemcpyDeviceToHost, h->stream));
    if (upper) HIP_TRY(hipMemcpyAsync(upper, h->B.u, h->N * 4, hipMemcpyDeviceToHost, h->stream));
    if (lower) HIP_TRY(hipMemcpyAsync(lower, h->B.lower, h->N * h->K * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RP_OK;
}

int rp_kmeans_centroids(rp_kmeans* h, uint32_t* counts, uint64_t* weight) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_centroids: NULL handle");
    int rc = need_centroids(h, "rp_kmeans_centroids");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    std::vector<uint32_t> w(h->K);
    if (counts) HIP_TRY(hipMemcpyAsync(counts, h->cs[h->cur].counts, (size_t)h->K * h->bins * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(w.data(), h->cs[h->cur].weight, h->K * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (weight) for (uint32_t k = 0; k < h->K; ++k) weight[k] = w[k];
    return RP_OK;
}

int rp_kmeans_metric(rp_kmeans* h, float* tri) {
    if (!h || !tri) return rp::fail(RP_ERR_INVALID, "rp_kmeans_metric: NULL argument");
    int rc = need_centroids(h, "rp_kmeans_metric");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    ck_begin(h, CK_PAIRWISE);
    hipLaunchKernelGGL(k_pairwise, dim3(h->K * h->K), dim3(64), 0, h->stream, h->cs[h->cur], h->K, h->M, h->kind, h->pairw, (const uint32_t*)nullptr, (uint32_t*)nullptr);
    ck_end(h, CK_PAIRWISE);
    HIP_TRY(hipGetLastError());
    std::vector<float> pw((size_t)h->K * h->K);
    HIP_TRY(hipMemcpyAsync(pw.data(), h->pairw, pw.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    // Layer::metric (layer.rs:85-101): (emd(x,y) + emd(y,x)) / 2, then Metric::from normalises by the max (metric.rs:127-141)
    float mx = RP_EPSILON;
    for (uint32_t i = 0; i < h->K; ++i)
        for (uint32_t j = 0; j < i; ++j) {
            float d = pw[(size_t)i * h->K + j] + pw[(size_t)j * h->K + i];
            d = d / 2.0f;
            tri[rp_tri_index(i, j)] = d;
            mx = rp_maxf(mx, d);
        }
    for (uint32_t t = 0; t < h->K * (h->K - 1) / 2; ++t) tri[t] = tri[t] / mx;
    ck_drain(h);
    return RP_OK;
}

int rp_kmeans_rms(rp_kmeans* h, float* out) {
    if (!h || !out) return rp::fail(RP_ERR_INVALID, "rp_kmeans_rms: NULL argument");
    int rc = need_bounds(h, "rp_kmeans_rms");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if (h->kind == RP_METRIC_VARIATION)
        hipLaunchKernelGGL(k_point_dist_var, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, h->P, h->cs[h->cur], h->K,
                           h->M, h->B.j, h->pdist);
    else
        hipLaunchKernelGGL(k_point_dist, dim3((unsigned)h->N), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M, h->kind, h->B.j,
                           h->pdist);
    HIP_TRY(hipGetLastError());
    std::vector<float> d(h->N);
    HIP_TRY(hipMemcpyAsync(d.data(), h->pdist, h->N * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    double acc = 0.0;  // f64 accumulation in point order (the reference's rayon f32 sum has no fixed order)
    for (uint64_t i = 0; i < h->N; ++i) acc += (double)(d[i] * d[i]);
    *out = (float)std::sqrt(acc / (double)h->N);
    return RP_OK;
}

static int read_stats(rp_kmeans* h, unsigned long long s[3]) {
    std::vector<unsigned long long> all((size_t)KM_STAT_STRIPES * STAT_STRIDE);
    HIP_TRY(hipMemcpyAsync(all.data(), h->stats, all.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    s[0] = s[1] = s[2] = 0;
    for (uint32_t q = 0; q < KM_STAT_STRIPES; ++q)
        for (uint32_t k = 0; k < 3; ++k) s[k] += all[(size_t)q * STAT_STRIDE + k];
    return RP_OK;
}

int rp_kmeans_stats(rp_kmeans* h, uint64_t* distances, uint64_t* sinkhorn_iterations) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_stats: NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    unsigned long long s[3];
    int rc = read_stats(h, s);
    if (rc) return rc;
    if (distances) *distances = s[0];
    if (sinkhorn_iterations) *sinkhorn_iterations = s[1];
    return RP_OK;
}

int rp_kmeans_exp_evals(rp_kmeans* h, uint64_t* evals) {
    if (!h || !evals) return rp::fail(RP_ERR_INVALID, "rp_kmeans_exp_evals: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    unsigned long long s[3];
    int rc = read_stats(h, s);
    if (rc) return rc;
    *evals = s[2];
    return RP_OK;
}

static int prune_stats_full(rp_kmeans* h, rp_prune_stats* out);
// the caller says how large ITS rp_prune_stats is: the struct grew in 0.3 (sampled_points, sample_mismatches) and may grow again;
// a host compiled against an older header gets the fields it knows, never a write past its struct
int rp_kmeans_prune_stats_sized(rp_kmeans* h, void* out, size_t out_bytes) {
    if (!h || !out || out_bytes < 8) return rp::fail(RP_ERR_INVALID, "rp_kmeans_prune_stats_sized: bad argument");
    rp_prune_stats full;
    const int rc = prune_stats_full(h, &full);
    if (rc) return rc;
    memcpy(out, &full, std::min(out_bytes, sizeof(full)));
    if (out_bytes > sizeof(full)) memset(reinterpret_cast<unsigned char*>(out) + sizeof(full), 0, out_bytes - sizeof(full));
    return RP_OK;
}
int rp_kmeans_prune_stats(rp_kmeans* h, rp_prune_stats* out) { return rp_kmeans_prune_stats_sized(h, out, sizeof(rp_prune_stats)); }
static int prune_stats_full(rp_kmeans* h, rp_prune_stats* out) {
    if (!h || !out) return rp::fail(RP_ERR_INVALID, "rp_kmeans_prune_stats: NULL argument");
    memset(out, 0, sizeof(*out));
    out->enabled = h->sb_on ? 1u : 0u;
    if (!h->sb_on) return RP_OK;
    HIP_TRY(hipSetDevice(h->device));
    std::vector<unsigned long long> all((size_t)KM_STAT_STRIPES * STAT_STRIDE);
    unsigned long long bad[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(all.data(), h->sb_stats, all.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(bad, h->sb_bad, 16, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    unsigned long long s[5] = {0, 0, 0, 0, 0};
    for (uint32_t q = 0; q < KM_STAT_STRIPES; ++q)
        for (uint32_t k = 0; k < 5; ++k) s[k] += all[(size_t)q * STAT_STRIDE + k];
    out->survivors = s[0];
    out->points = s[1];
    out->candidates = s[1] * h->K;
    out->block_iterations = s[2];
    out->cost_passes = s[3];
    out->mfma_instructions = s[4];
    out->audited_points = h->sb_audited;
    out->audit_mismatches = bad[0];
    out->sampled_points = h->sb_sampled;
    out->sample_mismatches = bad[1];
    return RP_OK;
}

int rp_kmeans_bound_intervals(rp_kmeans* h, float* lo, float* hi) {
    if (!h || !lo || !hi) return rp::fail(RP_ERR_INVALID, "rp_kmeans_bound_intervals: NULL argument");
    if (!h->sb_on) return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_bound_intervals: the layer has no MFMA bound (variation metric, or switched off)");
    int rc = need_centroids(h, "rp_kmeans_bound_intervals");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    float *d_lo = nullptr, *d_hi = nullptr;
    const size_t cells = (size_t)h->N * h->K;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_lo), cells * 4));
    if (hipMalloc(reinterpret_cast<void**>(&d_hi), cells * 4) != hipSuccess) {
        (void)hipFree(d_lo);
        return rp::fail(RP_ERR_HIP, "rp_kmeans_bound_intervals: out of device memory");
    }
    // points outside the bound's support classes keep [0, inf)
    std::vector<float> zeros(cells, 0.0f), infs(cells, INFINITY);
    (void)hipMemcpyAsync(d_lo, zeros.data(), cells * 4, hipMemcpyHostToDevice, h->stream);
    (void)hipMemcpyAsync(d_hi, infs.data(), cells * 4, hipMemcpyHostToDevice, h->stream);
    rc = launch_bound(h, d_lo, d_hi);
    if (!rc) {
        (void)hipMemcpyAsync(lo, d_lo, cells * 4, hipMemcpyDeviceToHost, h->stream);
        (void)hipMemcpyAsync(hi, d_hi, cells * 4, hipMemcpyDeviceToHost, h->stream);
    }
    const hipError_t e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_lo);
    (void)hipFree(d_hi);
    ck_drain(h);
    if (rc) return rc;
    if (e != hipSuccess) return rp::fail(RP_ERR_HIP, "rp_kmeans_bound_intervals: %s", hipGetErrorString(e));
    return RP_OK;
}

int rp_kmeans_set_stream(rp_kmeans* h, void* hip_stream) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_stream: NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->own_stream) {
        HIP_TRY(hipStreamDestroy(h->stream));
        h->own_stream = false;
    }
    if (hip_stream) h->stream = reinterpret_cast<hipStream_t>(hip_stream);
    else {
        HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    return RP_OK;
}

int rp_kmeans_profile(rp_kmeans* h, int enable) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_profile: NULL handle");
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    ck_drain(h);
    h->profiling = enable != 0;
    for (auto& c : h->clk) { c.total_ms = 0.0; c.launches = 0; }
    return RP_OK;
}

int rp_kmeans_kernel_time(rp_kmeans* h, const char* name, double* total_ms, uint64_t* launches) {
    if (!h || !name) return rp::fail(RP_ERR_INVALID, "rp_kmeans_kernel_time: NULL argument");
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    ck_drain(h);
    for (int i = 0; i < CK_COUNT; ++i)
        if (std::string(name) == CLOCK_NAMES[i]) {
            if (total_ms) *total_ms = h->clk[i].total_ms;
            if (launches) *launches = h->clk[i].launches;
            return RP_OK;
        }
    return rp::fail(RP_ERR_INVALID, "rp_kmeans_kernel_time: unknown kernel '%s'", name);
}

// ---- stand-alone batched distances -------------------------------------------------------------------
namespace {
std::atomic<int> g_pair_libm{RP_LIBM_CONTRACT};
int pair_common(uint32_t bins, uint64_t pairs, const uint32_t* mu, const uint32_t* nu, const float* tri, const rp_sinkhorn_hp* hp,
                int device, float* out, uint32_t* iterations, int divergence) {
    if (!mu || !nu || !out || pairs == 0 || bins == 0 || bins > MAXB || (!tri && bins > 1))
        return rp::fail(RP_ERR_INVALID, "rp_sinkhorn_*: bad argument");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_sinkhorn_*: no HIP device visible; no CPU fallback");
    rp_sinkhorn_hp hh;
    if (hp) hh = *hp; else rp_sinkhorn_hp_default(&hh);
    HIP_TRY(hipSetDevice(device));
    std::vector<float> C((size_t)bins * bins, 0.0f), R((size_t)bins * bins, 0.0f);
    for (uint32_t x = 0; x < bins; ++x)
        for (uint32_t y = 0; y < bins; ++y) {
            const float c = x == y ? 0.0f : tri[rp_tri_index(x, y)];
            C[(size_t)x * bins + y] = c;
            R[(size_t)x * bins + y] = c / hh.temperature;
        }
    float *dC = nullptr, *dR = nullptr, *dout = nullptr;
    uint32_t *dmu = nullptr, *dnu = nullptr, *dit = nullptr;
    unsigned long long *dstats = nullptr, *dscr = nullptr;
    const size_t hb = (size_t)pairs * bins * 4;
    HIP_TRY(hipMalloc(&dC, C.size() * 4));
    HIP_TRY(hipMalloc(&dR, R.size() * 4));
    HIP_TRY(hipMalloc(&dout, pairs * 4));
    HIP_TRY(hipMalloc(&dmu, hb));
    HIP_TRY(hipMalloc(&dnu, hb));
    HIP_TRY(hipMalloc(&dstats, 32));
    HIP_TRY(hipMemset(dstats, 0, 32));
    HIP_TRY(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dmu, mu, hb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dnu, nu, hb, hipMemcpyHostToDevice));
    Metric M{dC, dR, bins, hh.iterations, hh.tolerance, dstats, 1u};
    const bool glibc = g_pair_libm.load() == RP_LIBM_GLIBC;
    if (glibc) hipLaunchKernelGGL((k_pair_sinkhorn<1>), dim3((unsigned)pairs), dim3(64), 0, 0, dmu, dnu, M, divergence, dout, (uint32_t*)nullptr);
    else hipLaunchKernelGGL((k_pair_sinkhorn<0>), dim3((unsigned)pairs), dim3(64), 0, 0, dmu, dnu, M, divergence, dout, (uint32_t*)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, pairs * 4, hipMemcpyDeviceToHost));
    if (iterations) {
        HIP_TRY(hipMalloc(&dit, pairs * 4));
        HIP_TRY(hipMalloc(&dscr, pairs * 32));
        HIP_TRY(hipMemset(dscr, 0, pairs * 32));
        if (glibc) hipLaunchKernelGGL((k_pair_iters<1>), dim3((unsigned)pairs), dim3(64), 0, 0, dmu, dnu, M, dscr, dit);
        else hipLaunchKernelGGL((k_pair_iters<0>), dim3((unsigned)pairs), dim3(64), 0, 0, dmu, dnu, M, dscr, dit);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(iterations, dit, pairs * 4, hipMemcpyDeviceToHost));
        (void)hipFree(dit);
        (void)hipFree(dscr);
    }
    (void)hipFree(dC); (void)hipFree(dR); (void)hipFree(dout); (void)hipFree(dmu); (void)hipFree(dnu); (void)hipFree(dstats);
    return RP_OK;
}
}  // namespace

int rp_sinkhorn_set_libm(rp_libm_kind kind) {
    if (kind != RP_LIBM_CONTRACT && kind != RP_LIBM_GLIBC) return rp::fail(RP_ERR_INVALID, "rp_sinkhorn_set_libm: unknown kind %d", (int)kind);
    g_pair_libm.store(kind);
    return RP_OK;
}
int rp_sinkhorn_divergence(uint32_t bins, uint64_t pairs, const uint32_t* mu, const uint32_t* nu, const float* tri_metric,
                           const rp_sinkhorn_hp* hp, int device, float* out) {
    return pair_common(bins, pairs, mu, nu, tri_metric, hp, device, out, nullptr, 1);
}
int rp_sinkhorn_cost(uint32_t bins, uint64_t pairs, const uint32_t* mu, const uint32_t* nu, const float* tri_metric,
                     const rp_sinkhorn_hp* hp, int device, float* out, uint32_t* iterations) {
    return pair_common(bins, pairs, mu, nu, tri_metric, hp, device, out, iterations, 0);
}
int rp_sinkhorn_flow(uint32_t bins, const uint32_t* mu, const uint32_t* nu, const float* tri, const rp_sinkhorn_hp* hp, int device,
                     float* flow, float* coupling) {
    if (!mu || !nu || !flow || bins == 0 || bins > MAXB || (!tri && bins > 1)) return rp::fail(RP_ERR_INVALID, "rp_sinkhorn_flow: bad argument");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_sinkhorn_flow: no HIP device visible; no CPU fallback");
    rp_sinkhorn_hp hh;
    if (hp) hh = *hp; else rp_sinkhorn_hp_default(&hh);
    HIP_TRY(hipSetDevice(device));
    const size_t cells = (size_t)bins * bins;
    std::vector<float> C(cells, 0.0f), R(cells, 0.0f);
    for (uint32_t x = 0; x < bins; ++x)
        for (uint32_t y = 0; y < bins; ++y) {
            const float c = x == y ? 0.0f : tri[rp_tri_index(x, y)];
            C[(size_t)x * bins + y] = c;
            R[(size_t)x * bins + y] = c / hh.temperature;
        }
    float* buf = nullptr;  // C | R | flow | coupling
    uint32_t* hist = nullptr;
    unsigned long long* dstats = nullptr;
    HIP_TRY(hipMalloc(&buf, cells * 4 * 4));
    HIP_TRY(hipMalloc(&hist, (size_t)bins * 8));
    HIP_TRY(hipMalloc(&dstats, 32));
    HIP_TRY(hipMemset(dstats, 0, 32));
    HIP_TRY(hipMemset(buf + 2 * cells, 0, cells * 8));
    HIP_TRY(hipDeviceSynchronize());  // null-stream memsets before launches on other streams
    HIP_TRY(hipMemcpy(buf, C.data(), cells * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(buf + cells, R.data(), cells * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(hist, mu, (size_t)bins * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(hist + bins, nu, (size_t)bins * 4, hipMemcpyHostToDevice));
    Metric M{buf, buf + cells, bins, hh.iterations, hh.tolerance, dstats, 1u};
    if (g_pair_libm.load() == RP_LIBM_GLIBC)
        hipLaunchKernelGGL((k_pair_flow<1>), dim3(1), dim3(64), 0, 0, hist, hist + bins, M, buf + 2 * cells, buf + 3 * cells);
    else hipLaunchKernelGGL((k_pair_flow<0>), dim3(1), dim3(64), 0, 0, hist, hist + bins, M, buf + 2 * cells, buf + 3 * cells);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(flow, buf + 2 * cells, cells * 4, hipMemcpyDeviceToHost));
    if (coupling) HIP_TRY(hipMemcpy(coupling, buf + 3 * cells, cells * 4, hipMemcpyDeviceToHost));
    (void)hipFree(buf); (void)hipFree(hist); (void)hipFree(dstats);
    return RP_OK;
}
int rp_equity_variation(uint32_t bins, uint64_t pairs, const uint32_t* x, const uint32_t* y, int device, float* out) {
    if (!x || !y || !out || pairs == 0 || bins == 0) return rp::fail(RP_ERR_INVALID, "rp_equity_variation: bad argument");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_equity_variation: no HIP device visible; no CPU fallback");
    HIP_TRY(hipSetDevice(device));
    uint32_t *dx = nullptr, *dy = nullptr;
    float* dout = nullptr;
    const size_t hb = (size_t)pairs * bins * 4;
    HIP_TRY(hipMalloc(&dx, hb));
    HIP_TRY(hipMalloc(&dy, hb));
    HIP_TRY(hipMalloc(&dout, pairs * 4));
    HIP_TRY(hipMemcpy(dx, x, hb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dy, y, hb, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pair_variation, dim3((unsigned)((pairs + 63) / 64)), dim3(64), 0, 0, dx, dy, bins, pairs, dout);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, pairs * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout);
    return RP_OK;
}

}  // extern "C"

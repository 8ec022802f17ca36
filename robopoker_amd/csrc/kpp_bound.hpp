// kpp_bound.hpp — the second filter of the k-means++ rounds (Layer::init_centroids, layer.rs:170-178): a scaling-domain Sinkhorn
// INTERVAL for (new centroid, point) pairs on the vector ALU.  Included by lloyd_kernels.hpp (contract arithmetic only), after
// sinkhorn_bound.hpp whose window logic, noise model and margins (SbParams) it reuses.
//
// WHY.  A round updates potentials <- min(potentials, d(c_k, x)^2): the bit-faithful solve is needed only where d^2 can be below the
// current potential.  The column-marginal bound (k_kpp_filter) lets ~30 % of the N x K pairs through; the pairs whose distance really
// is below the potential are ~2 % (sum_k 1/k per point).  The scaling-domain iteration  u = mu ./ (K v),  v = nu ./ (K^T u)  is the
// same Gauss-Seidel trajectory as the reference's log-domain solve (sinkhorn.rs:77-139) without a single exponential — two
// multiply-adds per matrix entry per iteration against ~40 VALU instructions in the faithful kernel — and both supports are small here
// (the centroid of a k-means++ round IS a point), so the K sub-matrix of a pair lives in REGISTERS: no MFMA, no LDS tile.
//
// MAPPING.  ROWS lanes per pair (32: two pairs per wavefront, 64: one).  Lane r of a pair is row r of the centroid (holds K[x_r][y_j],
// j < n, and K.C of the same entries for the cost) AND column r of the point (holds K[x_i][y_r], i < m): both half-iterations are
// lane-local dot products against the other side's vector, broadcast through LDS (ds_read_b128, address uniform per pair).  The
// stopping statistic, the noise bound and the cost are all-reduced over the pair's lanes with DPP row operations and
// v_permlane16/32_swap.  The window is sinkhorn_bound.hpp's: every iteration at which the reference could stop contributes its cost;
// lo = min over the window - (dc_abs + dc_rel cost), pushed through the divergence's three f32 operations.  A pair is dropped from the
// round iff lo^2 >= potential (the solve could not lower it); as soon as an iterate inside the window falls below the potential the
// pair is kept without finishing the window.  The noise bound is the MFMA kernel's with (sum u + sum v)(max(ln u_max, ln v_max) + 4)
// in place of the two separate products: never smaller, so windows only widen.
//
// THE DUAL EXIT (round 6, prm.lip >= 2; sinkhorn_bound.hpp has the argument).  93 % of the pairs that reach this kernel end with
// lo^2 >= potential, after following their window to its end (58 iterations on average).  From the second iteration on, the pair
// f = T ln u', g = -T ln (K^T u') of the iterate just produced is feasible for the unregularised problem and the row marginals' L1 error
// e of the iterate the iteration started from does not grow, so every iterate from here on costs at least  <mu, f> + <nu, g> - max C e / 2;
// the iterates the reference could have stopped at before are the window's.  A pair whose divergence at that lower end already reaches
// the potential leaves at once.  One more all-reduction and two logarithms per iteration.
#pragma once

#define KB_WAVES_PER_CU 8u

template <uint32_t ROWS>
struct __attribute__((aligned(16))) KbLds {
    static constexpr uint32_t G = 64u / ROWS;
    uint32_t xC[64];        // centroid support bins (padded with xC[0])
    float muC[64];          // centroid densities, 0 past the support
    uint32_t yP[G][ROWS];   // the points' support bins
    float nuP[G][ROWS];
    float ub[G][ROWS];      // u of each pair, broadcast
    float vb[G][ROWS];      // v
    uint32_t n[G];
    uint32_t item;
};

#if defined(RP_EMUL)
// the execution model of tests/emul has no DPP: the same all-reductions as xor butterflies
template <uint32_t ROWS>
__device__ __forceinline__ float kb_allsum(float x) {
    for (int o = (int)ROWS / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
template <uint32_t ROWS>
__device__ __forceinline__ float kb_allmax(float x) {
    for (int o = (int)ROWS / 2; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}
#else
template <int CTRL>
__device__ __forceinline__ float kb_dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
// every lane of a row of 16 ends with the row's total: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float kb_row_sum(float x) {
    x += kb_dpp<0xB1>(x);
    x += kb_dpp<0x4E>(x);
    x += kb_dpp<0x141>(x);
    x += kb_dpp<0x140>(x);
    return x;
}
__device__ __forceinline__ float kb_row_max(float x) {
    x = fmaxf(x, kb_dpp<0xB1>(x));
    x = fmaxf(x, kb_dpp<0x4E>(x));
    x = fmaxf(x, kb_dpp<0x141>(x));
    x = fmaxf(x, kb_dpp<0x140>(x));
    return x;
}
// both halves of the exchange as scalars (an ext-vector ELEMENT handed to __builtin_bit_cast reads element 0: copy first)
__device__ __forceinline__ void kb_swap16(float x, float& a, float& b) {  // a: the even row of each row pair, b: the odd one, in all 32 lanes
    const uint32_t w = __builtin_bit_cast(uint32_t, x);
    const auto r = __builtin_amdgcn_permlane16_swap(w, w, false, false);
    const uint32_t r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void kb_swap32(float x, float& a, float& b) {  // a: lanes 0-31's value, b: lanes 32-63's, in all 64 lanes
    const uint32_t w = __builtin_bit_cast(uint32_t, x);
    const auto r = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    const uint32_t r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
template <uint32_t ROWS>
__device__ __forceinline__ float kb_allsum(float x) {
    float a, b;
    x = kb_row_sum(x);
    kb_swap16(x, a, b);  // rows 2q and 2q+1: v_permlane16_swap exchanges the odd rows of one operand with the even rows of the other
    x = a + b;
    if (ROWS == 64) {
        kb_swap32(x, a, b);
        x = a + b;
    }
    return x;
}
template <uint32_t ROWS>
__device__ __forceinline__ float kb_allmax(float x) {
    float a, b;
    x = kb_row_max(x);
    kb_swap16(x, a, b);
    x = fmaxf(a, b);
    if (ROWS == 64) {
        kb_swap32(x, a, b);
        x = fmaxf(a, b);
    }
    return x;
}
#endif

// sum_{j < R} a[j] * vec[j], vec in LDS (16-B aligned, uniform per pair), stopping at the wave-uniform bound `lim` (a multiple of 4)
template <uint32_t R>
__device__ __forceinline__ float kb_dot(const float (&a)[R], const float* vec, uint32_t lim) {
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < R; j += 8) {
        if (j < lim) {
            const float4 q = *reinterpret_cast<const float4*>(vec + j);
            s0 = __builtin_fmaf(a[j], q.x, s0);
            s1 = __builtin_fmaf(a[j + 1], q.y, s1);
            s0 = __builtin_fmaf(a[j + 2], q.z, s0);
            s1 = __builtin_fmaf(a[j + 3], q.w, s1);
        }
        if (j + 4 < lim) {
            const float4 q = *reinterpret_cast<const float4*>(vec + j + 4);
            s0 = __builtin_fmaf(a[j + 4], q.x, s0);
            s1 = __builtin_fmaf(a[j + 5], q.y, s1);
            s0 = __builtin_fmaf(a[j + 6], q.z, s0);
            s1 = __builtin_fmaf(a[j + 7], q.w, s1);
        }
    }
    return s0 + s1;
}

// the same sum and, beside it, max_j a[j] * vec[j] (the dual exit's c-transform: -T ln of it is the largest g(y) that keeps f + g <= C)
template <uint32_t R>
__device__ __forceinline__ float kb_dot_max(const float (&a)[R], const float* vec, uint32_t lim, float& mx) {
    float s0 = 0.0f, s1 = 0.0f, m0 = 0.0f, m1 = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < R; j += 4) {
        if (j < lim) {
            const float4 q = *reinterpret_cast<const float4*>(vec + j);
            s0 = __builtin_fmaf(a[j], q.x, s0);
            s1 = __builtin_fmaf(a[j + 1], q.y, s1);
            s0 = __builtin_fmaf(a[j + 2], q.z, s0);
            s1 = __builtin_fmaf(a[j + 3], q.w, s1);
            m0 = fmaxf(m0, fmaxf(a[j] * q.x, a[j + 2] * q.z));
            m1 = fmaxf(m1, fmaxf(a[j + 1] * q.y, a[j + 3] * q.w));
        }
    }
    mx = fmaxf(m0, m1);
    return s0 + s1;
}

// kstats (striped like Metric::stats): [0] pairs examined, [1] pairs kept (without the tripwire's), [2] pair-iterations, [3] cost passes
// in_list / in_count: the round's candidates of one support class (k_kpp_filter); out_list / out_count: the ones the solve is still
// needed for.  dbg_lo (tests): the lower bound of the divergence per point, 0 where the pair was kept without one.
template <uint32_t ROWS, uint32_t R>
__global__ __launch_bounds__(64) void k_kpp_bound(Points P, CentroidSet cs, uint32_t k, uint32_t bins, const float* Cm, SbParams prm,
                                                  const float* pot, const uint32_t* in_list, const unsigned int* in_count,
                                                  unsigned int* cursor, uint32_t* out_list, unsigned int* out_count,
                                                  unsigned long long* kstats, float* dbg_lo, int dual, float* claim, float claim_scale) {
    constexpr uint32_t G = 64u / ROWS;
    static_assert(R <= ROWS && R % 8 == 0, "register tile");
    __shared__ KbLds<ROWS> L;
    const uint32_t lane = threadIdx.x & 63u, g = lane / ROWS, r = lane % ROWS;
    const uint32_t have = in_count ? *in_count : 0u;
    const uint32_t m = cs.n[k];
    const float sc = cs.self[k];
    // the centroid: support and densities
    {
        const uint32_t x = lane < m ? (uint32_t)cs.sup[(size_t)k * MAXB + lane] : (m ? (uint32_t)cs.sup[(size_t)k * MAXB] : 0u);
        L.xC[lane] = x;
        L.muC[lane] = lane < m ? cs.densR[(size_t)k * MAXB + x] : 0.0f;
    }
    __syncthreads();
    const uint32_t xr = L.xC[r];
    const float mur = r < m ? L.muC[r] : 0.0f;
    const uint32_t mlim = (min(m, R) + 3u) & ~3u;
    unsigned long long my_iters = 0, my_costs = 0, my_pairs = 0, my_kept = 0;
    for (;;) {
        __syncthreads();  // the previous pairs' LDS is no longer read
        if (lane == 0) L.item = atomicAdd(cursor, G);
        __syncthreads();
        const uint32_t item0 = L.item;
        if (item0 >= have) break;
        // ---- the G points: support (ascending bins) and densities (Bins::density, bins.rs:58-60)
        uint64_t ipt[G];
#pragma unroll
        for (uint32_t h = 0; h < G; ++h) {
            const bool real = item0 + h < have;
            ipt[h] = real ? in_list[item0 + h] : in_list[item0];
            const uint8_t* counts = P.counts + ipt[h] * P.stride;
            const float fw = (float)P.weight[ipt[h]];
            uint32_t base = 0;
            for (uint32_t q = 0; q * 64 < bins; ++q) {
                const uint32_t bb = q * 64 + lane;
                const uint32_t cc = bb < bins ? (uint32_t)counts[bb] : 0u;
                const bool has = cc > 0;
                const unsigned long long msk = __ballot(has);
                if (has) {
                    const uint32_t rr = base + __popcll(msk & ((1ull << lane) - 1ull));
                    if (rr < ROWS) {
                        L.yP[h][rr] = bb;
                        L.nuP[h][rr] = (float)cc / fw;
                    }
                }
                base += __popcll(msk);
            }
            if (lane == 0) L.n[h] = real ? base : 0u;
        }
        __syncthreads();
        const uint32_t n = L.n[g];
        const uint64_t ip = G == 1 ? ipt[0] : (g == 0 ? ipt[0] : ipt[G - 1]);
        // a pair outside the register tile (or an empty slot) is not examined: a real one is kept
        const bool fits = n > 0 && n <= R && m > 0 && m <= R;
        uint32_t nmax = 0;
#pragma unroll
        for (uint32_t h = 0; h < G; ++h) nmax = max(nmax, min(L.n[h], R));
        const uint32_t nlim = (nmax + 3u) & ~3u;
        if (r >= n && r < ROWS) {  // padding: bin 0 of the point, zero mass
            L.yP[g][r] = n ? L.yP[g][0] : 0u;
            L.nuP[g][r] = 0.0f;
        }
        __syncthreads();
        const uint32_t yr = L.yP[g][r];
        const float nur = (fits && r < n) ? L.nuP[g][r] : 0.0f;
        // ---- K sub-matrix of the pair in registers: row r (centroid bin x_r against the point's bins) with K.C beside it, column r
        float Ku[R], KC[R], Kv[R];
        float cmx = 0.0f;  // max C over the pair's supports (the dual exit)
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) {
            const bool ok = fits && r < m && j < n;
            const uint32_t y = L.yP[g][j];
            const float kk = ok ? prm.Kmat[xr * 256u + y] : 0.0f;
            const float cc = ok ? Cm[xr * bins + y] : 0.0f;
            Ku[j] = kk;
            KC[j] = kk * cc;
            cmx = fmaxf(cmx, cc);
        }
        if (dual) cmx = kb_allmax<ROWS>(cmx) * 1.0001f;
#pragma unroll
        for (uint32_t i = 0; i < R; ++i) {
            const bool ok = fits && i < m && r < n;
            Kv[i] = ok ? prm.Kmat[L.xC[i] * 256u + yr] : 0.0f;
        }
        // Potential::uniform (phi.rs:34-39): exp(lhs) = 1/|supp mu|, exp(rhs) = 1/|supp nu|
        float u = (fits && r < m) ? 1.0f / (float)m : 0.0f;
        float v = (fits && r < n) ? 1.0f / (float)n : 0.0f;
        L.vb[g][r] = v;
        const float sp = P.self[ip];
        const float potv = pot[ip];
        const float fm = (float)m, fn = (float)n;
        float wmin = __builtin_inff(), nb_prev = SB_EPS23 * (8.0f + 0.37f * (fm + fn));
        int flatc = 0;
        bool opened = false, complete = false, keep = false;
        bool done = !fits;
        if (n > 0 && !fits) keep = true;
        bool dropped = false;  // left by the dual exit: dlq is its lower bound of the divergence
        float dlq = 0.0f;
        __syncthreads();
        for (uint32_t t = 0; t < prm.iters; ++t) {
            if (__ballot(!done) == 0) break;
            const bool last = t + 1 == prm.iters;
            // lhs: u <- mu ./ (K v)
            const float su = kb_dot<R>(Ku, L.vb[g], nlim);
            const float un = mur * sb_rcp(fmaxf(su, 1e-37f));
            float e = fabsf(un - u);
            // this lane's share of  T ln 2 (mu_x log2 u'_x - nu_y log2 (K^T u')_y) - max C / 2 * 1.01 |mu_x - u_x (K v)_x|   (old u: the row
            // marginal of the coupling the iteration starts from; log2 u' from the float's bits, a lower bound: sinkhorn_bound.hpp)
            float dq = 0.0f;
            if (dual) dq = (-prm.neg_t_ln2) * (mur * ((float)((int)__float_as_uint(un) - 0x3f800000) * 1.1920929e-7f)) - 0.505f * cmx * fabsf(mur - u * su);
            u = un;
            L.ub[g][r] = u;
            __syncthreads();
            // rhs: v <- nu ./ (K^T u), on the fresh u (Gauss-Seidel, sinkhorn.rs:80-87)
            float mxv = 0.0f;
            const float sv = dual ? kb_dot_max<R>(Kv, L.ub[g], mlim, mxv) : kb_dot<R>(Kv, L.ub[g], mlim);
            const float vn = nur * sb_rcp(fmaxf(sv, 1e-37f));
            e += fabsf(vn - v);
            v = vn;
            L.vb[g][r] = v;
            // g(y) = -T ln max_x K_xy u'_x (the c-transform of f; dual == 1: -T ln (K^T u')_y, the looser pair that needs no maximum)
            if (dual) dq -= (-prm.neg_t_ln2) * (nur * __builtin_amdgcn_logf(fmaxf(dual >= 2 ? mxv : sv, 1e-37f)));
            __syncthreads();
            const float err = kb_allsum<ROWS>(e);
            const float suv = kb_allsum<ROWS>(u + v);
            const float mx = kb_allmax<ROWS>(fmaxf(u, v));
            if (!done) my_iters += (r == 0);
            const float lmx = fmaxf(__builtin_amdgcn_logf(mx) * 0.6931472f, 0.0f);
            const float nb = SB_EPS23 * (suv * (lmx + 4.0f) + 0.37f * (fm + fn));
            const float noise = prm.kappa * (nb + nb_prev);
            nb_prev = nb;
            const bool possible = last || (err - noise < prm.tol * prm.rho);
            const bool certain = last || ((err + noise) * prm.rho < prm.tol);
            const bool flat = err <= prm.flat * SB_EPS23 * suv;
            const bool want = !done && (possible || flat);
            if (__ballot(want)) {  // the cost of this iterate: sum_x u_x sum_y K C v_y
                const float part = u * kb_dot<R>(KC, L.vb[g], nlim);
                const float cost = kb_allsum<ROWS>(part);
                if (want) {
                    my_costs += (r == 0);
                    wmin = cost == cost ? fminf(wmin, cost) : -__builtin_inff();  // a non-finite cost: the pair is kept
                    opened = true;
                    // an iterate of the window already below the potential: the solve is needed whatever the rest of the window holds
                    const float cl = wmin - (prm.dc_abs + prm.dc_rel * fabsf(wmin));
                    const float dl = rp_maxf(cl - 0.5f * sc - 0.5f * sp, 0.0f);
                    if (!(dl * dl >= potv)) {
                        done = true;
                        keep = true;
                    }
                }
            }
            flatc = flat ? flatc + 1 : 0;
            if (!done && (certain || flatc >= 2)) {
                done = true;
                complete = true;
            }
            if (dual && t >= 1) {  // (t >= 1: the coupling the iteration started from has had its rhs update; wave uniform: the reduction)
                const float lbc = kb_allsum<ROWS>(dq) - ((-prm.neg_t_ln2) * SB_DUAL_SLACK + 0.5f * cmx * 5e-5f);
                const float cmin = opened ? fminf(wmin, lbc) : lbc;
                const float cl = cmin - (prm.dc_abs + prm.dc_rel * fabsf(cmin));
                const float dd = rp_maxf(cl - 0.5f * sc - 0.5f * sp, 0.0f);
                if (!done && dd * dd >= potv) {  // every iterate the reference could stop at is at or above the potential (not a number: no)
                    done = true;
                    dropped = true;
                    dlq = dd;
                }
            }
        }
        float dl = 0.0f;
        if (dropped) {
            dl = dlq;
        } else if (fits && !keep) {
            if (opened && complete) {
                const float cl = wmin - (prm.dc_abs + prm.dc_rel * fabsf(wmin));
                dl = (cl == cl && cl > -__builtin_inff()) ? rp_maxf(cl - 0.5f * sc - 0.5f * sp, 0.0f) : 0.0f;
                keep = !(dl * dl >= potv);  // a NaN potential or bound keeps the pair
            } else {
                keep = true;
            }
        }
        if (r == 0 && n > 0) {
            my_pairs += 1;
            // the tripwire of the rounds (Metric::kpp_claim): every 521st point that is dropped goes to the solve all the same, with the
            // bound it was dropped with; the solve changes nothing if the bound holds (d^2 >= potential) and counts a disagreement if not
            bool solve = keep;
            if (!keep && claim && dl > 0.0f && ip % 521ull == 0ull) {
                claim[ip] = dl * claim_scale;  // (claim_scale: 1; a test sets it to prove that the wire trips)
                solve = true;
            }
            if (keep) my_kept += 1;
            if (solve) out_list[atomicAdd(out_count, 1u)] = (uint32_t)ip;
            if (dbg_lo) dbg_lo[ip] = keep ? 0.0f : dl;
        }
    }
    if (r == 0 && (my_pairs | my_iters)) {
        unsigned long long* s = kstats + (size_t)(blockIdx.x % KM_STAT_STRIPES) * STAT_STRIDE;
        atomicAdd(s + 0, my_pairs);
        atomicAdd(s + 1, my_kept);
        atomicAdd(s + 2, my_iters);
        atomicAdd(s + 3, my_costs);
    }
}

// sinkhorn_bound.hpp — the MFMA path of the lloyd pipeline: a scaling-domain Sinkhorn *bound* in front of the N x K
// neighbor passes (Elkan::init_bounds elkan.rs:39-47, Elkan::neighbor :68-77, Layer::lookup layer.rs:62-82).
// Included by lloyd.hip inside namespace rp, after Points / CentroidSet / Metric.
//
// WHAT IT IS.  The reference's distance is Sinkhorn::divergence (sinkhorn.rs:166-171) whose cross term is the cost of a
// log-domain Gauss-Seidel solve stopped by an L1 test on exp(potential) (sinkhorn.rs:77-139).  The bit-faithful kernel
// (wave_sinkhorn_cost) reproduces it float for float and is exp-bound.  The SAME iteration in the scaling domain,
//      u = mu ./ (K v),   v = nu ./ (K^T u),   K = exp(-C / T)    (u = exp(lhs), v = exp(rhs)),
// is two dense contractions per iteration with a K shared by every pair, i.e. GEMMs: for ONE point (nu, <= 64 support bins)
// against ALL centroids (mu_j, j < 256)
//      S[x][j] = sum_y K[x][y] V[y][j]      (256 x n_p) . (n_p x 256)
//      R[y][j] = sum_x K[y][x] U[x][j]      (n_p x 256) . (256 x 256)
// on v_mfma_f32_16x16x4_f32 (exact f32 products, the guide's 157.3 TF path).  It cannot be bit-faithful to the log-domain
// left folds, so it is not a replacement: it yields, per (point, centroid), an INTERVAL [lo, hi] that contains the value
// the faithful kernel would return, and a centroid whose lo exceeds the smallest hi cannot be the argmin.  Only the
// survivors (1.0x per point on the flop layer) go through the faithful kernel, in ascending centroid order, so buckets,
// distances and tie-breaks are bit-identical to the unpruned pass.
//
// WHY THE INTERVAL HOLDS (DESIGN.md §4b has the measurements behind every constant).
//  (1) Same trajectory.  Both computations iterate the same contraction from the same start (Potential::uniform) in the
//      same Gauss-Seidel order; f32 scaling-domain and f32 log-domain iterates differ from the real-number trajectory by
//      rounding only.  Measured against a float64 restatement: |cost_t(oracle) - cost_t(f64)| <= 4.5e-6 * cost at every
//      iteration t, the scaling-domain f32 iterate the same; the margin used is dc = dc_abs + dc_rel * cost (4e-6, 4e-5).
//  (2) Unknown stopping time.  The reference stops at the first t with err_t < tol, where err_t = sum |exp(f') - exp(f)|
//      is evaluated on f32 LOG potentials: its rounding noise is about sum_x u_x (|ln u_x| + c) 2^-23, which for far pairs
//      (u ~ 1e7) dwarfs tol, so the stop can come "early" (the log potentials stall bitwise) or never (128 iterations).
//      The bound therefore does not predict the stop.  It tracks a noise bound N_t (an over-estimate: kappa times the
//      measured worst case) and takes the cost range over EVERY iteration at which the reference could stop:
//         window = [ first t with err_t - N_t < tol * rho ,  first t with (err_t + N_t) * rho < tol ]   (or the cap, or
//      the iteration at which the scaling iterate itself stops moving).  lo/hi = min/max of the cost over the window -/+ dc.
//  (3) Window iterates without an evaluation (round 6, prm.lip).  One iteration moves the coupling P = diag(u) K diag(v) by
//      dlt_t = sum_x (K v)_x |u'_x - u_x| + sum_y (K^T u')_y |v'_y - v_y| in L1 (the lhs update, then the rhs update; both sums
//      ride on the iteration: (K v)_x and (K^T u')_y are its contractions), so |cost_t - cost_s| <= max C * sum_{s < tau <= t} dlt_tau
//      — exact arithmetic, no measured constant; max C over the point's rows.  A column keeps its last evaluated cost and that
//      sum; a window iterate whose divergence would exceed a published upper bound even at the lower end of the range enters the
//      window as the range instead of an evaluation.  Such a column cannot be the argmin whatever the exact value; every other
//      window iterate is evaluated as before.  On the flop layer 75 % of the wavefront iterations carried a cost contraction
//      before, 4 % do now.
//  A too-large N_t, rho or dc only widens intervals (more survivors, never a wrong answer).  RP_LLOYD_AUDIT=1 runs the
//  unpruned pass next to the pruned one and counts disagreements (rp_kmeans_prune_stats); the GPU tests do that.
//
// MAPPING.  One workgroup (2 or 4 wavefronts) per point; a wavefront iterates 16 centroid column SLOTS at a time (k_sinkhorn_bound below).
//  K[sup_y][.] of the point's support is gathered once into LDS (row stride 260 floats: the second contraction's A operand is a
//  conflict-free ds_read_b128, the first one's four ds_read_b32 down a column).  U, V live in registers in the MFMA C/D layout, which IS
//  the B-operand layout of the next contraction when the k index is permuted (row 4g + r of an accumulator feeds k = g of step r), so
//  the two GEMMs chain without any data movement: U never exists in memory.  Per wave: U 64 VGPRs, V 4*NT, accumulators 4*NT + 8.
//  mu_j(x) streams from L2 (256 KB table, b128 per lane).
//
// THE DUAL EXIT (round 6, prm.lip >= 2).  A far column used to iterate to the end of its stopping window only to be pruned.  For ANY
//  positive u the pair  f(x) = T ln u_x,  g(y) = -T ln (K^T u)_y  is feasible for the UNREGULARISED transport problem (f + g <= C because
//  u_x K_xy is one term of (K^T u)_y), so <mu, f> + <nu, g> <= OT_0(mu, nu) by weak duality — no iteration count, no measured constant.
//  After a rhs update the coupling's column marginals are nu exactly and its row marginals a_s are within e_s (L1) of mu, hence
//  cost_s = <P_s, C> >= OT_0(a_s, nu) >= OT_0(mu, nu) - max C e_s / 2, and e_s never grows (each half step pushes both marginals through
//  one stochastic kernel).  So from the second iteration on, every iterate the reference can still stop at costs at least
//  dual_t - max C e_{t-1} / 2; the iterates it could have stopped at already are the window's.  Both sums ride on the iteration (ln u from
//  the float's bits: a lower bound of the logarithm, which f may be).  A column whose divergence at that lower end exceeds a published
//  upper bound leaves at once with [bound, inf).  Full flop layer: column iterations 10.8 G -> 4.8 G.
#pragma once

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define SB_KS 260       // ksub row stride in floats: 260 mod 64 = 4 -> the 16 lanes of a ds_read_b128 group (address c KS + 4 g) hit 64 banks
                        // once, and so do the 64 lanes of a ds_read_b32 at (4 g) KS + c (4 KS mod 64 = 16)
#define SB_MAXROWS 64   // a point with more support bins than this is not pruned (all its centroids go to the exact kernel)
#define SB_EPS23 1.1920929e-7f
#define SB_MU_AHEAD 4
#define SB_TIGHT_EVERY 16u  // wavefront iterations between two c-transform evaluations of the dual exit (0: never); measured 2 / 4 / 8 / 12 / 16 / 24 / 32: 2.72 / 2.16 / 1.92 / 1.86 / 1.84 / 1.84 / 1.87 s, without 2.38
#define SB_DUAL_SLACK 1e-3f  // log2 units (x T ln 2 = 1.7e-5 of cost at T = 0.025)

struct SbParams {
    const float* Kmat;  // [256][256] exp(-C/T); 1.0 outside the bins x bins block
    float neg_t_ln2;    // C = neg_t_ln2 * log2(K)
    float tol;
    uint32_t iters;
    float kappa;   // multiplier of the stopping-statistic noise model
    float rho;     // slack factor of the stopping window (>= 1)
    float dc_abs;  // cost margin: dc_abs + dc_rel * cost
    float dc_rel;
    float flat;    // the scaling iterate counts as stationary when err <= flat * 2^-23 * (sum u + sum v), twice in a row
    int use_lb0;   // the column-marginal bound is valid for this metric / temperature (max C / T <= 64): sort and drop by it
    uint32_t tight;  // the dual exit's c-transform pair every `tight`-th wavefront iteration (0: never; RP_SB_TIGHT)
    int lip;       // >= 1: the cost of a window iterate is taken from the last evaluated one and a Lipschitz bound when that is enough;
                   // 2 (the default): and a column leaves as soon as a Kantorovich dual bound puts it above a published upper bound (see k_sinkhorn_bound)
};

// LDS per workgroup = per point, sized by the point's support class.  K's rows are kept in ONE orientation (until round 6 also transposed,
// so that both contractions' A operands were ds_read_b128: 46 / 78 / 110 / 146 KB); the first contraction's operand is four ds_read_b32
// now — the same LDS bandwidth — and a point takes 21 / 38 / 55 / 71 KB: FOUR points per CU for NT <= 2, two for NT >= 3.
template <int NT>
struct __attribute__((aligned(16))) SbLds {
    float ksub[NT * 16 * SB_KS];  // [y][x] = K[sup_y][x]
    float b[SB_MAXROWS];          // nu(sup_y), 0 on the padding rows
    float dlo[256];
    float dhi[256];
    float red[8];
    float lb0[256];   // per centroid: the column-marginal lower bound of the divergence (0 when not in use)
    uint32_t perm[256];  // column slot -> centroid, ascending lb0
    uint32_t ub;      // bits of the smallest upper bound published so far (non-negative floats order as integers)
    uint32_t kmin;    // bits of the smallest K of the point's rows: max C over the point's couplings = neg_t_ln2 * log2(kmin)
    uint32_t sup[SB_MAXROWS];
    uint32_t np;
    uint32_t item;
    uint32_t next;    // the point's column queue: the next column slot of perm[] (taken by whichever wavefront has a free slot)
    uint8_t crank[256];  // the centroids' place in an order that keeps similar centroids together (ties of lb0 follow it)
};
#define SB_LB_SAFETY 0.999f
#define SB_LB_SLACK 1e-6f

__device__ __forceinline__ f32x4 sb_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float sb_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// reductions over the four lanes (g = 0..3) that share a centroid column: lanes c, c + 16, c + 32, c + 48
#if defined(RP_EMUL)
__device__ __forceinline__ float sb_sum4(float x) {
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}
__device__ __forceinline__ float sb_max4(float x) {
    x = fmaxf(x, __shfl_xor(x, 16, 64));
    x = fmaxf(x, __shfl_xor(x, 32, 64));
    return x;
}
#else
// v_permlane16_swap / v_permlane32_swap (vector ALU) instead of two ds_bpermute round trips through LDS: seven of these reductions stand
// between one iteration's last MFMA and the next one's first.  The same additions as the xor butterfly: (x_l + x_{l^16}) + (the pair l^32's).
__device__ __forceinline__ void sb_swap16(float x, float& a, float& b) {  // a: the even row of each row pair, b: the odd one
    const uint32_t w = __builtin_bit_cast(uint32_t, x);
    const auto r = __builtin_amdgcn_permlane16_swap(w, w, false, false);
    const uint32_t r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void sb_swap32(float x, float& a, float& b) {  // a: lanes 0-31's value, b: lanes 32-63's
    const uint32_t w = __builtin_bit_cast(uint32_t, x);
    const auto r = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    const uint32_t r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ float sb_sum4(float x) {
    float a, b;
    sb_swap16(x, a, b);
    x = a + b;
    sb_swap32(x, a, b);
    return a + b;
}
__device__ __forceinline__ float sb_max4(float x) {
    float a, b;
    sb_swap16(x, a, b);
    x = fmaxf(a, b);
    sb_swap32(x, a, b);
    return fmaxf(a, b);
}
#endif

// One Gauss-Seidel iteration for a block of 16 centroid columns: U <- mu ./ (K V), then V <- nu ./ (K^T U).
// uo: U of the previous iteration in C/D layout (lane (c, g), register r of tile xt = U[16 xt + 4g + r][column c]);
// v likewise over the point's rows.  drow = this lane's centroid density row (256 floats, zero off the support).
// The A operands of tile xt + 1 (two ds_read_b128 per y tile) and its densities are requested before tile xt's MFMAs: with two
// wavefronts per SIMD the LDS round trip is otherwise exposed twice per tile.  The K V chain of a tile is split over two
// accumulators (its eight MFMAs depend on each other otherwise).  COST: the K .* C contraction of the cost (sb_cost) rides on
// the second contraction's operands — the same K tile, the same fresh U — and returns the cost of the iterate this call produces.
// LIP: also returns dlt = sum_x |mu_x - u_x (K v)_x| + sum_y |nu_y - v_y (K^T u')_y| with the OLD u, v: the L1 distance the coupling
// P = diag(u) K diag(v) moves in this iteration (first by the lhs update, then by the rhs update), so |cost' - cost| <= max C * dlt.
// LIP == 2 (the dual exit, see k_sinkhorn_bound): also eprev = the first of the two sums = the L1 error of the row marginals of the
// coupling this call STARTS from, and dual = the value of the Kantorovich pair  f(x) = T ln u'_x,  g(y) = -T ln (K^T u')_y  of the
// iterate this call PRODUCES (feasible for the unregularised problem: u'_x K_xy is one term of (K^T u')_y), with ln u' taken from the
// float's bits — (bits - bits(1)) 2^-23 <= log2: a lower bound, which f may be — and (K^T u')_y from the second contraction.
template <int NT, bool COST, int LIP>
__device__ __forceinline__ void sb_iterate(f32x4 (&uo)[16], f32x4 (&v)[NT], const float* __restrict__ drow, const SbLds<NT>& L, uint32_t c,
                                           uint32_t g, float& err, float& sumu, float& umax, float& sumv, float& vmax, float neg_t_ln2,
                                           float& cost, float& dlt, float& eprev, float& dual, float& fsum) {
    f32x4 racc[NT], w[NT];
#pragma unroll
    for (int yt = 0; yt < NT; ++yt) {
        racc[yt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        w[yt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    float eu = 0.0f, su = 0.0f, mu_ = 0.0f, dl = 0.0f, dlx = 0.0f, fd = 0.0f;
    const float* kq = &L.ksub[(4 * g) * SB_KS + c];  // first contraction: A[x = 16 xt + c][k -> y = 16 yt + 4 g + r], r = the MFMA step
    const float* ks = &L.ksub[c * SB_KS + 4 * g];    // second: A[y = 16 yt + c][k -> x = 16 xt + 4 g + r]
    f32x4 aS[NT], aR[NT];
    // the centroid's densities stream from L2 (a 256 KB table, 16 rows per wavefront: no L1 reuse): SB_MU_AHEAD tiles in flight
    f32x4 mq[SB_MU_AHEAD];
#pragma unroll
    for (int a = 0; a < SB_MU_AHEAD; ++a) mq[a] = *reinterpret_cast<const f32x4*>(drow + a * 16 + 4 * g);
#pragma unroll
    for (int yt = 0; yt < NT; ++yt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) aS[yt][r] = kq[(yt * 16 + r) * SB_KS];
        aR[yt] = *reinterpret_cast<const f32x4*>(ks + yt * 16 * SB_KS);
    }
#pragma unroll
    for (int xt = 0; xt < 16; ++xt) {
        f32x4 nS[NT], nR[NT];
        const f32x4 mcur = mq[xt % SB_MU_AHEAD];
        if (xt + SB_MU_AHEAD < 16) mq[xt % SB_MU_AHEAD] = *reinterpret_cast<const f32x4*>(drow + (xt + SB_MU_AHEAD) * 16 + 4 * g);
        if (xt + 1 < 16) {  // the A operands one tile ahead
#pragma unroll
            for (int yt = 0; yt < NT; ++yt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) nS[yt][r] = kq[(yt * 16 + r) * SB_KS + (xt + 1) * 16];
                nR[yt] = *reinterpret_cast<const f32x4*>(ks + yt * 16 * SB_KS + (xt + 1) * 16);
            }
        }
        // VALU and MFMA work of a tile are kept in separate bursts (sched_barrier): a VALU instruction issued between two MFMAs of an
        // accumulate chain costs tens of cycles of the matrix pipe (MI355X_MICROARCH: +43 per extra issue state)
        f32x4 kc[NT];
        if (COST) {
#pragma unroll
            for (int yt = 0; yt < NT; ++yt)
#pragma unroll
                for (int r = 0; r < 4; ++r) kc[yt][r] = aR[yt][r] * (neg_t_ln2 * __builtin_amdgcn_logf(aR[yt][r]));
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int yt = 0; yt < NT; ++yt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if ((yt * 4 + r) & 1) s1 = sb_mfma(aS[yt][r], v[yt][r], s1);
                else s0 = sb_mfma(aS[yt][r], v[yt][r], s0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sx = fmaxf(s0[r] + s1[r], 1e-37f);
            const float u0 = mcur[r] * sb_rcp(sx);
            const float du = fabsf(u0 - uo[xt][r]);
            if (LIP) dlx += sx * du;  // |mu_x - u_x (K v)_x| = (K v)_x |u'_x - u_x|   (u' (K v) = mu up to the reciprocal's rounding)
            if (LIP == 2) fd = fmaf(mcur[r], (float)((int)__float_as_uint(u0) - 0x3f800000), fd);  // mu_x * 2^23 * (a lower bound of log2 u'_x)
            eu += du;
            su += u0;
            mu_ = fmaxf(mu_, u0);
            uo[xt][r] = u0;
        }
        // the tile's statistics are due HERE: left to itself the compiler sinks them behind the last tile, where they need the old u, the new
        // u, (K v) and mu of all sixteen tiles at once (256 registers) and spills inside the loop
        asm volatile("" : "+v"(eu), "+v"(su), "+v"(mu_), "+v"(dlx), "+v"(fd));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int yt = 0; yt < NT; ++yt) {
                racc[yt] = sb_mfma(aR[yt][r], uo[xt][r], racc[yt]);
                if (COST) w[yt] = sb_mfma(kc[yt][r], uo[xt][r], w[yt]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the tiles apart: hoisting further tiles' operands costs more registers than the wave has
        if (xt + 1 < 16) {
#pragma unroll
            for (int yt = 0; yt < NT; ++yt) {
                aS[yt] = nS[yt];
                aR[yt] = nR[yt];
            }
        }
    }
    float ev = 0.0f, sv = 0.0f, mv = 0.0f, part = 0.0f, gd = 0.0f;
#pragma unroll
    for (int yt = 0; yt < NT; ++yt) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(&L.b[yt * 16 + 4 * g]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ry = fmaxf(racc[yt][r], 1e-37f);
            const float vn = bq[r] * sb_rcp(ry);
            const float dv = fabsf(vn - v[yt][r]);
            if (LIP) dl += ry * dv;
            if (LIP == 2) gd = fmaf(bq[r], __builtin_amdgcn_logf(ry), gd);  // nu_y log2 (K^T u')_y  (padding rows: 0 * finite)
            ev += dv;
            sv += vn;
            mv = fmaxf(mv, vn);
            v[yt][r] = vn;
            if (COST) part += vn * w[yt][r];
        }
    }
    err = sb_sum4(eu) + sb_sum4(ev);
    sumu = sb_sum4(su);
    sumv = sb_sum4(sv);
    umax = sb_max4(mu_);
    vmax = sb_max4(mv);
    cost = COST ? sb_sum4(part) : 0.0f;
    dlt = LIP ? sb_sum4(dl + dlx) : 0.0f;
    eprev = LIP == 2 ? sb_sum4(dlx) : 0.0f;
    // T ln 2 * (sum_x mu_x log2 u'_x - sum_y nu_y log2 (K^T u')_y), less SB_DUAL_SLACK for the roundings of (K^T u')_y (an MFMA
    // chain: <= 256 * 2^-24 relative = 2.2e-5 in log2), of v_log_f32 and of the two sums (<= 1e-4 in log2 at |log2| <= 40)
    fsum = LIP == 2 ? sb_sum4(fd) * 1.1920929e-7f : 0.0f;  // sum_x mu_x log2 u'_x (from below)
    dual = LIP == 2 ? (-neg_t_ln2) * ((fsum - sb_sum4(gd)) - SB_DUAL_SLACK) : 0.0f;
}

// The c-transform of f for the dual exit: sum_y nu_y log2 max_x K[y][x] u_x — the LARGEST g that keeps f + g <= C (g = -T ln of the
// maximum; sb_iterate's g uses the sum over x instead, which the second contraction delivers for free).  A max-product over the point's
// rows and all 256 x on the vector ALU: about one iteration's time for the sixteen columns of a wavefront, so it runs every
// SB_TIGHT_EVERY-th iteration only.  u: the iterate sb_iterate just produced (C/D layout: lane (c, g) holds x = 16 xt + 4 g + r).
template <int NT>
__device__ __forceinline__ float sb_dual_ctransform(const f32x4 (&uo)[16], const SbLds<NT>& L, uint32_t g, uint32_t np) {
    float gd = 0.0f;
    const float* ks = &L.ksub[4 * g];
    for (uint32_t y = 0; y < np; ++y) {  // (np is uniform over the workgroup)
        // products are >= 0 (or not a number): their order is their bit patterns' as unsigned integers — two packed multiplies and two
        // v_max3_u32 per tile (the float maximum canonicalises its operands: 115 instructions per row where these are 64); a NaN's
        // pattern is above every number's, reaches the logarithm and keeps the column
        uint32_t m0 = 0u, m1 = 0u;
#pragma unroll
        for (int xt = 0; xt < 16; ++xt) {
            const f32x4 k = *reinterpret_cast<const f32x4*>(ks + y * SB_KS + xt * 16);  // the same address in the 16 lanes of a g: a broadcast
            const f32x2 pa = f32x2{k[0], k[1]} * f32x2{uo[xt][0], uo[xt][1]};
            const f32x2 pb = f32x2{k[2], k[3]} * f32x2{uo[xt][2], uo[xt][3]};
            m0 = max(max(m0, __float_as_uint(pa[0])), __float_as_uint(pb[0]));
            m1 = max(max(m1, __float_as_uint(pa[1])), __float_as_uint(pb[1]));
        }
        const float mx = sb_max4(__uint_as_float(max(m0, m1)));
        gd = fmaf(L.b[y], __builtin_amdgcn_logf(fmaxf(mx, 1e-37f)), gd);
    }
    return gd;  // (every lane of the column holds the whole sum: no reduction over g left)
}

// cost of the current iterate: sum_y v_y sum_x K[y][x] C[y][x] u_x, with C recovered from K (C = -T ln K)
template <int NT>
__device__ __forceinline__ float sb_cost(const f32x4 (&uo)[16], const f32x4 (&v)[NT], const SbLds<NT>& L, uint32_t c, uint32_t g,
                                         float neg_t_ln2) {
    f32x4 w[NT];
#pragma unroll
    for (int yt = 0; yt < NT; ++yt) w[yt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int xt = 0; xt < 16; ++xt) {
        const float* ks = &L.ksub[c * SB_KS + xt * 16 + 4 * g];
#pragma unroll
        for (int yt = 0; yt < NT; ++yt) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ks + yt * 16 * SB_KS);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float kc = a[r] * (neg_t_ln2 * __builtin_amdgcn_logf(a[r]));
                w[yt] = sb_mfma(kc, uo[xt][r], w[yt]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // as in sb_iterate: the tiles' operands are not hoisted over each other
    }
    float part = 0.0f;
#pragma unroll
    for (int yt = 0; yt < NT; ++yt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part += v[yt][r] * w[yt][r];
    return sb_sum4(part);
}

struct SbCol {  // per-column window state (replicated in the four lanes of the column)
    float wmin, wmax, nb_prev;
    float cref, acc;  // the last evaluated cost of this column and max C * the coupling's L1 travel since (prm.lip): cost in [cref - acc, cref + acc]
    bool hasref;
    int flatc;
    bool opened, done;
    bool complete;  // the stopping window was followed to its end (not dropped)
};
// Column order.  The rigorous bound  cost >= sum_y nu(y) min_{x in supp mu_j} C(x, y)  (the column sums of the coupling are nu
// after every rhs update, at any iteration count) costs n reads per centroid and needs no iteration.  The point's columns are queued
// in ascending bound (the near ones first: their upper bounds are published while the far ones wait), and a column whose bound exceeds
// an upper bound already published by a finished column is dropped at once — its interval is [bound, inf), it cannot be the argmin.

// pstats: [0] survivors, [1] points, [2] wavefront iterations, [3] cost passes, [4] MFMA instructions, [5] column iterations (live
// slots summed over the wavefront iterations: [5] / (16 [2]) is the useful share)          (striped like Metric::stats)
// A wavefront iterates sixteen column SLOTS (64 + 12 NT accumulator registers) and a slot whose column is done takes the point's next
// column from an LDS counter: a point's columns differ in length by a factor of six.  Until round 6 a wavefront took sixteen columns
// together and iterated until the slowest was done: 0.62 of the slot-iterations were live then, 0.75 now (what is left is each
// wavefront's own drain at the end of the point: measured, neither a descending queue nor handing the last columns to the
// wavefronts of highest priority changes it — DESIGN 4d).
// NT <= 2: TWO wavefronts per point and four points per CU; NT >= 3: four wavefronts, two points per CU (SbLds).  With the dual exit the far
// columns leave after a few iterations and a point's time is its longest column's (the near ones: up to the cap of 128 iterations): on
// four wavefronts a point had 102 iterations per wavefront and 0.57 of its slots live; fewer slots per point and more points per CU
// fill them (one point's gather / drain is covered by the others' iterations).
template <int NT, int LIP>  // LIP = prm.lip, at compile time (one pair of inlined iterations in the loop, not two: registers)
__global__ __launch_bounds__((NT <= 2 ? 128 : 256), 2) void k_sinkhorn_bound(Points P, CentroidSet cs, uint32_t K, uint32_t bins, SbParams prm,
                                                                   const uint32_t* list, uint32_t count, unsigned int* cursor,
                                                                   unsigned long long* mask_out, float* dbg_lo, float* dbg_hi,
                                                                   unsigned long long* pstats, const float* ub0, const uint8_t* crank) {
    __shared__ SbLds<NT> L;
    constexpr uint32_t THREADS = NT <= 2 ? 128u : 256u, WAVES = THREADS / 64u;
    for (uint32_t q = threadIdx.x; q < 256u; q += THREADS) L.crank[q] = crank ? crank[q] : (uint8_t)q;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t c = lane & 15u, g = lane >> 4;
    unsigned long long my_cb_iters = 0, my_cost_passes = 0, my_col_iters = 0;
    for (;;) {
        __syncthreads();  // the previous point's LDS is no longer read
        if (tid == 0) {
            L.item = atomicAdd(cursor, 1u);
            L.next = 0;
            L.kmin = 0x3f800000u;
        }
        __syncthreads();
        const uint32_t item = L.item;
        if (item >= count) break;
        const uint64_t i = list[item];
        // ---- the point: support (ascending bins), densities, K rows in both orientations
        if (wave == 0) {
            const uint8_t* counts = P.counts + i * P.stride;
            const float fw = (float)P.weight[i];
            uint32_t base = 0;
            for (uint32_t q = 0; q * 64 < bins; ++q) {
                const uint32_t bb = q * 64 + lane;
                const uint32_t cc = bb < bins ? (uint32_t)counts[bb] : 0u;
                const bool has = cc > 0;
                const unsigned long long m = __ballot(has);
                if (has) {
                    const uint32_t r = base + __popcll(m & ((1ull << lane) - 1ull));
                    if (r < SB_MAXROWS) {
                        L.sup[r] = bb;
                        L.b[r] = (float)cc / fw;  // Bins::density (bins.rs:58-60)
                    }
                }
                base += __popcll(m);
            }
            if (lane == 0) L.np = base;
        }
        __syncthreads();
        const uint32_t np = L.np;
        if (np == 0 || np > NT * 16) {  // not a point of this list's class: leave it to the exact kernel, unpruned
            if (tid < 4) mask_out[i * 4 + tid] = ~0ull;
            continue;
        }
        if (tid >= np && tid < NT * 16) {
            L.sup[tid] = L.sup[0];
            L.b[tid] = 0.0f;
        }
        __syncthreads();
        float kmin = 1.0f;
#pragma unroll 8
        for (uint32_t e = tid; e < NT * 16 * 256; e += THREADS) {
            const uint32_t y = e >> 8, x = e & 255u;
            const float k = prm.Kmat[L.sup[y] * 256u + x];
            L.ksub[y * SB_KS + x] = k;
            kmin = fminf(kmin, k);
        }
        if (LIP) {  // (K in [0, 1]: non-negative floats order as their bits)
            for (int o = 32; o > 0; o >>= 1) kmin = fminf(kmin, __shfl_xor(kmin, o, 64));
            if (lane == 0) atomicMin(&L.kmin, __float_as_uint(kmin));
        }
        const float sp = P.self[i];
        for (uint32_t q = tid; q < 256u; q += THREADS) {
            float key = __builtin_inff();
            if (q < K) {
                key = 0.0f;
                if (prm.use_lb0) {
                    float acc = 0.0f;
                    for (uint32_t y = 0; y < np; ++y) acc += L.b[y] * cs.mincT[(size_t)L.sup[y] * 256u + q];
                    key = rp_maxf((acc * SB_LB_SAFETY - SB_LB_SLACK) - 0.5f * cs.self[q] - 0.5f * sp, 0.0f);
                    if (!(key == key)) key = 0.0f;
                }
            }
            L.lb0[q] = key;
        }
        // ub0: an upper bound of the point's smallest distance known before the pass (the exact distance to SOME centroid:
        // the Elkan assignment for lookup, the k-means++ potential for init_bounds); columns whose bound exceeds it never start
        if (tid == 0) {
            const float u = ub0 ? ub0[i] : __builtin_inff();
            L.ub = (u == u && u >= 0.0f) ? __float_as_uint(u) : 0x7f800000u;
        }
        __syncthreads();
        for (uint32_t q = tid; q < 256u; q += THREADS) {  // rank sort (256 keys, LDS broadcasts): ties follow the similarity order of the centroids
            const float mine = L.lb0[q];
            const uint32_t myr = L.crank[q];
            uint32_t rank = 0;
            for (uint32_t k = 0; k < 256; ++k) {
                const float o = L.lb0[k];
                rank += (o < mine || (o == mine && (uint32_t)L.crank[k] < myr)) ? 1u : 0u;
            }
            L.perm[rank] = q;
        }
        __syncthreads();
        // ---- columns in ascending bound (the near ones first: their upper bounds are published while the far ones wait).  A wavefront
        // iterates 16 column SLOTS; a slot whose column is done takes the point's next column from the queue at once (round 6: until
        // round 5 a wavefront took sixteen columns together and iterated until the slowest of them was done — 38 % of the
        // block-iterations ran columns that were finished).  Everything a column owns is per lane (the four lanes c, c+16, c+32, c+48).
        {
            f32x4 uo[16], v[NT];
            uint32_t j0 = 256u, jc = 0u, mj = 0u, tcol = 0u;
            float dlb0 = 0.0f, sc = 0.0f;
            bool valid = false, has = false;
            const float* drow = cs.densR;
            SbCol st;
            st.done = true;
            // takes the next live column for the slots whose `want` is set (wave uniform call, lane-varying want); columns that can be
            // settled without iterating (not a centroid, empty, or bound above a published upper bound) are written off on the way
            auto refill = [&](bool want) {
                bool fresh = false;
                for (;;) {  // the queue: a cheap loop (a pop, four LDS reads), repeated while a wanting slot popped a column that never starts
                    if (__ballot(want) == 0) break;
                    uint32_t idx = 256u;
                    if (want && g == 0) idx = atomicAdd(&L.next, 1u);
                    idx = (uint32_t)__shfl((int)idx, (int)c, 64);  // the column's four lanes follow lane c
                    if (want) {
                        if (idx >= 256u) {
                            want = false;  // the queue is empty: the slot stays idle (has == false)
                        } else {
                            j0 = L.perm[idx];
                            dlb0 = L.lb0[j0];
                            jc = j0 < K ? j0 : K - 1;
                            mj = cs.n[jc];
                            valid = j0 < K && mj > 0;
                            if (!valid || dlb0 > __uint_as_float(*(volatile uint32_t*)&L.ub)) {  // never starts: [bound, inf)
                                if (g == 0) {
                                    L.dlo[j0] = j0 < K ? (valid ? dlb0 : 0.0f) : __builtin_inff();
                                    L.dhi[j0] = __builtin_inff();
                                }
                            } else {
                                fresh = true;
                                want = false;
                            }
                        }
                    }
                }
                if (__ballot(fresh) == 0) return;
                if (fresh) {  // the start of a column, once per refill whatever the queue loop took
                    has = true;
                    drow = cs.densR + (size_t)jc * 256;
                    sc = cs.self[jc];
                    tcol = 0;
                    // Potential::uniform (phi.rs:34-39): exp(lhs) = 1/|supp mu| on the support, exp(rhs) = 1/|supp nu|
                    const float iu = 1.0f / (float)mj, iv = 1.0f / (float)np;
#pragma unroll
                    for (int xt = 0; xt < 16; ++xt) {
                        const f32x4 d0 = *reinterpret_cast<const f32x4*>(drow + xt * 16 + 4 * g);
#pragma unroll
                        for (int r = 0; r < 4; ++r) uo[xt][r] = d0[r] > 0.0f ? iu : 0.0f;
                    }
#pragma unroll
                    for (int yt = 0; yt < NT; ++yt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[yt][r] = (uint32_t)(yt * 16 + 4 * g + r) < np ? iv : 0.0f;
                    st.wmin = __builtin_inff();
                    st.wmax = -__builtin_inff();
                    st.nb_prev = SB_EPS23 * (8.0f + 0.37f * (float)np) + SB_EPS23 * 0.37f * (float)mj;
                    st.flatc = 0;
                    st.opened = false;
                    st.done = false;
                    st.complete = false;
                    st.cref = 0.0f;
                    st.acc = 0.0f;
                    st.hasref = false;
                }
            };
            // max C over the point's rows (an over-estimate of the column's own: every x, not only supp mu_j), a hair above the rounding of
            // the approximate log2; K = 0 (C / T beyond the float range) makes it infinite: every window iterate is evaluated, as without prm.lip
            const float cmaxp = LIP ? (prm.neg_t_ln2 * __builtin_amdgcn_logf(__uint_as_float(L.kmin))) * 1.0001f : 0.0f;
            // the divergence's lower end for a cost w (sinkhorn.rs:166-171 and the cost margin, as at the end of a column)
            auto lo_of = [&](float w) { return rp_maxf((w - (prm.dc_abs + prm.dc_rel * fabsf(w))) - 0.5f * sc - 0.5f * sp, 0.0f); };
#pragma unroll
            for (int xt = 0; xt < 16; ++xt) uo[xt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int yt = 0; yt < NT; ++yt) v[yt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            refill(true);
            for (;;) {
                if (__ballot(has && !st.done) == 0) break;
                const bool live = has && !st.done;
                const bool last = tcol + 1 == prm.iters;
                float err, sumu, umax, sumv, vmax, fused = 0.0f;
                // once a column of the wavefront is inside its stopping window every further iterate's cost is wanted: the cost contraction
                // then rides on the iteration itself; the first iterate of a window (not known in advance) takes the separate pass below
                // prm.lip: a column whose cost is known to lie in [cref - acc, cref + acc] and whose divergence is above a published upper
                // bound even at cref - acc needs no evaluation for this iterate — it cannot be the argmin whatever the exact value, and
                // its interval stays an interval (only wider).  The far columns, whose windows are the long ones, are of that kind;
                // acc grows by max C * dlt per iteration until the test fails and the next evaluation resets it.
                float dlt = 0.0f;
                const float ubn = __uint_as_float(*(volatile uint32_t*)&L.ub);
                bool near = !(LIP && st.hasref && lo_of(st.cref - st.acc) > ubn);
                // (LIP: evaluations are rare — 2 % of the iterations on the flop layer — and always take the separate pass: the fused
                // form's accumulators are what the register allocation of the whole loop is sized by)
                const bool with_cost = !LIP && __ballot(live && st.opened && near) != 0;
                float eprev = 0.0f, dual = 0.0f, fsum = 0.0f;
                if (with_cost) sb_iterate<NT, true, LIP>(uo, v, drow, L, c, g, err, sumu, umax, sumv, vmax, prm.neg_t_ln2, fused, dlt, eprev, dual, fsum);
                else sb_iterate<NT, false, LIP>(uo, v, drow, L, c, g, err, sumu, umax, sumv, vmax, prm.neg_t_ln2, fused, dlt, eprev, dual, fsum);
                if (LIP == 2 && prm.tight && prm.use_lb0 && (my_cb_iters % prm.tight) == prm.tight - 1u) {  // wave uniform
                    const float gdt = sb_dual_ctransform<NT>(uo, L, g, np);
                    const float dt = (-prm.neg_t_ln2) * ((fsum - gdt) - SB_DUAL_SLACK);
                    dual = fmaxf(dual, dt);  // (both are lower bounds; not a number: the looser one stays)
                }
                if (LIP) {
                    st.acc += cmaxp * dlt;
                    near = !(st.hasref && lo_of(st.cref - st.acc) > ubn);
                }
                my_cb_iters += 1;
                const float ln2 = 0.6931472f;
                const float lu = fmaxf(__builtin_amdgcn_logf(umax) * ln2, 0.0f), lv = fmaxf(__builtin_amdgcn_logf(vmax) * ln2, 0.0f);
                const float nb = SB_EPS23 * (sumu * (lu + 4.0f) + 0.37f * (float)mj + sumv * (lv + 4.0f) + 0.37f * (float)np);
                const float noise = prm.kappa * (nb + st.nb_prev);
                st.nb_prev = nb;
                const bool possible = last || (err - noise < prm.tol * prm.rho);
                const bool certain = last || ((err + noise) * prm.rho < prm.tol);
                const bool flat = err <= prm.flat * SB_EPS23 * (sumu + sumv);
                const bool want = live && (possible || flat);
                if (with_cost || __ballot(want && near)) {  // wave uniform: the cost of this iterate, for all sixteen columns at once
                    const float cost = with_cost ? fused : sb_cost<NT>(uo, v, L, c, g, prm.neg_t_ln2);  // (one more contraction with K .* C)
                    my_cost_passes += 1;
                    if (live) {  // every live column takes the evaluation as its new reference
                        st.cref = cost;
                        st.acc = 0.0f;
                        st.hasref = true;
                    }
                    if (want) {
                        // a non-finite cost (under/overflow of the scaling form) poisons the window: the column survives
                        st.wmin = cost == cost ? fminf(st.wmin, cost) : -__builtin_inff();
                        st.wmax = cost == cost ? fmaxf(st.wmax, cost) : __builtin_inff();
                        st.opened = true;
                    }
                } else if (want) {  // (prm.lip, a far column) the iterate's cost by its bounds; not a number: poisoned, as above
                    const float wl = st.cref - st.acc, wh = st.cref + st.acc;
                    st.wmin = wl == wl ? fminf(st.wmin, wl) : -__builtin_inff();
                    st.wmax = wh == wh ? fmaxf(st.wmax, wh) : __builtin_inff();
                    st.opened = true;
                }
                st.flatc = flat ? st.flatc + 1 : 0;
                if (LIP == 2 && prm.use_lb0 && live && tcol >= 1) {  // (RP_SB_NO_LB0 follows every column to the end of its window)
                    // The dual exit (round 6).  Weak duality: the pair (f, g) of sb_iterate bounds the UNREGULARISED optimum OT_0(mu, nu)
                    // from below whatever the iterate is.  After every rhs update the coupling's column marginals are nu exactly and its
                    // row marginals a_s are within e_s of mu in L1, so cost_s = <P_s, C> >= OT_0(a_s, nu) >= OT_0(mu, nu) - max C e_s / 2;
                    // and e_s does not grow with s (each half step pushes both marginals through one stochastic kernel), so with eprev =
                    // e of the iterate this iteration started from:  cost_s >= dual - max C eprev / 2  for THIS iterate and every later one
                    // (tcol >= 1: the start's coupling has had its rhs update).  The iterates at which the reference could have stopped
                    // earlier are the window's (wmin).  A column whose divergence at that lower end exceeds a published upper bound
                    // cannot be the argmin at any stopping time: its rigorous bound dlb0 is raised and the test below drops it.
                    // (1 % and 5e-5 on eprev: its own rounding, and the f32 iterates against the real-number trajectory's.)
                    const float lbc = dual - 0.5f * cmaxp * (1.01f * eprev + 5e-5f);
                    const float cmin = st.opened ? fminf(st.wmin, lbc) : lbc;
                    const float lbd = lo_of(cmin);
                    if (lbd > dlb0) dlb0 = lbd;  // (not a number: no change)
                }
                if (live) {
                    tcol += 1;
                    if (g == 0) my_col_iters += 1;
                    if (certain || st.flatc >= 2) {
                        st.done = true;
                        st.complete = true;
                        if (st.opened && g == 0) {  // publish this column's upper bound
                            const float ch = st.wmax + (prm.dc_abs + prm.dc_rel * fabsf(st.wmax));
                            const float hi = rp_maxf(ch - 0.5f * sc - 0.5f * sp, 0.0f);
                            if (hi == hi && hi < __builtin_inff()) atomicMin(&L.ub, __float_as_uint(hi));
                        }
                    }
                    // a column whose rigorous lower bound exceeds a published upper bound cannot be the argmin: stop iterating it
                    if (!st.done && dlb0 > __uint_as_float(*(volatile uint32_t*)&L.ub)) st.done = true;
                }
                const bool finished = has && st.done;
                if (__ballot(finished)) {
                    if (finished) {
                        // ---- interval of the divergence (sinkhorn.rs:166-171): the same three f32 operations, monotone in the cost
                        float lo = dlb0, hi = __builtin_inff();
                        if (st.opened && st.complete) {  // a column dropped inside its window keeps [dlb0, inf)
                            const float cl = st.wmin - (prm.dc_abs + prm.dc_rel * fabsf(st.wmin));
                            const float ch = st.wmax + (prm.dc_abs + prm.dc_rel * fabsf(st.wmax));
                            if (cl == cl && cl > -__builtin_inff()) lo = fmaxf(lo, rp_maxf(cl - 0.5f * sc - 0.5f * sp, 0.0f));
                            if (ch == ch) hi = rp_maxf(ch - 0.5f * sc - 0.5f * sp, 0.0f);
                        }
                        if (g == 0) {
                            L.dlo[j0] = lo;
                            L.dhi[j0] = hi;
                        }
                        has = false;
                    }
                    refill(finished);
                }
            }
        }
        __syncthreads();
        // ---- survivors: every centroid whose lower bound does not exceed the smallest upper bound
        {
            float ub = __builtin_inff();
            for (uint32_t q = tid; q < K; q += THREADS) ub = fminf(ub, L.dhi[q]);
            for (int o = 32; o > 0; o >>= 1) ub = fminf(ub, __shfl_xor(ub, o, 64));
            if (lane == 0) L.red[wave] = ub;
        }
        __syncthreads();
        {
            // (L.ub: the published upper bounds and ub0, the exact distance to SOME centroid — a column dropped on the way keeps a lower
            // bound that was above L.ub then, not necessarily above the smallest dhi)
            float ub = __uint_as_float(L.ub);
            for (uint32_t w = 0; w < WAVES; ++w) ub = fminf(ub, L.red[w]);
            for (uint32_t q = tid; q < 256u; q += THREADS) {  // (q >> 6 is wave uniform)
                const bool keep = q < K && !(L.dlo[q] > ub);
                const unsigned long long m = __ballot(keep);
                if (lane == 0) {
                    mask_out[i * 4 + (q >> 6)] = m;
                    atomicAdd(pstats + (size_t)(blockIdx.x % KM_STAT_STRIPES) * STAT_STRIDE + 0, (unsigned long long)__popcll(m));
                }
                if (dbg_lo && q < K) {
                    dbg_lo[i * K + q] = L.dlo[q];
                    dbg_hi[i * K + q] = L.dhi[q];
                }
            }
        }
        if (tid == 0) atomicAdd(pstats + (size_t)(blockIdx.x % KM_STAT_STRIPES) * STAT_STRIDE + 1, 1ull);
    }
    for (int o = 8; o > 0; o >>= 1) my_col_iters += __shfl_xor(my_col_iters, o, 64);  // the 16 columns of lanes g == 0
    if (lane == 0) {
        atomicAdd(pstats + (size_t)(blockIdx.x % KM_STAT_STRIPES) * STAT_STRIDE + 5, my_col_iters);
        atomicAdd(pstats + (size_t)(blockIdx.x % KM_STAT_STRIPES) * STAT_STRIDE + 2, my_cb_iters);
        atomicAdd(pstats + (size_t)(blockIdx.x % KM_STAT_STRIPES) * STAT_STRIDE + 3, my_cost_passes);
        // per block iteration 16 x-tiles x NT y-tiles x 4 k-steps in each of the two contractions; a cost pass is one more
        atomicAdd(pstats + (size_t)(blockIdx.x % KM_STAT_STRIPES) * STAT_STRIDE + 4,
                  (my_cb_iters * 2ull + my_cost_passes) * (unsigned long long)(16 * NT * 4));
    }
}

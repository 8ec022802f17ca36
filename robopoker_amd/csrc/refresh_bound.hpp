// refresh_bound.hpp — Elkan's stale-bound refresh (elkan.rs:113-117) decided by an INTERVAL (round 6).  Included by lloyd_kernels.hpp in
// the contract namespace only, after sinkhorn_bound.hpp / kpp_bound.hpp whose window logic, noise model and margins (SbParams) it reuses.
//
// WHY.  From the second iteration on every bound is stale, so every point that passes the filter u > s(c(x)) has u = d(x, c(x)) solved
// afresh — one bit-faithful Sinkhorn solve per point per iteration, 6.3 s (contract) / 7.6 s (glibc) of the flop layer — and the only
// consumers of that value are comparisons: u > l[k], u > P[j][k] / 2 in the candidate loop (bounds.rs:57-61), u + drift > s(c) in the
// next iteration's filter (elkan.rs:159).  f32 add / max are monotone, so an interval [lo, hi] that contains the value the reference
// would compute decides every comparison whose other side lies outside it.  On a flop-layer slice (profiles/r06_refresh_study_*.json)
// 38 % of the refreshes of iteration 1, 82 % of iteration 3 and 98.8 % of iteration 9 have hi <= min_k max(l[k], P[j][k] / 2): the
// candidate loop would find nothing whatever u's last bits are.  Those points keep (lo, hi) instead of u; everything else — the points
// whose interval reaches a threshold, the first iteration, supports of more than 32 bins — goes through the bit-faithful solve as before.
//
// WHAT IS KEPT EXACT.  Assignments, centroids, drifts, sizes and the lower bounds of every centroid a point is not assigned to are the
// reference's bit for bit (checked per step against the oracle).  B.u is the upper end and B.ulo the lower end of an interval around the
// reference's Bounds::error while B.uiv is set; l[c(x)] holds the lower end (the reference sets it to u at the refresh and reads it only
// after the point has moved away, which a point does only in a step that begins with an exact refresh).
//   An interval lives for ONE step: the next step either refreshes the point again (its lower end passes the filter: the reference
// refreshes too, whatever the exact value) or first replaces the interval by the exact value — one bit-faithful solve against the
// PREVIOUS centroids, still in the other CentroidSet, plus the drift that Bounds::update added (k_refresh_pairs in exactify mode) —
// and only then applies the reference's filter to it.  No comparison is ever decided by a value that straddles it.
//
// THE ITERATION.  distance(point, centroid): mu = the point (n <= 32 bins), nu = the centroid (up to 256 bins).  Scaling domain,
// Gauss-Seidel, the point's side first (sinkhorn.rs:77-92):  a = mu ./ (K b),  b = nu ./ (K^T a),  K = exp(-C/T) from the bound's table.
// One wavefront per pair.  Lane l owns centroid bins 4l .. 4l+3: b there, and the K sub-matrix K[y][4l..4l+3] of all NR point rows in
// registers (NR x 4).  K^T a is lane-local (NR x 4 multiply-adds against a broadcast from LDS); K b needs a sum over the 64 lanes
// for each of the NR rows: the partials go through LDS ([NR][68] floats: conflict-free writes, b128 reads), lane (y, h) adds half a row.
// ~256 multiply-adds + ~70 other VALU instructions per iteration against ~5 000 for the bit-faithful iteration of the same pair.
#pragma once

#define RB_STRIDE 68u  // floats per row of the partial-sum tile: 16-byte aligned rows, 68 y mod 64 = 4 y: rows start 4 banks apart

template <uint32_t NR>
struct __attribute__((aligned(16))) RbLds {
    float part[NR * RB_STRIDE];  // K b partials: [y][lane]
    float a[32];                 // the point side's scaling vector, broadcast
    float mu[32];                // the point's densities, 0 past the support
    uint32_t yP[32];             // the point's support bins
    uint32_t item;
};

// rstats (striped like Metric::stats): [0] pairs examined, [1] settled by the interval, [2] pair-iterations, [3] cost passes,
// [4] pairs whose interval was remembered (same centroid content: no iteration)
// code[i]: 1 on entry (refresh wanted); left 1 -> 2 (the bit-faithful refresh is needed) or set to 0 (settled)
template <uint32_t NR>
__global__ __launch_bounds__(64) void k_refresh_interval(Points P, CentroidSet cs, uint32_t K, uint32_t bins, const float* Cm, SbParams prm, Bounds B,
                                                         const float* pairw, const uint32_t* list, const uint32_t* list_end, unsigned int* cursor,
                                                         uint8_t* code, unsigned long long* rstats, unsigned long long* mstats) {
    __shared__ RbLds<NR> L;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t have = *list_end;
    unsigned long long my_pairs = 0, my_settled = 0, my_iters = 0, my_costs = 0, my_memo = 0;
    for (;;) {
        __syncthreads();  // the previous pair's LDS is no longer read
        if (lane == 0) L.item = atomicAdd(cursor, 1u);
        __syncthreads();
        const uint32_t item = L.item;
        if (item >= have) break;
        const uint32_t ip = list[item];
        if (ip == 0xffffffffu) continue;  // bucket padding
        const uint64_t i = ip;
        if (code[i] != 1) continue;
        const uint32_t j = B.j[i];
        // ---- the point: support (ascending bins) and densities (Bins::density, bins.rs:58-60)
        uint32_t n = 0;
        {
            const uint8_t* counts = P.counts + i * P.stride;
            const float fw = (float)P.weight[i];
            if (lane < 32u) {
                L.yP[lane] = 0u;
                L.mu[lane] = 0.0f;
            }
            __syncthreads();
            for (uint32_t q = 0; q * 64 < bins; ++q) {
                const uint32_t bb = q * 64 + lane;
                const uint32_t cc = bb < bins ? (uint32_t)counts[bb] : 0u;
                const bool has = cc > 0;
                const unsigned long long msk = __ballot(has);
                if (has) {
                    const uint32_t rr = n + __popcll(msk & ((1ull << lane) - 1ull));
                    if (rr < NR) {
                        L.yP[rr] = bb;
                        L.mu[rr] = (float)cc / fw;
                    }
                }
                n += __popcll(msk);
            }
        }
        __syncthreads();
        const uint32_t m = cs.n[j];
        if (n == 0 || n > NR || m == 0) {  // not a pair of this kernel's class: the bit-faithful refresh takes it
            if (lane == 0) code[i] = 2;
            continue;
        }
        if (lane >= n && lane < 32u) L.yP[lane] = L.yP[0];  // padding rows: a real bin, zero mass
        __syncthreads();
        // ---- the threshold: the candidate loop fires for centroid k iff u > l[k] and u > P[j][k] / 2 (bounds.rs:57-61)
        float thr = __builtin_inff();
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t k = 4u * lane + r;
            if (k < K && k != j) thr = fminf(thr, fmaxf(B.lower[i * K + k], 0.5f * pairw[(size_t)j * K + k]));
        }
        thr = -kb_allmax<64>(-thr);
        // the same point against the same centroid CONTENT as when an interval was last computed (cver, like the remembered exact
        // refreshes): the value it contains has not changed, only the thresholds have
        if (B.im_j[i] == (uint8_t)j && B.im_ver[i] == B.cver[j] && B.im_ver[i] != 0u) {
            const float dlo = B.im_lo[i], dhi = B.im_hi[i];
            const bool settled = dhi <= thr;
            my_pairs += 1;
            my_memo += 1;
            if (lane == 0) {
                if (settled) {
                    B.u[i] = dhi;
                    B.ulo[i] = dlo;
                    B.uiv[i] = 1;
                    B.lower[i * K + j] = dlo;
                    B.stale[i] = 0;
                    code[i] = 0;
                    my_settled += 1;
                } else {
                    code[i] = 2;
                }
            }
            continue;
        }
        // ---- K sub-matrix in registers, the centroid's densities, Potential::uniform (phi.rs:34-39)
        float Kr[NR][4];
#pragma unroll
        for (uint32_t y = 0; y < NR; ++y) {
            const float4 kq = *reinterpret_cast<const float4*>(prm.Kmat + (size_t)L.yP[y] * 256u + 4u * lane);
            const bool ok = y < n;
            Kr[y][0] = ok ? kq.x : 0.0f;
            Kr[y][1] = ok ? kq.y : 0.0f;
            Kr[y][2] = ok ? kq.z : 0.0f;
            Kr[y][3] = ok ? kq.w : 0.0f;
        }
        float nu[4], b[4];
        {
            const float4 dq = *reinterpret_cast<const float4*>(cs.densR + (size_t)j * MAXB + 4u * lane);
            nu[0] = dq.x, nu[1] = dq.y, nu[2] = dq.z, nu[3] = dq.w;
            const float ib = 1.0f / (float)m;
#pragma unroll
            for (uint32_t r = 0; r < 4; ++r) b[r] = nu[r] > 0.0f ? ib : 0.0f;
        }
        // lane (y, h) = (lane & 31, lane >> 5) finishes row y of K b; lanes with h = 0 hold a_y
        const uint32_t yrow = lane & 31u, hsel = lane >> 5;
        const float mu_y = L.mu[yrow];
        float a_y = yrow < n ? 1.0f / (float)n : 0.0f;
        const float sp = P.self[i], sc = cs.self[j];
        const float fm = (float)m, fn = (float)n;
        float wmin = __builtin_inff(), wmax = -__builtin_inff(), nb_prev = SB_EPS23 * (8.0f + 0.37f * (fm + fn));
        int flatc = 0;
        bool opened = false, complete = false, done = false;
        uint32_t t = 0;
        for (; t < prm.iters && !done; ++t) {
            const bool last = t + 1 == prm.iters;
            // a <- mu ./ (K b): partials of the NR rows over this lane's four bins, summed over the lanes through LDS
#pragma unroll
            for (uint32_t y = 0; y < NR; ++y) {
                float p = Kr[y][0] * b[0];
                p = __builtin_fmaf(Kr[y][1], b[1], p);
                p = __builtin_fmaf(Kr[y][2], b[2], p);
                p = __builtin_fmaf(Kr[y][3], b[3], p);
                L.part[y * RB_STRIDE + lane] = p;
            }
            __syncthreads();
            float s0 = 0.0f, s1 = 0.0f;
            if (yrow < NR) {
                const float* row = &L.part[yrow * RB_STRIDE + 32u * hsel];
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(row + 4u * q);
                    s0 += v.x;
                    s1 += v.y;
                    s0 += v.z;
                    s1 += v.w;
                }
            }
            const float sh = s0 + s1;
            const float su = half_take<0>(sh) + half_take<1>(sh);  // both halves of the row, in every lane of the row
            const float an = mu_y * sb_rcp(fmaxf(su, 1e-37f));
            float e = hsel == 0 ? fabsf(an - a_y) : 0.0f;
            a_y = an;
            if (hsel == 0) L.a[yrow] = an;
            __syncthreads();
            // b <- nu ./ (K^T a) on the fresh a (Gauss-Seidel): lane-local
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
#pragma unroll
            for (uint32_t y = 0; y < NR; y += 4) {
                const float4 av = *reinterpret_cast<const float4*>(&L.a[y]);
                t0 = __builtin_fmaf(Kr[y][0], av.x, t0);
                t1 = __builtin_fmaf(Kr[y][1], av.x, t1);
                t2 = __builtin_fmaf(Kr[y][2], av.x, t2);
                t3 = __builtin_fmaf(Kr[y][3], av.x, t3);
                t0 = __builtin_fmaf(Kr[y + 1][0], av.y, t0);
                t1 = __builtin_fmaf(Kr[y + 1][1], av.y, t1);
                t2 = __builtin_fmaf(Kr[y + 1][2], av.y, t2);
                t3 = __builtin_fmaf(Kr[y + 1][3], av.y, t3);
                t0 = __builtin_fmaf(Kr[y + 2][0], av.z, t0);
                t1 = __builtin_fmaf(Kr[y + 2][1], av.z, t1);
                t2 = __builtin_fmaf(Kr[y + 2][2], av.z, t2);
                t3 = __builtin_fmaf(Kr[y + 2][3], av.z, t3);
                t0 = __builtin_fmaf(Kr[y + 3][0], av.w, t0);
                t1 = __builtin_fmaf(Kr[y + 3][1], av.w, t1);
                t2 = __builtin_fmaf(Kr[y + 3][2], av.w, t2);
                t3 = __builtin_fmaf(Kr[y + 3][3], av.w, t3);
            }
            const float tt[4] = {t0, t1, t2, t3};
            float sb_ = 0.0f, mb = 0.0f;
#pragma unroll
            for (uint32_t r = 0; r < 4; ++r) {
                const float bn = nu[r] * sb_rcp(fmaxf(tt[r], 1e-37f));
                e += fabsf(bn - b[r]);
                b[r] = bn;
                sb_ += bn;
                mb = fmaxf(mb, bn);
            }
            const float err = kb_allsum<64>(e);
            const float suv = kb_allsum<64>(sb_ + (hsel == 0 ? a_y : 0.0f));
            const float mx = kb_allmax<64>(fmaxf(mb, a_y));
            my_iters += 1;
            const float lmx = fmaxf(__builtin_amdgcn_logf(mx) * 0.6931472f, 0.0f);
            const float nb = SB_EPS23 * (suv * (lmx + 4.0f) + 0.37f * (fm + fn));
            const float noise = prm.kappa * (nb + nb_prev);
            nb_prev = nb;
            const bool possible = last || (err - noise < prm.tol * prm.rho);
            const bool certain = last || ((err + noise) * prm.rho < prm.tol);
            const bool flat = err <= prm.flat * SB_EPS23 * suv;
            // (every lane holds the same err / suv / mx — butterfly sums give both partners the same bits — so the flag is wave
            // uniform; read through an SGPR it becomes a real branch: left as a lane predicate the compiler if-converted the cost
            // pass into the loop body and every iteration issued its ~450 instructions with an empty exec mask)
            if (__builtin_amdgcn_readfirstlane((int)(possible || flat))) {  // the cost of this iterate, sum_y a_y sum_x K C b_x
                float part = 0.0f;
                const uint32_t xb = 4u * lane < bins ? 4u * lane : 0u;  // bins is a multiple of four here; past it b = 0
#pragma unroll
                for (uint32_t y0 = 0; y0 < NR; y0 += 8) {  // eight rows of C in flight at a time: the loads of all NR rows at once cost
                                                           // 4 NR registers on top of K's, and a wavefront per SIMD with them
                    float4 cq[8];
#pragma unroll
                    for (uint32_t y = 0; y < 8; ++y) cq[y] = *reinterpret_cast<const float4*>(Cm + (size_t)L.yP[y0 + y] * bins + xb);
#pragma unroll
                    for (uint32_t y = 0; y < 8; ++y) {
                        float q = (Kr[y0 + y][0] * cq[y].x) * b[0];
                        q = __builtin_fmaf(Kr[y0 + y][1] * cq[y].y, b[1], q);
                        q = __builtin_fmaf(Kr[y0 + y][2] * cq[y].z, b[2], q);
                        q = __builtin_fmaf(Kr[y0 + y][3] * cq[y].w, b[3], q);
                        part = __builtin_fmaf(L.a[y0 + y], q, part);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float cost = kb_allsum<64>(part);
                my_costs += 1;
                wmin = cost == cost ? fminf(wmin, cost) : -__builtin_inff();  // a non-finite cost: nothing is settled
                wmax = cost == cost ? fmaxf(wmax, cost) : __builtin_inff();
                opened = true;
                // an iterate of the window already above the threshold: the exact value is needed whatever the rest of the window holds
                const float ch = wmax + (prm.dc_abs + prm.dc_rel * fabsf(wmax));
                const float dh = rp_maxf(ch - 0.5f * sp - 0.5f * sc, 0.0f);
                if (!(dh <= thr)) done = true;
            }
            flatc = flat ? flatc + 1 : 0;
            if (!done && (certain || flatc >= 2)) {
                done = true;
                complete = true;
            }
        }
        bool settled = false;
        float dlo = 0.0f, dhi = 0.0f;
        if (opened && complete) {
            const float cl = wmin - (prm.dc_abs + prm.dc_rel * fabsf(wmin));
            const float ch = wmax + (prm.dc_abs + prm.dc_rel * fabsf(wmax));
            dlo = rp_maxf(cl - 0.5f * sp - 0.5f * sc, 0.0f);
            dhi = rp_maxf(ch - 0.5f * sp - 0.5f * sc, 0.0f);
            settled = cl == cl && ch == ch && cl > -__builtin_inff() && ch < __builtin_inff() && dhi <= thr;
        }
        my_pairs += 1;
        if (lane == 0) {
            if (settled) {
                B.u[i] = dhi;
                B.ulo[i] = dlo;
                B.uiv[i] = 1;
                B.lower[i * K + j] = dlo;  // Bounds::refresh sets l[j] = u; read again only after the point has left j (header)
                B.stale[i] = 0;
                code[i] = 0;
                my_settled += 1;
            } else {
                code[i] = 2;
            }
            if (opened && complete && dlo == dlo && dhi == dhi && dhi < __builtin_inff()) {  // remembered whether it settled or not
                B.im_lo[i] = dlo;
                B.im_hi[i] = dhi;
                B.im_j[i] = (uint8_t)j;
                B.im_ver[i] = B.cver[j];
            }
        }
    }
    if (lane == 0 && my_pairs) {
        unsigned long long* s = rstats + (size_t)(blockIdx.x % KM_STAT_STRIPES) * STAT_STRIDE;
        atomicAdd(s + 0, my_pairs);
        atomicAdd(s + 1, my_settled);
        atomicAdd(s + 2, my_iters);
        atomicAdd(s + 3, my_costs);
        atomicAdd(s + 4, my_memo);
        // a settled refresh is a distance the reference evaluates at this point: counted beside the evaluated and the remembered ones
        atomicAdd(mstats + (size_t)(blockIdx.x % KM_STAT_STRIPES) * STAT_STRIDE + 5, my_settled);
    }
}

// The start of a step in interval mode (replaces k_refresh_memo): what each stale point with <= PAIR_ROWS bins needs.
//   code 0  nothing: the filter rejects it on an exact u (the reference skips it too), or its refresh is remembered (done here)
//   code 1  a refresh against the current centroids (interval first)
//   code 3  u is an interval whose lower end does not pass the filter: the exact value first (k_refresh_pairs, exactify mode)
// An interval-valued u whose exact value is remembered (the centroid's content has not changed since an exact solve) becomes exact here:
// the refresh it stands for returned memo_d, and Bounds::update added the last drift.
__global__ __launch_bounds__(256) void k_rb_prepare(Bounds B, const uint8_t* nsup, const float* mid, const float* drift_prev, uint64_t N, uint32_t K,
                                                    uint8_t* code, unsigned long long* mstats) {
    unsigned long long hits = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (uint64_t)gridDim.x * 256) {
        uint8_t c = 0;
        if (B.stale[i] && nsup[i] <= PAIR_ROWS) {
            const uint32_t j = B.j[i];
            const bool memo = memo_valid(B, i, j);
            if (B.uiv[i] && memo) {
                const float d = B.memo_d[i], dp = drift_prev[j];
                B.u[i] = d + dp;
                B.lower[i * K + j] = rp_maxf(d - dp, 0.0f);
                B.uiv[i] = 0;
            }
            if (!B.uiv[i]) {
                if (B.u[i] > mid[j]) {
                    if (memo) {  // Bounds::refresh without the solve
                        const float d = B.memo_d[i];
                        B.u[i] = d;
                        B.lower[i * K + j] = d;
                        B.stale[i] = 0;
                        hits += 1;
                    } else {
                        c = 1;
                    }
                }
            } else {
                c = B.ulo[i] > mid[j] ? 1 : 3;
            }
        } else if (B.stale[i] && B.uiv[i]) {
            B.uiv[i] = 0;  // (never set for these points: the interval kernel does not take them)
        }
        code[i] = c;
    }
    for (int o = 32; o > 0; o >>= 1) hits += __shfl_xor(hits, o, 64);
    if ((threadIdx.x & 63u) == 0 && hits) atomicAdd(mstats + (size_t)(blockIdx.x % KM_STAT_STRIPES) * STAT_STRIDE + 3, hits);
}

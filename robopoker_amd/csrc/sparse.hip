// sparse.hip — rp_profile_*: the update half of Solver::step on a row-addressed table in HBM (include/rp_mi355x.h,
// "sparse profile").  Reference: crates/mccfr/src/solver/solver.rs:96-105 (step), :143-192 (update_regret /
// update_weight / update_payoff / update_visits), regret/*.rs, policy/*.rs; oracle: ora_profile_* in
// oracle/rp_oracle_mccfr.c.
//
// MI355X mapping.  A table row is 16*A contiguous bytes {regret[A], weight[A], payoff[A], visits[A]}: the touches of
// a row read and write one contiguous span, so the HBM traffic of a batch is (rows touched) x 32*A bytes plus the
// Decisions themselves, independent of the table size.  A batch is brought into per-row, batch-ordered segments by a
// stable radix sort of (row, position) pairs (the library's own LSD radix sort and scans: csrc/sortscan.hpp) and a run-length
// encode; then a group of 16 lanes owns one row — lane a owns action a's four cells, so a row is loaded and stored
// with coalesced dwords — and walks the row's touches in order.  Four rows per wavefront.
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "sortscan.hpp"
#include "mccfr_kernels.hpp"

namespace rp {

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define GROUP 16u  // lanes per row (max_actions <= 16)
#define PF 8       // touches fetched ahead of the sequential chain
#define HOT_TOUCHES 128u  // ordered mode: rows with more touches than this go to k_apply_hot
#define HOT_CAP 4096  // rows per batch folded by k_hot_fold; further hot rows fold (serially) in k_seg_fold

struct SparseParams {
    float* tab;           // [n_rows][4A]
    uint32_t A;
    int R, W;
    float tf;             // (float)epoch
    float pow15, pow05;   // powf(tf, 1.5), powf(tf, 0.5): DiscountedRegret (host: rp_libm_glibc.h)
    float floor_r;
    float dr, dw;         // composed discounts
};
struct DevBatch {
    const uint32_t* row;
    const uint8_t* nact;
    const uint16_t* expanded;
    const float* regret;
    const float* policy;
    const float* payoff;
};
struct Segments {
    const uint32_t* rows;     // [n_segs] distinct rows ascending
    const uint32_t* counts;   // [n_segs]
    const uint32_t* offsets;  // [n_segs] exclusive scan of counts
    const uint32_t* n_segs;   // device scalar
    const uint32_t* perm;     // [n] batch positions sorted by (row, position)
};

__global__ void k_iota(uint32_t* p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

// the batch in sorted (row, position) order: the chains below then stream their inputs instead of chasing `perm`
struct SortedBatch {
    float* regret;       // [n][A]
    float* policy;       // [n][A]
    float* payoff;       // [n]
    uint16_t* expanded;  // [n]
};
__global__ void k_permute(DevBatch b, const uint32_t* perm, uint32_t n, uint32_t A, SortedBatch o) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)n * A) return;
    const uint32_t t = (uint32_t)(e / A), a = (uint32_t)(e % A);
    const uint32_t idx = perm[t];
    o.regret[e] = b.regret[(size_t)idx * A + a];
    o.policy[e] = b.policy[(size_t)idx * A + a];
    if (a == 0) {
        o.payoff[t] = b.payoff[idx];
        o.expanded[t] = b.expanded[idx];
    }
}

// ORDERED: Solver::update_* per touch, in batch order (solver.rs:143-192)
struct TouchChunk {
    float dv[PF], sv[PF], pv[PF];
    uint32_t ev_mask;
};
__device__ __forceinline__ void fetch_chunk(TouchChunk& c, const SortedBatch& sb, uint32_t base, uint32_t m, uint32_t A,
                                            uint32_t a, bool mine) {
    c.ev_mask = 0;
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u) {
        const uint32_t t = base + (u < m ? u : 0u);
        c.dv[u] = mine ? sb.regret[(size_t)t * A + a] : 0.0f;
        c.sv[u] = mine ? sb.policy[(size_t)t * A + a] : 0.0f;
        c.pv[u] = sb.payoff[t];
        c.ev_mask |= (((uint32_t)sb.expanded[t] >> a) & 1u) << u;
    }
}
// the same chunk straight from the unsorted batch, through the sort permutation: the composed update reads every touch exactly
// once, so gathering it here saves the write and the re-read of a sorted copy (k_permute) — 1.7 ms of the 16 M-Decision NLHE step
__device__ __forceinline__ void fetch_chunk_gather(TouchChunk& c, const DevBatch& b, const uint32_t* perm, uint32_t base, uint32_t m,
                                                   uint32_t A, uint32_t a, bool mine) {
    c.ev_mask = 0;
    uint32_t t[PF];
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u) t[u] = perm[base + (u < m ? u : 0u)];
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u) {
        c.dv[u] = mine ? b.regret[(size_t)t[u] * A + a] : 0.0f;
        c.sv[u] = mine ? b.policy[(size_t)t[u] * A + a] : 0.0f;
        c.pv[u] = b.payoff[t[u]];
        c.ev_mask |= (((uint32_t)b.expanded[t[u]] >> a) & 1u) << u;
    }
}
__global__ __launch_bounds__(256) void k_apply_ordered(SparseParams p, DevBatch b, Segments sg, SortedBatch sb, uint32_t* hot,
                                                       uint32_t* n_hot, uint32_t hot_cap) {
    const uint32_t n_segs = *sg.n_segs;
    const uint32_t a = threadIdx.x % GROUP;
    const uint32_t A = p.A;
    for (uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) / GROUP; g < n_segs; g += gridDim.x * blockDim.x / GROUP) {
        const uint32_t off = sg.offsets[g], cnt = sg.counts[g];
        if (cnt > HOT_TOUCHES) {  // a hot row: its chain gets a wavefront of its own and LDS-staged inputs (k_apply_hot)
            uint32_t slot = hot_cap;
            if (a == 0) slot = atomicAdd(n_hot, 1u);
            slot = __shfl(slot, (int)(threadIdx.x & 48u), 64);
            if (slot < hot_cap) {
                if (a == 0) hot[slot] = g;
                continue;
            }
        }
        float* row = p.tab + (size_t)sg.rows[g] * 4u * A;
        const uint32_t nact = b.nact[sg.perm[off]];
        const bool mine = a < nact;
        float r = 0.0f, w = 0.0f, ev = 0.0f;
        uint32_t v = 0;
        if (mine) {
            r = row[a];
            w = row[A + a];
            ev = row[2 * A + a];
            v = reinterpret_cast<const uint32_t*>(row)[3 * A + a];
        }
        // the inputs of a touch do not depend on the chain: chunk k+1 is in flight while chunk k is applied
        TouchChunk cur, nxt;
        fetch_chunk(cur, sb, off, min((uint32_t)PF, cnt), A, a, mine);
        for (uint32_t t0 = 0; t0 < cnt; t0 += PF) {
            const uint32_t m = min((uint32_t)PF, cnt - t0);
            if (t0 + PF < cnt) fetch_chunk(nxt, sb, off + t0 + PF, min((uint32_t)PF, cnt - t0 - PF), A, a, mine);
            if (mine) {
#pragma unroll
                for (uint32_t u = 0; u < PF; ++u) {
                    if (u >= m) break;
                    if ((cur.ev_mask >> u) & 1u) r = d_regret_gain(p.R, r, cur.dv[u], p.tf, p.pow15, p.pow05, p.floor_r);
                    w = d_weight_learn(p.W, w, cur.sv[u], p.tf);
                    ev += (cur.pv[u] - ev) / (float)(v + 1u);
                    v += 1u;
                }
            }
            cur = nxt;
        }
        if (mine) {
            row[a] = r;
            row[A + a] = w;
            row[2 * A + a] = ev;
            reinterpret_cast<uint32_t*>(row)[3 * A + a] = v;
        }
    }
}

// ORDERED, hot rows: one wavefront per row.  All 64 lanes stage the next tile of HT touches (coalesced, from the
// sorted batch) in LDS while the chain lanes (lane a = action a) apply the current tile in order: the chain never
// waits for global memory, its length is the only cost (the reference's sequential semantics).
#define HT 64u
// Two wavefronts per hot row: wave 0 runs the regret and weight chains (lane a = action a), wave 1 the Welford payoff
// chain — two short instruction streams on two SIMDs instead of one long one (a lone wave is issue-latency bound).
template <bool SIGNED>  // SIGNED: the regret discount depends on the accumulator's sign (Discounted / Asymmetric)
__global__ __launch_bounds__(128) void k_apply_hot(SparseParams p, DevBatch b, Segments sg, SortedBatch sb, const uint32_t* hot,
                                                   const uint32_t* n_hot, uint32_t hot_cap) {
    __shared__ float treg[2][HT * GROUP], tpol[2][HT * GROUP], tpay[2][HT], trcp[2][HT], tden[2][HT];
    __shared__ uint32_t texp[2][HT];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, A = p.A;
    const uint32_t nh = min(*n_hot, hot_cap);
    // The schedules as one instruction stream: acc <- max(acc * d + delta, floor).  x * 1.0f is exact, so Summed /
    // Floored / Constant are the same stream with d = 1; the discount is per-epoch (by sign for Discounted /
    // Asymmetric: regret/{linear,discounted,asymmetric}.rs) and is hoisted out of the chain.
    const float lin = p.tf / (p.tf + 1.0f);
    float dpos = 1.0f, dneg = 1.0f, dzero = 1.0f;
    if (p.R == RP_REGRET_LINEAR) dpos = dneg = dzero = lin;
    else if (p.R == RP_REGRET_ASYMMETRIC) dneg = dzero = lin;
    else if (p.R == RP_REGRET_DISCOUNTED) {
        const float xp = p.pow15, xn = p.pow05, xz = p.tf / 1.0f;
        dpos = xp / (xp + 1.0f);
        dneg = xn / (xn + 1.0f);
        dzero = xz / (xz + 1.0f);
    }
    const float dw = p.W == RP_WEIGHT_EXPONENTIAL ? 0.9999f : 1.0f;
    for (uint32_t hi = blockIdx.x; hi < nh; hi += gridDim.x) {
        const uint32_t g = hot[hi];
        const uint32_t off = sg.offsets[g], cnt = sg.counts[g];
        float* row = p.tab + (size_t)sg.rows[g] * 4u * A;
        const uint32_t nact = b.nact[sg.perm[off]];
        const bool mine = lane < nact;
        float r = 0.0f, w = 0.0f, ev = 0.0f;
        uint32_t v = 0;
        if (mine) {
            r = row[lane];
            w = row[A + lane];
            ev = row[2 * A + lane];
            v = reinterpret_cast<const uint32_t*>(row)[3 * A + lane];
        }
        // every action of a row is visited together, so its cells share `visits`: the Welford divisors are known in
        // advance; divisor and reciprocal are prepared by the staging lanes (rp_div_by_recip: the exact quotient
        // without a division on the chain).  A row that breaks the invariant divides plainly.
        const uint32_t v0 = __builtin_amdgcn_readfirstlane(v);
        const bool shared_v = __ballot(mine && v != v0) == 0ull;
        // Staging is split so that global latency hides behind the chains: the loads of tile k+1 are issued into
        // registers before the chains over tile k, and committed to LDS after them.  128 threads share the work.
        float lr[GROUP / 2], lp[GROUP / 2], lpay = 0.0f;
        uint32_t lexp = 0;
        auto stage_load = [&](uint32_t t0) {
            const uint32_t m = min(HT, cnt - t0);
#pragma unroll
            for (uint32_t k = 0; k < GROUP / 2; ++k) {
                const uint32_t e = k * 128u + tid;
                const bool in = e < m * A;
                lr[k] = in ? sb.regret[(size_t)(off + t0) * A + e] : 0.0f;
                lp[k] = in ? sb.policy[(size_t)(off + t0) * A + e] : 0.0f;
            }
            if (tid < m) {
                lpay = sb.payoff[off + t0 + tid];
                lexp = sb.expanded[off + t0 + tid];
            }
        };
        auto stage_store = [&](uint32_t buf, uint32_t t0) {
            const uint32_t m = min(HT, cnt - t0);
#pragma unroll
            for (uint32_t k = 0; k < GROUP / 2; ++k) {
                const uint32_t e = k * 128u + tid;
                if (e < m * A) {
                    treg[buf][e] = lr[k];
                    tpol[buf][e] = p.W == RP_WEIGHT_LINEAR ? lp[k] * p.tf : (p.W == RP_WEIGHT_QUADRATIC ? lp[k] * p.tf * p.tf : lp[k]);
                }
            }
            if (tid < m) {
                const float den = (float)(v0 + t0 + tid + 1u);
                tpay[buf][tid] = lpay;
                texp[buf][tid] = lexp;
                tden[buf][tid] = den;
                trcp[buf][tid] = 1.0f / den;
            }
        };
        stage_load(0);
        stage_store(0, 0);
        __syncthreads();
        uint32_t buf = 0;
        for (uint32_t t0 = 0; t0 < cnt; t0 += HT) {
            const uint32_t m = min(HT, cnt - t0);
            const bool more = t0 + HT < cnt;
            if (more) stage_load(t0 + HT);
            if (mine && wave == 0) {  // regret + weight
                uint32_t u = 0;
                for (; u + PF <= m; u += PF) {
                    float dv[PF], sv[PF];
                    uint32_t em = 0;
#pragma unroll
                    for (uint32_t q = 0; q < PF; ++q) {
                        dv[q] = treg[buf][(u + q) * A + lane];
                        sv[q] = tpol[buf][(u + q) * A + lane];
                        em |= ((texp[buf][u + q] >> lane) & 1u) << q;
                    }
#pragma unroll
                    for (uint32_t q = 0; q < PF; ++q) {
                        const float d = SIGNED ? (r > 0.0f ? dpos : (r < 0.0f ? dneg : dzero)) : dpos;
                        const float rn = rp_maxf(r * d + dv[q], p.floor_r);
                        r = ((em >> q) & 1u) ? rn : r;
                        w = rp_maxf(w * dw + sv[q], RP_EPSILON);
                    }
                }
                for (; u < m; ++u) {
                    if ((texp[buf][u] >> lane) & 1u) {
                        const float d = SIGNED ? (r > 0.0f ? dpos : (r < 0.0f ? dneg : dzero)) : dpos;
                        r = rp_maxf(r * d + treg[buf][u * A + lane], p.floor_r);
                    }
                    w = rp_maxf(w * dw + tpol[buf][u * A + lane], RP_EPSILON);
                }
            }
            if (mine && wave == 1) {  // payoff + visits: ev += (x - ev) / (visits + 1)  (solver.rs:174-192)
                const float ev_in = ev;
                bool bad = !shared_v;  // the reciprocal quotient is exact unless a tiny numerator breaks its precondition
                uint32_t u = 0;
                for (; u + PF <= m; u += PF) {
                    float pv[PF], rc[PF], dn[PF];
#pragma unroll
                    for (uint32_t q = 0; q < PF; ++q) {
                        pv[q] = tpay[buf][u + q];
                        rc[q] = trcp[buf][u + q];
                        dn[q] = tden[buf][u + q];
                    }
#pragma unroll
                    for (uint32_t q = 0; q < PF; ++q) {
                        const float num = pv[q] - ev;
                        bad = bad || !rp_div_by_recip_ok(num);
                        ev += rp_div_by_recip(num, dn[q], rc[q]);
                    }
                }
                for (; u < m; ++u) {
                    const float num = tpay[buf][u] - ev;
                    bad = bad || !rp_div_by_recip_ok(num);
                    ev += rp_div_by_recip(num, tden[buf][u], trcp[buf][u]);
                }
                if (bad) {  // replay the tile with IEEE divisions (none observed)
                    ev = ev_in;
                    for (uint32_t q = 0; q < m; ++q) ev += (tpay[buf][q] - ev) / (float)(v + q + 1u);
                }
                v += m;
            }
            if (more) stage_store(buf ^ 1u, t0 + HT);
            __syncthreads();
            buf ^= 1u;
        }
        if (mine && wave == 0) {
            row[lane] = r;
            row[A + lane] = w;
        }
        if (mine && wave == 1) {
            row[2 * A + lane] = ev;
            reinterpret_cast<uint32_t*>(row)[3 * A + lane] = v;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float composed_wdelta(int W, float sigma, float tf) {
    if (W == RP_WEIGHT_LINEAR) return sigma * tf;
    if (W == RP_WEIGHT_QUADRATIC) return sigma * tf * tf;
    return sigma;
}

// COMPOSED, first half: a row's touches -> one entry (oracle: ora_profile_summarize).  Blocks of RP_SPARSE_BLOCK
// consecutive touches are composed sequentially — every (row, block) pair by its own 16-lane group, so a hot row's
// thousands of touches spread over the chip — and the block records of a row are folded in block order.
// entry = [row][count][psum][n_actions][A regret maps][A weight maps]; a block record has the same layout.
__global__ void k_block_counts(const uint32_t* counts, const uint32_t* n_segs, uint32_t n, uint32_t* nblk) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) nblk[g] = g < *n_segs ? (counts[g] + RP_SPARSE_BLOCK - 1u) / RP_SPARSE_BLOCK : 0u;
}

// block -> row index, so that a block's group finds its row with one load
__global__ void k_block_index(const uint32_t* nblk, const uint32_t* boff, const uint32_t* n_segs, uint32_t* blkseg) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= *n_segs) return;
    const uint32_t nb = nblk[g], base = boff[g];
    for (uint32_t k = 0; k < nb; ++k) blkseg[base + k] = g;
}

// k_fold's arithmetic for ONE entry of a row and one of its cells (a single rank's entry is applied where it is produced: the
// step's last launch goes away).  The same operations in the same order as k_fold.
__device__ __forceinline__ void apply_cell(const SparseParams& p, uint32_t rowid, uint32_t A, uint32_t a, const Map& mr, const Map& mw, uint32_t c,
                                           float psum) {
    float* row = p.tab + (size_t)rowid * 4u * A;
    float r = row[a], w = row[A + a], ev = row[2 * A + a];
    uint32_t v = reinterpret_cast<const uint32_t*>(row)[3 * A + a];
    if (mr.n) r = rp_maxf(mr.a * r + mr.b, mr.m);
    if (mw.n) w = rp_maxf(mw.a * w + mw.b, mw.m);
    if (c) {
        const uint32_t n2 = v + c;
        ev = ev + (psum - (float)c * ev) / (float)n2;
        v = n2;
    }
    row[a] = r;
    row[A + a] = w;
    row[2 * A + a] = ev;
    reinterpret_cast<uint32_t*>(row)[3 * A + a] = v;
}

// The same three steps (touches per row -> blocks per row -> first block of every row -> block -> row index) inside ONE tiled scan:
// its tile-sum kernel derives the counts from the run starts and zeroes the hot-row counter on the way, its tile kernel writes the
// index.  Three launches instead of six (the counts of the run-length encoding, block counts, a three-launch scan, the index, the
// counter's memset); one launch and one workgroup when the batch has at most ss::SCAN_ONE Decisions.
__device__ __forceinline__ uint32_t blk_of_row(const uint32_t* starts, uint32_t runs, uint32_t n, uint32_t g, uint32_t* count) {
    const uint32_t cnt = g < runs ? (g + 1u < runs ? starts[g + 1u] : n) - starts[g] : 0u;
    *count = cnt;
    return (cnt + RP_SPARSE_BLOCK - 1u) / RP_SPARSE_BLOCK;
}
__global__ __launch_bounds__(256) void k_blk_sums(const uint32_t* starts, const uint32_t* n_segs, uint32_t n, uint32_t* counts, uint32_t* nblk,
                                                  uint64_t* sums, uint32_t* hot_counter) {
    __shared__ uint64_t wt[4];
    const uint32_t runs = *n_segs, base = blockIdx.x * ss::SCAN_TILE + threadIdx.x * 4u;
    uint64_t s = 0;
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t g = base + k;
        if (g >= n) break;
        uint32_t cnt;
        const uint32_t nb = blk_of_row(starts, runs, n, g, &cnt);
        if (g < runs) counts[g] = cnt;
        nblk[g] = nb;
        s += nb;
    }
    uint64_t tot;
    (void)ss::block_exscan64(s, wt, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
    if (blockIdx.x == 0 && threadIdx.x == 0) *hot_counter = 0;
}
__global__ __launch_bounds__(256) void k_blk_tiles(const uint32_t* nblk, const uint32_t* n_segs, uint32_t n, const uint64_t* bases, uint32_t* boff,
                                                   uint32_t* blkseg) {
    __shared__ uint64_t wt[4];
    const uint32_t runs = *n_segs, base = blockIdx.x * ss::SCAN_TILE + threadIdx.x * 4u;
    uint32_t v[4];
    uint64_t s = 0;
    for (uint32_t k = 0; k < 4u; ++k) {
        v[k] = base + k < n ? nblk[base + k] : 0u;
        s += v[k];
    }
    uint64_t tot;
    uint32_t run = (uint32_t)(bases[blockIdx.x] + ss::block_exscan64(s, wt, &tot));
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t g = base + k;
        if (g >= n) break;
        boff[g] = run;
        if (g < runs)
            for (uint32_t b = 0; b < v[k]; ++b) blkseg[run + b] = g;
        run += v[k];
    }
}
__device__ __forceinline__ void blk_one_body(const uint32_t* starts, const uint32_t* n_segs, uint32_t n, uint32_t* counts, uint32_t* nblk,
                                             uint32_t* boff, uint32_t* blkseg, uint32_t* hot_counter, uint64_t* wt) {
    const uint32_t runs = *n_segs, per = (n + blockDim.x - 1u) / blockDim.x, lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint64_t s = 0;
    for (uint32_t g = lo; g < hi; ++g) {
        uint32_t cnt;
        const uint32_t nb = blk_of_row(starts, runs, n, g, &cnt);
        if (g < runs) counts[g] = cnt;
        nblk[g] = nb;
        s += nb;
    }
    uint64_t tot;
    uint32_t run = (uint32_t)ss::block_exscan64(s, wt, &tot);
    for (uint32_t g = lo; g < hi; ++g) {
        uint32_t cnt;
        const uint32_t nb = blk_of_row(starts, runs, n, g, &cnt);
        boff[g] = run;
        if (g < runs)
            for (uint32_t b = 0; b < nb; ++b) blkseg[run + b] = g;
        run += nb;
    }
    if (threadIdx.x == 0) *hot_counter = 0;
}
__global__ __launch_bounds__(1024) void k_blk_one(const uint32_t* starts, const uint32_t* n_segs, uint32_t n, uint32_t* counts, uint32_t* nblk,
                                                  uint32_t* boff, uint32_t* blkseg, uint32_t* hot_counter) {
    __shared__ uint64_t wt[16];
    blk_one_body(starts, n_segs, n, counts, nblk, boff, blkseg, hot_counter, wt);
}
// A batch of at most ss::SORT_ONE Decisions (the reference's 128 trees emit ~10^4): the stable sort by row, its run lengths and the
// block index in ONE launch of one workgroup — what sort_and_segment + k_blk_one do in eight (six of them five workgroups wide).
__global__ __launch_bounds__(1024) void k_prep_one(const uint32_t* rows, uint32_t n, uint32_t bits, uint32_t* keys_out, uint32_t* perm,
                                                   uint32_t* seg_rows, uint32_t* seg_offsets, uint32_t* seg_counts, uint32_t* n_segs,
                                                   uint32_t* nblk, uint32_t* boff, uint32_t* blkseg, uint32_t* hot_counter) {
    __shared__ ss::SortOneLds L;
    __shared__ uint32_t wsum[16];
    ss::sort_one_body(rows, keys_out, perm, n, bits, L);
    const uint32_t* lkey = L.key;  // the sorted keys
    // ---- run lengths and block index.  Wavefront w owns the contiguous chunk w of the sorted keys and walks it in rounds of 64 (coalesced
    // reads, ballots and shuffles instead of a serial walk per work-item); the run starts stay in LDS (16 bits: n <= 16 384) for the block
    // counts, which therefore read no global memory at all.
    uint16_t* s_start = reinterpret_cast<uint16_t*>(L.wcount);  // [n]; the sort is done with the counters
    const uint32_t tid = threadIdx.x, ln = tid & 63u, wv = tid >> 6;
    const uint32_t chunk = ((n + 1023u) / 1024u) * 64u, lo = min(n, wv * chunk), hi = min(n, lo + chunk);
    uint32_t heads = 0;
    for (uint32_t i0 = lo; i0 < hi; i0 += 64u) {
        const uint32_t i = i0 + ln;
        const bool head = i < hi && (i == 0u || lkey[i] != lkey[i - 1u]);
        heads += (uint32_t)__popcll(__ballot(head));
    }
    if (ln == 0) wsum[wv] = heads;
    __syncthreads();
    uint32_t run = 0, runs = 0;
    for (uint32_t w = 0; w < 16u; ++w) {
        const uint32_t c = wsum[w];
        run += w < wv ? c : 0u;
        runs += c;
    }
    for (uint32_t i0 = lo; i0 < hi; i0 += 64u) {
        const uint32_t i = i0 + ln;
        const uint32_t key = i < hi ? lkey[i] : 0u;
        const bool head = i < hi && (i == 0u || key != lkey[i - 1u]);
        const unsigned long long m = __ballot(head);
        if (head) {
            const uint32_t r = run + (uint32_t)__popcll(m & ((1ull << ln) - 1ull));
            seg_rows[r] = key;
            seg_offsets[r] = i;
            s_start[r] = (uint16_t)i;
        }
        run += (uint32_t)__popcll(m);
    }
    if (tid == 0) {
        *n_segs = runs;
        *hot_counter = 0;
    }
    __syncthreads();  // every run start is in LDS; wsum is read
    // blocks per row and their exclusive scan, g in [0, n) (rows past the last run: no blocks), chunked the same way
    uint32_t blocks = 0;
    for (uint32_t g0 = lo; g0 < hi; g0 += 64u) {
        const uint32_t g = g0 + ln;
        const uint32_t cnt = g < hi && g < runs ? (g + 1u < runs ? (uint32_t)s_start[g + 1u] : n) - (uint32_t)s_start[g] : 0u;
        uint32_t nb = (cnt + RP_SPARSE_BLOCK - 1u) / RP_SPARSE_BLOCK;
        for (int d = 32; d > 0; d >>= 1) nb += (uint32_t)__shfl_xor((int)nb, d, 64);
        blocks += nb;
    }
    __syncthreads();  // (wsum's first use is over)
    if (ln == 0) wsum[wv] = blocks;
    __syncthreads();
    uint32_t brun = 0;
    for (uint32_t w = 0; w < wv; ++w) brun += wsum[w];
    for (uint32_t g0 = lo; g0 < hi; g0 += 64u) {
        const uint32_t g = g0 + ln;
        const bool in = g < hi;
        const uint32_t cnt = in && g < runs ? (g + 1u < runs ? (uint32_t)s_start[g + 1u] : n) - (uint32_t)s_start[g] : 0u;
        const uint32_t nb = (cnt + RP_SPARSE_BLOCK - 1u) / RP_SPARSE_BLOCK;
        uint32_t incl = nb;
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (ln >= d) incl += up;
        }
        const uint32_t at = brun + incl - nb;
        if (in) {
            if (g < runs) seg_counts[g] = cnt;
            nblk[g] = nb;
            boff[g] = at;
            for (uint32_t b = 0; b < nb; ++b) blkseg[at + b] = g;
        }
        brun += (uint32_t)__shfl((int)incl, 63, 64);
    }
}

template <bool GATHER, bool APPLY>  // GATHER: the touches are read from the unsorted batch through sg.perm (no sorted copy exists);
                                    // APPLY: a one-block row's entry goes straight into the table (single rank)
__global__ __launch_bounds__(256) void k_block_maps_sparse(SparseParams p, DevBatch b, Segments sg, const uint32_t* nblk,
                                                           const uint32_t* boff, const uint32_t* blkseg, SortedBatch sb,
                                                           unsigned char* entries, unsigned char* blocks, uint32_t entry_bytes,
                                                           uint32_t max_blocks) {
    const uint32_t n_segs = *sg.n_segs;
    if (n_segs == 0) return;
    const uint32_t total = boff[n_segs - 1] + nblk[n_segs - 1];
    const uint32_t a = threadIdx.x % GROUP;
    const uint32_t A = p.A;
    const float NEG_INF = rp_u2f(0xff800000u);
    const Map ident{1.0f, 0.0f, NEG_INF, 0u};
    for (uint32_t bi = (blockIdx.x * blockDim.x + threadIdx.x) / GROUP; bi < total && bi < max_blocks;
         bi += gridDim.x * blockDim.x / GROUP) {
        const uint32_t g = blkseg[bi], blk = bi - boff[g];
        const uint32_t cnt = sg.counts[g];
        const uint32_t t_lo = blk * RP_SPARSE_BLOCK, t_hi = min(cnt, t_lo + RP_SPARSE_BLOCK);
        const uint32_t off = sg.offsets[g];
        const uint32_t nact = b.nact[sg.perm[off]];
        const bool mine = a < nact;
        Map br = ident, bw = ident;
        float bp = 0.0f;
        // touches of a row are contiguous in the sorted batch: chunk k+1 is in flight while chunk k is composed
        TouchChunk cur, nxt;
        if (GATHER) fetch_chunk_gather(cur, b, sg.perm, off + t_lo, min((uint32_t)PF, t_hi - t_lo), A, a, mine);
        else fetch_chunk(cur, sb, off + t_lo, min((uint32_t)PF, t_hi - t_lo), A, a, mine);
        for (uint32_t t0 = t_lo; t0 < t_hi; t0 += PF) {
            const uint32_t m = min((uint32_t)PF, t_hi - t0);
            if (t0 + PF < t_hi) {
                if (GATHER) fetch_chunk_gather(nxt, b, sg.perm, off + t0 + PF, min((uint32_t)PF, t_hi - t0 - PF), A, a, mine);
                else fetch_chunk(nxt, sb, off + t0 + PF, min((uint32_t)PF, t_hi - t0 - PF), A, a, mine);
            }
#pragma unroll
            for (uint32_t u = 0; u < PF; ++u) {
                if (u >= m) break;
                if (mine) {
                    if ((cur.ev_mask >> u) & 1u) map_touch(br, p.dr, cur.dv[u], p.floor_r);
                    map_touch(bw, p.dw, composed_wdelta(p.W, cur.sv[u], p.tf), RP_EPSILON);
                }
                bp += cur.pv[u];
            }
            cur = nxt;
        }
        // a row with a single block is final: group = total = compose(identity, block) = block
        const bool single = nblk[g] == 1u;
        unsigned char* ent = single ? entries + (size_t)g * entry_bytes : blocks + (size_t)bi * entry_bytes;
        if (a == 0) {
            uint32_t* hdr = reinterpret_cast<uint32_t*>(ent);
            hdr[0] = sg.rows[g];
            hdr[1] = single ? cnt : t_hi - t_lo;
            hdr[2] = rp_f2u(single ? 0.0f + (0.0f + bp) : bp);  // group sum = 0 + block sum, total = 0 + group sum
            hdr[3] = nact;
        }
        if (a < A) {
            Map* mr = reinterpret_cast<Map*>(ent + 16);
            mr[a] = mine ? br : ident;
            mr[A + a] = mine ? bw : ident;
        }
        if (APPLY && single && mine) apply_cell(p, sg.rows[g], A, a, br, bw, cnt, 0.0f + (0.0f + bp));
    }
}

// rows with several blocks: fold the block records in block order into the entry
template <bool APPLY>
__global__ __launch_bounds__(256) void k_seg_fold(SparseParams p, Segments sg, const uint32_t* nblk, const uint32_t* boff,
                                                  unsigned char* entries, const unsigned char* blocks, uint32_t entry_bytes,
                                                  uint32_t* hot, uint32_t* n_hot, uint32_t hot_cap) {
    const uint32_t n_segs = *sg.n_segs;
    const uint32_t a = threadIdx.x % GROUP;
    const uint32_t A = p.A;
    const float NEG_INF = rp_u2f(0xff800000u);
    const Map ident{1.0f, 0.0f, NEG_INF, 0u};
    for (uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) / GROUP; g < n_segs; g += gridDim.x * blockDim.x / GROUP) {
        const uint32_t nb = nblk[g];
        if (nb <= 1u) continue;
        if (nb > RP_FOLD_GROUP) {  // a hot row: its groups fold in parallel in k_hot_fold
            uint32_t slot = hot_cap;
            if (a == 0) slot = atomicAdd(n_hot, 1u);
            slot = __shfl(slot, (int)(threadIdx.x & 48u), 64);
            if (slot < hot_cap) {
                if (a == 0) hot[slot] = g;
                continue;
            }
        }
        // two-level fold (RP_FOLD_GROUP): block records sequentially into a group, groups sequentially into the total
        Map tr = ident, tw = ident;
        float tp = 0.0f;
        uint32_t nact = 0;
        const unsigned char* rec0 = blocks + (size_t)boff[g] * entry_bytes;
        for (uint32_t g0 = 0; g0 < nb; g0 += RP_FOLD_GROUP) {
            const uint32_t g1 = min(nb, g0 + RP_FOLD_GROUP);
            Map sr = ident, sw = ident;
            float sp = 0.0f;
            for (uint32_t k0 = g0; k0 < g1; k0 += PF) {  // records are fetched PF at a time, folded in order
                const uint32_t m = min((uint32_t)PF, g1 - k0);
                Map mr[PF], mw[PF];
                float ps[PF];
#pragma unroll
                for (uint32_t u = 0; u < PF; ++u) {
                    const unsigned char* rec = rec0 + (size_t)(k0 + (u < m ? u : 0u)) * entry_bytes;
                    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(rec);
                    nact = hdr[3];
                    ps[u] = rp_u2f(hdr[2]);
                    mr[u] = a < A ? reinterpret_cast<const Map*>(rec + 16)[a] : ident;
                    mw[u] = a < A ? reinterpret_cast<const Map*>(rec + 16)[A + a] : ident;
                }
#pragma unroll
                for (uint32_t u = 0; u < PF; ++u) {
                    if (u >= m) break;
                    sr = map_compose(sr, mr[u]);
                    sw = map_compose(sw, mw[u]);
                    sp += ps[u];
                }
            }
            tr = map_compose(tr, sr);
            tw = map_compose(tw, sw);
            tp += sp;
        }
        unsigned char* ent = entries + (size_t)g * entry_bytes;
        if (a == 0) {
            uint32_t* hdr = reinterpret_cast<uint32_t*>(ent);
            hdr[0] = sg.rows[g];
            hdr[1] = sg.counts[g];
            hdr[2] = rp_f2u(tp);
            hdr[3] = nact;
        }
        if (a < A) {
            Map* mr = reinterpret_cast<Map*>(ent + 16);
            mr[a] = tr;
            mr[A + a] = tw;
        }
        if (APPLY && a < nact) apply_cell(p, sg.rows[g], A, a, tr, tw, sg.counts[g], tp);
    }
}

// hot rows (more than RP_FOLD_GROUP blocks): one workgroup per row, the groups of the two-level fold in parallel —
// thread (slot s, cell c) folds the records of group g0 + s, then the cell threads fold the group maps in order
template <bool APPLY>
__global__ __launch_bounds__(256) void k_hot_fold(SparseParams p, Segments sg, const uint32_t* nblk, const uint32_t* boff,
                                                  unsigned char* entries, const unsigned char* blocks, uint32_t entry_bytes,
                                                  const uint32_t* hot, const uint32_t* n_hot, uint32_t hot_cap) {
    __shared__ __attribute__((aligned(16))) Map sup[256];
    __shared__ float supp[8];
    const uint32_t A = p.A, W2 = 2 * A;
    const uint32_t tid = threadIdx.x, c = tid % 32u, s = tid / 32u;  // 8 slots x 32 cell lanes (2A <= 32)
    const float NEG_INF = rp_u2f(0xff800000u);
    const Map ident{1.0f, 0.0f, NEG_INF, 0u};
    const uint32_t nh = min(*n_hot, hot_cap);
    for (uint32_t hi = blockIdx.x; hi < nh; hi += gridDim.x) {
        const uint32_t g = hot[hi];
        const uint32_t nb = nblk[g], ngrp = (nb + RP_FOLD_GROUP - 1) / RP_FOLD_GROUP;
        const unsigned char* rec0 = blocks + (size_t)boff[g] * entry_bytes;
        const uint32_t nact = reinterpret_cast<const uint32_t*>(rec0)[3];
        Map tot = ident;
        float tp = 0.0f;
        for (uint32_t g0 = 0; g0 < ngrp; g0 += 8) {
            const uint32_t grp = g0 + s;
            if (grp < ngrp) {
                const uint32_t k_lo = grp * RP_FOLD_GROUP, k_hi = min(nb, k_lo + RP_FOLD_GROUP);
                Map m = ident;
                float gp = 0.0f;
                for (uint32_t k0 = k_lo; k0 < k_hi; k0 += PF) {
                    const uint32_t cnt = min((uint32_t)PF, k_hi - k0);
                    Map mm[PF];
                    float pp[PF];
#pragma unroll
                    for (uint32_t u = 0; u < PF; ++u) {
                        const unsigned char* rec = rec0 + (size_t)(k0 + (u < cnt ? u : 0u)) * entry_bytes;
                        mm[u] = c < W2 ? reinterpret_cast<const Map*>(rec + 16)[c] : ident;
                        pp[u] = rp_u2f(reinterpret_cast<const uint32_t*>(rec)[2]);
                    }
#pragma unroll
                    for (uint32_t u = 0; u < PF; ++u) {
                        if (u >= cnt) break;
                        m = map_compose(m, mm[u]);
                        gp += pp[u];
                    }
                }
                sup[s * 32u + c] = m;
                if (c == 0) supp[s] = gp;
            }
            __syncthreads();
            const uint32_t have = min(8u, ngrp - g0);
            if (tid < W2) {
                for (uint32_t k = 0; k < have; ++k) tot = map_compose(tot, sup[k * 32u + tid]);
            } else if (tid == 32u) {
                for (uint32_t k = 0; k < have; ++k) tp += supp[k];
            }
            __syncthreads();
        }
        unsigned char* ent = entries + (size_t)g * entry_bytes;
        if (tid == 32u) {
            uint32_t* hdr = reinterpret_cast<uint32_t*>(ent);
            hdr[0] = sg.rows[g];
            hdr[1] = sg.counts[g];
            hdr[2] = rp_f2u(tp);
            hdr[3] = nact;
        }
        if (tid < W2) reinterpret_cast<Map*>(ent + 16)[tid] = tot;
        if (APPLY) {  // the entry is read back by the cell lanes once the whole workgroup has written it
            __syncthreads();
            if (tid < nact) {
                const Map mr = reinterpret_cast<const Map*>(ent + 16)[tid], mw = reinterpret_cast<const Map*>(ent + 16)[A + tid];
                apply_cell(p, sg.rows[g], A, tid, mr, mw, sg.counts[g], rp_u2f(reinterpret_cast<const uint32_t*>(ent)[2]));
            }
        }
    }
}

// COMPOSED, second half: entries of one row, in the given (rank) order, folded into the table row
// (oracle: ora_profile_fold).  `order` == nullptr: entry g is its own segment (single rank: rows are distinct,
// *n_segs_dev entries).
__global__ __launch_bounds__(256) void k_fold(SparseParams p, const unsigned char* entries, uint32_t entry_bytes,
                                              const uint32_t* order, const uint32_t* offsets, const uint32_t* counts,
                                              const uint32_t* n_segs_dev) {
    const uint32_t n_segs = *n_segs_dev;
    const uint32_t a = threadIdx.x % GROUP;
    const uint32_t A = p.A;
    for (uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) / GROUP; g < n_segs; g += gridDim.x * blockDim.x / GROUP) {
        const uint32_t off = order ? offsets[g] : g, cnt = order ? counts[g] : 1u;
        const unsigned char* first = entries + (size_t)(order ? order[off] : g) * entry_bytes;
        const uint32_t rowid = reinterpret_cast<const uint32_t*>(first)[0];
        const uint32_t nact = reinterpret_cast<const uint32_t*>(first)[3];
        if (a >= nact) continue;
        float* row = p.tab + (size_t)rowid * 4u * A;
        float r = row[a], w = row[A + a], ev = row[2 * A + a];
        uint32_t v = reinterpret_cast<const uint32_t*>(row)[3 * A + a];
        for (uint32_t t = 0; t < cnt; ++t) {
            const unsigned char* ent = entries + (size_t)(order ? order[off + t] : g) * entry_bytes;
            const uint32_t* hdr = reinterpret_cast<const uint32_t*>(ent);
            const Map mr = reinterpret_cast<const Map*>(ent + 16)[a];
            const Map mw = reinterpret_cast<const Map*>(ent + 16)[A + a];
            if (mr.n) r = rp_maxf(mr.a * r + mr.b, mr.m);
            if (mw.n) w = rp_maxf(mw.a * w + mw.b, mw.m);
            const uint32_t c = hdr[1];
            if (c) {
                const uint32_t n2 = v + c;
                ev = ev + (rp_u2f(hdr[2]) - (float)c * ev) / (float)n2;
                v = n2;
            }
        }
        row[a] = r;
        row[A + a] = w;
        row[2 * A + a] = ev;
        reinterpret_cast<uint32_t*>(row)[3 * A + a] = v;
    }
}

__global__ void k_entry_rows(const unsigned char* entries, uint32_t entry_bytes, uint32_t n, uint32_t* rows) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rows[i] = *reinterpret_cast<const uint32_t*>(entries + (size_t)i * entry_bytes);
}

__global__ void k_init_rows(float* tab, uint64_t n_rows, uint32_t A, const float* default_regret) {
    const uint64_t cells = n_rows * 4u * A;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < cells; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)(e % (4u * A));
        tab[e] = c < A ? default_regret[c] : 0.0f;  // regret <- default, weight/payoff/visits <- 0 (book.rs:93-122)
    }
}

__global__ void k_gather_rows(const float* tab, uint32_t A, const uint32_t* rows, uint64_t n, rp_encounter* out) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * A) return;
    const uint64_t i = e / A;
    const uint32_t a = (uint32_t)(e % A);
    const float* row = tab + (size_t)rows[i] * 4u * A;
    out[e].regret = row[a];
    out[e].weight = row[A + a];
    out[e].payoff = row[2 * A + a];
    out[e].visits = reinterpret_cast<const uint32_t*>(row)[3 * A + a];
}

struct SpClock {
    double total_ms = 0.0;
    uint64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace rp

using namespace rp;

struct rp_profile {
    int device = 0;
    uint64_t n_rows = 0;
    uint32_t A = 0;
    int R = 0, W = 0;
    rp_hyper hp{};
    uint64_t epoch = 0;
    uint32_t max_batch = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    float* tab = nullptr;
    // sort / segment workspace (capacity `cap` items)
    uint32_t cap = 0;
    uint32_t *iota = nullptr, *keys_out = nullptr, *perm = nullptr, *seg_rows = nullptr, *seg_counts = nullptr,
             *seg_offsets = nullptr, *n_segs = nullptr, *ent_rows = nullptr, *nblk = nullptr, *boff = nullptr;
    unsigned char* blocks = nullptr;   // block records of multi-block rows
    uint32_t* blkseg = nullptr;        // [max blocks] row index of every block
    uint32_t* hot = nullptr;           // [HOT_CAP + 1] rows with more than RP_FOLD_GROUP blocks; last slot = counter
    bool no_prep_one = false;          // RP_SPARSE_NO_PREP_ONE=1 (tests, A/B): small batches through the tiled sort as well
    float *srt_regret = nullptr, *srt_policy = nullptr, *srt_payoff = nullptr;  // the batch in sorted order (ordered mode)
    uint16_t* srt_expanded = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_bytes = 0;
    unsigned char* entries = nullptr;  // local composed apply
    bool profiling = false;
    SpClock clk_sort, clk_apply;
};

namespace rp {

// nlmc.hip: the NLHE traversal reads (and initialises the rows of) the table it feeds
float* profile_table(rp_profile* h) { return h->tab; }
hipStream_t profile_stream(rp_profile* h) { return h->stream; }
uint64_t profile_epoch(const rp_profile* h) { return h->epoch; }
unsigned char* profile_entries(rp_profile* h) { return h->entries; }  // the local summary buffer: max_batch entries

static size_t entry_bytes_of(const rp_profile* h) { return 16 + (size_t)2 * h->A * sizeof(Map); }
// sum over rows of ceil(count / RP_SPARSE_BLOCK) <= rows + n / RP_SPARSE_BLOCK <= n + n / RP_SPARSE_BLOCK
static uint32_t max_blocks_of(uint32_t n) { return n + n / RP_SPARSE_BLOCK + 1u; }

static void sp_begin(rp_profile* h, SpClock& c) {
    if (!h->profiling) return;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, h->stream);
    c.pending.emplace_back(a, b);
}
static void sp_end(rp_profile* h, SpClock& c) {
    if (!h->profiling || c.pending.empty()) return;
    (void)hipEventRecord(c.pending.back().second, h->stream);
    c.launches += 1;
}
static void sp_drain(SpClock& c) {
    for (auto& pr : c.pending) {
        float ms = 0.0f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess)
            c.total_ms += ms;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    c.pending.clear();
}

static void free_workspace(rp_profile* h) {
    for (void* p : {(void*)h->iota, (void*)h->keys_out, (void*)h->perm, (void*)h->seg_rows, (void*)h->seg_counts,
                    (void*)h->seg_offsets, (void*)h->ent_rows, (void*)h->nblk, (void*)h->boff, (void*)h->blocks, (void*)h->blkseg,
                    h->sort_tmp,
                    (void*)h->entries, (void*)h->srt_regret, (void*)h->srt_policy, (void*)h->srt_payoff,
                    (void*)h->srt_expanded})
        if (p) (void)hipFree(p);
    h->iota = h->keys_out = h->perm = h->seg_rows = h->seg_counts = h->seg_offsets = h->ent_rows = h->nblk = h->boff = nullptr;
    h->blocks = nullptr;
    h->blkseg = nullptr;
    h->srt_regret = h->srt_policy = h->srt_payoff = nullptr;
    h->srt_expanded = nullptr;
    h->sort_tmp = nullptr;
    h->entries = nullptr;
    h->cap = 0;
}

// workspace for sorting / segmenting up to n items
static int ensure_capacity(rp_profile* h, uint32_t n) {
    if (n <= h->cap) return RP_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    free_workspace(h);
    const uint32_t cap = n;
    HIP_TRY(hipMalloc(&h->iota, (size_t)cap * 4));
    HIP_TRY(hipMalloc(&h->keys_out, (size_t)cap * 4));
    HIP_TRY(hipMalloc(&h->perm, (size_t)cap * 4));
    HIP_TRY(hipMalloc(&h->seg_rows, (size_t)cap * 4));
    HIP_TRY(hipMalloc(&h->seg_counts, (size_t)cap * 4));
    HIP_TRY(hipMalloc(&h->seg_offsets, (size_t)cap * 4));
    HIP_TRY(hipMalloc(&h->ent_rows, (size_t)cap * 4));
    HIP_TRY(hipMalloc(&h->entries, (size_t)cap * entry_bytes_of(h)));
    HIP_TRY(hipMalloc(&h->srt_regret, (size_t)cap * h->A * 4));
    HIP_TRY(hipMalloc(&h->srt_policy, (size_t)cap * h->A * 4));
    HIP_TRY(hipMalloc(&h->srt_payoff, (size_t)cap * 4));
    HIP_TRY(hipMalloc(&h->srt_expanded, (size_t)cap * 2));
    HIP_TRY(hipMalloc(&h->nblk, (size_t)cap * 4));
    HIP_TRY(hipMalloc(&h->boff, (size_t)cap * 4));
    HIP_TRY(hipMalloc(&h->blocks, (size_t)max_blocks_of(cap) * entry_bytes_of(h)));
    HIP_TRY(hipMalloc(&h->blkseg, (size_t)max_blocks_of(cap) * 4));
    // scratch of the sort (histograms, their scan, one ping-pong pair) + the run-length encoding's n flag words
    h->sort_bytes = ss::sort_scratch_bytes(cap) + ((ss::scan_scratch_bytes(cap) + 255) & ~(size_t)255) + (size_t)cap * 4 + 256;
    HIP_TRY(hipMalloc(&h->sort_tmp, h->sort_bytes));
    hipLaunchKernelGGL(k_iota, dim3((cap + 255) / 256), dim3(256), 0, h->stream, h->iota, cap);
    HIP_TRY(hipGetLastError());
    h->cap = cap;
    return RP_OK;
}

static uint32_t key_bits(uint64_t n_rows) {
    uint32_t bits = 1;
    while (bits < 32 && (1ull << bits) < n_rows) ++bits;
    return bits;
}

// rows[n] (device) -> perm (stable by row), distinct rows, counts, offsets, n_segs
static int sort_and_segment(rp_profile* h, const uint32_t* rows, uint32_t n, bool counts_later = false) {
    int rc = ensure_capacity(h, n);
    if (rc) return rc;
    unsigned char* sp = reinterpret_cast<unsigned char*>(h->sort_tmp);
    HIP_TRY(ss::sort_pairs(rows, h->iota, h->keys_out, h->perm, n, key_bits(h->n_rows), sp, h->stream));
    sp += ss::sort_scratch_bytes(h->cap);
    void* scan_tmp = sp;
    sp += (ss::scan_scratch_bytes(h->cap) + 255) & ~(size_t)255;
    // seg_offsets = the start of every run = the exclusive scan of the run lengths
    // counts_later: the caller is launch_summarize, whose block scan derives the counts from the starts on its way
    const bool derived = counts_later && (n + ss::SCAN_TILE - 1u) / ss::SCAN_TILE <= ss::SCAN_ONE;  // launch_summarize's condition
    HIP_TRY(ss::run_length_encode(h->keys_out, n, h->seg_rows, h->seg_offsets, derived ? nullptr : h->seg_counts, h->n_segs,
                                  reinterpret_cast<uint32_t*>(sp), scan_tmp, h->stream));
    return RP_OK;
}

static int composed_params(const rp_profile* h, SparseParams& p) {
    const float t = (float)h->epoch;
    switch (h->R) {
        case RP_REGRET_SUMMED:
        case RP_REGRET_FLOORED: p.dr = 1.0f; break;
        case RP_REGRET_LINEAR: p.dr = t / (t + 1.0f); break;
        default:
            return rp::fail(RP_ERR_UNSUPPORTED,
                            "composed update needs a sign-independent discount (Summed/Linear/Floored regret)");
    }
    p.dw = h->W == RP_WEIGHT_EXPONENTIAL ? 0.9999f : 1.0f;
    return RP_OK;
}
static SparseParams make_params(const rp_profile* h) {
    SparseParams p{};
    p.tab = h->tab;
    p.A = h->A;
    p.R = h->R;
    p.W = h->W;
    p.tf = (float)h->epoch;
    if (h->R == RP_REGRET_DISCOUNTED) {
        p.pow15 = rp_pow15(p.tf);
        p.pow05 = rp_pow05(p.tf);
    }
    p.floor_r = regret_floor_of(h->R, h->hp.regret_min);
    p.dr = p.dw = 1.0f;
    return p;
}
static int check_batch(const rp_profile* h, const rp_decisions* b, const char* who) {
    if (!h || !b) return rp::fail(RP_ERR_INVALID, "%s: null argument", who);
    if (b->n == 0) return RP_OK;
    if (!b->row || !b->n_actions || !b->expanded || !b->regret || !b->policy || !b->payoff)
        return rp::fail(RP_ERR_INVALID, "%s: null batch array", who);
    return RP_OK;
}
static uint32_t group_blocks(uint32_t n) { return std::max(1u, std::min((n * GROUP + 255u) / 256u, 65535u)); }

// segments (already built for this batch) -> entries[g] for g < n_segs
static int launch_summarize(rp_profile* h, const SparseParams& p, const DevBatch& b, const Segments& sg, uint32_t n,
                            unsigned char* entries, bool apply_local = false, bool prepped = false /* k_prep_one has built the block index */) {
    const uint32_t eb = (uint32_t)entry_bytes_of(h);
    const uint32_t mb = max_blocks_of(n);
    void* scan_tmp = reinterpret_cast<unsigned char*>(h->sort_tmp) + ss::sort_scratch_bytes(h->cap);
    const uint32_t tiles = (n + ss::SCAN_TILE - 1u) / ss::SCAN_TILE;
    if (prepped) {
    } else if (n <= ss::SCAN_ONE) {
        hipLaunchKernelGGL(k_blk_one, dim3(1), dim3(ss::ONE_THREADS), 0, h->stream, h->seg_offsets, h->n_segs, n, h->seg_counts, h->nblk, h->boff, h->blkseg,
                           h->hot + HOT_CAP);
    } else if (tiles <= ss::SCAN_ONE) {
        uint64_t* sums = reinterpret_cast<uint64_t*>(scan_tmp);
        hipLaunchKernelGGL(k_blk_sums, dim3(tiles), dim3(256), 0, h->stream, h->seg_offsets, h->n_segs, n, h->seg_counts, h->nblk, sums,
                           h->hot + HOT_CAP);
        hipLaunchKernelGGL(ss::k_scan_one64, dim3(1), dim3(256), 0, h->stream, sums, tiles);
        hipLaunchKernelGGL(k_blk_tiles, dim3(tiles), dim3(256), 0, h->stream, h->nblk, h->n_segs, n, sums, h->boff, h->blkseg);
    } else {
        hipLaunchKernelGGL(k_block_counts, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->seg_counts, h->n_segs, n, h->nblk);
        HIP_TRY(ss::exclusive_scan<uint32_t>(h->nblk, h->boff, n, scan_tmp, h->stream));
        hipLaunchKernelGGL(k_block_index, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->nblk, h->boff, h->n_segs, h->blkseg);
        HIP_TRY(hipMemsetAsync(h->hot + HOT_CAP, 0, 4, h->stream));
    }
    const SortedBatch sb{h->srt_regret, h->srt_policy, h->srt_payoff, h->srt_expanded};
#define RP_MAPS(G, AP)                                                                                                                  \
    hipLaunchKernelGGL((k_block_maps_sparse<G, AP>), dim3(group_blocks(mb)), dim3(256), 0, h->stream, p, b, sg, h->nblk, h->boff, h->blkseg, \
                       sb, entries, h->blocks, eb, mb)
    if (apply_local) RP_MAPS(true, true);  // the block maps read the touches through the sort permutation
    else RP_MAPS(true, false);
#undef RP_MAPS
    if (apply_local && prepped) {
        // a small batch: every row with several blocks folds in k_seg_fold (a hot-row capacity of 0: its own serial two-level fold, the
        // same grouping as k_hot_fold's) — at most 16 384 touches, and one launch less
        hipLaunchKernelGGL(k_seg_fold<true>, dim3(group_blocks(n)), dim3(256), 0, h->stream, p, sg, h->nblk, h->boff, entries, h->blocks, eb,
                           h->hot, h->hot + HOT_CAP, 0u);
    } else if (apply_local) {
        hipLaunchKernelGGL(k_seg_fold<true>, dim3(group_blocks(n)), dim3(256), 0, h->stream, p, sg, h->nblk, h->boff, entries, h->blocks, eb,
                           h->hot, h->hot + HOT_CAP, (uint32_t)HOT_CAP);
        hipLaunchKernelGGL(k_hot_fold<true>, dim3(64), dim3(256), 0, h->stream, p, sg, h->nblk, h->boff, entries, h->blocks, eb, h->hot,
                           h->hot + HOT_CAP, (uint32_t)HOT_CAP);
    } else {
        hipLaunchKernelGGL(k_seg_fold<false>, dim3(group_blocks(n)), dim3(256), 0, h->stream, p, sg, h->nblk, h->boff, entries, h->blocks, eb,
                           h->hot, h->hot + HOT_CAP, (uint32_t)HOT_CAP);
        hipLaunchKernelGGL(k_hot_fold<false>, dim3(64), dim3(256), 0, h->stream, p, sg, h->nblk, h->boff, entries, h->blocks, eb, h->hot,
                           h->hot + HOT_CAP, (uint32_t)HOT_CAP);
    }
    HIP_TRY(hipGetLastError());
    return RP_OK;
}

}  // namespace rp

extern "C" {

int rp_profile_create(int device, uint64_t n_rows, uint32_t max_actions, rp_regret_kind regret, rp_weight_kind weight,
                      const rp_hyper* hp, const float* default_regret, uint32_t max_batch, rp_profile** out) {
    if (!out) return rp::fail(RP_ERR_INVALID, "rp_profile_create: out is null");
    *out = nullptr;
    if (n_rows == 0 || n_rows > 0xffffffffull) return rp::fail(RP_ERR_INVALID, "rp_profile_create: n_rows must be in 1..2^32-1");
    if (max_actions == 0 || max_actions > GROUP) return rp::fail(RP_ERR_INVALID, "rp_profile_create: max_actions must be in 1..16");
    if ((int)regret < 0 || (int)regret > RP_REGRET_ASYMMETRIC || (int)weight < 0 || (int)weight > RP_WEIGHT_EXPONENTIAL)
        return rp::fail(RP_ERR_INVALID, "rp_profile_create: unknown schedule");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_profile_create: no HIP device (there is no CPU fallback)");
    rp_profile* h = new rp_profile();
    h->device = device;
    h->n_rows = n_rows;
    h->A = max_actions;
    h->R = regret;
    h->W = weight;
    if (hp) h->hp = *hp; else rp_hyper_default(&h->hp);
    h->max_batch = max_batch;
#define PF_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            int _rc = rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));        \
            rp_profile_destroy(h);                                                                \
            return _rc;                                                                           \
        }                                                                                         \
    } while (0)
    PF_TRY(hipSetDevice(device));
    PF_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    PF_TRY(hipMalloc(&h->tab, (size_t)n_rows * 4u * max_actions * 4u));
    PF_TRY(hipMalloc(&h->n_segs, 4));
    PF_TRY(hipMalloc(&h->hot, (HOT_CAP + 1) * 4));
    h->no_prep_one = getenv("RP_SPARSE_NO_PREP_ONE") != nullptr;
    float* d_def = nullptr;
    std::vector<float> def(max_actions, 0.0f);
    if (default_regret) def.assign(default_regret, default_regret + max_actions);
    PF_TRY(hipMalloc(&d_def, max_actions * 4u));
    PF_TRY(hipMemcpyAsync(d_def, def.data(), max_actions * 4u, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_init_rows, dim3(4096), dim3(256), 0, h->stream, h->tab, n_rows, max_actions, d_def);
    PF_TRY(hipGetLastError());
    PF_TRY(hipStreamSynchronize(h->stream));
    (void)hipFree(d_def);
    if (max_batch) {
        int rc = ensure_capacity(h, max_batch);
        if (rc) {
            rp_profile_destroy(h);
            return rc;
        }
    }
    // hipMemset on device memory returns before it has run, and a non-blocking stream does not wait for the null stream: the
    // first launch on this handle's stream could otherwise overtake the initialisation above and be overwritten by it
    (void)hipDeviceSynchronize();
    *out = h;
    return RP_OK;
}

int rp_profile_destroy(rp_profile* h) {
    if (!h) return RP_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    sp_drain(h->clk_sort);
    sp_drain(h->clk_apply);
    free_workspace(h);
    if (h->tab) (void)hipFree(h->tab);
    if (h->n_segs) (void)hipFree(h->n_segs);
    if (h->hot) (void)hipFree(h->hot);
    if (h->stream && h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return RP_OK;
}

int rp_profile_apply(rp_profile* h, const rp_decisions* batch, rp_update_mode mode) {
    int rc = check_batch(h, batch, "rp_profile_apply");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    SparseParams p = make_params(h);
    if (mode == RP_UPDATE_COMPOSED && (rc = composed_params(h, p))) return rc;
    if (batch->n) {
        const DevBatch b{batch->row, batch->n_actions, batch->expanded, batch->regret, batch->policy, batch->payoff};
        // a small batch (the reference's 128 trees): sort, run lengths and block index by one workgroup in one launch (k_prep_one)
        const bool one = mode == RP_UPDATE_COMPOSED && batch->n <= ss::SORT_ONE && key_bits(h->n_rows) <= 27u && !h->no_prep_one;
        sp_begin(h, h->clk_sort);
        if (one) {
            if ((rc = ensure_capacity(h, batch->n))) return rc;
            hipLaunchKernelGGL(k_prep_one, dim3(1), dim3(1024), 0, h->stream, batch->row, batch->n, key_bits(h->n_rows), h->keys_out, h->perm,
                               h->seg_rows, h->seg_offsets, h->seg_counts, h->n_segs, h->nblk, h->boff, h->blkseg, h->hot + HOT_CAP);
        } else if ((rc = sort_and_segment(h, batch->row, batch->n, mode == RP_UPDATE_COMPOSED))) {
            return rc;
        }
        sp_end(h, h->clk_sort);
        const Segments sg{h->seg_rows, h->seg_counts, h->seg_offsets, h->n_segs, h->perm};
        sp_begin(h, h->clk_apply);
        if (mode == RP_UPDATE_ORDERED) {
            const SortedBatch sb{h->srt_regret, h->srt_policy, h->srt_payoff, h->srt_expanded};
            hipLaunchKernelGGL(k_permute, dim3((unsigned)(((uint64_t)batch->n * h->A + 255) / 256)), dim3(256), 0, h->stream, b,
                               h->perm, batch->n, h->A, sb);
            HIP_TRY(hipMemsetAsync(h->hot + HOT_CAP, 0, 4, h->stream));
            hipLaunchKernelGGL(k_apply_ordered, dim3(group_blocks(batch->n)), dim3(256), 0, h->stream, p, b, sg, sb, h->hot,
                               h->hot + HOT_CAP, (uint32_t)HOT_CAP);
            if (h->R == RP_REGRET_DISCOUNTED || h->R == RP_REGRET_ASYMMETRIC)
                hipLaunchKernelGGL(k_apply_hot<true>, dim3(256), dim3(128), 0, h->stream, p, b, sg, sb, h->hot, h->hot + HOT_CAP,
                                   (uint32_t)HOT_CAP);
            else
                hipLaunchKernelGGL(k_apply_hot<false>, dim3(256), dim3(128), 0, h->stream, p, b, sg, sb, h->hot, h->hot + HOT_CAP,
                                   (uint32_t)HOT_CAP);
        } else {
            // a single rank's entries have distinct rows and are applied where they are produced (k_fold's arithmetic, apply_cell):
            // no k_fold launch (measured round 4: apply 0.118 -> 0.113 ms at 2^17 Decisions, profiles/r04_optin_ab.json)
            if ((rc = launch_summarize(h, p, b, sg, batch->n, h->entries, true, one))) return rc;
        }
        sp_end(h, h->clk_apply);
        HIP_TRY(hipGetLastError());
    }
    h->epoch += 1;  // CfrSampling::increment via Solver::advance (solver.rs:103-104)
    return RP_OK;
}

int rp_profile_sync(rp_profile* h) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_profile_sync: null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RP_OK;
}
int rp_profile_epoch(const rp_profile* h, uint64_t* epoch) {
    if (!h || !epoch) return rp::fail(RP_ERR_INVALID, "rp_profile_epoch: null argument");
    *epoch = h->epoch;
    return RP_OK;
}
int rp_profile_set_epoch(rp_profile* h, uint64_t epoch) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_profile_set_epoch: null handle");
    h->epoch = epoch;
    return RP_OK;
}

int rp_profile_get_rows(rp_profile* h, uint64_t n, const uint32_t* rows, rp_encounter* out) {
    if (!h || (n && (!rows || !out))) return rp::fail(RP_ERR_INVALID, "rp_profile_get_rows: null argument");
    if (n == 0) return RP_OK;
    for (uint64_t i = 0; i < n; ++i)
        if (rows[i] >= h->n_rows) return rp::fail(RP_ERR_INVALID, "rp_profile_get_rows: row %u out of range", rows[i]);
    HIP_TRY(hipSetDevice(h->device));
    uint32_t* d_rows = nullptr;
    rp_encounter* d_out = nullptr;
    HIP_TRY(hipMalloc(&d_rows, n * 4));
    HIP_TRY(hipMalloc(&d_out, n * h->A * sizeof(rp_encounter)));
    HIP_TRY(hipMemcpyAsync(d_rows, rows, n * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n * h->A + 255) / 256)), dim3(256), 0, h->stream, h->tab, h->A, d_rows, n, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_out, n * h->A * sizeof(rp_encounter), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    (void)hipFree(d_rows);
    (void)hipFree(d_out);
    return RP_OK;
}

int rp_profile_set_rows(rp_profile* h, uint64_t n, const uint32_t* rows, const rp_encounter* in) {
    if (!h || (n && (!rows || !in))) return rp::fail(RP_ERR_INVALID, "rp_profile_set_rows: null argument");
    if (n == 0) return RP_OK;
    for (uint64_t i = 0; i < n; ++i)
        if (rows[i] >= h->n_rows) return rp::fail(RP_ERR_INVALID, "rp_profile_set_rows: row %u out of range", rows[i]);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<float> row(4u * h->A);
    for (uint64_t i = 0; i < n; ++i) {  // a resynchronisation path (tests, hydrate): one small copy per row
        for (uint32_t a = 0; a < h->A; ++a) {
            const rp_encounter& e = in[i * h->A + a];
            row[a] = e.regret;
            row[h->A + a] = e.weight;
            row[2 * h->A + a] = e.payoff;
            memcpy(&row[3 * h->A + a], &e.visits, 4);
        }
        HIP_TRY(hipMemcpy(h->tab + (size_t)rows[i] * 4u * h->A, row.data(), row.size() * 4, hipMemcpyHostToDevice));
    }
    return RP_OK;
}

int rp_profile_set_stream(rp_profile* h, void* hip_stream) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_profile_set_stream: null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    if (hip_stream) {
        h->stream = reinterpret_cast<hipStream_t>(hip_stream);
        h->own_stream = false;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    return RP_OK;
}

int rp_profile_entry_bytes(const rp_profile* h, size_t* bytes) {
    if (!h || !bytes) return rp::fail(RP_ERR_INVALID, "rp_profile_entry_bytes: null argument");
    *bytes = entry_bytes_of(h);
    return RP_OK;
}

int rp_profile_summarize(rp_profile* h, const rp_decisions* batch, void* entries_dev, uint32_t* n_entries) {
    int rc = check_batch(h, batch, "rp_profile_summarize");
    if (rc) return rc;
    if (!entries_dev || !n_entries) return rp::fail(RP_ERR_INVALID, "rp_profile_summarize: null output");
    HIP_TRY(hipSetDevice(h->device));
    SparseParams p = make_params(h);
    if ((rc = composed_params(h, p))) return rc;
    *n_entries = 0;
    if (batch->n == 0) return RP_OK;
    const DevBatch b{batch->row, batch->n_actions, batch->expanded, batch->regret, batch->policy, batch->payoff};
    sp_begin(h, h->clk_sort);
    if ((rc = sort_and_segment(h, batch->row, batch->n, true))) return rc;
    sp_end(h, h->clk_sort);
    const Segments sg{h->seg_rows, h->seg_counts, h->seg_offsets, h->n_segs, h->perm};
    sp_begin(h, h->clk_apply);
    if ((rc = launch_summarize(h, p, b, sg, batch->n, reinterpret_cast<unsigned char*>(entries_dev)))) return rc;
    sp_end(h, h->clk_apply);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(n_entries, h->n_segs, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RP_OK;
}

int rp_profile_fold(rp_profile* h, const void* entries_dev, uint32_t n_entries) {
    if (!h || (n_entries && !entries_dev)) return rp::fail(RP_ERR_INVALID, "rp_profile_fold: null argument");
    HIP_TRY(hipSetDevice(h->device));
    SparseParams p = make_params(h);
    int rc = composed_params(h, p);
    if (rc) return rc;
    if (n_entries) {
        const unsigned char* ent = reinterpret_cast<const unsigned char*>(entries_dev);
        const uint32_t eb = (uint32_t)entry_bytes_of(h);
        if ((rc = ensure_capacity(h, n_entries))) return rc;
        sp_begin(h, h->clk_sort);
        hipLaunchKernelGGL(k_entry_rows, dim3((n_entries + 255) / 256), dim3(256), 0, h->stream, ent, eb, n_entries, h->ent_rows);
        if ((rc = sort_and_segment(h, h->ent_rows, n_entries))) return rc;
        sp_end(h, h->clk_sort);
        sp_begin(h, h->clk_apply);
        hipLaunchKernelGGL(k_fold, dim3(group_blocks(n_entries)), dim3(256), 0, h->stream, p, ent, eb, h->perm, h->seg_offsets,
                           h->seg_counts, h->n_segs);
        sp_end(h, h->clk_apply);
        HIP_TRY(hipGetLastError());
    }
    h->epoch += 1;
    return RP_OK;
}

int rp_profile_profile(rp_profile* h, int enable) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_profile_profile: null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    sp_drain(h->clk_sort);
    sp_drain(h->clk_apply);
    if (enable && !h->profiling) {
        h->clk_sort = SpClock{};
        h->clk_apply = SpClock{};
    }
    h->profiling = enable != 0;
    return RP_OK;
}
int rp_profile_kernel_time(rp_profile* h, const char* name, double* total_ms, uint64_t* launches) {
    if (!h || !name || !total_ms || !launches) return rp::fail(RP_ERR_INVALID, "rp_profile_kernel_time: null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    SpClock* c = std::string(name) == "sort" ? &h->clk_sort : (std::string(name) == "apply" ? &h->clk_apply : nullptr);
    if (!c) return rp::fail(RP_ERR_INVALID, "rp_profile_kernel_time: unknown kernel '%s'", name);
    sp_drain(*c);
    *total_ms = c->total_ms;
    *launches = c->launches;
    return RP_OK;
}

}  // extern "C"

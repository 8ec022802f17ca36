// mccfr.hip — external-sampling MCCFR on MI355X (gfx950): kernels + the rp_mccfr_* C ABI.
//
// Reference path (crates/mccfr): Solver::step (solver/solver.rs:96-105) = batch() (:225-250) then the
// sequential update_{regret,weight,payoff,visits} (:143-192).  MI355X mapping:
//
//   k_traverse   one LANE per sampled tree (trees of the table-driven games have <= a few dozen nodes):
//                TreeBuilder's explicit DFS stack (builder.rs:141-161) and the node list live in a
//                lane-interleaved HBM scratch (64 lanes = 64 consecutive dwords), the regret/strategy
//                tables are read through L1/L2.  Each walker infoset is evaluated exactly as CfrFlow::dfs /
//                recursed_value / ancestor_reach do (flow.rs:64-87,166-216) — same f32 operation order —
//                by a top-down reach sweep and a bottom-up value sweep over the contiguous subtree.
//   k_count / k_scan / k_compact   stable counting sort of the batch's Decisions into one tree-id-ordered
//                segment per infoset (slot map -> chunk counts -> offsets -> scatter), all CUs busy.
//   k_chain      one workgroup per infoset streams its segment through LDS tiles; one lane per table cell
//                applies the touches sequentially: the reference's order-dependent semantics
//                (R <- max(R*d + delta, floor) per touch, Welford payoff mean), bit for bit.
//   k_summarize / k_fold   the multi-GPU exchange: per-cell composed maps.
//
// Everything f32 is spelled with the primitives of include/rp_math.h and compiled -ffp-contract=off.
#include "mccfr_kernels.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

namespace rp {

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// device: profile reads (RefProf::{regret,weight} profile.rs:31-37; CfrFlow flow.rs:20-59)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float d_regret(const DevTables& t, uint32_t A, uint32_t info, uint32_t a) {
    return rp_maxf(t.regret[info * A + a], RP_EPSILON);
}
__device__ __forceinline__ float d_weight(const DevTables& t, uint32_t A, uint32_t info, uint32_t a) {
    return rp_maxf(t.weight[info * A + a], RP_EPSILON);
}
__device__ __forceinline__ float d_regret_denom(const DevTables& t, uint32_t A, uint32_t info, uint32_t n) {
    float s = 0.0f;
    for (uint32_t a = 0; a < n; ++a) s += d_regret(t, A, info, a);
    return s;
}
__device__ __forceinline__ float d_weight_denom(const DevTables& t, uint32_t A, uint32_t info, uint32_t n,
                                                float smoothing) {
    float s = 0.0f;
    for (uint32_t a = 0; a < n; ++a) s += d_weight(t, A, info, a);
    return s + smoothing;
}
__device__ __forceinline__ float d_sampling_weight(const DevTables& t, uint32_t A, uint32_t info, uint32_t a,
                                                   float denom, const StepParams& p) {
    return rp_maxf((d_weight(t, A, info, a) / p.temperature + p.smoothing) / denom, p.curiosity);
}
__device__ __forceinline__ float d_sampling_z(const DevTables& t, uint32_t A, uint32_t info, uint32_t n,
                                              float denom, const StepParams& p) {
    float z = 0.0f;
    for (uint32_t a = 0; a < n; ++a) z += d_sampling_weight(t, A, info, a, denom, p);
    return z;
}

// SamplingScheme::sample as a bitmask over child slots (sample/{mod,external,pruning,pluribus}.rs)
__device__ uint32_t d_sample_mask(const DevGame& g, const DevTables& t, const StepParams& p, uint64_t tree_id,
                                  uint32_t state, uint32_t turn, uint32_t n, uint32_t info, uint32_t off) {
    const uint32_t all = (1u << n) - 1u;
    const bool ref = p.ref_info != nullptr;
    if (n == 0) return 0;
    if (turn == RP_TURN_CHANCE) return 1u << d_draw_chance(p, ref, tree_id, state, n, info);  // a chance record carries chance_info in y
    if (turn != p.walker) {
        // weighted (external.rs:41-64): WeightedIndex over sampling_distribution().max(EPSILON)
        const float denom = d_weight_denom(t, g.A, info, n, p.smoothing);
        const float z = d_sampling_z(t, g.A, info, n, denom, p);
        float total = 0.0f;
        for (uint32_t a = 0; a < n; ++a)
            total += rp_maxf(d_sampling_weight(t, g.A, info, a, denom, p) / z, RP_EPSILON);
        const float x = d_draw_weight(p, ref, tree_id, info, total);
        float cum = 0.0f;
        uint32_t idx = 0;
        bool open = true;
        for (uint32_t a = 0; a + 1 < n; ++a) {
            cum += rp_maxf(d_sampling_weight(t, g.A, info, a, denom, p) / z, RP_EPSILON);
            open = open && (cum <= x);
            if (open) idx = a + 1;
        }
        return 1u << idx;
    }
    if (p.S == RP_SAMPLING_EXTERNAL) return all;
    if (p.S == RP_SAMPLING_PLURIBUS) {
        if (p.epoch < p.prune_warmup) return all;
        if (d_draw_coin(p, ref, tree_id, info) < p.prune_explore) return all;
    }
    uint32_t mask = 0;
    for (uint32_t a = 0; a < n; ++a) {
        bool keep = t.regret[info * g.A + a] > p.prune_threshold;
        if (p.S == RP_SAMPLING_PLURIBUS) {
            const uint4 c = g.states[g.children[off + a]];
            keep = keep || ((c.x & 0xffu) == RP_TURN_TERMINAL);
        }
        if (keep) mask |= 1u << a;
    }
    return mask ? mask : all;
}

__device__ __forceinline__ uint32_t lane_of() { return threadIdx.x & 63u; }

// Metrics (metrics/mod.rs:21-80; solver.rs:273): nodes / infos, one atomic per WAVE — a million lanes adding to the
// same two addresses would serialise in the L2 atomic unit
// The counters are STRIPED: 16 384 waves adding to ONE address serialise at its L2 channel (~9 ns per atomic: 0.29 ms
// of a 0.52 ms launch was spent there, found by ablation); stripe s owns its own 128-byte line, the host sums them.
#define METRIC_STRIPES 256u
#define METRIC_STRIDE 16u  // u64 per stripe (128 B)
__device__ __forceinline__ void count_metrics(const StepParams& p, uint32_t nn, uint32_t ndec, uint32_t err) {
    unsigned long long* c = p.counters + (size_t)(blockIdx.x % METRIC_STRIPES) * METRIC_STRIDE;
    if (__ballot(1) == ~0ull) {
        uint32_t a = nn, b = ndec;
        for (int d = 32; d > 0; d >>= 1) {
            a += __shfl_xor(a, d, 64);
            b += __shfl_xor(b, d, 64);
        }
        if (lane_of() == 0) {
            atomicAdd(&c[0], (unsigned long long)a);
            atomicAdd(&c[1], (unsigned long long)b);
        }
    } else {  // the ragged last wave
        atomicAdd(&c[0], (unsigned long long)nn);
        atomicAdd(&c[1], (unsigned long long)ndec);
    }
    if (err) atomicOr(&c[2], (unsigned long long)err);
}

#define META_PARENT(m) ((m)&0xffu)
#define META_EDGE(m) (((m) >> 8) & 0xffu)
#define META_PTYPE(m) (((m) >> 16) & 3u)
#define META_LEAF(m) (((m) >> 18) & 1u)
#define META_WALKER(m) (((m) >> 19) & 1u)
#define META_NACT(m) (((m) >> 24) & 0xffu)
#define NO_PARENT 0xffu

// ------------------------------------------------------------------------------------------------
// k_traverse: Solver::batch for one shard of trees
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_traverse(DevGame g, DevTables t, DevScratch sc, DevDecisions dc,
                                                  StepParams p) {
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= p.batch) return;
    const uint64_t tree_id = p.tree_base + lane;
    const size_t S = sc.stride;
    uint32_t err = 0;

    // ---- TreeBuilder::build (builder.rs:74-87,141-161): pop-last DFS -----------------------------
    uint32_t nn = 0, sp = 0;
    uint32_t cur_state = g.root;
    uint32_t cur_meta_in = NO_PARENT | (PT_NONE << 16);
    float cur_frel = 1.0f, cur_fsmp = 1.0f;
    for (;;) {
        const uint4 st = g.states[cur_state];
        const uint32_t turn = st.x & 0xffu, nch = (st.x >> 8) & 0xffu, info = st.y, off = st.z;
        const uint32_t me = nn;
        if (nn >= sc.maxn) {
            err |= ERR_NODE_CAPACITY;
            break;
        }
        const bool is_walker = turn == p.walker;
        uint32_t meta = cur_meta_in | ((nch == 0 ? 1u : 0u) << 18) | ((is_walker ? 1u : 0u) << 19) | (nch << 24);
        sc.n_meta[me * S + lane] = meta;
        sc.n_info[me * S + lane] = info;
        sc.n_frel[me * S + lane] = cur_frel;
        sc.n_fsmp[me * S + lane] = cur_fsmp;
        if (nch == 0) sc.n_pay[me * S + lane] = g.payoffs[off * g.n_players + p.walker];
        nn += 1;
        if (nch > 0) {
            const uint32_t mask = d_sample_mask(g, t, p, tree_id, cur_state, turn, nch, info, off);
            const bool chance = turn == RP_TURN_CHANCE;
            const uint32_t ptype = chance ? PT_CHANCE : (is_walker ? PT_WALKER : PT_OPP);
            float rd = 0.0f, denom = 0.0f, z = 0.0f;
            if (!chance) rd = d_regret_denom(t, g.A, info, nch);
            if (ptype == PT_OPP) {
                denom = d_weight_denom(t, g.A, info, nch, p.smoothing);
                z = d_sampling_z(t, g.A, info, nch, denom, p);
            }
            for (uint32_t k = 0; k < nch; ++k) {
                if (!((mask >> k) & 1u)) continue;
                if (sp >= sc.maxs) {
                    err |= ERR_STACK_CAPACITY;
                    break;
                }
                // reach factors of the edge parent->child (flow.rs:195-212)
                const float frel = chance ? 1.0f : d_regret(t, g.A, info, k) / rd;
                const float fsmp = ptype == PT_OPP ? d_sampling_weight(t, g.A, info, k, denom, p) / z : 1.0f;
                sc.s_state[sp * S + lane] = g.children[off + k];
                sc.s_meta[sp * S + lane] = me | (k << 8) | (ptype << 16);
                sc.s_frel[sp * S + lane] = frel;
                sc.s_fsmp[sp * S + lane] = fsmp;
                sp += 1;
            }
        }
        if (sp == 0 || err) break;
        sp -= 1;
        cur_state = sc.s_state[sp * S + lane];
        cur_meta_in = sc.s_meta[sp * S + lane];
        cur_frel = sc.s_frel[sp * S + lane];
        cur_fsmp = sc.s_fsmp[sp * S + lane];
    }

    // ---- Tree::partition + CfrFlow::dfs per walker infoset (tree.rs:88-98, flow.rs:64-87) --------
    uint32_t ndec = 0;
    if (!err) {
        for (uint32_t i = 0; i < nn; ++i) {
            const uint32_t mi = sc.n_meta[i * S + lane];
            if (!META_WALKER(mi) || META_LEAF(mi)) continue;
            const uint32_t info = sc.n_info[i * S + lane];
            bool head = true;
            for (uint32_t j = 0; j < i; ++j) {
                const uint32_t mj = sc.n_meta[j * S + lane];
                if (META_WALKER(mj) && !META_LEAF(mj) && sc.n_info[j * S + lane] == info) head = false;
            }
            if (!head) continue;
            if (ndec >= dc.maxdec) {
                err |= ERR_DEC_CAPACITY;
                break;
            }
            const uint32_t nact = META_NACT(mi);
            const uint32_t slot = ndec++;
            const size_t D = dc.stride;
            const float rd = d_regret_denom(t, g.A, info, nact);
            for (uint32_t a = 0; a < nact; ++a) {  // policy_vector = iterated_distribution (profile.rs:47-51)
                dc.policy[(slot * g.A + a) * D + lane] = d_regret(t, g.A, info, a) / rd;
                dc.regret[(slot * g.A + a) * D + lane] = 0.0f;
            }
            float payoff = 0.0f;
            uint32_t expanded = 0;
            for (uint32_t j = i; j < nn; ++j) {  // span in ascending node index
                const uint32_t mj = sc.n_meta[j * S + lane];
                if (!META_WALKER(mj) || META_LEAF(mj) || sc.n_info[j * S + lane] != info) continue;
                // top-down: reach products below root j, starting at 1 on j's children (flow.rs:72)
                uint32_t end = j;
                for (uint32_t n = j + 1; n < nn; ++n) {
                    const uint32_t mn = sc.n_meta[n * S + lane];
                    const uint32_t par = META_PARENT(mn);
                    if (par < j) break;
                    float rel = 1.0f, smp = 1.0f;
                    if (par != j) {
                        rel = sc.n_rel[par * S + lane] * sc.n_frel[n * S + lane];
                        smp = sc.n_smp[par * S + lane] * sc.n_fsmp[n * S + lane];
                    }
                    sc.n_rel[n * S + lane] = rel;
                    sc.n_smp[n * S + lane] = smp;
                    sc.n_acc[n * S + lane] = 0.0f;
                    end = n;
                }
                // bottom-up: children were created in reverse choices() order, so descending node index
                // adds them to the parent's sum in choices() order, as node.edges() does (node.rs:103-107)
                uint32_t kids = 0;
                for (uint32_t n = end; n > j; --n) {
                    const uint32_t mn = sc.n_meta[n * S + lane];
                    const float v = META_LEAF(mn)
                                        ? sc.n_rel[n * S + lane] / sc.n_smp[n * S + lane] * sc.n_pay[n * S + lane]
                                        : sc.n_acc[n * S + lane];
                    const uint32_t par = META_PARENT(mn);
                    if (par == j) {
                        sc.t_v[META_EDGE(mn) * S + lane] = v;
                        kids |= 1u << META_EDGE(mn);
                    } else {
                        sc.n_acc[par * S + lane] = sc.n_acc[par * S + lane] + v;
                    }
                }
                // ancestor_reach (flow.rs:166-174): upward over opponent decision ancestors
                float cf = 1.0f, sm = 1.0f;
                for (uint32_t n = j;;) {
                    const uint32_t mn = sc.n_meta[n * S + lane];
                    const uint32_t par = META_PARENT(mn);
                    if (par == NO_PARENT) break;
                    if (META_PTYPE(mn) == PT_OPP) {
                        cf = cf * sc.n_frel[n * S + lane];
                        sm = sm * sc.n_fsmp[n * S + lane];
                    }
                    n = par;
                }
                const float reach = cf / sm;
                float ev = 0.0f;
                for (uint32_t a = 0; a < nact; ++a) {
                    if (!((kids >> a) & 1u)) continue;
                    const float v = reach * sc.t_v[a * S + lane];
                    sc.t_v[a * S + lane] = v;
                }
                for (uint32_t a = 0; a < nact; ++a) {
                    if (!((kids >> a) & 1u)) continue;
                    ev += d_regret(t, g.A, info, a) / rd * sc.t_v[a * S + lane];
                }
                payoff += ev;
                for (uint32_t a = 0; a < nact; ++a) {
                    if (!((kids >> a) & 1u)) continue;
                    const size_t k = (slot * g.A + a) * D + lane;
                    dc.regret[k] = dc.regret[k] + (sc.t_v[a * S + lane] - ev);
                }
                expanded |= kids;
            }
            dc.info[slot * D + lane] = info;
            dc.mask[slot * D + lane] = expanded;
            dc.payoff[slot * D + lane] = payoff;
            if (dc.slotmap) dc.slotmap[(size_t)info * D + lane] = (uint8_t)(slot + 1);
        }
    }
    dc.ndec[lane] = (uint8_t)ndec;
    // Metrics: nodes / infos (metrics/mod.rs:21-80; solver.rs:273)
    count_metrics(p, nn, ndec, err);
}

// ------------------------------------------------------------------------------------------------
// k_prepare_infos: everything a node needs from its infoset, computed ONCE per epoch per infoset instead of at
// every visited node: regret-matching policy sigma(a) = regret(a)/sum (profile.rs:47-51), the normalised sampling
// distribution q(a) (flow.rs:33-42), the cumulative weights WeightedIndex draws from (external.rs:52-62) and the
// regret-based pruning mask (pruning.rs:57-63).  Same expressions, same order => same bits as the per-node code.
// ------------------------------------------------------------------------------------------------
struct DevInfoTab {
    float* sigma;    // [n_infos][A]
    float* q;        // [n_infos][A]
    float* cum;      // [n_infos][A] inclusive cumulative of max(q, EPSILON)
    float* total;    // [n_infos]
    uint32_t* keep;  // [n_infos] edges with cum_regret > prune_threshold
    float2* sq;      // [n_infos][A] (sigma, q) side by side: one load per edge in the traversal's sweeps
};

__device__ __forceinline__ void prepare_one(const DevGame& g, const DevTables& t, const StepParams& p, const DevInfoTab& it,
                                            uint32_t info) {
    const uint32_t A = g.A, n = g.info_actions[info];
    const float rd = d_regret_denom(t, A, info, n);
    const float denom = d_weight_denom(t, A, info, n, p.smoothing);
    const float z = d_sampling_z(t, A, info, n, denom, p);
    float total = 0.0f;
    uint32_t keep = 0;
    for (uint32_t a = 0; a < n; ++a) {
        it.sigma[info * A + a] = d_regret(t, A, info, a) / rd;
        const float qa = d_sampling_weight(t, A, info, a, denom, p) / z;
        it.q[info * A + a] = qa;
        it.sq[info * A + a] = make_float2(it.sigma[info * A + a], qa);
        total += rp_maxf(qa, RP_EPSILON);
        it.cum[info * A + a] = total;
        if (t.regret[info * A + a] > p.prune_threshold) keep |= 1u << a;
    }
    it.total[info] = total;
    it.keep[info] = keep;
}
__global__ void k_prepare_infos(DevGame g, DevTables t, StepParams p, DevInfoTab it) {
    const uint32_t info = blockIdx.x * blockDim.x + threadIdx.x;
    if (info >= g.n_infos) return;
    prepare_one(g, t, p, it, info);
}

// reference-seed mode: DefaultHasher after t.hash() and info.hash() for every infoset and every in-tree chance info
// (flow.rs:290-293); a node continues with node.seed().hash() and finish() (rp_ref_seed_finish).  Depends on the epoch: every step.
__global__ void k_prepare_ref(const rp_hash_stream* infos, uint32_t n_infos, const rp_hash_stream* chance, uint32_t n_chance,
                              uint64_t epoch, rp_sip_mid* info_mid, rp_sip_mid* chance_mid) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_infos) rp_ref_seed_prefix(&info_mid[i], epoch, infos[i].bytes, infos[i].len);
    else if (i < n_infos + n_chance) rp_ref_seed_prefix(&chance_mid[i - n_infos], epoch, chance[i - n_infos].bytes, chance[i - n_infos].len);
}

// SamplingScheme::sample with the per-infoset tables
__device__ __forceinline__ uint32_t d_sample_mask_tab(const DevGame& g, const DevInfoTab& it, const StepParams& p,
                                                      uint64_t tree_id, uint32_t state, uint32_t turn, uint32_t n,
                                                      uint32_t info, uint32_t off) {
    const uint32_t all = (1u << n) - 1u;
    const bool ref = p.ref_info != nullptr;
    if (turn == RP_TURN_CHANCE) return 1u << d_draw_chance(p, ref, tree_id, state, n, info);  // a chance record carries chance_info in y
    if (turn != p.walker) {
        const float x = d_draw_weight(p, ref, tree_id, info, it.total[info]);
        uint32_t idx = 0;
        bool open = true;
        for (uint32_t a = 0; a + 1 < n; ++a) {
            open = open && (it.cum[info * g.A + a] <= x);
            if (open) idx = a + 1;
        }
        return 1u << idx;
    }
    if (p.S == RP_SAMPLING_EXTERNAL) return all;
    if (p.S == RP_SAMPLING_PLURIBUS) {
        if (p.epoch < p.prune_warmup) return all;
        if (d_draw_coin(p, ref, tree_id, info) < p.prune_explore) return all;
    }
    uint32_t mask = it.keep[info] & all;
    if (p.S == RP_SAMPLING_PLURIBUS) {
        for (uint32_t a = 0; a < n; ++a) {
            if ((g.kids[off + a].x & 0xffu) == RP_TURN_TERMINAL) mask |= 1u << a;
        }
    }
    return mask ? mask : all;
}

// ------------------------------------------------------------------------------------------------
// k_traverse_lds: the same traversal with the per-tree scratch in LDS instead of HBM.
//
// The HBM variant moves ~1.1 GB per 262 144-tree launch (profiles/r01_mccfr_hbm_traffic.json) against ~72 MB of
// algorithmic bytes: the multi-pass evaluation re-reads the node list.  Here a node is 4 dwords
// (meta | frel | fsmp | value) in a lane-interleaved LDS array (bank = lane: conflict free), the leaf stack 4
// dwords per entry; reach products of a leaf are rebuilt by walking its (<= 10 node) path instead of being stored.
// One wave per workgroup; 4*maxn + 4*maxs + A dwords per lane (Leduc: 472 B/lane, 30 KB/wave, 5 waves/CU).
// Used when the game fits: <= 62 nodes per sampled tree, depth <= 10, <= 8191 infosets, <= 16 actions.
// ------------------------------------------------------------------------------------------------
#define LM_PARENT(m) ((m)&63u)
#define LM_EDGE(m) (((m) >> 6) & 15u)
#define LM_PTYPE(m) (((m) >> 10) & 3u)
#define LM_LEAF(m) (((m) >> 12) & 1u)
#define LM_WALKER(m) (((m) >> 13) & 1u)
#define LM_ISLOT(m) (((m) >> 14) & 31u)  // internal nodes: rank among the internal nodes (their reach-prefix slot)
#define LM_INFO(m) ((m) >> 19)
#define LM_NO_PARENT 63u

// TVREG:  at most 4 actions, the per-action values of a root live in registers, not LDS.
// The per-infoset sigma / q tables are read through L1 (a per-wave LDS copy measured slower on Leduc: 0.57 vs 0.54 ms per 2^20 trees).
// A node is TWO dwords (meta, value): the reach factor of its incoming edge is not stored but looked up as
// table[infoset(parent)][edge] whenever a sweep needs it.  Leduc: 78 dwords per lane = 8 waves/CU.
template <bool TVREG>
__global__ __launch_bounds__(64) void k_traverse_lds(DevGame g, DevInfoTab it, DevDecisions dc, StepParams p, uint32_t maxn,
                                                     uint32_t maxs, uint32_t maxi) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t ln = threadIdx.x;
    const uint32_t lane = blockIdx.x * 64 + ln;
    uint32_t* nm = lds;                                              // [maxn][64] meta
    float* nv = reinterpret_cast<float*>(nm + (size_t)maxn * 64);    // [maxn][64] leaf: payoff; internal: child-value sum
    float* tv = nv + (size_t)maxn * 64;                              // [A][64] (absent when TVREG)
    float tvr[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    auto tv_set = [&](uint32_t e, float v) {
        if (TVREG) {
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) tvr[q] = e == q ? v : tvr[q];
        } else {
            tv[e * 64 + ln] = v;
        }
    };
    auto tv_get = [&](uint32_t e) -> float {
        if (TVREG) {
            float r = tvr[0];
#pragma unroll
            for (uint32_t q = 1; q < 4; ++q) r = e == q ? tvr[q] : r;
            return r;
        }
        return tv[e * 64 + ln];
    };
    // build phase: the DFS stack; evaluation phase: per-root reach prefixes of internal nodes (same storage)
    uint32_t* ss = reinterpret_cast<uint32_t*>(tv + (TVREG ? 0 : (size_t)g.A * 64));  // [maxs][4][64] stack: record x | meta << 16, y, z, w
    // reach prefixes exist for INTERNAL nodes only: slot = rank of the node among the internal nodes (popcount of a
    // register mask)
    float* xr = reinterpret_cast<float*>(ss);                        // [maxi][64] relative reach root's child -> node
    float* xs = xr + (size_t)maxi * 64;                              // [maxi][64] sampling reach root's child -> node
    const uint32_t cells = g.n_infos * g.A;
    auto SIG = [&](uint32_t e) -> float { return it.sigma[e]; };
    if (lane >= p.batch) return;
    const uint64_t tree_id = p.tree_base + lane;
    uint32_t err = 0;
#define L(arr, slot) arr[(slot)*64 + ln]
#define STK(e, f) ss[((e)*4u + (f)) * 64u + ln]
    // reach factors of the edge into a node (meta mn): sigma / q of the parent's infoset at the node's edge
    // (sigma, q) of the edge into a node: (1, 1) below chance, (sigma, 1) below the walker, (sigma, q) below an opponent
    // mp: the meta of mn's parent
    auto f_of = [&](uint32_t mn, uint32_t mp) -> float2 {
        const uint32_t pt = LM_PTYPE(mn);
        if (pt != PT_WALKER && pt != PT_OPP) return make_float2(1.0f, 1.0f);
        const uint32_t e = LM_INFO(mp) * g.A + LM_EDGE(mn);
        float2 f = it.sq[e];
        if (pt != PT_OPP) f.y = 1.0f;
        return f;
    };
    // the same lookup issued AHEAD of its use, for metas that may lie past the subtree (stale LDS): a bounds-checked
    // buffer load (out of range -> 0) of the raw (sigma, q) pair; f_fix applies the parent-type rule once it is used
    const __amdgpu_buffer_rsrc_t sq_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(it.sq), 0, (int)(cells * 8u), 0x00020000);
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    auto f_issue = [&](uint32_t mn, uint32_t mp) -> float2 {
        const uint32_t e = LM_INFO(mp) * g.A + LM_EDGE(mn);
        const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(sq_rsrc, (int)(e * 8u), 0, 0);
        return make_float2(rp_u2f(raw.x), rp_u2f(raw.y));
    };
    auto f_fix = [&](uint32_t mn, float2 f) -> float2 {
        const uint32_t pt = LM_PTYPE(mn);
        if (pt != PT_WALKER && pt != PT_OPP) return make_float2(1.0f, 1.0f);
        if (pt != PT_OPP) f.y = 1.0f;
        return f;
    };

    // ---- TreeBuilder::build (builder.rs:74-87,141-161): pop-last DFS -----------------------------
    // A node arrives as its RECORD (DevGame::kids): the records of all sampled children are requested together
    // when their parent is expanded, so a tree pays one L2 round trip per internal node, not two or three per node.
    // The last child pushed is the next node popped: it is carried in registers instead of through the stack.
    uint32_t nn = 0, sp = 0;
    uint4 rec = g.root_rec;
    uint32_t cur_in = LM_NO_PARENT | (PT_NONE << 10);
    unsigned long long wmask = 0;  // walker decision nodes
    uint32_t n_int = 0;            // internal nodes so far
    for (;;) {
        const uint32_t turn = rec.x & 0xffu, nch = (rec.x >> 8) & 0xffu, info = rec.y, off = rec.z;
        const uint32_t me = nn;
        if (nn >= maxn) {
            err |= ERR_NODE_CAPACITY;
            break;
        }
        const bool is_walker = turn == p.walker;
        L(nm, me) = cur_in | ((nch == 0 ? 1u : 0u) << 12) | ((is_walker ? 1u : 0u) << 13) | ((n_int & 31u) << 14) |
                    ((turn < RP_TURN_CHANCE ? info : 0u) << 19);
        nn += 1;
        if (nch > 0) {
            n_int += 1;
            if (is_walker) wmask |= 1ull << me;
            uint32_t mask = d_sample_mask_tab(g, it, p, tree_id, rec.w, turn, nch, info, off);
            const bool chance = turn == RP_TURN_CHANCE;
            const uint32_t ptype = chance ? PT_CHANCE : (is_walker ? PT_WALKER : PT_OPP);
            const uint32_t last = 31u - (uint32_t)__builtin_clz(mask);
            mask &= ~(1u << last);
            rec = g.kids[off + last];
            while (mask) {
                const uint32_t k = (uint32_t)__builtin_ctz(mask);
                mask &= mask - 1u;
                if (sp >= maxs) {
                    err |= ERR_STACK_CAPACITY;
                    break;
                }
                const uint4 kr = g.kids[off + k];
                STK(sp, 0) = kr.x | ((me | (k << 6) | (ptype << 10)) << 16);
                STK(sp, 1) = kr.y;
                STK(sp, 2) = kr.z;
                STK(sp, 3) = kr.w;
                sp += 1;
            }
            if (err) break;
            cur_in = me | (last << 6) | (ptype << 10);
            continue;
        }
        L(nv, me) = g.n_players == 2 ? rp_u2f(p.walker == 0 ? rec.y : rec.z) : g.payoffs[off * g.n_players + p.walker];
        if (sp == 0) break;
        sp -= 1;
        const uint32_t xm = STK(sp, 0);
        rec = make_uint4(xm & 0xffffu, STK(sp, 1), STK(sp, 2), STK(sp, 3));
        cur_in = xm >> 16;
    }
#undef STK

    // ---- Tree::partition + CfrFlow::dfs per walker infoset (tree.rs:88-98, flow.rs:64-87) --------
    const bool fuse = g.A <= 2;  // every node has at most two children
    uint32_t ndec = 0;
    if (n_int > maxi || n_int > 32u) err |= ERR_NODE_CAPACITY;
    if (!err) {
        unsigned long long todo = wmask;
        while (todo) {
            const uint32_t i = (uint32_t)__builtin_ctzll(todo);  // head of the next infoset span
            const uint32_t mi = L(nm, i);
            const uint32_t info = LM_INFO(mi);
            if (ndec >= dc.maxdec) {
                err |= ERR_DEC_CAPACITY;
                break;
            }
            const uint32_t nact = g.info_actions[info];
            const uint32_t slot = ndec++;
            const size_t D = dc.stride;
            float payoff = 0.0f;
            uint32_t expanded = 0;
            unsigned long long span = todo;
            while (span) {  // roots of the span in ascending node index
                const uint32_t j = (uint32_t)__builtin_ctzll(span);
                span &= span - 1ull;
                if (j != i && LM_INFO(L(nm, j)) != info) continue;
                todo &= ~(1ull << j);
                // top-down over the (contiguous) subtree of j: reach products from j's child (flow.rs:195-212),
                // which start at 1 there; internal nodes also start their child-value sum at 0
                // With at most two children per node (fuse) a sum of child values does not depend on the order of its
                // additions (0 + x = x, x + y = y + x exactly), so a leaf hands its value to its parent right here and
                // the bottom-up sweep only moves the internal nodes' sums: one factor lookup per node instead of two.
                // The sweep is a three-stage software pipeline: while node n is processed, the factor pair of node n + 1,
                // the parent meta of node n + 2 and the meta of node n + 3 are in flight (a lone wave per SIMD pays every
                // LDS / L1 round trip in full otherwise).  Stages may run past the subtree: they only read.
                uint32_t end = j;
                uint32_t kids = 0;
                uint32_t mnA = L(nm, j + 1), mnB = L(nm, j + 2), mnC = L(nm, j + 3);
                uint32_t mpA = L(nm, LM_PARENT(mnA)), mpB = L(nm, LM_PARENT(mnB));
                float2 fA = f_issue(mnA, mpA);
                for (uint32_t n = j + 1; n < nn; ++n) {
                    const uint32_t mn = mnA, mp = mpA;
                    const float2 fraw = fA;
                    fA = f_issue(mnB, mpB);
                    mpA = mpB;
                    mpB = L(nm, LM_PARENT(mnC));
                    mnA = mnB;
                    mnB = mnC;
                    mnC = L(nm, n + 3);
                    const uint32_t par = LM_PARENT(mn);
                    if (par < j) break;
                    end = n;
                    const bool leaf = LM_LEAF(mn);
                    if (leaf && !fuse) continue;
                    float rel = 1.0f, smp = 1.0f;
                    if (par != j) {
                        const uint32_t ps = LM_ISLOT(mp);
                        const float2 f = f_fix(mn, fraw);
                        rel = L(xr, ps) * f.x;
                        smp = L(xs, ps) * f.y;
                    }
                    if (leaf) {
                        const float v = rel / smp * L(nv, n);
                        if (par == j) {
                            tv_set(LM_EDGE(mn), v);
                            kids |= 1u << LM_EDGE(mn);
                        } else {
                            L(nv, par) = L(nv, par) + v;
                        }
                        continue;
                    }
                    const uint32_t ns = LM_ISLOT(mn);
                    L(xr, ns) = rel;
                    L(xs, ns) = smp;
                    L(nv, n) = 0.0f;
                }
                // bottom-up: descending node index adds children in choices() order (node.rs:103-107)
                uint32_t mn_prev = L(nm, end);
                float v_prev = L(nv, end);
                for (uint32_t n = end; n > j; --n) {
                    const uint32_t mn = mn_prev;
                    const uint32_t par = LM_PARENT(mn);
                    float v = v_prev;
                    mn_prev = L(nm, n - 1);  // one node ahead; its value is patched below if this node is its child
                    v_prev = L(nv, n - 1);
                    if (fuse && LM_LEAF(mn)) continue;  // already with its parent
                    if (LM_LEAF(mn)) {
                        float rel = 1.0f, smp = 1.0f;
                        if (par != j) {
                            const uint32_t mp = L(nm, par), ps = LM_ISLOT(mp);
                            const float2 f = f_of(mn, mp);
                            rel = L(xr, ps) * f.x;
                            smp = L(xs, ps) * f.y;
                        }
                        v = rel / smp * v;
                    }
                    if (par == j) {
                        tv_set(LM_EDGE(mn), v);
                        kids |= 1u << LM_EDGE(mn);
                    } else {
                        const float sum = (par == n - 1 ? v_prev : L(nv, par)) + v;
                        L(nv, par) = sum;
                        if (par == n - 1) v_prev = sum;
                    }
                }
                // ancestor_reach (flow.rs:166-174)
                float cf = 1.0f, sm_ = 1.0f;
                for (uint32_t mn = L(nm, j);;) {
                    const uint32_t par = LM_PARENT(mn);
                    if (par == LM_NO_PARENT) break;
                    const uint32_t mp = L(nm, par);
                    if (LM_PTYPE(mn) == PT_OPP) {
                        const float2 f = f_of(mn, mp);
                        cf = cf * f.x;
                        sm_ = sm_ * f.y;
                    }
                    mn = mp;
                }
                const float reach = cf / sm_;
                float ev = 0.0f;
                for (uint32_t a = 0; a < nact; ++a) {
                    if (!((kids >> a) & 1u)) continue;
                    const float u = reach * tv_get(a);
                    tv_set(a, u);
                    ev += SIG(info * g.A + a) * u;
                }
                payoff += ev;
                for (uint32_t a = 0; a < nact; ++a) {
                    if (!((kids >> a) & 1u)) continue;
                    const size_t k = (slot * g.A + a) * D + lane;
                    // first root of the span writes, later roots accumulate (0 + x = x exactly)
                    const float prev = (expanded >> a) & 1u ? dc.regret[k] : 0.0f;
                    dc.regret[k] = prev + (tv_get(a) - ev);
                }
                expanded |= kids;
            }
            for (uint32_t a = 0; a < nact; ++a) {  // policy_vector = iterated_distribution (profile.rs:47-51)
                dc.policy[(slot * g.A + a) * D + lane] = SIG(info * g.A + a);
                if (!((expanded >> a) & 1u)) dc.regret[(slot * g.A + a) * D + lane] = 0.0f;
            }
            dc.info[slot * D + lane] = info;
            dc.mask[slot * D + lane] = expanded;
            dc.payoff[slot * D + lane] = payoff;
            if (dc.slotmap) dc.slotmap[(size_t)info * D + lane] = (uint8_t)(slot + 1);
        }
    }
#undef L
    dc.ndec[lane] = (uint8_t)ndec;
    count_metrics(p, nn, ndec, err);
}

// ------------------------------------------------------------------------------------------------
// schedules (regret/*.rs, policy/*.rs)
// ------------------------------------------------------------------------------------------------
// d_regret_gain / d_weight_learn / regret_floor_of live in mccfr_kernels.hpp (shared with sparse.hip)

// ------------------------------------------------------------------------------------------------
// Update pipeline (Solver::update_{regret,weight,payoff,visits}, solver.rs:96-105,143-192):
//   k_count    per (infoset, 1024-tree chunk): how many trees of the chunk produced Decisions for it
//   k_scan     per infoset: exclusive scan of the chunk counts -> offsets, segment length
//   k_compact  scatter the Decisions into ONE tree-id-ordered segment per infoset (stable counting sort)
//   k_chain    per infoset: stream the segment through LDS tiles and apply the touches sequentially,
//              one lane per table cell (the reference's order-dependent semantics, bit for bit)
// ------------------------------------------------------------------------------------------------
#define CH_TREES 256u     // trees per compaction chunk == RP_COMPOSE_CHUNK (a block of the composed update)
#define CH_THREADS 256u   // small-game kernels: one tree per thread
#define SLOT_THREADS (CH_TREES / 4u)  // slot-map kernels: four slot-map bytes per thread

struct DevSorted {
    float* rw;         // [cap][2A]  per Decisions: regret delta a=0..A-1, then weight delta a=0..A-1
    uint32_t* mask;    // [cap]      edges present in the regret vector
    float* payoff;     // [cap]
    uint32_t* counts;  // [n_infos][n_chunks]
    uint32_t* offs;    // [n_infos][n_chunks]
    uint32_t* total;   // [n_infos]  segment length
    uint32_t n_chunks;
};

__device__ __forceinline__ uint32_t chunk_slots(const DevDecisions& dc, uint32_t info, uint32_t t0, uint32_t batch) {
    uint32_t slots = 0;
    if (t0 + 4 <= batch) {
        slots = *reinterpret_cast<const uint32_t*>(&dc.slotmap[(size_t)info * dc.stride + t0]);
    } else {
        for (uint32_t k = 0; k < 4; ++k)
            if (t0 + k < batch) slots |= (uint32_t)dc.slotmap[(size_t)info * dc.stride + t0 + k] << (8 * k);
    }
    return slots;
}
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t v) {
    return ((v & 0xffu) != 0) + ((v & 0xff00u) != 0) + ((v & 0xff0000u) != 0) + ((v & 0xff000000u) != 0);
}
// block-wide exclusive scan over the blockDim.x threads in thread order; returns (exclusive prefix, total)
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t* wave_tot, uint32_t* total) {
    const uint32_t tid = threadIdx.x;
    uint32_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if ((int)(tid & 63) >= d) incl += o;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; ++w) {
        const uint32_t c = wave_tot[w];
        if (w < (tid >> 6)) wbase += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return wbase + incl - v;
}

__global__ __launch_bounds__(SLOT_THREADS) void k_count(DevGame g, DevDecisions dc, DevSorted so, StepParams p) {
    __shared__ uint32_t wave_tot[CH_THREADS / 64];
    const uint32_t info = blockIdx.y, chunk = blockIdx.x;
    if (g.info_player[info] != p.walker) return;
    const uint32_t t0 = chunk * CH_TREES + threadIdx.x * 4;
    uint32_t cnt = nonzero_bytes(chunk_slots(dc, info, t0, p.batch));
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        for (uint32_t w = 0; w < blockDim.x / 64; ++w) c += wave_tot[w];
        so.counts[(size_t)info * so.n_chunks + chunk] = c;
    }
}

__global__ __launch_bounds__(CH_THREADS) void k_scan(DevGame g, DevSorted so, StepParams p) {
    __shared__ uint32_t wave_tot[CH_THREADS / 64];
    const uint32_t info = blockIdx.x;
    if (g.info_player[info] != p.walker) {
        if (threadIdx.x == 0) so.total[info] = 0;
        return;
    }
    uint32_t carry = 0;
    for (uint32_t base = 0; base < so.n_chunks; base += CH_THREADS) {
        const uint32_t c = base + threadIdx.x;
        const uint32_t v = c < so.n_chunks ? so.counts[(size_t)info * so.n_chunks + c] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exscan(v, wave_tot, &tot);
        if (c < so.n_chunks) so.offs[(size_t)info * so.n_chunks + c] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) so.total[info] = carry;
}

__global__ __launch_bounds__(SLOT_THREADS) void k_compact(DevGame g, DevDecisions dc, DevSorted so, StepParams p) {
    __shared__ uint32_t wave_tot[CH_THREADS / 64];
    __shared__ uint32_t sh_base;
    const uint32_t info = blockIdx.y, chunk = blockIdx.x;
    if (g.info_player[info] != p.walker) return;
    // segment base = sum of the lengths of all lower infosets
    uint32_t part = 0;
    for (uint32_t i = threadIdx.x; i < info; i += blockDim.x) part += so.total[i];
    for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t b0 = so.offs[(size_t)info * so.n_chunks + chunk];
        for (uint32_t w = 0; w < blockDim.x / 64; ++w) b0 += wave_tot[w];
        sh_base = b0;
    }
    __syncthreads();
    const uint32_t A = g.A, nact = g.info_actions[info];
    const uint32_t t0 = chunk * CH_TREES + threadIdx.x * 4;
    const uint32_t slots = chunk_slots(dc, info, t0, p.batch);
    uint32_t tot;
    uint32_t rank = block_exscan(nonzero_bytes(slots), wave_tot, &tot);
    const float tf = (float)p.epoch;
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t sl = (slots >> (8 * k)) & 0xffu;
        if (!sl) continue;
        const uint32_t tree = t0 + k, slot = sl - 1;
        const size_t pos = (size_t)sh_base + rank;
        for (uint32_t a = 0; a < A; ++a) {
            float rd = 0.0f, wd = 0.0f;
            if (a < nact) {
                rd = dc.regret[(slot * A + a) * dc.stride + tree];
                const float sg = dc.policy[(slot * A + a) * dc.stride + tree];
                // WeightSchedule::accumulate's immediate term (policy/{linear,quadratic}.rs): sigma * t, sigma * t * t
                wd = p.W == RP_WEIGHT_LINEAR ? sg * tf : (p.W == RP_WEIGHT_QUADRATIC ? sg * tf * tf : sg);
            }
            so.rw[pos * 2 * A + a] = rd;
            so.rw[pos * 2 * A + A + a] = wd;
        }
        so.mask[pos] = dc.mask[slot * dc.stride + tree];
        so.payoff[pos] = dc.payoff[slot * dc.stride + tree];
        rank += 1;
    }
}

// ---- games whose per-chunk bitmap fits in LDS (56 B per infoset): the same stable counting sort without the
// per-infoset slot map in HBM --------
// One workgroup per chunk of CH_TREES trees.  A tree's Decisions are marked in an LDS bitmap [infoset][tree]; the
// rank of a Decisions inside its (chunk, infoset) bucket — its place in tree-id order — is a prefix popcount of that
// bitmap row.  Every Decisions is read once; nothing is scanned per infoset.
static_assert(CH_TREES == RP_COMPOSE_CHUNK, "a block of the composed update is one compaction chunk of trees");
#define SM_WORDS (CH_TREES / 32u)
#define CM_PASSES 4u  // k_chunk_maps: infosets per thread; 4 * 256 infosets * 56 B is past its 64 KB LDS budget
__device__ __forceinline__ void chunk_bitmap(const DevDecisions& dc, uint32_t n_infos, uint32_t chunk, uint32_t batch,
                                             uint32_t* bits) {
    for (uint32_t e = threadIdx.x; e < n_infos * SM_WORDS; e += CH_THREADS) bits[e] = 0;
    __syncthreads();
    for (uint32_t lt = threadIdx.x; lt < CH_TREES; lt += CH_THREADS) {
        const uint32_t tree = chunk * CH_TREES + lt;  // coalesced over threads
        if (tree >= batch) continue;
        const uint32_t nd = dc.ndec[tree];
        for (uint32_t slot = 0; slot < nd; ++slot)
            atomicOr(&bits[dc.info[slot * dc.stride + tree] * SM_WORDS + (lt >> 5)], 1u << (lt & 31u));
    }
    __syncthreads();
}
// exclusive scan of in[0..n) into out[0..n) (both LDS), any n, by the whole workgroup
__device__ __forceinline__ void lds_exscan(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* wave_tot) {
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < n; b0 += blockDim.x) {
        const uint32_t i = b0 + threadIdx.x;
        uint32_t tot;
        const uint32_t ex = block_exscan(i < n ? in[i] : 0u, wave_tot, &tot);
        if (i < n) out[i] = carry + ex;
        carry += tot;
    }
    __syncthreads();
}
__global__ __launch_bounds__(CH_THREADS) void k_count_small(DevGame g, DevDecisions dc, DevSorted so, StepParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sm_lds[];
    uint32_t* bits = sm_lds;  // [n_infos][SM_WORDS]
    const uint32_t chunk = blockIdx.x;
    chunk_bitmap(dc, g.n_infos, chunk, p.batch, bits);
    for (uint32_t info = threadIdx.x; info < g.n_infos; info += CH_THREADS) {
        uint32_t c = 0;
        for (uint32_t w = 0; w < SM_WORDS; ++w) c += __popc(bits[info * SM_WORDS + w]);
        so.counts[(size_t)info * so.n_chunks + chunk] = c;
    }
}
__global__ __launch_bounds__(CH_THREADS) void k_compact_small(DevGame g, DevDecisions dc, DevSorted so, StepParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sm_lds[];
    __shared__ uint32_t wave_tot[CH_THREADS / 64];
    const uint32_t NI = g.n_infos;
    uint32_t* bits = sm_lds;                                        // [NI][SM_WORDS]
    uint32_t* base = bits + NI * SM_WORDS;                          // [NI] first position of the (chunk, infoset) bucket
    uint32_t* tots = base + NI;                                     // [NI] segment lengths
    uint16_t* pre = reinterpret_cast<uint16_t*>(tots + NI);         // [NI][SM_WORDS] trees before word w that visited the infoset
    const uint32_t chunk = blockIdx.x;
    chunk_bitmap(dc, g.n_infos, chunk, p.batch, bits);
    for (uint32_t info = threadIdx.x; info < g.n_infos; info += CH_THREADS) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < SM_WORDS; ++w) {
            pre[info * SM_WORDS + w] = (uint16_t)run;
            run += __popc(bits[info * SM_WORDS + w]);
        }
    }
    // bucket base = sum of the lengths of all lower infosets + this chunk's offset inside the infoset's segment
    for (uint32_t info = threadIdx.x; info < NI; info += CH_THREADS) tots[info] = so.total[info];
    __syncthreads();
    lds_exscan(tots, base, NI, wave_tot);
    for (uint32_t info = threadIdx.x; info < NI; info += CH_THREADS) base[info] += so.offs[(size_t)info * so.n_chunks + chunk];
    __syncthreads();
    const uint32_t A = g.A;
    const float tf = (float)p.epoch;
    for (uint32_t lt = threadIdx.x; lt < CH_TREES; lt += CH_THREADS) {
        const uint32_t tree = chunk * CH_TREES + lt;
        if (tree >= p.batch) continue;
        const uint32_t nd = dc.ndec[tree];
        for (uint32_t slot = 0; slot < nd; ++slot) {
            const uint32_t info = dc.info[slot * dc.stride + tree];
            const uint32_t nact = g.info_actions[info];
            const uint32_t rank = pre[info * SM_WORDS + (lt >> 5)] + __popc(bits[info * SM_WORDS + (lt >> 5)] & ((1u << (lt & 31u)) - 1u));
            const size_t pos = (size_t)base[info] + rank;
            for (uint32_t a = 0; a < A; ++a) {
                float rd = 0.0f, wd = 0.0f;
                if (a < nact) {
                    rd = dc.regret[(slot * A + a) * dc.stride + tree];
                    const float sg = dc.policy[(slot * A + a) * dc.stride + tree];
                    wd = p.W == RP_WEIGHT_LINEAR ? sg * tf : (p.W == RP_WEIGHT_QUADRATIC ? sg * tf * tf : sg);
                }
                so.rw[pos * 2 * A + a] = rd;
                so.rw[pos * 2 * A + A + a] = wd;
            }
            so.mask[pos] = dc.mask[slot * dc.stride + tree];
            so.payoff[pos] = dc.payoff[slot * dc.stride + tree];
        }
    }
}

// per-epoch discount constants of a RegretSchedule (regret/{linear,discounted,asymmetric}.rs)
struct Discount {
    float pos, neg, zero;
};
__device__ __forceinline__ Discount regret_discount(int R, float t, float pow15, float pow05) {
    Discount d{1.0f, 1.0f, 1.0f};
    const float lin = t / (t + 1.0f);
    if (R == RP_REGRET_LINEAR) d = Discount{lin, lin, lin};
    else if (R == RP_REGRET_ASYMMETRIC) d = Discount{1.0f, lin, lin};
    else if (R == RP_REGRET_DISCOUNTED) {
        const float xp = pow15, xn = pow05, xz = t / 1.0f;
        d = Discount{xp / (xp + 1.0f), xn / (xn + 1.0f), xz / (xz + 1.0f)};
    }
    return d;
}
__device__ __forceinline__ uint32_t seg_base(const DevSorted& so, uint32_t info) {
    uint32_t part = 0;
    for (uint32_t i = lane_of(); i < info; i += 64) part += so.total[i];
    for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
    return part;
}

#define TILE_FLOATS 1024u  // regret/weight deltas per LDS tile (16 per lane)
#define TILE_REGS (TILE_FLOATS / 64u)
#define TILE_PAD 4u        // row padding of the stream-major tile (keeps 16-B alignment, staggers banks)
#define PTILE 1024u        // payoffs per LDS tile

// LDS carve of k_chain (bytes): regret/weight tiles, mask tiles, payoff / reciprocal / divisor tiles
#define CHAIN_TILE_WORDS (TILE_FLOATS + 2u * RP_MAX_ACTIONS * TILE_PAD)
#define CHAIN_LDS_WORDS (2u * CHAIN_TILE_WORDS + 2u * (TILE_FLOATS / 2u) + 7u * PTILE)

// wave 0: regret + weight cells, universal op acc <- max(acc * d + delta, floor) (x * 1.0f is exact, so Summed /
// Floored / Constant / Linear-weight schedules are the same instruction stream with d = 1).  wave 1: payoff + visits.
// Tiles are double buffered: the global loads of tile t+1 are issued into registers BEFORE the chain over tile t
// and committed to LDS after it, so HBM/L2 latency hides under the serial chain.  In LDS a tile is stream-major
// ([cell][entry]) so each chain lane reads its own stream 4 entries at a time (ds_read_b128), 16 entries ahead.
template <bool SIGNED, bool PRUNED>
__global__ __launch_bounds__(128) void k_chain(DevGame g, DevTables t, DevSorted so, StepParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t info = blockIdx.x;
    if (g.info_player[info] != p.walker) return;
    const uint32_t len = so.total[info];
    if (len == 0) return;
    const uint32_t A = g.A, nact = g.info_actions[info], W2 = 2 * A;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const size_t base = seg_base(so, info);
    const float tf = (float)p.epoch;
    float* tile = reinterpret_cast<float*>(smem);                                  // [2][CHAIN_TILE_WORDS]
    uint32_t* mtile = reinterpret_cast<uint32_t*>(tile + 2 * CHAIN_TILE_WORDS);    // [2][TILE_FLOATS / 2]
    float* ptile = reinterpret_cast<float*>(mtile + TILE_FLOATS);                  // [2][3][PTILE]: payoff, 1/b, b
    if (wave == 0) {
        const uint32_t T = (TILE_FLOATS / W2) & ~3u;  // Decisions per tile, multiple of 4
        const uint32_t TP = T + TILE_PAD;             // row stride of the stream-major tile
        const bool isreg = lane < A;
        const uint32_t a = lane % A;
        const bool chain = lane < W2 && a < nact;
        const size_t cell = (size_t)info * A + a;
        float acc = 0.0f, fl = RP_EPSILON;
        Discount d{1.0f, 1.0f, 1.0f};
        if (chain) {
            if (isreg) {
                acc = t.regret[cell];
                fl = regret_floor_of(p.R, p.regret_min);
                d = regret_discount(p.R, tf, p.pow15, p.pow05);
            } else {
                acc = t.weight[cell];
                const float dw = p.W == RP_WEIGHT_EXPONENTIAL ? 0.9999f : 1.0f;
                d = Discount{dw, dw, dw};
            }
        }
        const uint32_t ntiles = (len + T - 1) / T;
        float rg[TILE_REGS];
        uint32_t mk[TILE_REGS / 2];
        auto issue = [&](uint32_t tl) {
            const size_t e0 = (base + (size_t)tl * T) * W2;
            const uint32_t ne = min(T, len - tl * T), nfl = ne * W2;
#pragma unroll
            for (uint32_t r = 0; r < TILE_REGS; ++r) {
                const uint32_t k = lane + 64 * r;
                rg[r] = k < nfl ? so.rw[e0 + k] : 0.0f;
            }
            if (PRUNED) {
#pragma unroll
                for (uint32_t r = 0; r < TILE_REGS / 2; ++r) {
                    const uint32_t k = lane + 64 * r;
                    mk[r] = k < ne ? so.mask[base + (size_t)tl * T + k] : 0u;
                }
            }
        };
        auto commit = [&](uint32_t buf) {  // entry-major registers -> stream-major LDS
#pragma unroll
            for (uint32_t r = 0; r < TILE_REGS; ++r) {
                const uint32_t k = lane + 64 * r;
                if (k < T * W2) tile[buf * CHAIN_TILE_WORDS + (k % W2) * TP + k / W2] = rg[r];
            }
            if (PRUNED) {
#pragma unroll
                for (uint32_t r = 0; r < TILE_REGS / 2; ++r) mtile[buf * (TILE_FLOATS / 2) + lane + 64 * r] = mk[r];
            }
        };
        auto step = [&](float delta, uint32_t m) {
            float dd = d.zero;
            if (SIGNED) dd = acc > 0.0f ? d.pos : (acc < 0.0f ? d.neg : d.zero);
            const float nv = rp_maxf(acc * dd + delta, fl);
            if (PRUNED) acc = (isreg && !((m >> a) & 1u)) ? acc : nv;
            else acc = nv;
        };
        issue(0);
        commit(0);
        __builtin_amdgcn_wave_barrier();
        for (uint32_t tl = 0; tl < ntiles; ++tl) {
            const uint32_t buf = tl & 1u;
            const bool more = tl + 1 < ntiles;
            if (more) issue(tl + 1);
            const uint32_t n = min(T, len - tl * T);
            const float* row = tile + buf * CHAIN_TILE_WORDS + (chain ? lane : 0u) * TP;
            const uint32_t* mrow = mtile + buf * (TILE_FLOATS / 2);
            if (chain) {
                const uint32_t n16 = n & ~15u;
                uint32_t i = 0;
                if (n16) {
                    float4 c0 = *reinterpret_cast<const float4*>(row + 0), c1 = *reinterpret_cast<const float4*>(row + 4);
                    float4 c2 = *reinterpret_cast<const float4*>(row + 8), c3 = *reinterpret_cast<const float4*>(row + 12);
                    for (; i < n16; i += 16) {
                        float4 x0 = c0, x1 = c1, x2 = c2, x3 = c3;
                        if (i + 16 < n16) {  // the next 16 entries travel from LDS while these 16 are chained
                            c0 = *reinterpret_cast<const float4*>(row + i + 16);
                            c1 = *reinterpret_cast<const float4*>(row + i + 20);
                            c2 = *reinterpret_cast<const float4*>(row + i + 24);
                            c3 = *reinterpret_cast<const float4*>(row + i + 28);
                        }
                        uint32_t m[16];
#pragma unroll
                        for (uint32_t q = 0; q < 16; ++q) m[q] = PRUNED ? mrow[i + q] : 0xffffffffu;
                        step(x0.x, m[0]); step(x0.y, m[1]); step(x0.z, m[2]); step(x0.w, m[3]);
                        step(x1.x, m[4]); step(x1.y, m[5]); step(x1.z, m[6]); step(x1.w, m[7]);
                        step(x2.x, m[8]); step(x2.y, m[9]); step(x2.z, m[10]); step(x2.w, m[11]);
                        step(x3.x, m[12]); step(x3.y, m[13]); step(x3.z, m[14]); step(x3.w, m[15]);
                    }
                }
                for (; i < n; ++i) step(row[i], PRUNED ? mrow[i] : 0xffffffffu);
            }
            if (more) commit(buf ^ 1u);
            __builtin_amdgcn_wave_barrier();
        }
        if (chain) {
            if (isreg) t.regret[cell] = acc;
            else t.weight[cell] = acc;
        }
    } else {
        // Welford mean with the pre-increment visit count (solver.rs:174-192): ev += (payoff - ev) / (n + 1).
        // The divisor sequence is known in advance, so the whole wave precomputes b = (float)(n+1) and the
        // correctly rounded 1/b per entry; the serial chain then needs mul + 2 fma per division
        // (rp_div_by_recip1) and each quotient carries an exact off-path proof that it equals IEEE a / b.
        const bool chain = lane < nact;
        const size_t cell = (size_t)info * A + lane;
        float ev = 0.0f;
        uint32_t visits = 0;
        if (chain) {
            ev = t.payoff[cell];
            visits = t.visits[cell];
        }
        const uint32_t v0 = __shfl(visits, 0, 64);
        const float ev0 = __shfl(ev, 0, 64);
        // every edge of an infoset is always visited together, so all its (payoff, visits) cells hold the same
        // value and ONE chain serves them; anything else (a hand-made import) takes the plain path below
        const bool uniform = __all(!chain || (visits == v0 && rp_f2u(ev) == rp_f2u(ev0)));
        float* hist = ptile + 6 * PTILE;  // [PTILE] ev after each touch of the current tile
        if (uniform) ev = ev0;
        const float ev_start = ev;
        if (uniform) {
            const uint32_t ntiles = (len + PTILE - 1) / PTILE;
            float rg[PTILE / 64];
            auto issue = [&](uint32_t tl) {
                const uint32_t n = min(PTILE, len - tl * PTILE);
#pragma unroll
                for (uint32_t r = 0; r < PTILE / 64; ++r) {
                    const uint32_t k = lane + 64 * r;
                    rg[r] = k < n ? so.payoff[base + (size_t)tl * PTILE + k] : 0.0f;
                }
            };
            auto commit = [&](uint32_t tl, uint32_t buf) {
                float* pt = ptile + buf * 3 * PTILE;
#pragma unroll
                for (uint32_t r = 0; r < PTILE / 64; ++r) {
                    const uint32_t k = lane + 64 * r;
                    const float b = (float)(v0 + tl * PTILE + k + 1u);  // (n + 1) as f32 (solver.rs:179)
                    pt[k] = rg[r];
                    pt[PTILE + k] = 1.0f / b;
                    pt[2 * PTILE + k] = b;
                }
            };
            issue(0);
            commit(0, 0);
            __builtin_amdgcn_wave_barrier();
            for (uint32_t tl = 0; tl < ntiles; ++tl) {
                const uint32_t buf = tl & 1u;
                const bool more = tl + 1 < ntiles;
                if (more) issue(tl + 1);
                const uint32_t n = min(PTILE, len - tl * PTILE);
                const float* pt = ptile + buf * 3 * PTILE;
#ifndef RP_EXPERIMENT_SKIP_PAYOFF
                // (1) the serial chain: 5 VALU ops per touch (sub, mul, fma, fma, add); lane 0 logs ev after each touch
                const float ev_tile = ev;
                {
                    auto fast = [&](float pv, float rv, float bv) {
                        const float s = pv - ev;
                        const float q0 = s * rv;
                        const float e0 = fmaf(-bv, q0, s);
                        ev += fmaf(e0, rv, q0);
                        return ev;
                    };
                    const uint32_t n4 = n & ~3u;
                    uint32_t i = 0;
                    for (; i < n4; i += 4) {
                        const float4 pv = *reinterpret_cast<const float4*>(pt + i);
                        const float4 rv = *reinterpret_cast<const float4*>(pt + PTILE + i);
                        const float4 bv = *reinterpret_cast<const float4*>(pt + 2 * PTILE + i);
                        float4 h;
                        h.x = fast(pv.x, rv.x, bv.x); h.y = fast(pv.y, rv.y, bv.y);
                        h.z = fast(pv.z, rv.z, bv.z); h.w = fast(pv.w, rv.w, bv.w);
                        if (lane == 0) *reinterpret_cast<float4*>(hist + i) = h;
                    }
                    for (; i < n; ++i) {
                        const float e = fast(pt[i], pt[PTILE + i], pt[2 * PTILE + i]);
                        if (lane == 0) hist[i] = e;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // (2) the proof, lane-parallel and off the chain: every touch's quotient is re-derived from the logged
                //     ev and checked with the exact-residual criterion of rp_div_by_recip1 (== IEEE s / b when proven)
                bool bad = false;
                for (uint32_t k = lane; k < n; k += 64) {
                    const float prev = k ? hist[k - 1] : ev_tile;
                    int proven;
                    const float q = rp_div_by_recip1(pt[k] - prev, pt[2 * PTILE + k], pt[PTILE + k], &proven);
                    bad |= !proven || (prev + q != hist[k]);
                }
                if (__any(bad)) {  // essentially never: redo this tile with IEEE divisions
                    ev = ev_tile;
                    for (uint32_t k = 0; k < n; ++k) ev += (pt[k] - ev) / pt[2 * PTILE + k];
                }
                __builtin_amdgcn_wave_barrier();
#endif
                if (more) commit(tl + 1, buf ^ 1u);
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (chain) {
            if (!uniform) {  // edges of one infoset with different visit counts (only after a hand-made import)
                ev = ev_start;
                uint32_t v = visits;
                for (uint32_t i = 0; i < len; ++i) {
                    ev += (so.payoff[base + i] - ev) / (float)(v + 1u);
                    v += 1u;
                }
            }
            t.payoff[cell] = ev;
            t.visits[cell] = visits + len;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Composed update (include/rp_mi355x.h rp_compose_block; oracle: ora_mccfr_step_local): the serial chain of the
// ordered mode is replaced by a two-level composition of per-cell maps F(x) = max(a x + b, m).
//   k_block_maps  one workgroup per (infoset, block of T consecutive Decisions): sequential composition inside the
//                 block, all blocks of all infosets in parallel
//   k_combine2    (infoset, part) workgroups: block maps -> group maps -> Cell / InfoSum blob, or straight into the tables
// ------------------------------------------------------------------------------------------------
// Map / map_compose live in mccfr_kernels.hpp
// block = the Decisions of infoset blockIdx.y produced by chunk blockIdx.x (RP_COMPOSE_CHUNK == CH_TREES trees):
// in the sorted layout a contiguous group.  Large games (per-infoset slot map); small games fuse the sort away, below.
template <bool PRUNED>
__global__ __launch_bounds__(128) void k_block_maps(DevGame g, DevSorted so, StepParams p, Map* bmaps, float* bpsum,
                                                    uint32_t* bcnt, uint32_t nblk_max) {
    __shared__ __attribute__((aligned(16))) float tile[TILE_FLOATS + 2 * RP_MAX_ACTIONS * TILE_PAD];
    __shared__ uint32_t mtile[TILE_FLOATS / 2];
    __shared__ __attribute__((aligned(16))) float ptile[TILE_FLOATS / 2];
    const uint32_t info = blockIdx.y, blk = blockIdx.x;
    if (g.info_player[info] != p.walker) return;
    const uint32_t A = g.A, nact = g.info_actions[info], W2 = 2 * A;
    const uint32_t T = compose_block(A);  // touches per LDS tile
    const uint32_t n = so.counts[(size_t)info * so.n_chunks + blk];
    const uint32_t TP = T + TILE_PAD;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const size_t base = seg_base(so, info) + so.offs[(size_t)info * so.n_chunks + blk];
    const float NEG_INF = rp_u2f(0xff800000u);
    const bool isreg = lane < A;
    const uint32_t a = lane % A;
    const bool chain = lane < W2 && a < nact;
    const float tf = (float)p.epoch;
    const float fl = isreg ? regret_floor_of(p.R, p.regret_min) : RP_EPSILON;
    const float d = isreg ? (p.R == RP_REGRET_LINEAR ? tf / (tf + 1.0f) : 1.0f) : (p.W == RP_WEIGHT_EXPONENTIAL ? 0.9999f : 1.0f);
    float ma = 1.0f, mb = 0.0f, mm = NEG_INF, psum = 0.0f;
    uint32_t cnt = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += T) {  // the chain runs through the block tile by tile
        const uint32_t m = min(T, n - t0);
        if (wave == 0) {
            const size_t e0 = (base + t0) * W2;
            const uint32_t nfl = m * W2;
            float lr[TILE_FLOATS / 64];
#pragma unroll
            for (uint32_t q = 0; q < TILE_FLOATS / 64; ++q) {
                const uint32_t k = lane + 64 * q;
                lr[q] = k < nfl ? so.rw[e0 + k] : 0.0f;
            }
#pragma unroll
            for (uint32_t q = 0; q < TILE_FLOATS / 64; ++q) {
                const uint32_t k = lane + 64 * q;
                if (k < nfl) tile[(k % W2) * TP + k / W2] = lr[q];
            }
            if (PRUNED)
                for (uint32_t k = lane; k < m; k += 64) mtile[k] = so.mask[base + t0 + k];
        } else {
            for (uint32_t k = lane; k < m; k += 64) ptile[k] = so.payoff[base + t0 + k];
        }
        __syncthreads();
        if (wave == 0 && chain) {
            const float* row = tile + lane * TP;
            for (uint32_t i = 0; i < m; ++i) {
                const float delta = row[i];
                const bool skip = PRUNED && isreg && !((mtile[i] >> a) & 1u);
                // first touch of the block: (d, delta, floor); then a <- a d, b <- b d + delta, m <- max(m d + delta, floor)
                const float na = cnt ? ma * d : d;
                const float nb = cnt ? mb * d + delta : delta;
                const float nm = cnt ? rp_maxf(mm * d + delta, fl) : fl;
                ma = skip ? ma : na;
                mb = skip ? mb : nb;
                mm = skip ? mm : nm;
                cnt += skip ? 0u : 1u;
            }
        }
        if (wave == 1 && lane == 0)
            for (uint32_t i = 0; i < m; ++i) psum += ptile[i];
        __syncthreads();
    }
    const size_t slot = (size_t)info * nblk_max + blk;
    if (wave == 0 && lane < W2) bmaps[slot * W2 + lane] = Map{ma, mb, mm, chain ? cnt : 0u};
    if (wave == 1 && lane == 0) {
        bpsum[slot] = psum;
        bcnt[slot] = n;
    }
}

// Small games: block maps straight from the lane-interleaved Decisions of one chunk — no sorted copy in HBM at all.
// One thread per tree.  The chunk's Decisions get their place in per-infoset, tree-ordered lists (LDS bitmap + prefix
// popcount, as k_compact_small); then, cell by cell, every thread drops its trees' values at those places in an LDS
// array (global reads coalesced over trees) and one thread per infoset composes its list sequentially out of LDS.
template <bool PRUNED, uint32_t PASSES>
__global__ __launch_bounds__(CH_THREADS) void k_chunk_maps(DevGame g, DevDecisions dc, StepParams p, Map* bmaps, float* bpsum,
                                                           uint32_t* bcnt, uint32_t nblk_max) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cm_lds[];
    __shared__ uint32_t wave_tot[CH_THREADS / 64];
    const uint32_t NI = g.n_infos, A = g.A, W2 = 2 * A, chunk = blockIdx.x, tid = threadIdx.x, MD = dc.maxdec;
    uint32_t* bits = cm_lds;                                              // [NI][SM_WORDS]
    uint32_t* lcount = bits + NI * SM_WORDS;                              // [NI]
    uint32_t* lbase = lcount + NI;                                        // [NI]
    float* vals = reinterpret_cast<float*>(lbase + NI);                   // [MD * CH_TREES] one cell's values, list order
    uint16_t* pre = reinterpret_cast<uint16_t*>(vals + MD * CH_TREES);    // [NI][SM_WORDS]
    uint16_t* posl = pre + NI * SM_WORDS;                                 // [MD][CH_TREES] list position of (slot, tree)
    uint16_t* lmask = posl + MD * CH_TREES;                               // [MD * CH_TREES] expanded-edge masks (PRUNED)
    chunk_bitmap(dc, NI, chunk, p.batch, bits);
    for (uint32_t info = tid; info < NI; info += CH_THREADS) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < SM_WORDS; ++w) {
            pre[info * SM_WORDS + w] = (uint16_t)run;
            run += __popc(bits[info * SM_WORDS + w]);
        }
        lcount[info] = run;
    }
    __syncthreads();
    lds_exscan(lcount, lbase, NI, wave_tot);
    const uint32_t lt = tid, tree = chunk * CH_TREES + lt;  // CH_TREES == CH_THREADS: one tree per thread
    const uint32_t nd = tree < p.batch ? dc.ndec[tree] : 0u;
    for (uint32_t slot = 0; slot < nd; ++slot) {
        const uint32_t info = dc.info[slot * dc.stride + tree];
        const uint32_t rank = pre[info * SM_WORDS + (lt >> 5)] + __popc(bits[info * SM_WORDS + (lt >> 5)] & ((1u << (lt & 31u)) - 1u));
        const uint32_t pos = lbase[info] + rank;
        posl[slot * CH_TREES + lt] = (uint16_t)pos;
        if (PRUNED) lmask[pos] = (uint16_t)dc.mask[slot * dc.stride + tree];
    }
    const float NEG_INF = rp_u2f(0xff800000u);
    const float tf = (float)p.epoch;
    // chain phase: thread t owns infosets t, t + 256, ... (PASSES = 1 when the game has at most 256 infosets)
    uint32_t my_nact[PASSES], my_n[PASSES], my_base[PASSES];
#pragma unroll
    for (uint32_t q = 0; q < PASSES; ++q) {
        const uint32_t info = tid + q * CH_THREADS;
        const bool mine = info < NI && g.info_player[info < NI ? info : 0u] == p.walker;
        my_nact[q] = mine ? g.info_actions[info] : 0u;  // 0: not this walker's infoset
        my_n[q] = mine ? lcount[info] : 0u;
        my_base[q] = mine ? lbase[info] : 0u;
    }
    for (uint32_t c = 0; c <= W2; ++c) {  // regret cells, weight cells, then the payoff sum
        const bool isreg = c < A, ispay = c == W2;
        const uint32_t a = c % A;
        __syncthreads();  // the previous cell's chains are done with `vals`
        for (uint32_t slot = 0; slot < nd; ++slot) {
            float v;
            if (ispay) v = dc.payoff[slot * dc.stride + tree];
            else if (isreg) v = dc.regret[(slot * A + a) * dc.stride + tree];
            else {
                const float sg = dc.policy[(slot * A + a) * dc.stride + tree];
                v = p.W == RP_WEIGHT_LINEAR ? sg * tf : (p.W == RP_WEIGHT_QUADRATIC ? sg * tf * tf : sg);
            }
            vals[posl[slot * CH_TREES + lt]] = v;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < PASSES; ++q) {
            const uint32_t info = tid + q * CH_THREADS;
            if (!my_nact[q]) continue;
            const uint32_t n = my_n[q], base = my_base[q];
            const size_t slot_out = (size_t)info * nblk_max + chunk;
            if (ispay) {  // payoff sum of the block, left fold from 0.0f
                float sum = 0.0f;
                for (uint32_t e = 0; e < n; ++e) sum += vals[base + e];
                bpsum[slot_out] = sum;
                bcnt[slot_out] = n;
                continue;
            }
            const bool chain = a < my_nact[q];
            const float fl = isreg ? regret_floor_of(p.R, p.regret_min) : RP_EPSILON;
            const float d = isreg ? (p.R == RP_REGRET_LINEAR ? tf / (tf + 1.0f) : 1.0f) : (p.W == RP_WEIGHT_EXPONENTIAL ? 0.9999f : 1.0f);
            float ma = 1.0f, mb = 0.0f, mm = NEG_INF;
            uint32_t cnt = 0;
            if (chain) {
                for (uint32_t e = 0; e < n; ++e) {
                    const float delta = vals[base + e];
                    const bool skip = PRUNED && isreg && !((lmask[base + e] >> a) & 1u);
                    // first touch of the block: (d, delta, floor); then a <- a d, b <- b d + delta, m <- max(m d + delta, floor)
                    const float na = cnt ? ma * d : d;
                    const float nb = cnt ? mb * d + delta : delta;
                    const float nm = cnt ? rp_maxf(mm * d + delta, fl) : fl;
                    ma = skip ? ma : na;
                    mb = skip ? mb : nb;
                    mm = skip ? mm : nm;
                    cnt += skip ? 0u : 1u;
                }
            }
            bmaps[slot_out * W2 + c] = Map{ma, mb, mm, chain ? cnt : 0u};
        }
    }
}

}  // namespace rp
#include "traverse_static.hpp"
namespace rp {

// The two-level fold of the block maps (include/rp_mi355x.h RP_FOLD_GROUP: the block maps of a cell composed sequentially
// inside groups of RP_FOLD_GROUP consecutive blocks, the group maps then in group order), spread over (infoset, part)
// workgroups, and — APPLY — the rest of the step with it.
//   stage 1  a part owns CB2_GPW(2A) consecutive groups: its block maps are one contiguous run of HBM, copied to LDS by all
//            256 threads at once (one round trip instead of a chain of eight per thread), then thread (group, cell) composes
//            its RP_FOLD_GROUP maps out of LDS in block order; the group maps go to HBM with agent-scope stores;
//   stage 2  the LAST part of an infoset to finish (arrival counter; which one is timing, what it computes is not) folds the
//            infoset's group maps in group order into the summary cell maps, payoff sum and count;
//   APPLY    single-GPU step: that workgroup also applies the summary to the infoset's table row (k_fold with world = 1)
//            and refreshes the row of the per-infoset tables the next traversal reads (k_prepare_infos): one launch
//            instead of three.  Otherwise it writes the summary blob (rp_mccfr_step_local).
// Cross-workgroup data is a few KB per infoset: agent-scope (sc1) stores + s_waitcnt before the arrival, agent-scope
// loads after it.  (A release FENCE at agent scope writes back a whole XCD's L2: tried on the block maps, 4x slower.)
// This hand-over is written against the gfx942 / gfx950 memory system, not against the portable memory model: relaxed agent-scope
// stores are sc1 write-through stores that `s_waitcnt vmcnt(0)` waits for (no separate store counter), relaxed agent-scope loads
// bypass the XCD's non-coherent lines.  On any other target the arrival counter would need release / acquire semantics:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "k_combine2's cross-workgroup hand-over relies on gfx942/gfx950 store counting and sc1 semantics: use an acq_rel arrival counter on other targets"
#endif
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* q, uint32_t v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ Map ld_map(const Map* m) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(m);
    return Map{rp_u2f(ld_agent(w)), rp_u2f(ld_agent(w + 1)), rp_u2f(ld_agent(w + 2)), ld_agent(w + 3)};
}
__device__ __forceinline__ void st_map(Map* m, const Map& v) {
    uint32_t* w = reinterpret_cast<uint32_t*>(m);
    st_agent(w, rp_f2u(v.a));
    st_agent(w + 1, rp_f2u(v.b));
    st_agent(w + 2, rp_f2u(v.m));
    st_agent(w + 3, v.n);
}
__host__ __device__ inline uint32_t cb2_gpw(uint32_t W2) { return 32u / W2 ? 32u / W2 : 1u; }  // groups per part: a 32 KB tile
struct FoldScratch {
    Map* gmaps;      // [n_infos][ngrp_max][2A]
    float* gpsum;    // [n_infos][ngrp_max]
    uint32_t* gcnt;  // [n_infos][ngrp_max]
    uint32_t* done;  // [n_infos] parts finished
    uint32_t ngrp_max;
};
template <bool APPLY>
__global__ __launch_bounds__(256) void k_combine2(DevGame g, DevTables t, DevInfoTab it, StepParams p, const Map* bmaps, const float* bpsum,
                                                  const uint32_t* bcnt, uint32_t nblk_max, FoldScratch fs, Cell* cells, InfoSum* sums) {
    extern __shared__ __attribute__((aligned(16))) uint4 cb_lds[];
    __shared__ uint32_t role, sh_len;
    __shared__ float sh_ps;
    const uint32_t info = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    const uint32_t A = g.A, W2 = 2 * A, GPW = cb2_gpw(W2);
    const Map ident{1.0f, 0.0f, rp_u2f(0xff800000u), 0u};
    if (g.info_player[info] != p.walker) {  // not this walker's infoset: nothing happened to it
        if (!APPLY && part == 0) {
            if (tid < W2) {
                Cell* cl = &cells[(size_t)info * A + tid % A];
                if (tid < A) { cl->ra = ident.a; cl->rb = ident.b; cl->rm = ident.m; cl->rn = 0u; }
                else { cl->wa = ident.a; cl->wb = ident.b; cl->wm = ident.m; cl->wn = 0u; }
            } else if (tid == W2) {
                sums[info] = InfoSum{0u, 0.0f};
            }
        }
        return;
    }
    const uint32_t nb = (p.batch + RP_COMPOSE_CHUNK - 1) / RP_COMPOSE_CHUNK;  // one block per chunk of trees
    const uint32_t ngrp = (nb + RP_FOLD_GROUP - 1) / RP_FOLD_GROUP, nparts = (ngrp + GPW - 1) / GPW;
    if (part >= nparts) return;
    const uint32_t g0 = part * GPW, b_lo = g0 * RP_FOLD_GROUP, b_hi = min(nb, (g0 + GPW) * RP_FOLD_GROUP), count = b_hi - b_lo;
    uint4* tile = cb_lds;                                                                // [GPW * 64][W2] block maps
    float* ps_t = reinterpret_cast<float*>(tile + (size_t)GPW * RP_FOLD_GROUP * W2);     // [GPW * 64]
    uint32_t* cn_t = reinterpret_cast<uint32_t*>(ps_t + GPW * RP_FOLD_GROUP);            // [GPW * 64]
    {
        const uint4* src = reinterpret_cast<const uint4*>(bmaps + ((size_t)info * nblk_max + b_lo) * W2);
        for (uint32_t e = tid; e < count * W2; e += 256u) tile[e] = src[e];
        for (uint32_t e = tid; e < count; e += 256u) {
            ps_t[e] = bpsum[(size_t)info * nblk_max + b_lo + e];
            cn_t[e] = bcnt[(size_t)info * nblk_max + b_lo + e];
        }
    }
    __syncthreads();
    if (tid < GPW * W2) {
        const uint32_t s = tid / W2, c = tid % W2, grp = g0 + s;
        if (grp < ngrp) {
            const uint32_t lo = s * RP_FOLD_GROUP, hi = min(count, lo + RP_FOLD_GROUP);
            Map m = ident;
            for (uint32_t b0 = lo; b0 < hi; b0 += 8u) {  // eight LDS reads in flight ahead of the dependent chain
                uint4 v[8];
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) v[q] = tile[min(b0 + q, hi - 1u) * W2 + c];
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q)
                    if (b0 + q < hi) m = map_compose(m, Map{rp_u2f(v[q].x), rp_u2f(v[q].y), rp_u2f(v[q].z), v[q].w});
            }
            st_map(&fs.gmaps[((size_t)info * fs.ngrp_max + grp) * W2 + c], m);
        }
    } else if (tid < GPW * W2 + GPW) {
        const uint32_t s = tid - GPW * W2, grp = g0 + s;
        if (grp < ngrp) {
            const uint32_t lo = s * RP_FOLD_GROUP, hi = min(count, lo + RP_FOLD_GROUP);
            float gp = 0.0f;
            uint32_t gc = 0;
            for (uint32_t b = lo; b < hi; ++b) {
                gp += ps_t[b];
                gc += cn_t[b];
            }
            st_agent(reinterpret_cast<uint32_t*>(fs.gpsum) + (size_t)info * fs.ngrp_max + grp, rp_f2u(gp));
            st_agent(fs.gcnt + (size_t)info * fs.ngrp_max + grp, gc);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the group maps are in memory before this part counts itself in
    __syncthreads();
    if (tid == 0) role = atomicAdd(&fs.done[info], 1u) == nparts - 1u ? 1u : 0u;
    __syncthreads();
    if (!role) return;
    if (tid == 0) fs.done[info] = 0;  // for the next launch
    // the infoset's group maps: into LDS by all threads at once (one round trip), then folded in group order
    Map tot = ident;
    float ps = 0.0f;
    uint32_t len = 0;
    const uint32_t TILE_G = GPW * RP_FOLD_GROUP;
    const uint32_t* gm = reinterpret_cast<const uint32_t*>(fs.gmaps + (size_t)info * fs.ngrp_max * W2);
    for (uint32_t k0 = 0; k0 < ngrp; k0 += TILE_G) {
        const uint32_t n = min(TILE_G, ngrp - k0);
        __syncthreads();
        for (uint32_t e = tid; e < n * W2; e += 256u) {
            const uint32_t* w = gm + ((size_t)k0 * W2 + e) * 4u;
            tile[e] = make_uint4(ld_agent(w), ld_agent(w + 1), ld_agent(w + 2), ld_agent(w + 3));
        }
        for (uint32_t e = tid; e < n; e += 256u) {
            ps_t[e] = rp_u2f(ld_agent(reinterpret_cast<const uint32_t*>(fs.gpsum) + (size_t)info * fs.ngrp_max + k0 + e));
            cn_t[e] = ld_agent(fs.gcnt + (size_t)info * fs.ngrp_max + k0 + e);
        }
        __syncthreads();
        if (tid < W2) {
            for (uint32_t k0b = 0; k0b < n; k0b += 8u) {
                uint4 v[8];
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) v[q] = tile[min(k0b + q, n - 1u) * W2 + tid];
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q)
                    if (k0b + q < n) tot = map_compose(tot, Map{rp_u2f(v[q].x), rp_u2f(v[q].y), rp_u2f(v[q].z), v[q].w});
            }
        } else if (tid == W2) {
            for (uint32_t k = 0; k < n; ++k) {
                ps += ps_t[k];
                len += cn_t[k];
            }
        }
    }
    if (tid == W2) {
        sh_ps = ps;
        sh_len = len;
    }
    if (!APPLY) {
        if (tid < W2) {
            Cell* cl = &cells[(size_t)info * A + tid % A];
            if (tid < A) { cl->ra = tot.a; cl->rb = tot.b; cl->rm = tot.m; cl->rn = tot.n; }
            else { cl->wa = tot.a; cl->wb = tot.b; cl->wm = tot.m; cl->wn = tot.n; }
        }
        __syncthreads();
        if (tid == 0) sums[info] = InfoSum{sh_len, sh_ps};
        return;
    }
    // k_fold, world = 1, for this infoset's row
    __syncthreads();
    const uint32_t nact = g.info_actions[info];
    if (tid < W2 && tid % A < nact) {
        const uint32_t cell = info * A + tid % A;
        if (tid < A) {
            float r = t.regret[cell];
            if (tot.n) r = rp_maxf(tot.a * r + tot.b, tot.m);
            t.regret[cell] = r;
            float ev = t.payoff[cell];
            uint32_t visits = t.visits[cell];
            if (sh_len) {
                const uint32_t n2 = visits + sh_len;
                ev = ev + (sh_ps - (float)sh_len * ev) / (float)n2;
                visits = n2;
            }
            t.payoff[cell] = ev;
            t.visits[cell] = visits;
        } else {
            float w = t.weight[cell];
            if (tot.n) w = rp_maxf(tot.a * w + tot.b, tot.m);
            t.weight[cell] = w;
        }
    }
    __syncthreads();  // workgroup scope: the row just written is what prepare_one reads
    if (tid == 0) prepare_one(g, t, p, it, info);
}

// ------------------------------------------------------------------------------------------------
// k_fold: the receiving side of the multi-GPU exchange (oracle: ora_mccfr_step_apply)
// ------------------------------------------------------------------------------------------------
// one thread per table cell; `blob` holds `world` summaries back to back: [cells][sums]
__global__ void k_fold(DevGame g, DevTables t, const unsigned char* blob, size_t blob_stride, uint32_t world) {
    const uint32_t cell = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ncell = g.n_infos * g.A;
    if (cell >= ncell) return;
    const uint32_t info = cell / g.A, a = cell % g.A;
    if (a >= g.info_actions[info]) return;
    float r = t.regret[cell], w = t.weight[cell], ev = t.payoff[cell];
    uint32_t visits = t.visits[cell];
    for (uint32_t rk = 0; rk < world; ++rk) {
        const unsigned char* b = blob + (size_t)rk * blob_stride;
        const Cell c = reinterpret_cast<const Cell*>(b)[cell];
        const InfoSum s = reinterpret_cast<const InfoSum*>(b + (size_t)ncell * sizeof(Cell))[info];
        if (c.rn) r = rp_maxf(c.ra * r + c.rb, c.rm);
        if (c.wn) w = rp_maxf(c.wa * w + c.wb, c.wm);
        if (s.count) {
            const uint32_t n2 = visits + s.count;
            ev = ev + (s.psum - (float)s.count * ev) / (float)n2;
            visits = n2;
        }
    }
    t.regret[cell] = r;
    t.weight[cell] = w;
    t.payoff[cell] = ev;
    t.visits[cell] = visits;
}

// the exchange window (rp_mccfr_window_local): acc <- step o acc per table cell — the maps of consecutive local
// steps composed in step order, touch counts, payoff sums and visit counts added (oracle: ora_mccfr_window_accumulate)
__global__ void k_accumulate(DevGame g, unsigned char* acc, const unsigned char* step, uint32_t first) {
    const uint32_t cell = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ncell = g.n_infos * g.A;
    if (cell >= ncell) return;
    const Cell s = reinterpret_cast<const Cell*>(step)[cell];
    Cell* ac = reinterpret_cast<Cell*>(acc) + cell;
    if (first) {
        *ac = s;
    } else {
        const Cell a = *ac;
        const Map r = map_compose(Map{a.ra, a.rb, a.rm, a.rn}, Map{s.ra, s.rb, s.rm, s.rn});
        const Map w = map_compose(Map{a.wa, a.wb, a.wm, a.wn}, Map{s.wa, s.wb, s.wm, s.wn});
        *ac = Cell{r.a, r.b, r.m, w.a, w.b, w.m, r.n, w.n};
    }
    if (cell % g.A == 0) {
        const uint32_t info = cell / g.A;
        const InfoSum si = reinterpret_cast<const InfoSum*>(step + (size_t)ncell * sizeof(Cell))[info];
        InfoSum* ai = reinterpret_cast<InfoSum*>(acc + (size_t)ncell * sizeof(Cell)) + info;
        if (first) *ai = si;
        else *ai = InfoSum{ai->count + si.count, ai->psum + si.psum};
    }
}

}  // namespace rp

// =================================================================================================
// host side
// =================================================================================================
using namespace rp;

struct KernelClock {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    uint64_t launches = 0;
};

struct rp_mccfr {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // host copy of the game (exploitability, validation)
    std::vector<rp_state> states;
    std::vector<uint32_t> children;
    std::vector<float> payoffs;
    std::vector<uint8_t> info_actions, info_player;
    std::vector<float> default_regret;
    rp_game_table tbl{};
    // device
    DevGame g{};
    DevTables t{};
    DevScratch sc{};
    DevDecisions dc{};
    void* d_states = nullptr;
    void* d_children = nullptr;
    void* d_kids = nullptr;
    void* d_flat = nullptr;  // DevGame::flat
    void* d_payoffs = nullptr;
    void* d_info_actions = nullptr;
    void* d_info_player = nullptr;
    void* d_scratch = nullptr;
    void* d_dec = nullptr;
    void* d_summary = nullptr;
    void* d_window = nullptr;      // rp_mccfr_step_comm: this rank's window summary, then every rank's (all-gathered)
    uint32_t window_world = 0;
    void* d_sorted = nullptr;
    void* d_bmaps = nullptr;
    uint32_t maxint = 1;  // most internal nodes of a sampled tree
    void* d_itab = nullptr;
    DevInfoTab itab{};
    DevSorted so{};
    unsigned long long* d_counters = nullptr;
    int R = 0, W = 0, S = 0;
    rp_hyper hp{};
    uint64_t seed = 0;
    // rp_mccfr_set_rng: RP_RNG_REFERENCE draws from the reference's own chain (include/rp_refrng.h)
    rp_rng_kind rng = RP_RNG_COUNTER;
    uint32_t n_chance_infos = 0;
    void* d_info_streams = nullptr;    // rp_hash_stream[n_infos]
    void* d_chance_streams = nullptr;  // rp_hash_stream[n_chance_infos]
    void* d_ref_mid = nullptr;         // rp_sip_mid[n_infos + n_chance_infos], of epoch ref_mid_epoch
    uint64_t ref_mid_epoch = ~0ull;
    uint32_t batch = 1, capacity = 0;
    uint64_t epoch = 0;
    uint32_t rank = 0, world = 1;
    uint32_t maxdec = 1;
    rp_update_mode mode = RP_UPDATE_ORDERED;
    bool use_lds_traverse = false;
    // the per-infoset tables of the traversal (DevInfoTab) are a function of the regret/strategy tables: refreshed when stale
    uint64_t tables_version = 1, itab_version = 0;
    static constexpr uint32_t cell_pad = 7;  // words between the cells' value arrays in k_traverse_maps_static's LDS (bank spread)
    bool fuse_maps = true;  // composed update: traversal + block maps in one kernel when the game allows (RP_TRAV_UNFUSED=1: never)
    int static_skel = 0;  // 0: none (k_traverse_lds / k_traverse), 1: KuhnSkel, 2: LeducSkel (traverse_static.hpp)
    bool profiling = false;
    KernelClock clk_traverse, clk_compact, clk_update;
};

namespace {

int set_device(const rp_mccfr* h) {
    HIP_TRY(hipSetDevice(h->device));
    return RP_OK;
}

size_t chain_lds_bytes(uint32_t A) { (void)A; return (size_t)CHAIN_LDS_WORDS * 4; }

size_t summary_bytes_of(const rp_mccfr* h) {
    return (size_t)h->tbl.n_infos * h->tbl.max_actions * sizeof(Cell) + (size_t)h->tbl.n_infos * sizeof(InfoSum);
}

// largest number of walker nodes / stack entries an externally sampled tree can have
void sampled_tree_bounds(const rp_mccfr* h, uint32_t* maxdec, uint32_t* maxstack, uint32_t* maxint) {
    const rp_game_table& t = h->tbl;
    uint32_t best_dec = 1, best_stack = 1, best_int = 1;
    for (uint32_t w = 0; w < t.n_players; ++w) {
        std::function<uint32_t(uint32_t)> wn = [&](uint32_t s) -> uint32_t {
            const rp_state& st = h->states[s];
            uint32_t acc = 0;
            for (uint32_t k = 0; k < st.n_children; ++k) {
                uint32_t c = wn(h->children[st.offset + k]);
                acc = st.turn == w ? acc + c : std::max(acc, c);
            }
            return acc + (st.turn == w && st.n_children ? 1u : 0u);
        };
        // pending leaves: at a walker node all children are pushed, one is popped and explored first
        std::function<uint32_t(uint32_t)> stk = [&](uint32_t s) -> uint32_t {
            const rp_state& st = h->states[s];
            if (!st.n_children) return 0;
            uint32_t deepest = 0;
            for (uint32_t k = 0; k < st.n_children; ++k) deepest = std::max(deepest, stk(h->children[st.offset + k]));
            uint32_t pushed = st.turn == w ? st.n_children : 1u;
            return pushed - 1 + std::max(deepest, 1u);
        };
        // internal (expanded) nodes of a sampled tree
        std::function<uint32_t(uint32_t)> ins = [&](uint32_t s) -> uint32_t {
            const rp_state& st = h->states[s];
            if (!st.n_children) return 0;
            uint32_t acc = 0;
            for (uint32_t k = 0; k < st.n_children; ++k) {
                uint32_t c = ins(h->children[st.offset + k]);
                acc = st.turn == w ? acc + c : std::max(acc, c);
            }
            return acc + 1;
        };
        best_int = std::max(best_int, ins(t.train_root));
        best_dec = std::max(best_dec, wn(t.train_root));
        best_stack = std::max(best_stack, stk(t.train_root));
    }
    *maxdec = best_dec;
    *maxstack = best_stack + 1;
    *maxint = best_int;
}

size_t chunk_maps_lds_bytes(const rp_mccfr* h);
int alloc_batch_buffers(rp_mccfr* h, uint32_t batch) {
    if (batch <= h->capacity) return RP_OK;
    if (h->d_scratch) HIP_TRY(hipFree(h->d_scratch));
    if (h->d_dec) HIP_TRY(hipFree(h->d_dec));
    h->d_scratch = h->d_dec = nullptr;
    const size_t stride = ((size_t)batch + 255) & ~(size_t)255;
    const uint32_t A = h->tbl.max_actions;
    DevScratch& sc = h->sc;
    sc.stride = stride;
    const size_t node_words = (size_t)sc.maxn * stride, stack_words = (size_t)sc.maxs * stride;
    const size_t total_words = 8 * node_words + 4 * stack_words + (size_t)A * stride;
    HIP_TRY(hipMalloc(&h->d_scratch, total_words * 4));
    uint32_t* base = reinterpret_cast<uint32_t*>(h->d_scratch);
    sc.n_meta = base; base += node_words;
    sc.n_info = base; base += node_words;
    sc.n_frel = reinterpret_cast<float*>(base); base += node_words;
    sc.n_fsmp = reinterpret_cast<float*>(base); base += node_words;
    sc.n_pay = reinterpret_cast<float*>(base); base += node_words;
    sc.n_rel = reinterpret_cast<float*>(base); base += node_words;
    sc.n_smp = reinterpret_cast<float*>(base); base += node_words;
    sc.n_acc = reinterpret_cast<float*>(base); base += node_words;
    sc.s_state = base; base += stack_words;
    sc.s_meta = base; base += stack_words;
    sc.s_frel = reinterpret_cast<float*>(base); base += stack_words;
    sc.s_fsmp = reinterpret_cast<float*>(base); base += stack_words;
    sc.t_v = reinterpret_cast<float*>(base);

    DevDecisions& dc = h->dc;
    dc.stride = stride;
    dc.maxdec = h->maxdec;
    const size_t slot_words = (size_t)dc.maxdec * stride;
    // chunk-local sort: no per-infoset slot map (RP_MCCFR_SLOTMAP=1 forces the large-game path, for tests)
    const bool small = chunk_maps_lds_bytes(h) <= 64 * 1024 && h->tbl.n_infos <= CM_PASSES * CH_THREADS && !getenv("RP_MCCFR_SLOTMAP");
    const size_t dec_bytes = (3 * slot_words + 2 * slot_words * A) * 4 + (small ? 0 : (size_t)h->tbl.n_infos * stride) + stride;
    HIP_TRY(hipMalloc(&h->d_dec, dec_bytes));
    uint32_t* d = reinterpret_cast<uint32_t*>(h->d_dec);
    dc.info = d; d += slot_words;
    dc.mask = d; d += slot_words;
    dc.payoff = reinterpret_cast<float*>(d); d += slot_words;
    dc.regret = reinterpret_cast<float*>(d); d += slot_words * A;
    dc.policy = reinterpret_cast<float*>(d); d += slot_words * A;
    dc.ndec = reinterpret_cast<uint8_t*>(d);
    dc.slotmap = small ? nullptr : dc.ndec + stride;
    // tree-ordered segments: at most maxdec Decisions per tree
    if (h->d_sorted) HIP_TRY(hipFree(h->d_sorted));
    h->d_sorted = nullptr;
    DevSorted& so = h->so;
    so.n_chunks = (uint32_t)((stride + CH_TREES - 1) / CH_TREES);
    const size_t cap = slot_words;
    const size_t nic = (size_t)h->tbl.n_infos * so.n_chunks;
    const size_t sorted_words = cap * 2 * A + 2 * cap + 2 * nic + h->tbl.n_infos;
    HIP_TRY(hipMalloc(&h->d_sorted, sorted_words * 4));
    uint32_t* sw = reinterpret_cast<uint32_t*>(h->d_sorted);
    so.rw = reinterpret_cast<float*>(sw); sw += cap * 2 * A;
    so.mask = sw; sw += cap;
    so.payoff = reinterpret_cast<float*>(sw); sw += cap;
    so.counts = sw; sw += nic;
    so.offs = sw; sw += nic;
    so.total = sw;
    // per-(infoset, block) maps of the composed update
    if (h->d_bmaps) HIP_TRY(hipFree(h->d_bmaps));
    h->d_bmaps = nullptr;
    const size_t nblk_max = (stride + RP_COMPOSE_CHUNK - 1) / RP_COMPOSE_CHUNK;
    // ... followed by the group maps and the arrival counters of k_combine2
    const size_t ngrp_max = (nblk_max + RP_FOLD_GROUP - 1) / RP_FOLD_GROUP;
    const size_t per_slot = 2 * A * sizeof(Map) + sizeof(float) + sizeof(uint32_t);
    const size_t bytes = (size_t)h->tbl.n_infos * (nblk_max + ngrp_max) * per_slot + (size_t)h->tbl.n_infos * sizeof(uint32_t);
    HIP_TRY(hipMalloc(&h->d_bmaps, bytes));
    HIP_TRY(hipMemset(h->d_bmaps, 0, bytes));
    HIP_TRY(hipDeviceSynchronize());  // the memset runs on the null stream; the solver's stream is non-blocking (rp_mccfr_set_batch)
    h->capacity = batch;
    return RP_OK;
}

StepParams make_params(const rp_mccfr* h) {
    StepParams p{};
    p.seed = h->seed;
    p.epoch = h->epoch;
    p.tree_base = (uint64_t)h->rank * h->batch;
    p.batch = h->batch;
    p.walker = (uint32_t)(h->epoch % h->tbl.n_players);  // CfrSampling::walker (book.rs:142-144)
    p.R = h->R; p.W = h->W; p.S = h->S;
    p.temperature = h->hp.temperature; p.smoothing = h->hp.smoothing; p.curiosity = h->hp.curiosity;
    p.prune_threshold = h->hp.prune_threshold; p.prune_explore = h->hp.prune_explore;
    p.prune_warmup = h->hp.prune_warmup;
    p.regret_min = h->hp.regret_min;
    if (h->R == RP_REGRET_DISCOUNTED) {  // (t / PERIOD).powf(ALPHA | BETA), PERIOD = 1 (discounted.rs:23,33,37)
        p.pow15 = rp_pow15((float)h->epoch);
        p.pow05 = rp_pow05((float)h->epoch);
    }
    p.counters = h->d_counters;
    if (h->rng == RP_RNG_REFERENCE) {
        p.ref_info = reinterpret_cast<const rp_sip_mid*>(h->d_ref_mid);
        p.ref_chance = p.ref_info + h->tbl.n_infos;
    }
    return p;
}

void clock_begin(rp_mccfr* h, KernelClock& c) {
    if (!h->profiling) return;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, h->stream);
    c.pending.emplace_back(a, b);
}
void clock_end(rp_mccfr* h, KernelClock& c) {
    if (!h->profiling) return;
    (void)hipEventRecord(c.pending.back().second, h->stream);
    c.launches += 1;
}
void clock_drain(KernelClock& c) {
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

// Does the game's state table match skeleton G node for node, for EVERY chance outcome?  (traverse_static.hpp)  Also: an
// infoset belongs to one skeleton node only, so a sampled tree meets each walker infoset at most once (a span of the
// reference's Tree::partition has a single root) — the static kernel writes one Decisions per live walker node.
template <class G>
bool skel_matches(const rp_game_table* game, const std::vector<uint32_t>& children) {
    constexpr Skeleton S = SkelOf<G>::S;
    if (game->n_players != 2 || game->max_actions != 2) return false;
    std::vector<std::vector<int>> kids(S.n);
    for (int s = 1; s < S.n; ++s) kids[S.parent[s]].push_back(s);
    std::vector<int> node_of_info(game->n_infos, -1);
    std::function<bool(uint32_t, int)> match = [&](uint32_t sid, int s) -> bool {
        const rp_state& st = game->states[sid];
        if (S.kind[s] == SK_TERMINAL) return st.n_children == 0;
        if (S.kind[s] == SK_CHANCE) {
            if (st.turn != RP_TURN_CHANCE || st.n_children == 0 || kids[s].size() != 1) return false;
            for (uint32_t k = 0; k < st.n_children; ++k)
                if (!match(children[st.offset + k], kids[s][0])) return false;
            return true;
        }
        if (st.turn != (uint8_t)(S.kind[s] - SK_P0) || st.n_children != 2 || kids[s].size() != 2 || st.info >= game->n_infos) return false;
        if (node_of_info[st.info] >= 0 && node_of_info[st.info] != s) return false;
        node_of_info[st.info] = s;
        for (int c : kids[s])
            if (S.edge[c] > 1 || !match(children[st.offset + (uint32_t)S.edge[c]], c)) return false;
        return kids[s][0] != kids[s][1] && S.edge[kids[s][0]] != S.edge[kids[s][1]];
    };
    return match(game->train_root, 0);
}

// DevGame::flat for a game that matches skeleton G: records re-indexed by (skeleton node, chance outcomes on its path).  Returns
// false (no table) when two instances of a skeleton chance node differ in their number of outcomes.
template <class G>
bool build_flat(const rp_game_table* game, const std::vector<uint32_t>& children, const std::function<uint4(uint32_t)>& rec_of,
                std::vector<uint4>& flat, uint32_t* base, uint32_t* fan) {
    constexpr Skeleton S = SkelOf<G>::S;
    std::vector<std::vector<int>> kids(S.n);
    for (int s = 1; s < S.n; ++s) kids[S.parent[s]].push_back(s);
    for (int s = 0; s < S.n; ++s) fan[s] = 0;
    bool uniform = true;
    std::function<void(uint32_t, int)> fans = [&](uint32_t sid, int s) {
        const rp_state& st = game->states[sid];
        if (S.kind[s] == SK_CHANCE) {
            if (fan[s] == 0) fan[s] = st.n_children;
            else if (fan[s] != st.n_children) uniform = false;
            for (uint32_t k = 0; k < st.n_children; ++k) fans(children[st.offset + k], kids[s][0]);
        } else if (S.kind[s] != SK_TERMINAL) {
            for (int c : kids[s]) fans(children[st.offset + (uint32_t)S.edge[c]], c);
        }
    };
    fans(game->train_root, 0);
    if (!uniform) return false;
    // entries of node s = product of the fans of its chance ancestors
    std::vector<uint64_t> count(S.n, 1);
    uint64_t total = 0;
    for (int s = 0; s < S.n; ++s) {
        for (int c = 0; c < s; ++c)
            if (S.kind[c] == SK_CHANCE && s <= S.end[c]) count[s] *= fan[c];
        base[s] = (uint32_t)total;
        total += count[s];
    }
    if (total > (1ull << 26)) return false;
    flat.assign(total, make_uint4(0, 0, 0, 0));
    std::function<void(uint32_t, int, uint64_t)> fill = [&](uint32_t sid, int s, uint64_t idx) {
        const rp_state& st = game->states[sid];
        flat[base[s] + idx] = rec_of(sid);
        if (S.kind[s] == SK_CHANCE) {
            for (uint32_t k = 0; k < st.n_children; ++k) fill(children[st.offset + k], kids[s][0], idx * fan[s] + k);
        } else if (S.kind[s] != SK_TERMINAL) {
            for (int c : kids[s]) fill(children[st.offset + (uint32_t)S.edge[c]], c, idx);
        }
    };
    fill(game->train_root, 0, 0);
    return true;
}

size_t traverse_lds_bytes(const rp_mccfr* h) {
    const size_t shared = std::max<size_t>(2 * (size_t)h->maxint, 4 * (size_t)h->sc.maxs);  // stack, then reach prefixes
    return ((size_t)2 * h->sc.maxn + shared + (h->tbl.max_actions <= 4 ? 0 : h->tbl.max_actions)) * 64 * 4;
}
bool traverse_fits_lds(const rp_mccfr* h) {
    return h->sc.maxn <= 62 && h->maxint <= 32 && h->tbl.max_depth <= 10 && h->tbl.n_infos <= 8191 && h->tbl.max_actions <= 16 &&
           traverse_lds_bytes(h) <= 64 * 1024;
}

// k_prepare_infos, unless the tables have not changed since the per-infoset tables were last derived from them (an exchange
// window traverses against a frozen table; k_combine2<APPLY> refreshes the rows it changes itself)
void launch_prepare_ref(rp_mccfr* h, const StepParams& p) {
    if (h->rng == RP_RNG_REFERENCE && h->ref_mid_epoch != p.epoch) {
        const uint32_t n = h->tbl.n_infos + h->n_chance_infos;
        rp_sip_mid* mid = reinterpret_cast<rp_sip_mid*>(h->d_ref_mid);
        hipLaunchKernelGGL(k_prepare_ref, dim3((n + 63) / 64), dim3(64), 0, h->stream, reinterpret_cast<const rp_hash_stream*>(h->d_info_streams),
                           h->tbl.n_infos, reinterpret_cast<const rp_hash_stream*>(h->d_chance_streams), h->n_chance_infos, p.epoch, mid,
                           mid + h->tbl.n_infos);
        h->ref_mid_epoch = p.epoch;
    }
}
void launch_prepare(rp_mccfr* h, const StepParams& p) {
    launch_prepare_ref(h, p);
    if (h->itab_version == h->tables_version) return;
    hipLaunchKernelGGL(k_prepare_infos, dim3((h->tbl.n_infos + 63) / 64), dim3(64), 0, h->stream, h->g, h->t, p, h->itab);
    h->itab_version = h->tables_version;
}

int launch_traverse(rp_mccfr* h, const StepParams& p) {
    if (h->dc.slotmap) HIP_TRY(hipMemsetAsync(h->dc.slotmap, 0, (size_t)h->tbl.n_infos * h->dc.stride, h->stream));
    clock_begin(h, h->clk_traverse);
    if (h->static_skel) {
        launch_prepare(h, p);
        const dim3 grid((h->batch + 255) / 256), block(256);
        const bool pr = p.S != RP_SAMPLING_EXTERNAL, rf = p.ref_info != nullptr;
#define LAUNCH_STATIC(G, WK)                                                                                                   \
    do {                                                                                                                       \
        if (pr && rf) hipLaunchKernelGGL((k_traverse_static<G, WK, true, true>), grid, block, 0, h->stream, h->g, h->itab, h->dc, p);    \
        else if (pr) hipLaunchKernelGGL((k_traverse_static<G, WK, true, false>), grid, block, 0, h->stream, h->g, h->itab, h->dc, p);   \
        else if (rf) hipLaunchKernelGGL((k_traverse_static<G, WK, false, true>), grid, block, 0, h->stream, h->g, h->itab, h->dc, p);   \
        else hipLaunchKernelGGL((k_traverse_static<G, WK, false, false>), grid, block, 0, h->stream, h->g, h->itab, h->dc, p);          \
    } while (0)
        if (h->static_skel == 1) {
            if (p.walker == 0) LAUNCH_STATIC(KuhnSkel, 0);
            else LAUNCH_STATIC(KuhnSkel, 1);
        } else {
            if (p.walker == 0) LAUNCH_STATIC(LeducSkel, 0);
            else LAUNCH_STATIC(LeducSkel, 1);
        }
#undef LAUNCH_STATIC
    } else if (h->use_lds_traverse) {
        launch_prepare(h, p);
        const size_t lds = traverse_lds_bytes(h);
        const dim3 grid((h->batch + 63) / 64), block(64);
#define LAUNCH_TRAVERSE(TV) \
    hipLaunchKernelGGL((k_traverse_lds<TV>), grid, block, lds, h->stream, h->g, h->itab, h->dc, p, h->sc.maxn, h->sc.maxs, h->maxint)
        if (h->tbl.max_actions <= 4) LAUNCH_TRAVERSE(true);
        else LAUNCH_TRAVERSE(false);
#undef LAUNCH_TRAVERSE
    } else {
        launch_prepare_ref(h, p);
        const uint32_t threads = 256, blocks = (h->batch + threads - 1) / threads;
        hipLaunchKernelGGL(k_traverse, dim3(blocks), dim3(threads), 0, h->stream, h->g, h->t, h->sc, h->dc, p);
    }
    clock_end(h, h->clk_traverse);
    HIP_TRY(hipGetLastError());
    return RP_OK;
}

// stable counting sort of the batch's Decisions into per-infoset, tree-ordered segments
int launch_sort(rp_mccfr* h, const StepParams& p) {
    const uint32_t nchunks = (h->batch + CH_TREES - 1) / CH_TREES;
    h->so.n_chunks = nchunks;
    clock_begin(h, h->clk_compact);
    if (!h->dc.slotmap) {
        const size_t NI = h->tbl.n_infos;
        hipLaunchKernelGGL(k_count_small, dim3(nchunks), dim3(CH_THREADS), NI * SM_WORDS * 4, h->stream, h->g, h->dc, h->so, p);
        hipLaunchKernelGGL(k_scan, dim3(h->tbl.n_infos), dim3(CH_THREADS), 0, h->stream, h->g, h->so, p);
        hipLaunchKernelGGL(k_compact_small, dim3(nchunks), dim3(CH_THREADS), NI * SM_WORDS * 4 + 2 * NI * 4 + NI * SM_WORDS * 2,
                           h->stream, h->g, h->dc, h->so, p);
    } else {
        hipLaunchKernelGGL(k_count, dim3(nchunks, h->tbl.n_infos), dim3(SLOT_THREADS), 0, h->stream, h->g, h->dc, h->so, p);
        hipLaunchKernelGGL(k_scan, dim3(h->tbl.n_infos), dim3(CH_THREADS), 0, h->stream, h->g, h->so, p);
        hipLaunchKernelGGL(k_compact, dim3(nchunks, h->tbl.n_infos), dim3(SLOT_THREADS), 0, h->stream, h->g, h->dc, h->so, p);
    }
    clock_end(h, h->clk_compact);
    HIP_TRY(hipGetLastError());
    return RP_OK;
}

int launch_chain(rp_mccfr* h, const StepParams& p) {
    const size_t lds = chain_lds_bytes(h->tbl.max_actions);
    const bool sgn = h->R == RP_REGRET_DISCOUNTED || h->R == RP_REGRET_ASYMMETRIC;
    const bool prn = h->S != RP_SAMPLING_EXTERNAL;
    const dim3 grid(h->tbl.n_infos), block(128);
    clock_begin(h, h->clk_update);
    if (sgn && prn) hipLaunchKernelGGL((k_chain<true, true>), grid, block, lds, h->stream, h->g, h->t, h->so, p);
    else if (sgn) hipLaunchKernelGGL((k_chain<true, false>), grid, block, lds, h->stream, h->g, h->t, h->so, p);
    else if (prn) hipLaunchKernelGGL((k_chain<false, true>), grid, block, lds, h->stream, h->g, h->t, h->so, p);
    else hipLaunchKernelGGL((k_chain<false, false>), grid, block, lds, h->stream, h->g, h->t, h->so, p);
    clock_end(h, h->clk_update);
    HIP_TRY(hipGetLastError());
    return RP_OK;
}

// block bookkeeping of the composed update: one block per (infoset, chunk of RP_COMPOSE_CHUNK trees)
uint32_t chunks_of(size_t trees) { return (uint32_t)((trees + RP_COMPOSE_CHUNK - 1) / RP_COMPOSE_CHUNK); }
size_t chunk_maps_lds_bytes(const rp_mccfr* h) {
    const size_t NI = h->tbl.n_infos;
    return NI * SM_WORDS * 4 + 2 * NI * 4 + (size_t)CH_TREES * h->maxdec * 4 + NI * SM_WORDS * 2 +
           2 * (size_t)CH_TREES * h->maxdec * 2;
}

// Decisions of the batch -> one composed map per table cell (+ payoff sum and count per infoset) in `blob_dev`.
// Small games compose straight from the traversal's output; large ones from the sorted segments (launch_sort first).
// the traversal and the block maps in one kernel (k_traverse_maps_static): composed update of a small game whose traversal is
// instantiated over its skeleton, external sampling
size_t traverse_maps_lds_bytes(const rp_mccfr* h) {
    const size_t NI = h->tbl.n_infos;
    const size_t masks = h->S != RP_SAMPLING_EXTERNAL ? (size_t)h->maxdec * 256 * 4 : 0;  // the expanded edges of every list entry
    return (NI * 8 + 2 * NI) * 4 + NI * 8 * 2 + (NI + (NI & 1)) * 2 + (size_t)5 * (h->maxdec * 256 + h->cell_pad) * 4 + masks;
}
bool traverse_maps_fused(const rp_mccfr* h) {
    return h->static_skel && !h->dc.slotmap && h->tbl.n_infos <= CH_THREADS &&
           traverse_maps_lds_bytes(h) <= 64 * 1024 && h->fuse_maps;
}

// apply = true: the summary is applied to the tables by the fold itself (single-GPU step), `blob_dev` is not written
int launch_summarize(rp_mccfr* h, const StepParams& p, void* blob_dev, bool fused = false, bool apply = false) {
    unsigned char* blob = reinterpret_cast<unsigned char*>(blob_dev);
    Cell* cells = reinterpret_cast<Cell*>(blob);
    InfoSum* sums = reinterpret_cast<InfoSum*>(blob + (size_t)h->tbl.n_infos * h->tbl.max_actions * sizeof(Cell));
    const uint32_t A = h->tbl.max_actions;
    const uint32_t nblk_max = chunks_of(h->dc.stride), nblk = chunks_of(h->batch);
    const size_t slots = (size_t)h->tbl.n_infos * nblk_max;
    Map* bmaps = reinterpret_cast<Map*>(h->d_bmaps);
    float* bpsum = reinterpret_cast<float*>(bmaps + slots * 2 * A);
    uint32_t* bcnt = reinterpret_cast<uint32_t*>(bpsum + slots);
    const bool pruned = h->S != RP_SAMPLING_EXTERNAL;
    if (fused) {
        clock_begin(h, h->clk_traverse);
        launch_prepare(h, p);
        const size_t lds = traverse_maps_lds_bytes(h);
#define LAUNCH_FUSED_AS(G, WK, PR, RF)                                                                                                  \
    hipLaunchKernelGGL((k_traverse_maps_static<G, WK, PR, RF>), dim3(nblk), dim3(256), lds, h->stream, h->g, h->itab, p, bmaps, bpsum, \
                       bcnt, nblk_max, h->maxdec, h->cell_pad)
#define LAUNCH_FUSED(G, WK)                                    \
    do {                                                       \
        const bool rf = p.ref_info != nullptr;                 \
        if (pruned && rf) LAUNCH_FUSED_AS(G, WK, true, true);  \
        else if (pruned) LAUNCH_FUSED_AS(G, WK, true, false);  \
        else if (rf) LAUNCH_FUSED_AS(G, WK, false, true);      \
        else LAUNCH_FUSED_AS(G, WK, false, false);             \
    } while (0)
        if (h->static_skel == 1) {
            if (p.walker == 0) LAUNCH_FUSED(KuhnSkel, 0);
            else LAUNCH_FUSED(KuhnSkel, 1);
        } else {
            if (p.walker == 0) LAUNCH_FUSED(LeducSkel, 0);
            else LAUNCH_FUSED(LeducSkel, 1);
        }
#undef LAUNCH_FUSED
#undef LAUNCH_FUSED_AS
        clock_end(h, h->clk_traverse);
    }
    clock_begin(h, h->clk_update);
    if (fused) {
    } else if (!h->dc.slotmap) {
        const size_t lds = chunk_maps_lds_bytes(h);
#define LAUNCH_CHUNK_MAPS(PR, PS) \
    hipLaunchKernelGGL((k_chunk_maps<PR, PS>), dim3(nblk), dim3(CH_THREADS), lds, h->stream, h->g, h->dc, p, bmaps, bpsum, bcnt, nblk_max)
        const bool one = h->tbl.n_infos <= CH_THREADS;
        if (pruned && one) LAUNCH_CHUNK_MAPS(true, 1);
        else if (pruned) LAUNCH_CHUNK_MAPS(true, CM_PASSES);
        else if (one) LAUNCH_CHUNK_MAPS(false, 1);
        else LAUNCH_CHUNK_MAPS(false, CM_PASSES);
#undef LAUNCH_CHUNK_MAPS
    } else if (pruned) {
        hipLaunchKernelGGL((k_block_maps<true>), dim3(nblk, h->tbl.n_infos), dim3(128), 0, h->stream, h->g, h->so, p, bmaps, bpsum, bcnt, nblk_max);
    } else {
        hipLaunchKernelGGL((k_block_maps<false>), dim3(nblk, h->tbl.n_infos), dim3(128), 0, h->stream, h->g, h->so, p, bmaps, bpsum, bcnt, nblk_max);
    }
    {
        const uint32_t W2 = 2 * A, GPW = cb2_gpw(W2);
        const size_t ngrp_max = (nblk_max + RP_FOLD_GROUP - 1) / RP_FOLD_GROUP, gslots = (size_t)h->tbl.n_infos * ngrp_max;
        FoldScratch fs;
        fs.gmaps = reinterpret_cast<Map*>(bcnt + slots);
        fs.gpsum = reinterpret_cast<float*>(fs.gmaps + gslots * W2);
        fs.gcnt = reinterpret_cast<uint32_t*>(fs.gpsum + gslots);
        fs.done = fs.gcnt + gslots;
        fs.ngrp_max = (uint32_t)ngrp_max;
        const uint32_t ngrp = (nblk + RP_FOLD_GROUP - 1) / RP_FOLD_GROUP, nparts = (ngrp + GPW - 1) / GPW;
        const size_t lds = (size_t)GPW * RP_FOLD_GROUP * (W2 * 16 + 8);
        const dim3 grid(h->tbl.n_infos, std::max(nparts, 1u));
        if (apply) {
            hipLaunchKernelGGL((k_combine2<true>), grid, dim3(256), lds, h->stream, h->g, h->t, h->itab, p, bmaps, bpsum, bcnt, nblk_max, fs, cells, sums);
            h->tables_version += 1;  // the kernel refreshes the rows of the per-infoset tables it changes
            if (h->itab_version + 1 == h->tables_version) h->itab_version = h->tables_version;
        } else {
            hipLaunchKernelGGL((k_combine2<false>), grid, dim3(256), lds, h->stream, h->g, h->t, h->itab, p, bmaps, bpsum, bcnt, nblk_max, fs, cells, sums);
        }
    }
    clock_end(h, h->clk_update);
    HIP_TRY(hipGetLastError());
    return RP_OK;
}

// Solver::batch + the composed maps of its Decisions -> one summary blob
int launch_batch_summary(rp_mccfr* h, const StepParams& p, void* blob_dev, bool apply = false) {
    if (traverse_maps_fused(h)) return launch_summarize(h, p, blob_dev, true, apply);
    int rc = launch_traverse(h, p);
    if (rc) return rc;
    if (h->dc.slotmap && (rc = launch_sort(h, p))) return rc;
    return launch_summarize(h, p, blob_dev, false, apply);
}

int enqueue_step(rp_mccfr* h) {
    const StepParams p = make_params(h);
    int rc;
    // the ordered chains and the large-game composed path read the sorted segments; small games compose from the
    // traversal's output directly
    if (h->mode == RP_UPDATE_ORDERED) {
        if ((rc = launch_traverse(h, p))) return rc;
        if ((rc = launch_sort(h, p))) return rc;
        if ((rc = launch_chain(h, p))) return rc;
        h->tables_version += 1;
    } else {
        // the fold applies the summary to the tables itself (k_combine2<APPLY>: fold + k_fold + k_prepare_infos in one launch)
        if ((rc = launch_batch_summary(h, p, h->d_summary, true))) return rc;
    }
    h->epoch += 1;  // CfrSampling::increment via Solver::advance (solver.rs:103-104)
    return RP_OK;
}

// nodes, infos, error flags: the sums (resp. the OR) over the stripes
int read_counters(rp_mccfr* h, unsigned long long c[3]) {
    std::vector<unsigned long long> all((size_t)METRIC_STRIPES * METRIC_STRIDE);
    HIP_TRY(hipMemcpyAsync(all.data(), h->d_counters, all.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    c[0] = c[1] = c[2] = 0;
    for (uint32_t s = 0; s < METRIC_STRIPES; ++s) {
        c[0] += all[(size_t)s * METRIC_STRIDE];
        c[1] += all[(size_t)s * METRIC_STRIDE + 1];
        c[2] |= all[(size_t)s * METRIC_STRIDE + 2];
    }
    return RP_OK;
}

int check_device_errors(rp_mccfr* h) {
    unsigned long long c[3];
    int rc = read_counters(h, c);
    if (rc) return rc;
    if (c[2]) return rp::fail(RP_ERR_CAPACITY, "mccfr kernel capacity exceeded (flags %llu): nodes/stack/decisions", c[2]);
    return RP_OK;
}

int composed_supported(const rp_mccfr* h) {
    if (h->R == RP_REGRET_DISCOUNTED || h->R == RP_REGRET_ASYMMETRIC)
        return rp::fail(RP_ERR_UNSUPPORTED,
                        "composed update needs a sign-independent discount (Summed/Linear/Floored regret)");
    return RP_OK;
}

int fetch_tables(rp_mccfr* h, std::vector<float>& regret, std::vector<float>& weight, std::vector<float>& payoff,
                 std::vector<uint32_t>& visits) {
    const size_t cells = (size_t)h->tbl.n_infos * h->tbl.max_actions;
    regret.resize(cells); weight.resize(cells); payoff.resize(cells); visits.resize(cells);
    HIP_TRY(hipMemcpyAsync(regret.data(), h->t.regret, cells * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(weight.data(), h->t.weight, cells * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(payoff.data(), h->t.payoff, cells * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(visits.data(), h->t.visits, cells * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RP_OK;
}

}  // namespace

extern "C" {

int rp_mccfr_create(const rp_game_table* game, rp_regret_kind r, rp_weight_kind w, rp_sampling_kind s,
                    uint32_t batch_size, const rp_hyper* hp, uint64_t seed, int device, rp_mccfr** out) {
    if (!game || !out) return rp::fail(RP_ERR_INVALID, "rp_mccfr_create: NULL argument");
    int rc = rp_game_table_check(game);
    if (rc) return rc;
    if ((int)r < 0 || (int)r > 4 || (int)w < 0 || (int)w > 3 || (int)s < 0 || (int)s > 2)
        return rp::fail(RP_ERR_INVALID, "rp_mccfr_create: unknown schedule / sampling kind");
    if (batch_size == 0) return rp::fail(RP_ERR_INVALID, "rp_mccfr_create: batch_size must be > 0");
    if (rp_device_count() <= 0)
        return rp::fail(RP_ERR_NO_DEVICE, "rp_mccfr_create: no HIP device visible; the MI355X path has no CPU fallback");
    if (game->max_tree_nodes == 0 || game->max_tree_nodes >= 255)
        return rp::fail(RP_ERR_CAPACITY, "rp_mccfr_create: sampled trees of up to %u nodes exceed the per-lane limit (254)",
                        game->max_tree_nodes);
    rp_mccfr* h = new rp_mccfr();
    h->device = device;
    h->tbl = *game;
    h->states.assign(game->states, game->states + game->n_states);
    h->children.assign(game->children, game->children + game->n_children);
    h->payoffs.assign(game->payoffs, game->payoffs + (size_t)game->n_terminals * game->n_players);
    h->info_actions.assign(game->info_actions, game->info_actions + game->n_infos);
    h->info_player.assign(game->info_player, game->info_player + game->n_infos);
    const size_t cells = (size_t)game->n_infos * game->max_actions;
    h->default_regret.assign(cells, 0.0f);
    if (game->default_regret) h->default_regret.assign(game->default_regret, game->default_regret + cells);
    h->tbl.states = h->states.data();
    h->tbl.children = h->children.data();
    h->tbl.payoffs = h->payoffs.data();
    h->tbl.info_actions = h->info_actions.data();
    h->tbl.info_player = h->info_player.data();
    h->tbl.default_regret = h->default_regret.data();
    h->R = r; h->W = w; h->S = s;
    if (hp) h->hp = *hp; else rp_hyper_default(&h->hp);
    h->seed = seed;
    h->batch = batch_size;

#define CREATE_TRY(expr)                                                                             \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            int _rc = rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));           \
            rp_mccfr_destroy(h);                                                                     \
            return _rc;                                                                              \
        }                                                                                            \
    } while (0)

    CREATE_TRY(hipSetDevice(device));
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    // game table -> HBM (states repacked to 16 B)
    std::vector<uint4> packed(game->n_states);
    for (uint32_t i = 0; i < game->n_states; ++i) {
        const rp_state& st = game->states[i];
        packed[i] = make_uint4((uint32_t)st.turn | ((uint32_t)st.n_children << 8), st.turn == RP_TURN_CHANCE ? (uint32_t)st.chance_info : st.info, st.offset, 0u);
    }
    CREATE_TRY(hipMalloc(&h->d_states, packed.size() * sizeof(uint4)));
    CREATE_TRY(hipMemcpy(h->d_states, packed.data(), packed.size() * sizeof(uint4), hipMemcpyHostToDevice));
    CREATE_TRY(hipMalloc(&h->d_children, h->children.size() * 4));
    CREATE_TRY(hipMemcpy(h->d_children, h->children.data(), h->children.size() * 4, hipMemcpyHostToDevice));
    // child records: one load on arrival at a node instead of children[] -> states[] -> payoffs[]
    std::vector<uint4> kids(h->children.size());
    for (size_t c = 0; c < h->children.size(); ++c) {
        const uint32_t sid = h->children[c];
        uint4 r = packed[sid];
        r.w = sid;
        if (game->states[sid].n_children == 0 && game->n_players == 2) {
            r.y = rp_f2u(h->payoffs[(size_t)game->states[sid].offset * 2 + 0]);
            r.z = rp_f2u(h->payoffs[(size_t)game->states[sid].offset * 2 + 1]);
        }
        kids[c] = r;
    }
    CREATE_TRY(hipMalloc(&h->d_kids, std::max<size_t>(kids.size(), 1) * sizeof(uint4)));
    CREATE_TRY(hipMemcpy(h->d_kids, kids.data(), kids.size() * sizeof(uint4), hipMemcpyHostToDevice));
    CREATE_TRY(hipMalloc(&h->d_payoffs, h->payoffs.size() * 4));
    CREATE_TRY(hipMemcpy(h->d_payoffs, h->payoffs.data(), h->payoffs.size() * 4, hipMemcpyHostToDevice));
    CREATE_TRY(hipMalloc(&h->d_info_actions, game->n_infos));
    CREATE_TRY(hipMemcpy(h->d_info_actions, h->info_actions.data(), game->n_infos, hipMemcpyHostToDevice));
    CREATE_TRY(hipMalloc(&h->d_info_player, game->n_infos));
    CREATE_TRY(hipMemcpy(h->d_info_player, h->info_player.data(), game->n_infos, hipMemcpyHostToDevice));
    h->g.states = reinterpret_cast<const uint4*>(h->d_states);
    h->g.children = reinterpret_cast<const uint32_t*>(h->d_children);
    h->g.kids = reinterpret_cast<const uint4*>(h->d_kids);
    h->g.root_rec = packed[game->train_root];
    h->g.root_rec.w = game->train_root;
    h->g.payoffs = reinterpret_cast<const float*>(h->d_payoffs);
    h->g.info_actions = reinterpret_cast<const uint8_t*>(h->d_info_actions);
    h->g.info_player = reinterpret_cast<const uint8_t*>(h->d_info_player);
    h->g.n_players = game->n_players;
    h->g.n_infos = game->n_infos;
    h->g.A = game->max_actions;
    h->g.root = game->train_root;
    // tables: a missing Encounter reads (weight 0, regret default_regret, payoff 0, visits 0) (book.rs:93-122)
    CREATE_TRY(hipMalloc(&h->t.regret, cells * 4));
    CREATE_TRY(hipMalloc(&h->t.weight, cells * 4));
    CREATE_TRY(hipMalloc(&h->t.payoff, cells * 4));
    CREATE_TRY(hipMalloc(&h->t.visits, cells * 4));
    CREATE_TRY(hipMemcpy(h->t.regret, h->default_regret.data(), cells * 4, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemset(h->t.weight, 0, cells * 4));
    CREATE_TRY(hipMemset(h->t.payoff, 0, cells * 4));
    CREATE_TRY(hipMemset(h->t.visits, 0, cells * 4));
    CREATE_TRY(hipMalloc(&h->d_counters, (size_t)METRIC_STRIPES * METRIC_STRIDE * sizeof(unsigned long long)));
    CREATE_TRY(hipMemset(h->d_counters, 0, (size_t)METRIC_STRIPES * METRIC_STRIDE * sizeof(unsigned long long)));
    CREATE_TRY(hipMalloc(&h->d_summary, summary_bytes_of(h)));
    CREATE_TRY(hipMalloc(&h->d_itab, (5 * cells + 2 * (size_t)game->n_infos + 4) * 4));
    {
        float* f = reinterpret_cast<float*>(h->d_itab);
        h->itab.sigma = f;
        h->itab.q = f + cells;
        h->itab.cum = f + 2 * cells;
        h->itab.total = f + 3 * cells;
        h->itab.keep = reinterpret_cast<uint32_t*>(f + 3 * cells + game->n_infos);
        const size_t used = 3 * cells + 2 * (size_t)game->n_infos;
        h->itab.sq = reinterpret_cast<float2*>(f + ((used + 3) & ~(size_t)3));  // 16-byte aligned: a two-action row is one float4
    }
    uint32_t maxstack = 1;
    sampled_tree_bounds(h, &h->maxdec, &maxstack, &h->maxint);
    h->sc.maxn = game->max_tree_nodes;
    h->sc.maxs = maxstack;
    if (h->maxdec > 254) {
        rp_mccfr_destroy(h);
        return rp::fail(RP_ERR_CAPACITY, "rp_mccfr_create: more than 254 walker infosets per tree");
    }
    if (chain_lds_bytes(game->max_actions) > 64 * 1024) {
        rp_mccfr_destroy(h);
        return rp::fail(RP_ERR_CAPACITY, "rp_mccfr_create: chain tiles need %zu B of LDS", chain_lds_bytes(game->max_actions));
    }
    h->use_lds_traverse = traverse_fits_lds(h) && getenv("RP_MCCFR_HBM_SCRATCH") == nullptr;
    h->fuse_maps = getenv("RP_TRAV_UNFUSED") == nullptr;
    if (h->use_lds_traverse && getenv("RP_TRAV_GENERIC") == nullptr) {
        if (skel_matches<KuhnSkel>(game, h->children)) h->static_skel = 1;
        else if (skel_matches<LeducSkel>(game, h->children)) h->static_skel = 2;
    }
    h->g.flat = nullptr;
    if (h->static_skel) {
        const std::function<uint4(uint32_t)> rec_of = [&](uint32_t sid) {
            uint4 r = packed[sid];
            r.w = sid;
            if (game->states[sid].n_children == 0 && game->n_players == 2) {
                r.y = rp_f2u(h->payoffs[(size_t)game->states[sid].offset * 2 + 0]);
                r.z = rp_f2u(h->payoffs[(size_t)game->states[sid].offset * 2 + 1]);
            }
            return r;
        };
        std::vector<uint4> flat;
        const bool ok = h->static_skel == 1 ? build_flat<KuhnSkel>(game, h->children, rec_of, flat, h->g.flat_base, h->g.flat_fan)
                                            : build_flat<LeducSkel>(game, h->children, rec_of, flat, h->g.flat_base, h->g.flat_fan);
        if (ok) {
            CREATE_TRY(hipMalloc(&h->d_flat, flat.size() * sizeof(uint4)));
            CREATE_TRY(hipMemcpy(h->d_flat, flat.data(), flat.size() * sizeof(uint4), hipMemcpyHostToDevice));
            h->g.flat = reinterpret_cast<const uint4*>(h->d_flat);
        }
    }
    rc = alloc_batch_buffers(h, batch_size);
    if (rc) {
        rp_mccfr_destroy(h);
        return rc;
    }
#undef CREATE_TRY
    // hipMemset on device memory returns before it has run, and a non-blocking stream does not wait for the null stream: the
    // first launch on this handle's stream could otherwise overtake the initialisation above and be overwritten by it
    (void)hipDeviceSynchronize();
    *out = h;
    return RP_OK;
}

int rp_mccfr_destroy(rp_mccfr* h) {
    if (!h) return RP_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    clock_drain(h->clk_traverse);
    clock_drain(h->clk_compact);
    clock_drain(h->clk_update);
    void* ptrs[] = {h->d_info_streams, h->d_chance_streams, h->d_ref_mid, h->d_flat, h->d_states, h->d_children, h->d_kids, h->d_payoffs, h->d_info_actions, h->d_info_player, h->d_scratch,
                    h->d_dec, h->d_sorted, h->d_bmaps, h->d_itab, h->d_summary, h->d_window, h->d_counters, h->t.regret, h->t.weight, h->t.payoff,
                    h->t.visits};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return RP_OK;
}

int rp_mccfr_set_rng(rp_mccfr* h, rp_rng_kind kind, const rp_hash_streams* st) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_mccfr_set_rng: NULL handle");
    int rc = set_device(h);
    if (rc) return rc;
    if (kind == RP_RNG_COUNTER) {
        h->rng = RP_RNG_COUNTER;
        return RP_OK;
    }
    if (kind != RP_RNG_REFERENCE) return rp::fail(RP_ERR_INVALID, "rp_mccfr_set_rng: unknown rp_rng_kind");
    if (!st || !st->infos || st->n_infos != h->tbl.n_infos || (st->n_chance && !st->chance))
        return rp::fail(RP_ERR_INVALID, "rp_mccfr_set_rng: RP_RNG_REFERENCE needs the hash stream of each of the game's %u infosets", h->tbl.n_infos);
    for (uint32_t i = 0; i < st->n_infos; ++i)
        if (st->infos[i].len > RP_HASH_STREAM_MAX) return rp::fail(RP_ERR_INVALID, "rp_mccfr_set_rng: stream %u is longer than %u bytes", i, RP_HASH_STREAM_MAX);
    for (uint32_t i = 0; i < st->n_chance; ++i)
        if (st->chance[i].len > RP_HASH_STREAM_MAX) return rp::fail(RP_ERR_INVALID, "rp_mccfr_set_rng: chance stream %u is longer than %u bytes", i, RP_HASH_STREAM_MAX);
    for (const rp_state& s : h->states)
        if (s.turn == RP_TURN_CHANCE && s.chance_info > st->n_chance)
            return rp::fail(RP_ERR_INVALID, "rp_mccfr_set_rng: a chance state names chance info %u of %u", s.chance_info, st->n_chance);
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (void** q : {&h->d_info_streams, &h->d_chance_streams, &h->d_ref_mid}) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    HIP_TRY(hipMalloc(&h->d_info_streams, sizeof(rp_hash_stream) * std::max<size_t>(1, st->n_infos)));
    HIP_TRY(hipMalloc(&h->d_chance_streams, sizeof(rp_hash_stream) * std::max<size_t>(1, st->n_chance)));
    HIP_TRY(hipMalloc(&h->d_ref_mid, sizeof(rp_sip_mid) * ((size_t)st->n_infos + st->n_chance + 1)));
    HIP_TRY(hipMemcpy(h->d_info_streams, st->infos, sizeof(rp_hash_stream) * st->n_infos, hipMemcpyHostToDevice));
    if (st->n_chance) HIP_TRY(hipMemcpy(h->d_chance_streams, st->chance, sizeof(rp_hash_stream) * st->n_chance, hipMemcpyHostToDevice));
    h->n_chance_infos = st->n_chance;
    h->ref_mid_epoch = ~0ull;
    h->rng = RP_RNG_REFERENCE;
    return RP_OK;
}

int rp_mccfr_step_async(rp_mccfr* h, uint32_t steps) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_mccfr_step_async: NULL handle");
    int rc = set_device(h);
    if (rc) return rc;
    if (h->mode == RP_UPDATE_COMPOSED && (rc = composed_supported(h))) return rc;
    for (uint32_t i = 0; i < steps; ++i)
        if ((rc = enqueue_step(h))) return rc;
    return RP_OK;
}

int rp_mccfr_sync(rp_mccfr* h) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_mccfr_sync: NULL handle");
    int rc = set_device(h);
    if (rc) return rc;
    rc = check_device_errors(h);
    clock_drain(h->clk_traverse);
    clock_drain(h->clk_compact);
    clock_drain(h->clk_update);
    return rc;
}

int rp_mccfr_step(rp_mccfr* h) {
    int rc = rp_mccfr_step_async(h, 1);
    if (rc) return rc;
    return rp_mccfr_sync(h);
}

int rp_mccfr_solve(rp_mccfr* h, uint64_t trees) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_mccfr_solve: NULL handle");
    uint64_t steps = trees / h->batch;
    while (steps) {  // bounded queue depth; the loop shape of Solver::solve (solver.rs:111-122)
        const uint32_t n = (uint32_t)std::min<uint64_t>(steps, 256);
        int rc = rp_mccfr_step_async(h, n);
        if (rc) return rc;
        if ((rc = rp_mccfr_sync(h))) return rc;
        steps -= n;
    }
    return RP_OK;
}

int rp_mccfr_spend(rp_mccfr* h, double seconds, uint64_t* iterations, double* elapsed) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_mccfr_spend: NULL handle");
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t it = 0;
    double el = 0.0;
    for (;;) {
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el >= seconds) break;
        int rc = rp_mccfr_step(h);
        if (rc) return rc;
        it += 1;
    }
    if (iterations) *iterations = it;
    if (elapsed) *elapsed = el;
    return RP_OK;
}

// Checkpoint's Display / Progress::format: rp::format_progress (common.cpp), shared with rp_nlhe_train
using rp::format_progress;

int rp_mccfr_train(rp_mccfr* h, uint64_t max_steps, double max_seconds, double log_interval, double flush_interval,
                   rp_train_event_fn on_event, void* user, const volatile int* interrupt, char* summary, size_t summary_cap) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_mccfr_train: NULL handle");
    using clock = std::chrono::steady_clock;
    const auto start = clock::now();
    auto prior = start, flushed = start;
    uint64_t prior_infos = 0, steps = 0, nodes = 0, infos = 0;
    auto secs_since = [](clock::time_point t) { return std::chrono::duration<double>(clock::now() - t).count(); };
    for (;;) {
        int rc = rp_mccfr_step(h);  // Trainer::train: step, then checkpoint, then flush, then the interrupt test
        if (rc) return rc;
        steps += 1;
        if (secs_since(prior) >= log_interval) {  // Metrics::checkpoint (metrics/mod.rs:67-80)
            if ((rc = rp_mccfr_counters(h, &nodes, &infos))) return rc;
            const double secs = std::max(1.0, std::floor(secs_since(prior)));  // elapsed().as_secs().max(1)
            rp_checkpoint cp{h->epoch, nodes, infos, (double)(infos - prior_infos) / secs};
            prior = clock::now();
            prior_infos = infos;
            if (on_event) {
                char line[96];
                format_progress(line, sizeof line, cp.epoch, cp.nodes, cp.infos, cp.rate);
                on_event(RP_TRAIN_CHECKPOINT, &cp, line, user);
            }
        }
        if (secs_since(flushed) >= flush_interval) {  // FastSession::flush cadence (forge/src/fast.rs:97-125)
            flushed = clock::now();
            if (on_event) {
                rp_checkpoint cp{h->epoch, nodes, infos, 0.0};
                on_event(RP_TRAIN_FLUSH, &cp, "", user);
            }
        }
        const bool stop = (interrupt && *interrupt) || (max_steps && steps >= max_steps) ||
                          (max_seconds > 0.0 && secs_since(start) >= max_seconds);
        if (stop) break;
    }
    int rc = rp_mccfr_counters(h, &nodes, &infos);
    if (rc) return rc;
    if (summary && summary_cap) {  // Progress::summary (progress.rs:24-26): rate over the whole run
        char line[96];
        const double secs = std::max(1.0, std::floor(secs_since(start)));
        format_progress(line, sizeof line, h->epoch, nodes, infos, (double)infos / secs);
        snprintf(summary, summary_cap, "training stopped\n%s", line);
    }
    return RP_OK;
}

int rp_mccfr_epoch(rp_mccfr* h, uint64_t* epoch) {
    if (!h || !epoch) return rp::fail(RP_ERR_INVALID, "rp_mccfr_epoch: NULL argument");
    *epoch = h->epoch;
    return RP_OK;
}

int rp_mccfr_counters(rp_mccfr* h, uint64_t* nodes, uint64_t* infos) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_mccfr_counters: NULL handle");
    int rc = set_device(h);
    if (rc) return rc;
    unsigned long long c[3];
    if ((rc = read_counters(h, c))) return rc;
    if (nodes) *nodes = c[0];
    if (infos) *infos = c[1];
    return RP_OK;
}

int rp_mccfr_get(rp_mccfr* h, uint32_t info, uint32_t edge, rp_encounter* out) {
    if (!h || !out) return rp::fail(RP_ERR_INVALID, "rp_mccfr_get: NULL argument");
    if (info >= h->tbl.n_infos || edge >= h->info_actions[info]) return rp::fail(RP_ERR_INVALID, "rp_mccfr_get: out of range");
    int rc = set_device(h);
    if (rc) return rc;
    const size_t c = (size_t)info * h->tbl.max_actions + edge;
    HIP_TRY(hipMemcpyAsync(&out->regret, h->t.regret + c, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(&out->weight, h->t.weight + c, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(&out->payoff, h->t.payoff + c, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(&out->visits, h->t.visits + c, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RP_OK;
}

int rp_mccfr_set(rp_mccfr* h, uint32_t info, uint32_t edge, const rp_encounter* in) {
    if (!h || !in) return rp::fail(RP_ERR_INVALID, "rp_mccfr_set: NULL argument");
    if (info >= h->tbl.n_infos || edge >= h->info_actions[info]) return rp::fail(RP_ERR_INVALID, "rp_mccfr_set: out of range");
    int rc = set_device(h);
    if (rc) return rc;
    const size_t c = (size_t)info * h->tbl.max_actions + edge;
    HIP_TRY(hipMemcpyAsync(h->t.regret + c, &in->regret, 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->t.weight + c, &in->weight, 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->t.payoff + c, &in->payoff, 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->t.visits + c, &in->visits, 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->tables_version += 1;
    return RP_OK;
}

int rp_mccfr_export(rp_mccfr* h, rp_encounter* rows, uint64_t cap) {
    if (!h || !rows) return rp::fail(RP_ERR_INVALID, "rp_mccfr_export: NULL argument");
    const size_t cells = (size_t)h->tbl.n_infos * h->tbl.max_actions;
    if (cap < cells) return rp::fail(RP_ERR_INVALID, "rp_mccfr_export: need %zu rows", cells);
    int rc = set_device(h);
    if (rc) return rc;
    std::vector<float> r, w, p;
    std::vector<uint32_t> v;
    if ((rc = fetch_tables(h, r, w, p, v))) return rc;
    for (size_t c = 0; c < cells; ++c) rows[c] = rp_encounter{w[c], r[c], p[c], v[c]};
    return RP_OK;
}

int rp_mccfr_import(rp_mccfr* h, const rp_encounter* rows, uint64_t n, uint64_t epoch) {
    if (!h || !rows) return rp::fail(RP_ERR_INVALID, "rp_mccfr_import: NULL argument");
    const size_t cells = (size_t)h->tbl.n_infos * h->tbl.max_actions;
    if (n != cells) return rp::fail(RP_ERR_INVALID, "rp_mccfr_import: expected %zu rows", cells);
    int rc = set_device(h);
    if (rc) return rc;
    std::vector<float> r(cells), w(cells), p(cells);
    std::vector<uint32_t> v(cells);
    for (size_t c = 0; c < cells; ++c) {
        w[c] = rows[c].weight; r[c] = rows[c].regret; p[c] = rows[c].payoff; v[c] = rows[c].visits;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(h->t.regret, r.data(), cells * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->t.weight, w.data(), cells * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->t.payoff, p.data(), cells * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->t.visits, v.data(), cells * 4, hipMemcpyHostToDevice));
    h->tables_version += 1;
    h->epoch = epoch;
    return RP_OK;
}

int rp_mccfr_policy(rp_mccfr* h, uint32_t info, rp_dist_kind kind, float* out, uint32_t* n) {
    if (!h || !out) return rp::fail(RP_ERR_INVALID, "rp_mccfr_policy: NULL argument");
    if (info >= h->tbl.n_infos) return rp::fail(RP_ERR_INVALID, "rp_mccfr_policy: info out of range");
    int rc = set_device(h);
    if (rc) return rc;
    const uint32_t A = h->tbl.max_actions, na = h->info_actions[info];
    float reg[RP_MAX_ACTIONS], wgt[RP_MAX_ACTIONS];
    HIP_TRY(hipMemcpyAsync(reg, h->t.regret + (size_t)info * A, na * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(wgt, h->t.weight + (size_t)info * A, na * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (kind == RP_DIST_ITERATED) {  // profile.rs:47-51
        float denom = 0.0f;
        for (uint32_t a = 0; a < na; ++a) denom += rp_maxf(reg[a], RP_EPSILON);
        for (uint32_t a = 0; a < na; ++a) out[a] = rp_maxf(reg[a], RP_EPSILON) / denom;
    } else if (kind == RP_DIST_AVERAGED) {  // profile.rs:40-44
        float sum = 0.0f;
        for (uint32_t a = 0; a < na; ++a) sum += rp_maxf(wgt[a], RP_EPSILON);
        for (uint32_t a = 0; a < na; ++a) out[a] = rp_maxf(wgt[a], RP_EPSILON) / sum;
    } else if (kind == RP_DIST_SAMPLING) {  // flow.rs:33-42
        float denom = 0.0f;
        for (uint32_t a = 0; a < na; ++a) denom += rp_maxf(wgt[a], RP_EPSILON);
        denom = denom + h->hp.smoothing;
        float raw[RP_MAX_ACTIONS], z = 0.0f;
        for (uint32_t a = 0; a < na; ++a) {
            raw[a] = rp_maxf((rp_maxf(wgt[a], RP_EPSILON) / h->hp.temperature + h->hp.smoothing) / denom, h->hp.curiosity);
            z += raw[a];
        }
        for (uint32_t a = 0; a < na; ++a) out[a] = raw[a] / z;
    } else {
        return rp::fail(RP_ERR_INVALID, "rp_mccfr_policy: unknown distribution kind");
    }
    if (n) *n = na;
    return RP_OK;
}

int rp_mccfr_sum_regret(rp_mccfr* h, float* out) {
    if (!h || !out) return rp::fail(RP_ERR_INVALID, "rp_mccfr_sum_regret: NULL argument");
    int rc = set_device(h);
    if (rc) return rc;
    std::vector<float> r, w, p;
    std::vector<uint32_t> v;
    if ((rc = fetch_tables(h, r, w, p, v))) return rc;
    float s = 0.0f;
    for (uint32_t info = 0; info < h->tbl.n_infos; ++info)
        for (uint32_t a = 0; a < h->info_actions[info]; ++a) s += rp_maxf(r[(size_t)info * h->tbl.max_actions + a], 0.0f);
    *out = s / (float)(h->epoch > 1 ? h->epoch : 1);
    return RP_OK;
}

int rp_mccfr_set_batch(rp_mccfr* h, uint32_t batch_size) {
    if (!h || batch_size == 0) return rp::fail(RP_ERR_INVALID, "rp_mccfr_set_batch: bad argument");
    int rc = set_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if ((rc = alloc_batch_buffers(h, batch_size))) return rc;
    h->batch = batch_size;
    return RP_OK;
}

// rp_mccfr_create with the update mode chosen up front (a caller that wants the fast, shardable composed update does not have to
// know about a second call; schedules the composed mode cannot express are refused here, not at the first step)
int rp_mccfr_create_mode(const rp_game_table* game, rp_regret_kind r, rp_weight_kind w, rp_sampling_kind s, uint32_t batch_size,
                         const rp_hyper* hp, uint64_t seed, int device, rp_update_mode mode, rp_mccfr** out) {
    if (mode != RP_UPDATE_ORDERED && mode != RP_UPDATE_COMPOSED) return rp::fail(RP_ERR_INVALID, "rp_mccfr_create_mode: unknown update mode");
    int rc = rp_mccfr_create(game, r, w, s, batch_size, hp, seed, device, out);
    if (rc) return rc;
    (*out)->mode = mode;
    if (mode == RP_UPDATE_COMPOSED && (rc = composed_supported(*out))) {
        rp_mccfr_destroy(*out);
        *out = nullptr;
        return rc;
    }
    return RP_OK;
}

int rp_mccfr_set_update_mode(rp_mccfr* h, rp_update_mode mode) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_mccfr_set_update_mode: NULL handle");
    if (mode != RP_UPDATE_ORDERED && mode != RP_UPDATE_COMPOSED) return rp::fail(RP_ERR_INVALID, "unknown update mode");
    h->mode = mode;
    return RP_OK;
}

int rp_mccfr_set_stream(rp_mccfr* h, void* hip_stream) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_mccfr_set_stream: NULL handle");
    int rc = set_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->own_stream) {
        HIP_TRY(hipStreamDestroy(h->stream));
        h->own_stream = false;
    }
    if (hip_stream) {
        h->stream = reinterpret_cast<hipStream_t>(hip_stream);
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    return RP_OK;
}

int rp_mccfr_set_shard(rp_mccfr* h, uint32_t rank, uint32_t world) {
    if (!h || world == 0 || rank >= world) return rp::fail(RP_ERR_INVALID, "rp_mccfr_set_shard: bad rank/world");
    h->rank = rank;
    h->world = world;
    return RP_OK;
}

int rp_mccfr_summary_bytes(rp_mccfr* h, size_t* bytes) {
    if (!h || !bytes) return rp::fail(RP_ERR_INVALID, "rp_mccfr_summary_bytes: NULL argument");
    *bytes = summary_bytes_of(h);
    return RP_OK;
}

int rp_mccfr_step_local(rp_mccfr* h, void* summary_dev) {
    if (!h || !summary_dev) return rp::fail(RP_ERR_INVALID, "rp_mccfr_step_local: NULL argument");
    int rc = set_device(h);
    if (rc) return rc;
    if ((rc = composed_supported(h))) return rc;
    const StepParams p = make_params(h);
    return launch_batch_summary(h, p, summary_dev);
}

int rp_mccfr_step_apply(rp_mccfr* h, const void* gathered_dev, uint32_t world) {
    if (!h || !gathered_dev || world == 0) return rp::fail(RP_ERR_INVALID, "rp_mccfr_step_apply: bad argument");
    int rc = set_device(h);
    if (rc) return rc;
    const uint32_t ncell = h->tbl.n_infos * h->tbl.max_actions;
    hipLaunchKernelGGL(k_fold, dim3((ncell + 255) / 256), dim3(256), 0, h->stream, h->g, h->t,
                       reinterpret_cast<const unsigned char*>(gathered_dev), summary_bytes_of(h), world);
    HIP_TRY(hipGetLastError());
    h->tables_version += 1;
    h->epoch += 1;
    return RP_OK;
}

// The periodic exchange (north_star: "periodic RCCL all-reduce of regret/strategy tables"): a rank runs `window` local
// steps against the table as it stood at the start of the window — the table is not touched, the epoch advances (the
// sampled trees, the walker and the discounts are those of each step) — and folds each step's composed maps into ONE
// window summary; the summaries are all-gathered once per window and applied in rank order.  window = 1 is
// step_local + step_apply.
int rp_mccfr_window_local(rp_mccfr* h, void* window_dev, int first) {
    if (!h || !window_dev) return rp::fail(RP_ERR_INVALID, "rp_mccfr_window_local: NULL argument");
    int rc = set_device(h);
    if (rc) return rc;
    if ((rc = composed_supported(h))) return rc;
    const StepParams p = make_params(h);
    if ((rc = launch_batch_summary(h, p, h->d_summary))) return rc;
    const uint32_t ncell = h->tbl.n_infos * h->tbl.max_actions;
    hipLaunchKernelGGL(k_accumulate, dim3((ncell + 255) / 256), dim3(256), 0, h->stream, h->g, reinterpret_cast<unsigned char*>(window_dev),
                       reinterpret_cast<const unsigned char*>(h->d_summary), first ? 1u : 0u);
    HIP_TRY(hipGetLastError());
    h->epoch += 1;
    return RP_OK;
}

int rp_mccfr_window_apply(rp_mccfr* h, const void* gathered_dev, uint32_t world) {
    if (!h || !gathered_dev || world == 0) return rp::fail(RP_ERR_INVALID, "rp_mccfr_window_apply: bad argument");
    int rc = set_device(h);
    if (rc) return rc;
    const uint32_t ncell = h->tbl.n_infos * h->tbl.max_actions;
    hipLaunchKernelGGL(k_fold, dim3((ncell + 255) / 256), dim3(256), 0, h->stream, h->g, h->t,
                       reinterpret_cast<const unsigned char*>(gathered_dev), summary_bytes_of(h), world);
    HIP_TRY(hipGetLastError());
    h->tables_version += 1;
    return RP_OK;
}

int rp_mccfr_step_comm(rp_mccfr* h, rp_comm* c, uint32_t steps, uint32_t window) {
    if (!h || !c) return rp::fail(RP_ERR_INVALID, "rp_mccfr_step_comm: NULL argument");
    if (window == 0) window = 1;
    int rc = set_device(h);
    if (rc) return rc;
    if ((rc = composed_supported(h))) return rc;
    const uint32_t world = (uint32_t)rp::comm_world(c);
    h->rank = (uint32_t)rp::comm_rank(c);
    h->world = world;
    const size_t nb = summary_bytes_of(h);
    if (!h->d_window || h->window_world != world) {  // [mine][all ranks]
        if (h->d_window) (void)hipFree(h->d_window);
        h->d_window = nullptr;
        HIP_TRY(hipMalloc(&h->d_window, nb * (world + 1)));
        h->window_world = world;
    }
    unsigned char* mine = reinterpret_cast<unsigned char*>(h->d_window);
    unsigned char* all = mine + nb;
    uint32_t pending = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        if ((rc = rp_mccfr_window_local(h, mine, pending == 0))) return rc;
        pending += 1;
        if (pending == window || s + 1 == steps) {
            if ((rc = rp::comm_all_gather(c, mine, all, nb, h->stream))) return rc;
            if ((rc = rp_mccfr_window_apply(h, all, world))) return rc;
            pending = 0;
        }
    }
    return RP_OK;
}

int rp_mccfr_profile(rp_mccfr* h, int enable) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_mccfr_profile: NULL handle");
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    clock_drain(h->clk_traverse);
    clock_drain(h->clk_compact);
    clock_drain(h->clk_update);
    h->profiling = enable != 0;
    h->clk_traverse.total_ms = h->clk_compact.total_ms = h->clk_update.total_ms = 0.0;
    h->clk_traverse.launches = h->clk_compact.launches = h->clk_update.launches = 0;
    return RP_OK;
}

int rp_game_skeleton(const rp_game_table* game, int* out) {
    if (!game || !out) return rp::fail(RP_ERR_INVALID, "rp_game_skeleton: NULL argument");
    int rc = rp_game_table_check(game);
    if (rc) return rc;
    const std::vector<uint32_t> children(game->children, game->children + game->n_children);
    *out = skel_matches<KuhnSkel>(game, children) ? 1 : (skel_matches<LeducSkel>(game, children) ? 2 : 0);
    return RP_OK;
}

int rp_mccfr_traversal_variant(rp_mccfr* h, int* out) {
    if (!h || !out) return rp::fail(RP_ERR_INVALID, "rp_mccfr_traversal_variant: NULL argument");
    *out = h->static_skel ? 2 : (h->use_lds_traverse ? 1 : 0);
    return RP_OK;
}

int rp_mccfr_kernel_time(rp_mccfr* h, const char* name, double* total_ms, uint64_t* launches) {
    if (!h || !name) return rp::fail(RP_ERR_INVALID, "rp_mccfr_kernel_time: NULL argument");
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    clock_drain(h->clk_traverse);
    clock_drain(h->clk_compact);
    clock_drain(h->clk_update);
    const KernelClock* c = nullptr;
    if (std::string(name) == "traverse") c = &h->clk_traverse;
    else if (std::string(name) == "compact") c = &h->clk_compact;
    else if (std::string(name) == "update") c = &h->clk_update;
    else return rp::fail(RP_ERR_INVALID, "rp_mccfr_kernel_time: unknown kernel '%s'", name);
    if (total_ms) *total_ms = c->total_ms;
    if (launches) *launches = c->launches;
    return RP_OK;
}

// ---- Solver::exploitability (solver.rs:327-337) on the host: validation, not the hot path -------------
// Full-tree best response exactly as CfrNash does it (nash.rs:31-38,103-194): per-infoset argmax of
// sum_{n in span} external_reach * average-strategy value, then evaluation with those choices.
namespace {
struct XNode {
    uint32_t state;
    int32_t parent, edge;
    int32_t kids[RP_MAX_ACTIONS];
};
struct XCtx {
    const rp_mccfr* h;
    const float* weight;
    std::vector<XNode> nodes;
    float wt(uint32_t info, uint32_t a) const { return rp_maxf(weight[(size_t)info * h->tbl.max_actions + a], RP_EPSILON); }
    float averaged(uint32_t info, uint32_t a) const {
        float sum = 0.0f;
        for (uint32_t k = 0; k < h->info_actions[info]; ++k) sum += wt(info, k);
        return wt(info, a) / sum;
    }
    float value(int32_t n, uint32_t hero, const int32_t* br) const {
        const XNode& nd = nodes[n];
        const rp_state& st = h->states[nd.state];
        if (st.n_children == 0) return h->payoffs[(size_t)st.offset * h->tbl.n_players + hero];
        if (st.turn == RP_TURN_CHANCE) {
            float s = 0.0f;
            for (uint32_t k = 0; k < st.n_children; ++k) s += value(nd.kids[k], hero, br);
            return s / (float)st.n_children;
        }
        if (st.turn == hero && br) return value(nd.kids[br[st.info]], hero, br);
        float s = 0.0f;
        for (uint32_t k = 0; k < st.n_children; ++k) s += averaged(st.info, k) * value(nd.kids[k], hero, br);
        return s;
    }
    float reach(int32_t n, uint32_t hero) const {
        float p = 1.0f;
        const XNode* nd = &nodes[n];
        while (nd->parent >= 0) {
            const XNode& par = nodes[nd->parent];
            const rp_state& ps = h->states[par.state];
            if (ps.turn != RP_TURN_CHANCE && ps.turn != hero) p = p * averaged(ps.info, (uint32_t)nd->edge);
            nd = &par;
        }
        return p;
    }
};
}  // namespace

int rp_mccfr_exploitability(rp_mccfr* h, float* out) {
    if (!h || !out) return rp::fail(RP_ERR_INVALID, "rp_mccfr_exploitability: NULL argument");
    int rc = set_device(h);
    if (rc) return rc;
    std::vector<float> r, w, p;
    std::vector<uint32_t> v;
    if ((rc = fetch_tables(h, r, w, p, v))) return rc;
    XCtx x{h, w.data(), {}};
    // VanillaSampling tree in TreeBuilder's pop-last order (builder.rs:141-161) so spans ascend like the reference's
    struct Leaf { uint32_t state; int32_t parent, edge; };
    std::vector<Leaf> todo;
    auto push = [&](uint32_t state, int32_t parent, int32_t edge) {
        XNode nd{state, parent, edge, {}};
        for (auto& k : nd.kids) k = -1;
        x.nodes.push_back(nd);
        const int32_t me = (int32_t)x.nodes.size() - 1;
        if (parent >= 0) x.nodes[parent].kids[edge] = me;
        const rp_state& st = h->states[state];
        for (uint32_t k = 0; k < st.n_children; ++k) todo.push_back(Leaf{h->children[st.offset + k], me, (int32_t)k});
    };
    push(h->tbl.exploit_root, -1, -1);
    while (!todo.empty()) {
        Leaf lf = todo.back();
        todo.pop_back();
        push(lf.state, lf.parent, lf.edge);
    }
    std::vector<int32_t> br(h->tbl.n_infos, 0);
    std::vector<float> cfv((size_t)h->tbl.n_infos * RP_MAX_ACTIONS);
    float total = 0.0f;
    for (uint32_t hero = 0; hero < h->tbl.n_players; ++hero) {
        std::fill(cfv.begin(), cfv.end(), 0.0f);
        for (size_t i = 0; i < x.nodes.size(); ++i) {
            const XNode& nd = x.nodes[i];
            const rp_state& st = h->states[nd.state];
            if (st.turn != hero || st.n_children == 0) continue;
            for (uint32_t a = 0; a < st.n_children; ++a) {
                const int32_t c = nd.kids[a];
                cfv[(size_t)st.info * RP_MAX_ACTIONS + a] += x.reach(c, hero) * x.value(c, hero, nullptr);
            }
        }
        for (uint32_t info = 0; info < h->tbl.n_infos; ++info) {
            br[info] = 0;
            if (h->info_player[info] != hero) continue;
            float best = cfv[(size_t)info * RP_MAX_ACTIONS];
            for (uint32_t a = 1; a < h->info_actions[info]; ++a)
                if (cfv[(size_t)info * RP_MAX_ACTIONS + a] >= best) {  // max_by keeps the last maximum (nash.rs:184-193)
                    best = cfv[(size_t)info * RP_MAX_ACTIONS + a];
                    br[info] = (int32_t)a;
                }
        }
        total += x.value(0, hero, br.data());
    }
    *out = total / (float)h->tbl.n_players;
    return RP_OK;
}

}  // extern "C"

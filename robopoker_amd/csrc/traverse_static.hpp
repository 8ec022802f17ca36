// traverse_static.hpp — Solver::batch (crates/mccfr/src/solver/solver.rs:225-250) for games whose ACTION tree does not
// depend on the cards: the traversal instantiated per game, as the reference's Solver<G> is monomorphised per CfrGame.
//
// k_traverse_lds walks each sampled tree with a per-lane DFS: 64 lanes = 64 different node sequences, half of the lanes
// idle on average (profiles/r01_mccfr_sq_counters.txt) and every step of a lane waits for the one before it.  In Kuhn and
// Leduc the sampled tree of ANY deal is a sub-tree of one fixed skeleton (the public betting tree with its chance nodes
// collapsed to the sampled outcome): what varies per tree is which skeleton nodes are live (the opponent's sampled
// actions), their infoset ids and their payoffs.  So the skeleton is a compile-time constant, the node loop is unrolled
// over it, every lane executes the same instruction stream (no divergence, no LDS, no stack), all per-node state sits in
// registers with static indices, and the loads of independent sub-trees overlap.
//
// Same arithmetic as k_traverse_lds, operation for operation (TreeBuilder::build builder.rs:74-87,141-161 in pop-last
// order = pre-order with children in DESCENDING edge order; CfrFlow::dfs / recursed_value / ancestor_reach
// flow.rs:64-87,166-216): the Decisions are bit-identical (tests/test_gpu_mccfr.py::test_static_skeleton_equals_generic).
// Used when the game's tables match the skeleton node for node (skel_matches, checked once at rp_mccfr_create) and
// max_actions == 2; everything else takes k_traverse_lds.  PRUNED = false: external sampling (the walker expands every action).
// PRUNED = true: PrunableSampling / PluribusSampling (sample/pruning.rs:44-66, pluribus.rs:72-101) — a walker node's surviving
// edges come from the per-infoset keep masks (k_prepare_infos), the explore draw of (epoch, infoset, tree) and, for Pluribus,
// the edges whose child is a terminal node of the SKELETON (a compile-time fact); a pruned child is a dead skeleton node,
// exactly like an opponent action that was not sampled, and its edge takes no part in the node's value, regret or mask.
#pragma once

#include <type_traits>

namespace rp {

enum : int { SK_CHANCE = 0, SK_P0 = 1, SK_P1 = 2, SK_TERMINAL = 3 };
#define SK_MAXN 48

struct Skeleton {
    int n = 0;
    int kind[SK_MAXN] = {};
    int parent[SK_MAXN] = {};  // -1 at the root
    int edge[SK_MAXN] = {};    // the action leading here; ignored below a chance node (the sampled outcome)
    int end[SK_MAXN] = {};     // last node of the sub-tree (pre-order: the sub-tree of s is [s, end[s]])
    constexpr int add(int k, int p, int e) {
        const int i = n++;
        kind[i] = k;
        parent[i] = p;
        edge[i] = e;
        end[i] = i;
        return i;
    }
    constexpr void close(int s) { end[s] = n - 1; }
};

// ---- Kuhn (crates/kuhn/src/game.rs:7-15,131-151): Start -> Dealt -> Open{Check{T, CheckBet{T,T}}, Bet{T,T}} ----
struct KuhnSkel {
    static constexpr Skeleton make() {
        Skeleton b;
        const int start = b.add(SK_CHANCE, -1, 0);
        const int dealt = b.add(SK_CHANCE, start, 0);
        const int open = b.add(SK_P0, dealt, 0);
        {  // children in descending edge order (the last child pushed is the first popped)
            const int bet = b.add(SK_P1, open, 1);
            b.add(SK_TERMINAL, bet, 1);
            b.add(SK_TERMINAL, bet, 0);
            b.close(bet);
            const int check = b.add(SK_P1, open, 0);
            const int cb = b.add(SK_P0, check, 1);
            b.add(SK_TERMINAL, cb, 1);
            b.add(SK_TERMINAL, cb, 0);
            b.close(cb);
            b.add(SK_TERMINAL, check, 0);
            b.close(check);
        }
        b.close(open);
        b.close(dealt);
        b.close(start);
        return b;
    }
};

// ---- Leduc (crates/leduc/src/game.rs:7-12,152-223): two betting rounds of the same shape, a Deal between them ----
struct LeducSkel {
    enum { OPEN, CHECKED, RAISED, CHECKRAISED };
    // one betting round below `parent`; `next` = 1: a closed round continues with the Deal and round 2, 0: it ends
    static constexpr void round(Skeleton& b, int parent, int edge, int spot, int next) {
        const int actor = (spot == OPEN || spot == CHECKRAISED) ? SK_P0 : SK_P1;
        const int s = b.add(actor, parent, edge);
        auto closes = [&](int e) {  // the child that closes the round: Deal -> round 2, or the showdown
            if (next) {
                const int deal = b.add(SK_CHANCE, s, e);
                round(b, deal, 0, OPEN, 0);
                b.close(deal);
            } else {
                b.add(SK_TERMINAL, s, e);
            }
        };
        switch (spot) {
            case OPEN:  // [Check, Raise]
                round(b, s, 1, RAISED, next);
                round(b, s, 0, CHECKED, next);
                break;
            case CHECKED:  // [Check -> closes, Raise]
                round(b, s, 1, CHECKRAISED, next);
                closes(0);
                break;
            default:  // RAISED / CHECKRAISED: [Fold, Call -> closes]
                closes(1);
                b.add(SK_TERMINAL, s, 0);
                break;
        }
        b.close(s);
    }
    static constexpr Skeleton make() {
        Skeleton b;
        const int start = b.add(SK_CHANCE, -1, 0);
        const int dealt = b.add(SK_CHANCE, start, 0);
        round(b, dealt, 0, OPEN, 1);
        b.close(dealt);
        b.close(start);
        return b;
    }
};

template <class G>
struct SkelOf {
    static constexpr Skeleton S = G::make();
};

// the skeleton child of node s along edge e is a terminal node?  (walker nodes: the Pluribus exemption, pluribus.rs:96)
template <class G>
constexpr bool sk_child_terminal(int s, int e) {
    for (int c = 0; c < SkelOf<G>::S.n; ++c)
        if (SkelOf<G>::S.parent[c] == s && SkelOf<G>::S.edge[c] == e) return SkelOf<G>::S.kind[c] == SK_TERMINAL;
    return false;
}

static_assert(SkelOf<KuhnSkel>::S.n == 11, "Kuhn: two deals, four decision nodes, five terminals");
static_assert(SkelOf<LeducSkel>::S.n == 38, "Leduc: two deals, 4 + 3 x 4 decision nodes, three board draws, 2 + 3 x 5 terminals");
static_assert(SkelOf<LeducSkel>::S.end[0] == 37 && SkelOf<LeducSkel>::S.kind[2] == SK_P0, "pre-order, the first decision is P0's");

// compile-time loops: f(std::integral_constant<int, I>) for I = LO .. HI-1, ascending / descending
template <int I, int HI, class F>
__device__ __forceinline__ void sk_for(F&& f) {
    if constexpr (I < HI) {
        f(std::integral_constant<int, I>{});
        sk_for<I + 1, HI>(f);
    }
}
template <int I, int LO, class F>
__device__ __forceinline__ void sk_for_down(F&& f) {  // I = HI-1 down to LO
    if constexpr (I >= LO) {
        f(std::integral_constant<int, I>{});
        sk_for_down<I - 1, LO>(f);
    }
}

// The traversal of one tree by one lane.  W = the walker (epoch % 2).  `present` = this lane has a tree (the ragged last
// workgroup).  on_built(info_of, live_of) runs once the tree is sampled — before any value is computed — with two callables
// over the skeleton's node index; on_decision(j, info, regret0, regret1, sigma0, sigma1, payoff) once per LIVE walker node,
// ascending node index (= the order of Tree::partition's spans).  Returns the number of nodes of the sampled tree.
// on_decision's `mask`: the expanded edges (3 under external sampling).
// REF: the draws come from the reference's own chain (rp_rng_kind RP_RNG_REFERENCE, mccfr_kernels.hpp d_draw_*).
template <class G, int W, bool PRUNED, bool REF, class OnBuilt, class OnDecision>
__device__ __forceinline__ uint32_t static_traverse(const DevGame& g, const DevInfoTab& it, const StepParams& p, uint64_t tree_id,
                                                    bool present, OnBuilt&& on_built, OnDecision&& on_decision) {
    using SK = SkelOf<G>;
    constexpr int N = SK::S.n;
    constexpr int K_WALKER = W == 0 ? SK_P0 : SK_P1;
    constexpr int K_OPP = W == 0 ? SK_P1 : SK_P0;

    // ---- TreeBuilder::build over the skeleton ----------------------------------------------------------------------
    uint32_t rx[N], ry[N], rz[N], rw[N];  // the node's record (DevGame::kids): turn | n_children << 8, info / payoff0, offset / payoff1, state
    uint32_t pick[N];                     // chance: the sampled outcome; opponent: the sampled action; walker (PRUNED): the surviving edges
    bool live[N];
    float sg0[N], sg1[N], q0[N], q1[N];   // (sigma, q) of a player node's two edges
    // every draw of this tree: rp_node_hash(seed, epoch, tree, key) with the (seed, epoch, tree) part hashed once
    const uint64_t th = rp_node_hash_tree(rp_node_hash_step(p.seed, p.epoch), tree_id);
    rx[0] = g.root_rec.x;
    ry[0] = g.root_rec.y;
    rz[0] = g.root_rec.z;
    rw[0] = g.root_rec.w;
    live[0] = present;
    sk_for<0, N>([&](auto I) __attribute__((always_inline)) {
        constexpr int s = I;
        constexpr int par = SK::S.parent[s];
        if constexpr (par >= 0) {
            uint4 r;
            if (g.flat) {  // wave-uniform: the record by the chance outcomes on the path (DevGame::flat)
                uint32_t idx = 0;
                sk_for<0, s>([&](auto C) __attribute__((always_inline)) {
                    constexpr int c = C;
                    if constexpr (SK::S.kind[c] == SK_CHANCE && s <= SK::S.end[c]) idx = idx * g.flat_fan[c] + pick[c];
                });
                r = g.flat[g.flat_base[s] + idx];
            } else {
                uint32_t k = (uint32_t)SK::S.edge[s];
                if constexpr (SK::S.kind[par] == SK_CHANCE) k = pick[par];
                r = g.kids[rz[par] + k];
            }
            rx[s] = r.x;
            ry[s] = r.y;
            rz[s] = r.z;
            rw[s] = r.w;
            if constexpr (SK::S.kind[par] == K_OPP) live[s] = live[par] && pick[par] == (uint32_t)SK::S.edge[s];
            else if constexpr (PRUNED && SK::S.kind[par] == K_WALKER) live[s] = live[par] && ((pick[par] >> SK::S.edge[s]) & 1u);
            else live[s] = live[par];
        }
        if constexpr (SK::S.kind[s] == SK_CHANCE) {  // SamplingScheme::sample at a chance node: uniform (external.rs:41-64)
            const uint32_t nout = (rx[s] >> 8) & 0xffu, ci = ry[s];  // a chance record's y = chance_info; 0: the root deal (thread RNG in the reference)
            if (REF && ci) pick[s] = rp_ref_draw_range(rp_ref_seed_finish(&p.ref_chance[ci - 1u], tree_id), nout);
            else pick[s] = rp_pick_uniform(rp_node_hash_key(th, 0x80000000ull | rw[s]), nout);
        } else if constexpr (SK::S.kind[s] == SK_P0 || SK::S.kind[s] == SK_P1) {
            const uint32_t info = ry[s];
            const float4 f = *reinterpret_cast<const float4*>(&it.sq[info * 2u]);
            sg0[s] = f.x;
            q0[s] = f.y;
            sg1[s] = f.z;
            q1[s] = f.w;
            if constexpr (SK::S.kind[s] == K_OPP) {  // WeightedIndex over max(q, EPSILON): two actions = one threshold
                const float x = REF ? rp_ref_draw_weight(rp_ref_seed_finish(&p.ref_info[info], tree_id), it.total[info])
                                    : rp_u01(rp_node_hash_key(th, info)) * it.total[info];
                pick[s] = it.cum[info * 2u] <= x ? 1u : 0u;
            } else if constexpr (PRUNED) {  // SamplingScheme::sample at a walker node (d_sample_mask_tab, the same draw and masks)
                uint32_t mask = 3u;
                bool prune = true;
                if (p.S == RP_SAMPLING_PLURIBUS)
                    prune = p.epoch >= p.prune_warmup &&
                            !((REF ? rp_ref_draw_f32(rp_ref_seed_finish(&p.ref_info[info], tree_id)) : rp_u01(rp_node_hash_key(th, info))) <
                              p.prune_explore);
                if (prune) {
                    mask = it.keep[info] & 3u;
                    if (p.S == RP_SAMPLING_PLURIBUS)
                        mask |= (sk_child_terminal<G>(s, 0) ? 1u : 0u) | (sk_child_terminal<G>(s, 1) ? 2u : 0u);
                    mask = mask ? mask : 3u;  // pruning.rs:64, pluribus.rs:99
                }
                pick[s] = mask;
            }
        }
    });

    // ---- Tree::partition + CfrFlow::dfs per walker decision node (tree.rs:88-98, flow.rs:64-87) ---------------------
    uint32_t nn = 0;
    sk_for<0, N>([&](auto I) __attribute__((always_inline)) { nn += live[decltype(I)::value] ? 1u : 0u; });
    on_built([&](auto I) __attribute__((always_inline)) { return ry[decltype(I)::value]; },
             [&](auto I) __attribute__((always_inline)) { return live[decltype(I)::value]; });
    sk_for<0, N>([&](auto J) __attribute__((always_inline)) {
        constexpr int j = J;
        if constexpr (SK::S.kind[j] == K_WALKER) {
            constexpr int E = SK::S.end[j];
            float rel[N], smp[N], acc[N], tv[2] = {0.0f, 0.0f};
            // top-down over the sub-tree: reach products from j's children (flow.rs:195-212); a leaf hands its value to its
            // parent right here, an internal node starts its sum at 0 (the two-children argument of k_traverse_lds)
            sk_for<j + 1, E + 1>([&](auto Nn) __attribute__((always_inline)) {
                constexpr int n = Nn;
                constexpr int par = SK::S.parent[n];
                constexpr int e = SK::S.edge[n];
                float r = 1.0f, sm = 1.0f;
                if constexpr (par != j) {
                    r = rel[par];
                    sm = smp[par];
                    if constexpr (SK::S.kind[par] == K_WALKER) r = r * (e ? sg1[par] : sg0[par]);
                    if constexpr (SK::S.kind[par] == K_OPP) {
                        r = r * (e ? sg1[par] : sg0[par]);
                        sm = sm * (e ? q1[par] : q0[par]);
                    }
                }
                if constexpr (SK::S.kind[n] == SK_TERMINAL) {
                    const float v = r / sm * rp_u2f(W == 0 ? ry[n] : rz[n]);
                    if constexpr (par == j) tv[e] = v;
                    else acc[par] = live[n] ? acc[par] + v : acc[par];
                } else {
                    rel[n] = r;
                    smp[n] = sm;
                    acc[n] = 0.0f;
                }
            });
            // bottom-up: the internal nodes' sums, descending node index (node.rs:103-107)
            sk_for_down<E, j + 1>([&](auto Nn) __attribute__((always_inline)) {
                constexpr int n = Nn;
                constexpr int par = SK::S.parent[n];
                if constexpr (SK::S.kind[n] != SK_TERMINAL) {
                    if constexpr (par == j) tv[SK::S.edge[n]] = acc[n];
                    else acc[par] = live[n] ? acc[par] + acc[n] : acc[par];
                }
            });
            // ancestor_reach (flow.rs:166-174): the opponent's edges on the way up
            float cf = 1.0f, sm_ = 1.0f;
            sk_for<0, N>([&](auto Q) __attribute__((always_inline)) {
                // ancestors of j in the order j, parent(j), ...: node index DESCENDS along the chain, so visit candidates from
                // j downwards and keep those on the chain
                constexpr int n = j - decltype(Q)::value;
                if constexpr (n > 0) {
                    // is n on the parent chain of j (n == j or an ancestor)?  pre-order: n <= j <= end[n]
                    if constexpr (n <= j && j <= SK::S.end[n]) {
                        constexpr int par = SK::S.parent[n];
                        if constexpr (SK::S.kind[par] == K_OPP) {
                            cf = cf * (SK::S.edge[n] ? sg1[par] : sg0[par]);
                            sm_ = sm_ * (SK::S.edge[n] ? q1[par] : q0[par]);
                        }
                    }
                }
            });
            const float reach = cf / sm_;
            const float u0 = reach * tv[0], u1 = reach * tv[1];
            if constexpr (!PRUNED) {
                float ev = 0.0f;
                ev += sg0[j] * u0;
                ev += sg1[j] * u1;
                const float payoff = 0.0f + ev;
                const float g0 = 0.0f + (u0 - ev), g1 = 0.0f + (u1 - ev);
                if (live[j]) on_decision(J, ry[j], g0, g1, sg0[j], sg1[j], payoff, 3u);
            } else {  // only the expanded edges enter the node's value and receive a regret (k_traverse_lds, the same order)
                const uint32_t mask = pick[j];
                float ev = 0.0f;
                if (mask & 1u) ev += sg0[j] * u0;
                if (mask & 2u) ev += sg1[j] * u1;
                const float payoff = 0.0f + ev;
                const float g0 = (mask & 1u) ? 0.0f + (u0 - ev) : 0.0f, g1 = (mask & 2u) ? 0.0f + (u1 - ev) : 0.0f;
                if (live[j]) on_decision(J, ry[j], g0, g1, sg0[j], sg1[j], payoff, mask);
            }
        }
    });
    return nn;
}

// Decisions to HBM (DevDecisions), for the ordered update, the sorted large-game path and the debugging views.
// One lane per tree, 256 trees per workgroup, no LDS.
template <class G, int W, bool PRUNED, bool REF>
__global__ __launch_bounds__(256) void k_traverse_static(DevGame g, DevInfoTab it, DevDecisions dc, StepParams p) {
    const uint32_t lane = blockIdx.x * 256u + threadIdx.x;
    if (lane >= p.batch) return;
    const size_t D = dc.stride;
    uint32_t ndec = 0;
    const uint32_t nn = static_traverse<G, W, PRUNED, REF>(
        g, it, p, p.tree_base + lane, true, [](auto, auto) __attribute__((always_inline)) {},
        [&](auto, uint32_t info, float g0, float g1, float s0, float s1, float payoff, uint32_t mask) __attribute__((always_inline)) {
            const uint32_t slot = ndec;
            dc.regret[((size_t)slot * 2u + 0u) * D + lane] = g0;
            dc.regret[((size_t)slot * 2u + 1u) * D + lane] = g1;
            dc.policy[((size_t)slot * 2u + 0u) * D + lane] = s0;
            dc.policy[((size_t)slot * 2u + 1u) * D + lane] = s1;
            dc.info[(size_t)slot * D + lane] = info;
            dc.mask[(size_t)slot * D + lane] = mask;
            dc.payoff[(size_t)slot * D + lane] = payoff;
            if (dc.slotmap) dc.slotmap[(size_t)info * D + lane] = (uint8_t)(slot + 1u);
            ndec += 1u;
        });
    dc.ndec[lane] = (uint8_t)ndec;
    count_metrics(p, nn, ndec, 0u);
}

// Composed update, small games: the traversal AND the block maps of its 256-tree chunk in one kernel — the Decisions never
// reach HBM.  What k_chunk_maps does with the Decisions it reads back (per-infoset, tree-ordered lists by LDS bitmap +
// prefix popcount; one sequential composition per (infoset, cell) from the identity: include/rp_mi355x.h "Composed
// update") happens here on the values as they are produced: the lists' places are known once the trees are sampled
// (on_built), each Decisions drops its five values (two regret deltas, two weight deltas, the payoff) at its place in LDS,
// and the chains of ALL cells run side by side, one thread per (infoset, cell).  Same lists, same order, same operations
// as k_chunk_maps: bmaps / bpsum / bcnt are bit-identical (tests/test_gpu_mccfr.py).
// A chain is sequential and as long as its list (a root infoset meets a third of the chunk's trees, a river infoset a few):
// the (infoset, cell) tasks are handed out in descending order of list length (a counting sort by log2 class), so the 64
// chains of a wave have similar lengths instead of every wave waiting for one long chain.
// LDS (dynamic): bits u32[NI][8] | lcount u32[NI] | lbase u32[NI] | pre u16[NI][8] | order u16[NI + (NI & 1)] | vals f32[5][maxdec * 256]
//                | (PRUNED) lmask u32[maxdec * 256]: the expanded edges of every list entry — a regret cell skips the entries
//                  whose edge was pruned (no touch at all: a touch would apply the discount), as k_chunk_maps<true> does
template <class G, int W, bool PRUNED, bool REF>
__global__ __launch_bounds__(256, 4) void k_traverse_maps_static(DevGame g, DevInfoTab it, StepParams p, Map* bmaps, float* bpsum,
                                                              uint32_t* bcnt, uint32_t nblk_max, uint32_t maxdec, uint32_t lpad) {
    extern __shared__ __attribute__((aligned(16))) uint32_t tm_lds[];
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t cls_n[16], cls_at[16];
    const uint32_t NI = g.n_infos, chunk = blockIdx.x, lt = threadIdx.x;
    uint32_t* bits = tm_lds;
    uint32_t* lcount = bits + NI * 8u;
    uint32_t* lbase = lcount + NI;
    uint16_t* pre = reinterpret_cast<uint16_t*>(lbase + NI);
    uint16_t* order = pre + NI * 8u;
    float* vals = reinterpret_cast<float*>(order + NI + (NI & 1u));
    // places per cell.  The five chains of an infoset read vals[c L + base + e] in the same instruction: with L a multiple of 32 they
    // share a bank (a 5-way conflict at every step); lpad = 7 words moves the cells apart (measured round 4, profiles/r04_optin_ab.json:
    // pads 0 / 3 / 7 / 13 within 1 % of each other — the conflicts were not the limiter)
    const uint32_t L = maxdec * 256u + lpad;
    uint32_t* lmask = reinterpret_cast<uint32_t*>(vals + 5u * L);
    for (uint32_t e = lt; e < NI * 8u; e += 256u) bits[e] = 0;
    if (lt < 16u) cls_n[lt] = 0;
    __syncthreads();
    const uint32_t lane = chunk * 256u + lt;
    const float tf = (float)p.epoch;
    uint32_t ndec = 0;
    const uint32_t nn = static_traverse<G, W, PRUNED, REF>(
        g, it, p, p.tree_base + lane, lane < p.batch,
        [&](auto info_of, auto live_of) __attribute__((always_inline)) {
            sk_for<0, SkelOf<G>::S.n>([&](auto J) __attribute__((always_inline)) {
                if constexpr (SkelOf<G>::S.kind[decltype(J)::value] == (W == 0 ? SK_P0 : SK_P1)) {
                    if (live_of(J)) atomicOr(&bits[info_of(J) * 8u + (lt >> 5)], 1u << (lt & 31u));
                }
            });
            __syncthreads();
            for (uint32_t info = lt; info < NI; info += 256u) {
                uint32_t run = 0;
                for (uint32_t w = 0; w < 8u; ++w) {
                    pre[info * 8u + w] = (uint16_t)run;
                    run += __popc(bits[info * 8u + w]);
                }
                lcount[info] = run;
                atomicAdd(&cls_n[15u - (run ? 32u - (uint32_t)__builtin_clz(run) : 0u)], 1u);  // class 15 - bit length: long lists first
            }
            __syncthreads();
            lds_exscan(lcount, lbase, NI, wave_tot);
            if (lt == 0) {
                uint32_t at = 0;
                for (uint32_t k = 0; k < 16u; ++k) {
                    cls_at[k] = at;
                    at += cls_n[k];
                }
            }
            __syncthreads();
            for (uint32_t info = lt; info < NI; info += 256u) {
                const uint32_t run = lcount[info];
                order[atomicAdd(&cls_at[15u - (run ? 32u - (uint32_t)__builtin_clz(run) : 0u)], 1u)] = (uint16_t)info;
            }
        },
        [&](auto, uint32_t info, float g0, float g1, float s0, float s1, float payoff, uint32_t mask) __attribute__((always_inline)) {
            const uint32_t pos = lbase[info] + pre[info * 8u + (lt >> 5)] + __popc(bits[info * 8u + (lt >> 5)] & ((1u << (lt & 31u)) - 1u));
            if constexpr (PRUNED) lmask[pos] = mask;
            vals[pos] = g0;
            vals[L + pos] = g1;
            vals[2u * L + pos] = p.W == RP_WEIGHT_LINEAR ? s0 * tf : (p.W == RP_WEIGHT_QUADRATIC ? s0 * tf * tf : s0);
            vals[3u * L + pos] = p.W == RP_WEIGHT_LINEAR ? s1 * tf : (p.W == RP_WEIGHT_QUADRATIC ? s1 * tf * tf : s1);
            vals[4u * L + pos] = payoff;
            ndec += 1u;
        });
    __syncthreads();
    // the chains: task = (cell c, infoset); cells 0,1 regret, 2,3 weight, 4 the payoff sum.  The payoff sums are handed out after all
    // the map chains, so that no wavefront mixes the two loops (measured round 4: 0.582 -> 0.562 ms per launch against task % 5)
    const float NEG_INF = rp_u2f(0xff800000u);
    for (uint32_t task = lt; task < 5u * NI; task += 256u) {
        const uint32_t c = task < 4u * NI ? task & 3u : 4u;
        const uint32_t info = order[task < 4u * NI ? task >> 2 : task - 4u * NI];
        if (g.info_player[info] != p.walker) continue;
        const uint32_t n = lcount[info], base = lbase[info];
        const size_t slot_out = (size_t)info * nblk_max + chunk;
        const float* v = vals + (size_t)c * L + base;
        if (c == 4u) {  // payoff sum of the block, left fold from 0.0f
            float sum = 0.0f;
            for (uint32_t e = 0; e < n; ++e) sum += v[e];
            bpsum[slot_out] = sum;
            bcnt[slot_out] = n;
            continue;
        }
        const bool isreg = c < 2u;
        const float fl = isreg ? regret_floor_of(p.R, p.regret_min) : RP_EPSILON;
        const float d = isreg ? (p.R == RP_REGRET_LINEAR ? tf / (tf + 1.0f) : 1.0f) : (p.W == RP_WEIGHT_EXPONENTIAL ? 0.9999f : 1.0f);
        float ma = 1.0f, mb = 0.0f, mm = NEG_INF;
        uint32_t cnt = 0;
        for (uint32_t e = 0; e < n; ++e) {
            const float delta = v[e];
            const bool skip = PRUNED && isreg && !((lmask[base + e] >> c) & 1u);
            // first touch of the block: (d, delta, floor); then a <- a d, b <- b d + delta, m <- max(m d + delta, floor)
            const float na = cnt ? ma * d : d;
            const float nb = cnt ? mb * d + delta : delta;
            const float nm = cnt ? rp_maxf(mm * d + delta, fl) : fl;
            ma = skip ? ma : na;
            mb = skip ? mb : nb;
            mm = skip ? mm : nm;
            cnt += skip ? 0u : 1u;
        }
        bmaps[slot_out * 4u + c] = Map{ma, mb, mm, cnt};
    }
    count_metrics(p, nn, ndec, 0u);
}

}  // namespace rp

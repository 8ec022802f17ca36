// nlmc_common.hpp — what the NLHE traversal kernels (nlmc_level.hpp: level-synchronous for large batches, one tree per workgroup for
// small ones) share: the NlheInfo -> row table, the draw / bucket / choices helpers, the per-step parameters.
//
// Reference: nlhe/src/encoder.rs:30-68 (NlheEncoder::info), nlhe/src/info.rs:145-160 (the key), kicker/src/edge.rs:61-72 +
// bias.rs:47-70 (default regrets), kicker/src/game.rs:513-576,724-753,835-854 (permissions, choices, actionize, snap).
#ifndef RP_NLMC_COMMON_HPP
#define RP_NLMC_COMMON_HPP

#include <hip/hip_runtime.h>

#include "../../include/rp_math.h"
#include "../../include/rp_refrng.h"
#include "../../include/rp_mi355x.h"
#include "nlhe_engine.hpp"
#include "obs.hpp"

namespace rp {

#define NLMC_A 9u
typedef NlGameT<2, 2> G2;  // heads-up: two seats at compile time
enum : uint32_t { NK_TERMINAL = 0, NK_CHANCE = 1, NK_WALKER = 2, NK_OPP = 3 };
enum : uint32_t {
    NERR_NODES = 1u, NERR_STACK = 2u, NERR_WALKERS = 4u, NERR_DECISIONS = 8u, NERR_ILLEGAL = 16u, NERR_TABLE_FULL = 32u,
    NERR_LOOKUP = 64u, NERR_LEVELS = 128u, NERR_LISTS = 256u, NERR_CHAINS = 512u
};

// One infoset = one 32-byte slot (a single HBM sector per probe) + one profile row.  `state`: 0 empty, 1 being written,
// 2 ready; `born` = the launch tag of the kernel that inserted the key (a reader that meets a key born in its OWN launch
// must not trust cached row bytes: see nl_row_of).  state and born share one aligned 64-bit word so they are read together.
struct __align__(32) NlSlot {
    uint64_t past, choices;
    uint32_t present, pad;
    uint32_t state, born;
};
struct NlTable {  // NlheInfo -> row: open addressing, linear probing; slot index = row of the profile
    NlSlot* slots;
    uint32_t mask;
    unsigned int* n_keys;
    float* rows;  // the profile's table: row r = rows + r * 4A: regret[A] weight[A] payoff[A] visits[A]
};

struct NlParams {
    uint64_t seed, epoch;
    uint64_t step_hash;  // rp_node_hash_step(seed, epoch)
    uint32_t batch, walker;
    uint64_t tree_base;  // first tree id of this rank's shard (rank * batch)
    float temperature, smoothing, curiosity;
    int sampling;        // rp_sampling_kind at walker nodes (sample/{external,pruning,pluribus}.rs)
    float prune_threshold, prune_explore;
    uint64_t prune_warmup;
    int encoder;  // 0: hash of the canonical observation, 1: lookup tables
    const uint64_t* tkeys[4];
    const uint8_t* tabs[4];
    uint64_t tn[4];
    // reference-seed mode (rp_nlhe_set_rng RP_RNG_REFERENCE, include/rp_refrng.h): DefaultHasher after t.hash() (flow.rs:290-291);
    // a sampled node continues with NlheInfo's stream — subgame: Path(u64), choices: Path(u64), Abstraction(u16)
    // (nlhe/src/info.rs:41-42, public.rs:19-23, secret.rs:10-11, kicker/src/{path.rs:21-22, abstraction.rs:19-20}) — and the tree id
    uint32_t ref_rng;
    uint64_t ref_v[4];
    uint32_t check_legal;             // evaluate Game::is_allowed on every applied action (RP_NLHE_CHECK_LEGAL=1)
    uint32_t tag;                     // launch tag of the kernel about to run (never 0)
};

// the 2-seat game in five dwords (chips fit a byte: the stack is 200)
struct Packed {
    uint32_t w0, w1, w2, blo, bhi;
};
__device__ __forceinline__ Packed pack_game(const G2& g) {
    Packed p;
    p.w0 = (uint32_t)g.ticker | ((uint32_t)g.pot << 8) | ((uint32_t)g.state[0] << 24) | ((uint32_t)g.state[1] << 26);
    p.w1 = (uint32_t)g.stack[0] | ((uint32_t)g.stake[0] << 8) | ((uint32_t)g.spent[0] << 16);
    p.w2 = (uint32_t)g.stack[1] | ((uint32_t)g.stake[1] << 8) | ((uint32_t)g.spent[1] << 16);
    p.blo = (uint32_t)g.board;
    p.bhi = (uint32_t)(g.board >> 32);
    return p;
}
__device__ __forceinline__ void unpack_game(const Packed& p, G2& g) {
    g.n = 2;
    g.dealer = 0;
    g.ticker = (int)(p.w0 & 0xffu);
    g.pot = (int)((p.w0 >> 8) & 0xffffu);
    g.state[0] = (int)((p.w0 >> 24) & 3u);
    g.state[1] = (int)((p.w0 >> 26) & 3u);
    g.stack[0] = (int)(p.w1 & 0xffu); g.stake[0] = (int)((p.w1 >> 8) & 0xffu); g.spent[0] = (int)((p.w1 >> 16) & 0xffu);
    g.stack[1] = (int)(p.w2 & 0xffu); g.stake[1] = (int)((p.w2 >> 8) & 0xffu); g.spent[1] = (int)((p.w2 >> 16) & 0xffu);
    g.board = (uint64_t)p.blo | ((uint64_t)p.bhi << 32);
}

// kicker/src/edge.rs:61-72 with BiasHyperParams::default (bias.rs:47-70)
__device__ __forceinline__ float nl_default_regret(uint32_t e) {
    return e == NE_FOLD ? 100.0f : (e == NE_SHOVE ? 0.0f : ((e == NE_CHECK || e == NE_CALL) ? 50.0f : 10.0f));
}
// the u64 a sampled node's SmallRng is seeded with, reference-seed mode (flow.rs:285-295)
__device__ __forceinline__ uint64_t nl_ref_seed(const NlParams& p, uint64_t past, uint64_t choices, uint32_t present, uint64_t tree_id) {
    rp_sip s;
    s.v0 = p.ref_v[0]; s.v1 = p.ref_v[1]; s.v2 = p.ref_v[2]; s.v3 = p.ref_v[3];
    s.tail = 0; s.ntail = 0; s.len = 8;
    rp_defaulthasher_write_u64(&s, past);
    rp_defaulthasher_write_u64(&s, choices);
    rp_defaulthasher_write_u16(&s, (uint16_t)present);
    rp_defaulthasher_write_u64(&s, tree_id);
    return rp_defaulthasher_finish(&s);
}
// the two draws of a sampled NLHE node in either rp_rng_kind: Pluribus' coin (walker) and WeightedIndex's x (opponent)
__device__ __forceinline__ float nl_draw_coin(const NlParams& p, uint64_t tree_id, uint64_t khash, uint64_t past, uint64_t choices, uint32_t present) {
    if (p.ref_rng) return rp_ref_draw_f32(nl_ref_seed(p, past, choices, present, tree_id));
    return rp_u01(rp_node_hash_draw(p.step_hash, tree_id, khash));
}
__device__ __forceinline__ float nl_draw_weight(const NlParams& p, uint64_t tree_id, uint64_t khash, uint64_t past, uint64_t choices, uint32_t present,
                                                float total) {
    if (p.ref_rng) return rp_ref_draw_weight(nl_ref_seed(p, past, choices, present, tree_id), total);
    return rp_u01(rp_node_hash_draw(p.step_hash, tree_id, khash)) * total;
}
__device__ __forceinline__ uint64_t nl_key_hash(uint64_t past, uint64_t choices, uint32_t present) {
    return rp_mix64(rp_mix64(past ^ 0x9e3779b97f4a7c15ull) ^ rp_mix64(choices + 0xd1342543de82ef95ull) ^ ((uint64_t)present * 0xaf251af3b0f025b5ull));
}

// a load the compiler may neither hoist out of the probe loop nor fold with an earlier one; an ordinary cached load in the ISA
__device__ __forceinline__ uint64_t nl_peek64(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ uint32_t nl_peek32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }

// Find or insert (encoder.rs:38-68 looks the infoset up; book.rs:93-122: a missing Encounter reads as the default row).
//   * Optimistic read first: a slot that shows state 2 and the probe's own 20 key bytes IS that infoset (keys are written once,
//     before the state is published, and never change) — no atomic, no fence.  Anything else (empty, being written, a
//     different or a torn / stale key) is re-examined through the compare-and-swap path, whose answer is authoritative.
//   * Divergence-safe: the winner of a slot writes and publishes it INSIDE the loop body and leaves through the loop condition,
//     so lanes of one wavefront that wait for the same slot always see the winner run (no early return from the loop).
//   * A key born in the running launch (`born == tag`): the L2 of this lane's XCD may still hold the row's line from before the
//     insertion (rows are 144 B, neighbours share lines) — an acquire fence drops it before the caller reads the row.  Keys from
//     earlier launches need nothing: kernel boundaries already ordered their bytes.
__device__ __forceinline__ uint32_t nl_row_of(const NlTable& t, uint64_t past, uint64_t choices, uint32_t present, uint64_t key_hash, uint32_t nch,
                                              uint32_t tag, uint32_t* err, bool* settled = nullptr) {
    // *settled: the key was found at its home slot by the optimistic read and was not born in this launch — row bytes a caller
    // loaded from the home slot BEFORE the probe (in parallel with it) are then the row's
    uint32_t s = (uint32_t)key_hash & t.mask, probes = 0, row = 0;
    bool done = false, quiet = false;
    while (!done) {
        NlSlot* sl = t.slots + s;
        // the whole slot in two 16-byte loads.  Whatever they return — current, stale or torn — only a full match of state 2 and
        // the 20 key bytes is trusted; everything else goes through the compare-and-swap path below
        const uint4 lo = *reinterpret_cast<const uint4*>(sl), hi = *(reinterpret_cast<const uint4*>(sl) + 1);
        const uint64_t kp = (uint64_t)lo.x | ((uint64_t)lo.y << 32), kc = (uint64_t)lo.z | ((uint64_t)lo.w << 32);
        if (hi.z == 2u && kp == past && kc == choices && hi.x == present) {
            if (hi.w == tag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            else quiet = probes == 0;
            row = s;
            done = true;
        } else {
            const uint32_t st = atomicCAS(&sl->state, 0u, 1u);
            if (st == 0u) {
                sl->past = past;
                sl->choices = choices;
                sl->present = present;
                sl->born = tag;
                float* r = t.rows + (size_t)s * 4u * NLMC_A;
                for (uint32_t a = 0; a < nch; ++a) r[a] = nl_default_regret((uint32_t)(choices >> (5u * a)) & 31u);
                __threadfence();
                atomicExch(&sl->state, 2u);
                atomicAdd(t.n_keys, 1u);
                row = s;
                done = true;
            } else if (st == 2u) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (nl_peek64(&sl->past) == past && nl_peek64(&sl->choices) == choices && nl_peek32(&sl->present) == present) {
                    row = s;
                    done = true;
                } else {
                    s = (s + 1u) & t.mask;
                    if (++probes > t.mask) {
                        *err |= NERR_TABLE_FULL;
                        done = true;
                    }
                }
            }
            // st == 1: another lane (possibly of this wavefront) is writing the slot: look again
        }
    }
    if (settled) *settled = quiet;
    return row;
}
// regret[9] | weight[9] of a row as aligned 16-byte loads (a row is 144 B = 9 x 16): f[0..8] regrets, f[9..17] weights
__device__ __forceinline__ void nl_load_row(const float* rows, uint32_t row, bool weights, float* f) {
    const float4* q = reinterpret_cast<const float4*>(rows + (size_t)row * 4u * NLMC_A);
    float4* o = reinterpret_cast<float4*>(f);
    o[0] = q[0];
    o[1] = q[1];
    o[2] = q[2];
    if (weights) {
        o[3] = q[3];
        o[4] = q[4];
    }
}

__device__ __forceinline__ uint64_t nl_draw(uint64_t deck, int k, const NlParams& p, uint64_t tree, uint64_t key) {  // tree = its id in the epoch
    uint64_t out = 0;
    for (int c = 0; c < k; ++c) {
        const uint32_t pick = rp_pick_uniform(rp_node_hash_draw(p.step_hash, tree, key + (uint64_t)c), (uint32_t)__popcll(deck));
        // the pick-th lowest card of the deck: a popcount search (a loop clearing `pick` bits runs up to 51 rounds)
        uint32_t k = pick, w = (uint32_t)deck, base = 0;
        const uint32_t plo = (uint32_t)__popc(w);
        if (k >= plo) {
            k -= plo;
            w = (uint32_t)(deck >> 32);
            base = 32;
        }
#pragma unroll
        for (uint32_t half = 16; half >= 1; half >>= 1) {
            const uint32_t c = (uint32_t)__popc(w & ((1u << half) - 1u));
            const bool up = k >= c;
            k -= up ? c : 0u;
            w = up ? w >> half : w & ((1u << half) - 1u);
            base += up ? half : 0u;
        }
        const uint64_t card = 1ull << base;
        out |= card;
        deck &= ~card;
    }
    return out;
}
// NlheEncoder::abstraction (nlhe/src/encoder.rs:30-36); Abstraction = [8 bits street][8 bits index] (kicker/src/abstraction.rs:14-24).
// A lookup miss is the reference's panic ("isomorphism not found"): 0xffff and NERR_LOOKUP, the step fails.
__device__ __forceinline__ uint32_t nl_bucket(const NlParams& p, int street, uint64_t pocket, uint64_t board, uint32_t* err) {
    uint64_t cp, cb;
    canonical(pocket, board, &cp, &cb);
    if (p.encoder == 0) {
        // z mod the street's bucket count (169 / 256 / 256 / 101), each with its own compile-time divisor: a 64-bit remainder by
        // a run-time divisor is a software routine of a hundred instructions
        const uint64_t z = rp_mix64((uint64_t)obs_encode(cp, cb) ^ (0x51ed270b5ull * (uint64_t)(street + 1)));
        const uint32_t idx = street == 0 ? (uint32_t)(z % 169ull) : (street == 3 ? (uint32_t)(z % 101ull) : (uint32_t)(z & 255ull));
        return ((uint32_t)street << 8) | idx;
    }
    const int64_t at = table_find(p.tkeys[street], p.tn[street], search_key(cp, cb));
    if (at < 0) {
        *err |= NERR_LOOKUP;
        return 0xffffu;
    }
    return ((uint32_t)street << 8) | (uint32_t)p.tabs[street][at];
}

// What NlheGame::apply needs from the state of a DECISION node, computed once per node: every child of the node is
// game.apply(game.snap(game.actionize(edge))) on the SAME game (nlhe/src/game.rs:50-70), and actor / amounts / permissions
// (kicker game.rs:513-576) do not depend on the edge.  nl_choices_v / nl_action_v are GameN::choices (game.rs:724-739) and
// actionize + snap (:741-753, :835-854) over those cached values: the same decisions as the engine's own functions
// (nlhe_engine.hpp), which recompute them from the seats at every call.
struct NlView {
    int to_call, to_shove, to_raise, pot, street;
    bool may_fold, may_call, may_check, may_raise, may_shove, must_post;
};
__device__ __forceinline__ NlView nl_view(const G2& g) {  // g.turn() is a player
    NlView v;
    const int me = g.actor(), ms = g.max_stake();
    v.to_call = ms - g.at(g.stake, me);
    v.to_shove = g.at(g.stack, me);
    v.to_raise = g.to_raise();
    v.pot = g.pot;
    v.street = g.street();
    v.may_fold = v.to_call > 0;
    v.may_call = v.may_fold && v.to_call < v.to_shove;
    v.may_check = ms == g.at(g.stake, me);
    v.may_raise = v.to_raise < v.to_shove;
    v.may_shove = v.to_shove > 0;
    v.must_post = g.must_post();
    return v;
}
__device__ __forceinline__ int nl_choices_v(const NlView& v, int depth, uint32_t* out) {
    int k = 0;
    if (v.must_post) return 0;
    if (v.may_raise) k += nl_raise_edges(v.street, depth, out + k);
    if (v.may_shove) out[k++] = NE_SHOVE;
    if (v.may_call) out[k++] = NE_CALL;
    if (v.may_fold) out[k++] = NE_FOLD;
    if (v.may_check) out[k++] = NE_CHECK;
    return k;
}
// the same list as a choices Path (5-bit fields, first edge lowest: path.rs), built without an edge array: the raise grid's row
// arrives packed (NL_GRIDP), the four fixed edges are shifted in behind it
__device__ __forceinline__ uint32_t nl_choices_path(const NlView& v, int depth, uint64_t* path) {
    uint64_t p = 0;
    uint32_t k = 0;
    if (!v.must_post) {
        if (v.may_raise && depth <= 3) {  // MAX_RAISE_REPEATS
            const uint32_t gp = (v.street == 0 && depth == 0) ? NL_OPENSP : NL_GRIDP[v.street * 3 + min(depth, 2)];
            p = gp & 0x07ffffffu;
            k = gp >> 27;
        }
        if (v.may_shove) p |= (uint64_t)NE_SHOVE << (5u * k++);
        if (v.may_call) p |= (uint64_t)NE_CALL << (5u * k++);
        if (v.may_fold) p |= (uint64_t)NE_FOLD << (5u * k++);
        if (v.may_check) p |= (uint64_t)NE_CHECK << (5u * k++);
    }
    *path = p;
    return k;
}
__device__ __forceinline__ NlAction nl_action_v(const NlView& v, uint32_t e) {  // snap(actionize(e)), e is not a draw
    const NlAction shove{NA_SHOVE, v.to_shove, 0}, calls{NA_CALL, v.to_call, 0};
    const NlAction passive{v.may_check ? NA_CHECK : NA_FOLD, 0, 0};
    if (e == NE_FOLD) return v.may_fold ? NlAction{NA_FOLD, 0, 0} : NlAction{NA_CHECK, 0, 0};
    if (e == NE_CHECK) return v.may_check ? NlAction{NA_CHECK, 0, 0} : (v.may_call ? calls : NlAction{NA_FOLD, 0, 0});
    if (e == NE_CALL) return v.may_call ? calls : (v.may_shove ? shove : passive);
    bool is_shove = e == NE_SHOVE;
    int chips = 0;
    if (!is_shove) {  // a raise edge
        chips = nl_edge_chips(e, v.pot);
        if (chips >= v.to_shove || !v.may_raise) is_shove = true;  // Raise turns into Shove, which is snapped once more
        else return NlAction{NA_RAISE, chips < v.to_raise ? v.to_raise : chips, 0};
    }
    return v.may_shove ? shove : (v.may_call ? calls : passive);
}

}  // namespace rp

#endif

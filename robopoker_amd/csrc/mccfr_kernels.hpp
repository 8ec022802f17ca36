// mccfr_kernels.hpp — device-side data layout shared by mccfr.hip's kernels and host code.
#ifndef RP_MCCFR_KERNELS_HPP
#define RP_MCCFR_KERNELS_HPP

#include <hip/hip_runtime.h>

#include "../../include/rp_math.h"
#include "../../include/rp_refrng.h"
#include "../../include/rp_libm_glibc.h"
#include "rp_internal.h"

namespace rp {

// rp_state repacked to one 16-byte load: x = turn | n_children << 8, y = info (chance states: rp_state.chance_info), z = offset
struct DevGame {
    const uint4* states;
    const uint32_t* children;
    const uint4* kids;            // [n_children] the child's state record, w = its state id; a terminal child of a
                                  // two-player game carries its payoffs in y, z (no further load on arrival)
    const float* payoffs;         // [n_terminals][n_players]
    const uint8_t* info_actions;  // [n_infos]
    const uint8_t* info_player;   // [n_infos]
    uint32_t n_players;
    uint32_t n_infos;
    uint32_t A;                   // table row stride (max_actions)
    uint32_t root;                // train_root
    uint4 root_rec;               // states[root] with w = root
    // games that match a compile-time skeleton (traverse_static.hpp), when every instance of a skeleton chance node has the
    // same number of outcomes: the record of skeleton node s for the chance outcomes (o_1, o_2, ...) on its path, root first,
    // sits at flat[flat_base[s] + ((o_1 * fan_2 + o_2) * fan_3 + ...)] — a node's record then depends on the sampled outcomes
    // only, not on its parent's record (a chain of ten dependent loads becomes four)
    const uint4* flat;            // NULL: follow the child records
    uint32_t flat_base[48];
    uint32_t flat_fan[48];        // outcomes of skeleton chance node s
};

// regret/strategy tables, SoA by field, row-major [info][A] (Encounter, solver/encounter.rs:22-27)
struct DevTables {
    float* regret;
    float* weight;
    float* payoff;
    uint32_t* visits;
};

// per-tree scratch, lane-interleaved: element (slot, tree) lives at base[slot * stride + tree] so the
// 64 lanes of a wave (64 consecutive trees) touch 64 consecutive dwords
struct DevScratch {
    uint32_t* n_meta;  // parent | edge << 8 | ptype << 16 | leaf << 18 | walker << 19 | nact << 24
    uint32_t* n_info;
    float* n_frel;
    float* n_fsmp;
    float* n_pay;
    float* n_rel;
    float* n_smp;
    float* n_acc;
    uint32_t* s_state;  // leaf stack (TreeBuilder::todo, builder.rs:54)
    uint32_t* s_meta;   // parent | edge << 8 | ptype << 16
    float* s_frel;
    float* s_fsmp;
    float* t_v;         // [A] action values of the span root being evaluated
    size_t stride;
    uint32_t maxn;      // node capacity per tree
    uint32_t maxs;      // stack capacity per tree
};

// Decisions of one batch (solver/decisions.rs:23-32), slot-major and lane-interleaved like the scratch
struct DevDecisions {
    uint32_t* info;     // [maxdec][stride]   0xffffffff = empty slot
    uint32_t* mask;     // [maxdec][stride]   edges present in the regret vector
    float* payoff;      // [maxdec][stride]
    float* regret;      // [maxdec][A][stride]
    float* policy;      // [maxdec][A][stride]
    uint8_t* slotmap;   // [n_infos][stride]  slot + 1 of the tree's Decisions for that infoset, 0 = none (large games)
    uint8_t* ndec;      // [stride]           Decisions produced by the tree
    size_t stride;
    uint32_t maxdec;
};

struct StepParams {
    uint64_t seed;
    uint64_t epoch;
    uint64_t tree_base;  // first tree id of this shard (rank * batch)
    uint32_t batch;
    uint32_t walker;
    int R, W, S;
    float temperature, smoothing, curiosity;
    float prune_threshold, prune_explore;
    uint64_t prune_warmup;
    float regret_min;
    float pow15, pow05;  // powf(t, 1.5), powf(t, 0.5) of t = (float)epoch: DiscountedRegret (host: rp_libm_glibc.h)
    unsigned long long* counters;  // [0] nodes, [1] infos, [2] error flags
    // reference-seed mode (rp_rng_kind RP_RNG_REFERENCE), NULL otherwise: DefaultHasher after t.hash() and info.hash()
    // (flow.rs:290-293) per infoset / per in-tree chance info, refreshed by k_prepare_ref every step
    const rp_sip_mid* ref_info;
    const rp_sip_mid* ref_chance;
};

// ------------------------------------------------------------------------------------------------
// the three draws of SamplingScheme::sample (sample/{mod,external,pluribus}.rs) in either rp_rng_kind.  `ci` = a chance state's
// chance_info (the record's y; 0 = the root deal, which keeps the counter hash in both modes), n its number of outcomes.
// REF is a compile-time switch in the skeleton kernels and p.ref_info != NULL elsewhere.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t d_draw_chance(const StepParams& p, bool ref, uint64_t tree_id, uint32_t state, uint32_t n, uint32_t ci) {
    if (ref && ci) return rp_ref_draw_range(rp_ref_seed_finish(&p.ref_chance[ci - 1u], tree_id), n);  // rng.random_range(0..n)
    return rp_pick_uniform(rp_node_hash(p.seed, p.epoch, tree_id, 0x80000000ull | state), n);
}
__device__ __forceinline__ float d_draw_weight(const StepParams& p, bool ref, uint64_t tree_id, uint32_t info, float total) {
    if (ref) return rp_ref_draw_weight(rp_ref_seed_finish(&p.ref_info[info], tree_id), total);  // Uniform::new(0, total).sample(rng)
    return rp_u01(rp_node_hash(p.seed, p.epoch, tree_id, info)) * total;
}
__device__ __forceinline__ float d_draw_coin(const StepParams& p, bool ref, uint64_t tree_id, uint32_t info) {
    if (ref) return rp_ref_draw_f32(rp_ref_seed_finish(&p.ref_info[info], tree_id));  // rng.random::<f32>()
    return rp_u01(rp_node_hash(p.seed, p.epoch, tree_id, info));
}

// per-cell composed map of the multi-GPU exchange (rp_mccfr_step_local / step_apply)
struct Cell {
    float ra, rb, rm;
    float wa, wb, wm;
    uint32_t rn, wn;
};
struct InfoSum {
    uint32_t count;
    float psum;
};

enum : uint32_t { PT_CHANCE = 0, PT_WALKER = 1, PT_OPP = 2, PT_NONE = 3 };
enum : uint32_t { ERR_NODE_CAPACITY = 1u, ERR_STACK_CAPACITY = 2u, ERR_DEC_CAPACITY = 4u };

// ------------------------------------------------------------------------------------------------
// schedules (regret/*.rs, policy/*.rs) and the per-cell map algebra of the composed update
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float d_regret_gain(int kind, float acc, float imm, float t, float pow15, float pow05, float floor_r) {
    float v;
    switch (kind) {
        case RP_REGRET_LINEAR: {
            const float discount = t / (t + 1.0f);
            v = acc * discount + imm;
        } break;
        case RP_REGRET_DISCOUNTED: {
            float x;
            if (acc > 0.0f) x = pow15;
            else if (acc < 0.0f) x = pow05;
            else x = t / 1.0f;
            const float discount = x / (x + 1.0f);
            v = acc * discount + imm;
        } break;
        case RP_REGRET_ASYMMETRIC: {
            if (acc > 0.0f) v = acc + imm;
            else {
                const float discount = t / (t + 1.0f);
                v = acc * discount + imm;
            }
        } break;
        default: v = acc + imm; break;  // Summed, Floored
    }
    return rp_maxf(v, floor_r);
}
__device__ __forceinline__ float d_weight_learn(int kind, float acc, float imm, float t) {
    float v;
    switch (kind) {
        case RP_WEIGHT_LINEAR: v = acc + imm * t; break;
        case RP_WEIGHT_QUADRATIC: v = acc + imm * t * t; break;
        case RP_WEIGHT_EXPONENTIAL: v = acc * 0.9999f + imm; break;
        default: v = acc + imm; break;
    }
    return rp_maxf(v, RP_EPSILON);
}
__host__ __device__ inline float regret_floor_of(int R, float regret_min) {
    if (R == RP_REGRET_FLOORED) return 0.0f;
    if (R == RP_REGRET_SUMMED) return rp_u2f(0xff800000u);
    return regret_min;
}


struct Map {
    float a, b, m;
    uint32_t n;
};
// touches per LDS tile of the block-map chains
__host__ __device__ inline uint32_t compose_block(uint32_t max_actions) { return (1024u / (2u * max_actions)) & ~3u; }
__device__ __forceinline__ Map map_compose(const Map& first, const Map& second) {  // `first` is applied first
    if (second.n == 0) return first;
    if (first.n == 0) return second;
    Map r;
    r.a = second.a * first.a;
    r.b = second.a * first.b + second.b;
    const float t = rp_f2u(first.m) == 0xff800000u ? first.m : second.a * first.m + second.b;
    r.m = rp_maxf(t, second.m);
    r.n = first.n + second.n;
    return r;
}

// first touch of an empty map sets (d, delta, floor); later touches compose one step (oracle: map_touch)
__device__ __forceinline__ void map_touch(Map& mp, float d, float delta, float floor_v) {
    if (mp.n == 0) {
        mp.a = d;
        mp.b = delta;
        mp.m = floor_v;
    } else {
        mp.a = mp.a * d;
        mp.b = mp.b * d + delta;
        mp.m = rp_maxf(mp.m * d + delta, floor_v);
    }
    mp.n += 1;
}

}  // namespace rp

#endif

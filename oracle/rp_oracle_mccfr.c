/* rp_oracle_mccfr.c — CPU restatement of robopoker's `mccfr` hot path.            TEST INFRASTRUCTURE.
 *
 * This file is the ORACLE for the MI355X MCCFR path: a plain-C, single-threaded, deterministic
 * restatement of the reference algorithm.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (librp_mi355x.so) never does.
 *
 * What it follows (paths under /root/reference/crates, cited per function):
 *   mccfr/src/solver/solver.rs   Solver::{step, batch, update_*, exploitability}
 *   mccfr/src/solver/builder.rs  TreeBuilder (explicit DFS stack, node numbering)
 *   mccfr/src/strategy/flow.rs   CfrFlow::{regret_denom, weight_denom, sampling_weight, dfs,
 *                                ancestor_reach, recursed_value, rng}
 *   mccfr/src/strategy/profile.rs, book.rs, nash.rs; mccfr/src/{sample,regret,policy}/ (all files)
 *
 * PARITY STATUS.  The reference cannot be compiled or run here (no Rust toolchain) and its MCCFR is
 * not bit-reproducible against itself (thread RNG root deals, RandomState HashMaps).
 * The oracle is therefore pinned against every known-answer test the reference holds for this path —
 * Kuhn analytic Nash (kuhn/src/solver.rs:176-203), the 44 Kuhn / 3 Leduc exploitability thresholds,
 * RPS equilibrium, sampling-distribution normalisation — see tests/test_oracle_mccfr.py.  The seed -> sample chain
 * (DefaultHasher / SmallRng / WeightedIndex / random_range / random::<f32>(), flow.rs:285-295) is third-party but published:
 * include/rp_refrng.h restates it and pins it to the published vectors (tests/test_refrng.py); ora_mccfr_set_rng(RP_RNG_REFERENCE)
 * draws every sampled branch through it ("reference-seed" mode; the default keeps include/rp_math.h's counter hash, same
 * structure: one hash per (epoch, info, tree)).  DiscountedRegret's powers (discounted.rs:33,37) are what a build of the reference
 * computes: powf(t, 1.5) = glibc's powf, restated in include/rp_libm_glibc.h and equal to this machine's on every positive float
 * (tests/test_libm_glibc.py); powf(t, 0.5) = sqrt(t), LLVM's fold of pow(x, 0.5) at the workspace's opt-level 3.
 *
 * f32 operation order follows the reference expression by expression (sums fold left from 0 in
 * `choices()` order; petgraph's newest-edge-first adjacency over children pushed in reverse pop order
 * yields `choices()` order — mccfr/src/state/node.rs:103-107, solver/builder.rs:141-161).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rp_math.h"
#include "../include/rp_mi355x.h"
#include "../include/rp_refrng.h"
#include "../include/rp_libm_glibc.h"

#define ORA_MAXA 16

typedef struct ora_node {
    uint32_t state;
    int32_t parent;
    int32_t edge;             /* child slot taken at the parent */
    int32_t kids[ORA_MAXA];   /* node index per child slot, -1 if not expanded */
} ora_node;

typedef struct ora_tree {
    ora_node* nodes;
    uint32_t n, cap;
    uint64_t id;
} ora_tree;

typedef struct ora_decision {
    uint32_t info;
    uint32_t n_actions;
    uint32_t expanded;        /* bitmask of edges present in the regret vector */
    float regret[ORA_MAXA];
    float policy[ORA_MAXA];
    float payoff;
    uint64_t tree;
} ora_decision;

typedef struct ora_mccfr {
    rp_game_table g;
    rp_state* states;
    uint32_t* children;
    float* payoffs;
    uint8_t* info_actions;
    uint8_t* info_player;
    float* default_regret;
    int R, W, S;
    rp_hyper hp;
    uint64_t seed;
    int rng;                       /* rp_rng_kind */
    rp_hash_stream* info_streams;  /* RP_RNG_REFERENCE: I::hash byte stream per infoset / per in-tree chance info */
    rp_hash_stream* chance_streams;
    uint32_t n_chance_streams;
    uint32_t batch;
    uint64_t epoch;
    uint64_t nodes, infos;
    float* regret;
    float* weight;
    float* payoff;
    uint32_t* visits;
    /* scratch */
    ora_tree tree;
    ora_decision* dec;
    uint64_t ndec, capdec;
} ora_mccfr;

/* ------------------------------------------------------------------ profile reads */
/* RefProf::{regret, weight} (profile.rs:31-37): floored at EPSILON */
static float p_regret(const ora_mccfr* h, uint32_t info, uint32_t a) {
    return rp_maxf(h->regret[info * h->g.max_actions + a], RP_EPSILON);
}
static float p_weight(const ora_mccfr* h, uint32_t info, uint32_t a) {
    return rp_maxf(h->weight[info * h->g.max_actions + a], RP_EPSILON);
}
/* CfrFlow::regret_denom (flow.rs:20-22) */
static float regret_denom(const ora_mccfr* h, uint32_t info) {
    float s = 0.0f;
    for (uint32_t a = 0; a < h->info_actions[info]; ++a) s += p_regret(h, info, a);
    return s;
}
/* CfrFlow::weight_denom (flow.rs:24-26) */
static float weight_denom(const ora_mccfr* h, uint32_t info) {
    float s = 0.0f;
    for (uint32_t a = 0; a < h->info_actions[info]; ++a) s += p_weight(h, info, a);
    return s + h->hp.smoothing;
}
/* CfrFlow::sampling_weight (flow.rs:30-32) */
static float sampling_weight(const ora_mccfr* h, uint32_t info, uint32_t a, float denom) {
    return rp_maxf((p_weight(h, info, a) / h->hp.temperature + h->hp.smoothing) / denom, h->hp.curiosity);
}
static float sampling_z(const ora_mccfr* h, uint32_t info, float denom) {
    float z = 0.0f;
    for (uint32_t a = 0; a < h->info_actions[info]; ++a) z += sampling_weight(h, info, a, denom);
    return z;
}
/* CfrFlow::instant_policy (flow.rs:46-48) */
static float instant_policy(const ora_mccfr* h, uint32_t info, uint32_t a) {
    return p_regret(h, info, a) / regret_denom(h, info);
}
/* CfrFlow::sampling (flow.rs:52-59) */
static float sampling_prob(const ora_mccfr* h, uint32_t info, uint32_t a) {
    float denom = weight_denom(h, info);
    float z = sampling_z(h, info, denom);
    return sampling_weight(h, info, a, denom) / z;
}
/* RefProf::averaged_distribution(..).density(edge) (profile.rs:40-44, nash.rs:14-16) */
static float averaged_policy(const ora_mccfr* h, uint32_t info, uint32_t a) {
    float sum = 0.0f;
    for (uint32_t k = 0; k < h->info_actions[info]; ++k) sum += p_weight(h, info, k);
    return p_weight(h, info, a) / sum;
}

/* ------------------------------------------------------------------ tree */
static uint32_t tree_push(ora_tree* t, uint32_t state, int32_t parent, int32_t edge) {
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 64;
        t->nodes = (ora_node*)realloc(t->nodes, (size_t)t->cap * sizeof(ora_node));
    }
    ora_node* nd = &t->nodes[t->n];
    nd->state = state;
    nd->parent = parent;
    nd->edge = edge;
    for (int k = 0; k < ORA_MAXA; ++k) nd->kids[k] = -1;
    if (parent >= 0) t->nodes[parent].kids[edge] = (int32_t)t->n;
    return t->n++;
}
static int node_width(const ora_node* nd) {
    int w = 0;
    for (int k = 0; k < ORA_MAXA; ++k) w += nd->kids[k] >= 0;
    return w;
}

/* CfrFlow::rng (flow.rs:285-295): one hash per (epoch, info, tree id); chance nodes key on the state */
static uint64_t node_hash(const ora_mccfr* h, uint64_t tree_id, const rp_state* st, uint32_t state) {
    uint64_t key = st->turn == RP_TURN_CHANCE ? (0x80000000ull | state) : (uint64_t)st->info;
    return rp_node_hash(h->seed, h->epoch, tree_id, key);
}
/* reference-seed mode: does this node draw from the reference's own chain?  (a chance state without a chance_info is the root
 * deal, which the reference takes from the thread RNG — kuhn/src/game.rs:115-123 — and keeps the counter hash) */
static int node_is_ref(const ora_mccfr* h, const rp_state* st) {
    if (h->rng != RP_RNG_REFERENCE) return 0;
    return st->turn == RP_TURN_CHANCE ? st->chance_info != 0 : 1;
}
/* flow.rs:285-295: DefaultHasher over t, info, node.seed() -> the u64 SmallRng::seed_from_u64 takes */
static uint64_t node_seed_ref(const ora_mccfr* h, uint64_t tree_id, const rp_state* st) {
    const rp_hash_stream* hs = st->turn == RP_TURN_CHANCE ? &h->chance_streams[st->chance_info - 1] : &h->info_streams[st->info];
    return rp_ref_node_seed(h->epoch, hs->bytes, hs->len, tree_id);
}
/* the three draws of SamplingScheme::sample, in either mode */
static uint32_t draw_range(const ora_mccfr* h, uint64_t tree_id, const rp_state* st, uint32_t state, uint32_t n) {
    if (node_is_ref(h, st)) return rp_ref_draw_range(node_seed_ref(h, tree_id, st), n); /* rng.random_range(0..n) */
    return rp_pick_uniform(node_hash(h, tree_id, st, state), n);
}
static float draw_weight(const ora_mccfr* h, uint64_t tree_id, const rp_state* st, uint32_t state, float total) {
    if (node_is_ref(h, st)) return rp_ref_draw_weight(node_seed_ref(h, tree_id, st), total); /* Uniform::new(0, total).sample */
    return rp_u01(node_hash(h, tree_id, st, state)) * total;
}
static float draw_f32(const ora_mccfr* h, uint64_t tree_id, const rp_state* st, uint32_t state) {
    if (node_is_ref(h, st)) return rp_ref_draw_f32(node_seed_ref(h, tree_id, st)); /* rng.random::<f32>() */
    return rp_u01(node_hash(h, tree_id, st, state));
}

/* SamplingScheme::sample: returns a bitmask over child slots to expand.
 * vanilla != 0 -> VanillaSampling (sample/vanilla.rs), used only by exploitability. */
static uint32_t sample_mask(const ora_mccfr* h, uint64_t tree_id, uint32_t state, uint32_t walker, int vanilla) {
    const rp_state* st = &h->states[state];
    uint32_t n = st->n_children;
    uint32_t all = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
    if (n == 0) return 0;
    if (vanilla) return all;
    if (st->turn == RP_TURN_CHANCE) {
        /* randomly (sample/mod.rs:68-82) */
        return 1u << draw_range(h, tree_id, st, state, n);
    }
    if (st->turn != walker) {
        /* weighted (sample/external.rs:41-64): WeightedIndex over sampling_distribution.max(EPSILON) */
        uint32_t info = st->info;
        float denom = weight_denom(h, info);
        float raw[ORA_MAXA];
        float z = 0.0f;
        for (uint32_t a = 0; a < n; ++a) {
            raw[a] = sampling_weight(h, info, a, denom);
            z += raw[a];
        }
        float cum[ORA_MAXA];
        float total = 0.0f;
        for (uint32_t a = 0; a < n; ++a) {
            total += rp_maxf(raw[a] / z, RP_EPSILON);
            cum[a] = total;
        }
        float x = draw_weight(h, tree_id, st, state, total);
        uint32_t idx = 0;
        while (idx + 1 < n && cum[idx] <= x) ++idx;
        return 1u << idx;
    }
    /* walker node */
    if (h->S == RP_SAMPLING_EXTERNAL) return all;
    uint32_t info = st->info;
    if (h->S == RP_SAMPLING_PLURIBUS) {
        /* sample/pluribus.rs:72-101 */
        if (h->epoch < h->hp.prune_warmup) return all;
        if (draw_f32(h, tree_id, st, state) < h->hp.prune_explore) return all;
    }
    uint32_t mask = 0;
    for (uint32_t a = 0; a < n; ++a) {
        int keep = h->regret[info * h->g.max_actions + a] > h->hp.prune_threshold;
        if (h->S == RP_SAMPLING_PLURIBUS) {
            const rp_state* c = &h->states[h->children[st->offset + a]];
            keep = keep || c->turn == RP_TURN_TERMINAL;
        }
        if (keep) mask |= 1u << a;
    }
    return mask ? mask : all; /* sample/pruning.rs:64, pluribus.rs:99 */
}

/* TreeBuilder::{new, next, build} (builder.rs:74-87,141-161): explicit stack, pop-last order */
typedef struct ora_leaf {
    uint32_t state;
    int32_t parent;
    int32_t edge;
} ora_leaf;

static void build_tree(ora_mccfr* h, uint64_t tree_id, uint32_t root, uint32_t walker, int vanilla) {
    ora_tree* t = &h->tree;
    t->n = 0;
    t->id = tree_id;
    size_t cap = 256, top = 0;
    ora_leaf* todo = (ora_leaf*)malloc(cap * sizeof(ora_leaf));
    uint32_t r = tree_push(t, root, -1, -1);
    uint32_t cur = r;
    uint32_t cur_state = root;
    for (;;) {
        const rp_state* st = &h->states[cur_state];
        uint32_t mask = sample_mask(h, tree_id, cur_state, walker, vanilla);
        for (uint32_t k = 0; k < st->n_children; ++k) {
            if (!(mask >> k & 1u)) continue;
            if (top == cap) {
                cap *= 2;
                todo = (ora_leaf*)realloc(todo, cap * sizeof(ora_leaf));
            }
            todo[top].state = h->children[st->offset + k];
            todo[top].parent = (int32_t)cur;
            todo[top].edge = (int32_t)k;
            ++top;
        }
        if (top == 0) break;
        ora_leaf lf = todo[--top];
        cur = tree_push(t, lf.state, lf.parent, lf.edge);
        cur_state = lf.state;
    }
    free(todo);
}

/* ------------------------------------------------------------------ counterfactual values */
/* CfrNash::terminal_value (nash.rs:66-79), terminal case only (trees are never depth limited here) */
static float terminal_value(const ora_mccfr* h, uint32_t state, uint32_t hero) {
    return h->payoffs[h->states[state].offset * h->g.n_players + hero];
}

/* CfrFlow::recursed_value (flow.rs:182-216) */
static float recursed_value(const ora_mccfr* h, uint32_t walker, uint32_t hero, int32_t node, float rel, float smp) {
    const ora_node* nd = &h->tree.nodes[node];
    const rp_state* st = &h->states[nd->state];
    if (node_width(nd) == 0) return rel / smp * terminal_value(h, nd->state, hero);
    int chance = st->turn == RP_TURN_CHANCE;
    int is_walker = st->turn == walker;
    float rd = 0.0f, denom = 0.0f, z = 0.0f;
    if (!chance) rd = regret_denom(h, st->info);
    if (!chance && !is_walker) {
        denom = weight_denom(h, st->info);
        z = sampling_z(h, st->info, denom);
    }
    float sum = 0.0f;
    for (uint32_t k = 0; k < st->n_children; ++k) {
        if (nd->kids[k] < 0) continue;
        float r2 = rel * (chance ? 1.0f : p_regret(h, st->info, k) / rd);
        float s2 = smp * ((!chance && !is_walker) ? sampling_weight(h, st->info, k, denom) / z : 1.0f);
        sum += recursed_value(h, walker, hero, nd->kids[k], r2, s2);
    }
    return sum;
}

/* CfrFlow::ancestor_reach (flow.rs:166-174): upward over non-walker decision ancestors */
static float ancestor_reach(const ora_mccfr* h, uint32_t walker, int32_t node) {
    float cf = 1.0f, sm = 1.0f;
    const ora_node* nd = &h->tree.nodes[node];
    while (nd->parent >= 0) {
        const ora_node* par = &h->tree.nodes[nd->parent];
        const rp_state* ps = &h->states[par->state];
        if (ps->turn != RP_TURN_CHANCE && ps->turn != walker) {
            cf = cf * instant_policy(h, ps->info, (uint32_t)nd->edge);
            sm = sm * sampling_prob(h, ps->info, (uint32_t)nd->edge);
        }
        nd = par;
    }
    return cf / sm;
}

static ora_decision* push_decision(ora_mccfr* h) {
    if (h->ndec == h->capdec) {
        h->capdec = h->capdec ? h->capdec * 2 : 1024;
        h->dec = (ora_decision*)realloc(h->dec, (size_t)h->capdec * sizeof(ora_decision));
    }
    ora_decision* d = &h->dec[h->ndec++];
    memset(d, 0, sizeof(*d));
    return d;
}

/* Solver::record_infosets + update_vector (solver.rs:263-275,296-305) with CfrFlow::dfs (flow.rs:64-87):
 * Tree::partition groups non-leaf nodes by info in ascending node index (tree.rs:88-98); groups whose head
 * is a walker node become Decisions. */
static void tree_decisions(ora_mccfr* h, uint32_t walker) {
    const ora_tree* t = &h->tree;
    for (uint32_t i = 0; i < t->n; ++i) {
        const ora_node* nd = &t->nodes[i];
        const rp_state* st = &h->states[nd->state];
        if (st->turn != walker || node_width(nd) == 0) continue;
        uint32_t info = st->info;
        int head = 1;
        for (uint32_t j = 0; j < i; ++j) {
            const ora_node* o = &t->nodes[j];
            const rp_state* os = &h->states[o->state];
            if (os->turn < RP_TURN_CHANCE && os->info == info && node_width(o) > 0) head = 0;
        }
        if (!head) continue;
        ora_decision* d = push_decision(h);
        d->info = info;
        d->n_actions = h->info_actions[info];
        d->tree = t->id;
        float rd = regret_denom(h, info);
        /* policy_vector = iterated_distribution (flow.rs:118-120, profile.rs:47-51) */
        for (uint32_t a = 0; a < d->n_actions; ++a) d->policy[a] = p_regret(h, info, a) / rd;
        float payoff = 0.0f;
        for (uint32_t j = i; j < t->n; ++j) { /* span, ascending node index */
            const ora_node* root = &t->nodes[j];
            const rp_state* rs = &h->states[root->state];
            if (rs->turn >= RP_TURN_CHANCE || rs->info != info || node_width(root) == 0) continue;
            float reach = ancestor_reach(h, walker, (int32_t)j);
            float v[ORA_MAXA];
            float ev = 0.0f;
            for (uint32_t a = 0; a < rs->n_children; ++a) {
                if (root->kids[a] < 0) continue;
                v[a] = reach * recursed_value(h, walker, rs->turn, root->kids[a], 1.0f, 1.0f);
            }
            for (uint32_t a = 0; a < rs->n_children; ++a) {
                if (root->kids[a] < 0) continue;
                ev += p_regret(h, info, a) / rd * v[a];
            }
            payoff += ev;
            for (uint32_t a = 0; a < rs->n_children; ++a) {
                if (root->kids[a] < 0) continue;
                d->expanded |= 1u << a;
                d->regret[a] += v[a] - ev;
            }
        }
        d->payoff = payoff;
        h->infos += 1; /* inc_infos(1) per walker infoset (solver.rs:273) */
    }
}

/* ------------------------------------------------------------------ schedules */
/* RegretSchedule::accumulate (regret/{summed,linear,discounted,floored,asymmetric}.rs) */
static float regret_accumulate(int kind, float acc, float imm, uint64_t epoch) {
    float t = (float)epoch;
    switch (kind) {
        case RP_REGRET_SUMMED:
        case RP_REGRET_FLOORED:
            return acc + imm;
        case RP_REGRET_LINEAR: {
            float discount = t / (t + 1.0f);
            return acc * discount + imm;
        }
        case RP_REGRET_DISCOUNTED: {
            float p = 1.0f;
            float x;
            if (acc > 0.0f) x = rp_pow15(t / p);      /* (t / p).powf(ALPHA): a call to libm's powf = glibc's, restated (rp_libm_glibc.h) */
            else if (acc < 0.0f) x = rp_pow05(t / p); /* (t / p).powf(BETA): LLVM folds pow(x, 0.5) into sqrt(x) at opt-level 3 */
            else x = t / p;
            float discount = x / (x + 1.0f);
            return acc * discount + imm;
        }
        case RP_REGRET_ASYMMETRIC: {
            if (acc > 0.0f) return acc + imm;
            float discount = t / (t + 1.0f);
            return acc * discount + imm;
        }
    }
    return acc + imm;
}
/* RegretSchedule::accumulate as a pure function (known answers for the reference: scripts/make_reference_kat.py) */
__attribute__((visibility("default"))) float ora_regret_accumulate(int kind, float acc, float imm, uint64_t epoch) {
    return regret_accumulate(kind, acc, imm, epoch);
}
static float regret_floor(const ora_mccfr* h) {
    if (h->R == RP_REGRET_FLOORED) return 0.0f;
    if (h->R == RP_REGRET_SUMMED) return rp_u2f(0xff800000u);
    return h->hp.regret_min;
}
/* RegretSchedule::gain (regret/mod.rs:22-24) */
static float regret_gain(const ora_mccfr* h, float acc, float imm) {
    return rp_maxf(regret_accumulate(h->R, acc, imm, h->epoch), regret_floor(h));
}
/* WeightSchedule::learn (policy/mod.rs:22-24) over policy/{constant,linear,quadratic,exponential}.rs */
static float weight_learn(const ora_mccfr* h, float acc, float imm) {
    float t = (float)h->epoch;
    float v;
    switch (h->W) {
        case RP_WEIGHT_LINEAR: v = acc + imm * t; break;
        case RP_WEIGHT_QUADRATIC: v = acc + imm * t * t; break;
        case RP_WEIGHT_EXPONENTIAL: v = acc * 0.9999f + imm; break;
        default: v = acc + imm; break;
    }
    return rp_maxf(v, RP_EPSILON);
}

/* Solver::update_{regret,weight,payoff,visits} (solver.rs:143-192) for one Decisions */
static void apply_decision(ora_mccfr* h, const ora_decision* d) {
    uint32_t A = h->g.max_actions;
    for (uint32_t a = 0; a < d->n_actions; ++a) {
        if (!(d->expanded >> a & 1u)) continue;
        float* r = &h->regret[d->info * A + a];
        *r = regret_gain(h, *r, d->regret[a]);
    }
    for (uint32_t a = 0; a < d->n_actions; ++a) {
        float* w = &h->weight[d->info * A + a];
        *w = weight_learn(h, *w, d->policy[a]);
    }
    for (uint32_t a = 0; a < d->n_actions; ++a) {
        uint32_t n = h->visits[d->info * A + a];
        float* ev = &h->payoff[d->info * A + a];
        *ev += (d->payoff - *ev) / (float)(n + 1u);
    }
    for (uint32_t a = 0; a < d->n_actions; ++a) h->visits[d->info * A + a] += 1u;
}

/* ------------------------------------------------------------------ public API */
#define ORA_API __attribute__((visibility("default")))

ORA_API ora_mccfr* ora_mccfr_create(const rp_game_table* g, int R, int W, int S, uint32_t batch,
                                    const rp_hyper* hp, uint64_t seed) {
    if (!g || g->max_actions > ORA_MAXA) return NULL;
    ora_mccfr* h = (ora_mccfr*)calloc(1, sizeof(ora_mccfr));
    h->g = *g;
    h->states = (rp_state*)malloc(sizeof(rp_state) * g->n_states);
    memcpy(h->states, g->states, sizeof(rp_state) * g->n_states);
    h->children = (uint32_t*)malloc(4u * g->n_children);
    memcpy(h->children, g->children, 4u * g->n_children);
    h->payoffs = (float*)malloc(4u * g->n_terminals * g->n_players);
    memcpy(h->payoffs, g->payoffs, 4u * g->n_terminals * g->n_players);
    h->info_actions = (uint8_t*)malloc(g->n_infos);
    memcpy(h->info_actions, g->info_actions, g->n_infos);
    h->info_player = (uint8_t*)malloc(g->n_infos);
    memcpy(h->info_player, g->info_player, g->n_infos);
    size_t cells = (size_t)g->n_infos * g->max_actions;
    h->default_regret = (float*)calloc(cells, 4);
    if (g->default_regret) memcpy(h->default_regret, g->default_regret, cells * 4);
    h->R = R;
    h->W = W;
    h->S = S;
    h->hp = *hp;
    h->seed = seed;
    h->batch = batch ? batch : 1;
    h->regret = (float*)malloc(cells * 4);
    memcpy(h->regret, h->default_regret, cells * 4); /* cum_regret of a missing entry = default_regret (book.rs:101-106) */
    h->weight = (float*)calloc(cells, 4);
    h->payoff = (float*)calloc(cells, 4);
    h->visits = (uint32_t*)calloc(cells, 4);
    return h;
}

/* rp_mccfr_set_rng: which generator draws the sampled branches (include/rp_mi355x.h rp_rng_kind) */
ORA_API int ora_mccfr_set_rng(ora_mccfr* h, int kind, const rp_hash_streams* st) {
    if (!h) return -1;
    if (kind == RP_RNG_COUNTER) {
        h->rng = RP_RNG_COUNTER;
        return 0;
    }
    if (kind != RP_RNG_REFERENCE || !st || st->n_infos != h->g.n_infos || !st->infos) return -1;
    for (uint32_t s = 0; s < h->g.n_states; ++s)
        if (h->states[s].turn == RP_TURN_CHANCE && h->states[s].chance_info > st->n_chance) return -1;
    free(h->info_streams);
    free(h->chance_streams);
    h->info_streams = (rp_hash_stream*)malloc(sizeof(rp_hash_stream) * (st->n_infos ? st->n_infos : 1));
    memcpy(h->info_streams, st->infos, sizeof(rp_hash_stream) * st->n_infos);
    h->chance_streams = (rp_hash_stream*)malloc(sizeof(rp_hash_stream) * (st->n_chance ? st->n_chance : 1));
    if (st->n_chance) memcpy(h->chance_streams, st->chance, sizeof(rp_hash_stream) * st->n_chance);
    h->n_chance_streams = st->n_chance;
    h->rng = RP_RNG_REFERENCE;
    return 0;
}

ORA_API void ora_mccfr_destroy(ora_mccfr* h) {
    if (!h) return;
    free(h->info_streams); free(h->chance_streams);
    free(h->states); free(h->children); free(h->payoffs); free(h->info_actions); free(h->info_player);
    free(h->default_regret); free(h->regret); free(h->weight); free(h->payoff); free(h->visits);
    free(h->tree.nodes); free(h->dec);
    free(h);
}

/* Solver::batch for tree ids [first, first+count) (solver.rs:225-250); Decisions land in h->dec */
static void batch_range(ora_mccfr* h, uint64_t first, uint64_t count) {
    uint32_t walker = (uint32_t)(h->epoch % h->g.n_players); /* CfrSampling::walker (book.rs:142-144) */
    for (uint64_t i = 0; i < count; ++i) {
        build_tree(h, first + i, h->g.train_root, walker, 0);
        h->nodes += h->tree.n;
        tree_decisions(h, walker);
    }
}

/* Solver::step (solver.rs:96-105) */
ORA_API void ora_mccfr_step(ora_mccfr* h) {
    h->ndec = 0;
    batch_range(h, 0, h->batch);
    for (uint64_t i = 0; i < h->ndec; ++i) apply_decision(h, &h->dec[i]);
    h->epoch += 1;
}

/* Solver::step with the reference's parallel structure (solver.rs:225-250: rayon over the trees of a batch, then the
 * sequential update on one thread): `threads` workers traverse contiguous tree-id ranges into private Decisions lists,
 * which are applied in tree-id order — the same result as ora_mccfr_step, bit for bit (the batch is pure w.r.t. the
 * profile).  bench.py's all-core CPU baseline. */
ORA_API void ora_mccfr_step_mt(ora_mccfr* h, uint32_t threads) {
    if (threads < 2) {
        ora_mccfr_step(h);
        return;
    }
    ora_mccfr* shadow = (ora_mccfr*)malloc(sizeof(ora_mccfr) * threads);
    for (uint32_t t = 0; t < threads; ++t) {
        shadow[t] = *h; /* shares the game and the tables (read only here); private scratch and counters */
        memset(&shadow[t].tree, 0, sizeof(ora_tree));
        shadow[t].dec = NULL;
        shadow[t].ndec = shadow[t].capdec = 0;
        shadow[t].nodes = shadow[t].infos = 0;
    }
#pragma omp parallel for schedule(static, 1) num_threads(threads)
    for (uint32_t t = 0; t < threads; ++t) {
        uint64_t lo = (uint64_t)h->batch * t / threads, hi = (uint64_t)h->batch * (t + 1) / threads;
        batch_range(&shadow[t], lo, hi - lo);
    }
    for (uint32_t t = 0; t < threads; ++t) {
        for (uint64_t i = 0; i < shadow[t].ndec; ++i) apply_decision(h, &shadow[t].dec[i]);
        h->nodes += shadow[t].nodes;
        h->infos += shadow[t].infos;
        free(shadow[t].tree.nodes);
        free(shadow[t].dec);
    }
    free(shadow);
    h->epoch += 1;
}

ORA_API void ora_mccfr_solve(ora_mccfr* h, uint64_t trees) {
    for (uint64_t i = 0; i < trees / h->batch; ++i) ora_mccfr_step(h);
}

/* the batch without the update: Decisions of the current epoch, tree-id major (for kernel debugging) */
ORA_API uint64_t ora_mccfr_batch(ora_mccfr* h, ora_decision** out) {
    h->ndec = 0;
    uint64_t nodes = h->nodes, infos = h->infos;
    batch_range(h, 0, h->batch);
    h->nodes = nodes;
    h->infos = infos;
    *out = h->dec;
    return h->ndec;
}

/* ---- the multi-GPU exchange semantics (include/rp_mi355x.h, rp_mccfr_step_local/apply) --------------
 * `world` ranks each sample tree ids [r*B, (r+1)*B); every table cell receives one composed map per rank,
 * F(x) = max(a*x + b, m), built sequentially over that rank's touches in tree order; ranks are folded in
 * rank order.  Exact in real arithmetic for schedules with a sign-independent discount (Summed, Linear,
 * Floored x all weight schedules); Discounted/Asymmetric are rejected (-1). */
typedef struct ora_cell {
    float ra, rb, rm; /* regret map   */
    float wa, wb, wm; /* weight map   */
    uint32_t rn, wn;  /* touch counts */
} ora_cell;

static int composed_discount(const ora_mccfr* h, float* dr, float* dw) {
    float t = (float)h->epoch;
    switch (h->R) {
        case RP_REGRET_SUMMED:
        case RP_REGRET_FLOORED: *dr = 1.0f; break;
        case RP_REGRET_LINEAR: *dr = t / (t + 1.0f); break;
        default: return -1;
    }
    *dw = h->W == RP_WEIGHT_EXPONENTIAL ? 0.9999f : 1.0f;
    return 0;
}
static float composed_wdelta(const ora_mccfr* h, float sigma) {
    float t = (float)h->epoch;
    switch (h->W) {
        case RP_WEIGHT_LINEAR: return sigma * t;
        case RP_WEIGHT_QUADRATIC: return sigma * t * t;
        default: return sigma;
    }
}

typedef struct ora_isum {
    uint32_t count;
    float psum;
} ora_isum;

/* bytes of one rank's summary blob: [n_infos*A ora_cell][n_infos ora_isum] — the same layout as the device's
 * rp_mccfr_summary_bytes, so host logic written against one works against the other */
ORA_API size_t ora_mccfr_summary_bytes(const ora_mccfr* h) {
    return (size_t)h->g.n_infos * h->g.max_actions * sizeof(ora_cell) + (size_t)h->g.n_infos * sizeof(ora_isum);
}

/* composition of two per-cell maps, `first` applied before `second`:
 * an untouched map is the identity and is skipped exactly; -inf floors stay -inf (no 0 * inf) */
typedef struct ora_map {
    float a, b, m;
    uint32_t n;
} ora_map;
static ora_map map_compose(ora_map first, ora_map second) {
    if (second.n == 0) return first;
    if (first.n == 0) return second;
    ora_map r;
    r.a = second.a * first.a;
    r.b = second.a * first.b + second.b;
    float t = rp_f2u(first.m) == 0xff800000u ? first.m : second.a * first.m + second.b;
    r.m = rp_maxf(t, second.m);
    r.n = first.n + second.n;
    return r;
}
static void map_touch(ora_map* mp, float d, float delta, float floor_v) {
    if (mp->n == 0) { mp->a = d; mp->b = delta; mp->m = floor_v; }
    else { mp->a = mp->a * d; mp->b = mp->b * d + delta; mp->m = rp_maxf(mp->m * d + delta, floor_v); }
    mp->n += 1;
}
static const ora_map MAP_ID = {1.0f, 0.0f, -INFINITY, 0};

/* rp_mccfr_step_local: this rank's trees [rank*B, (rank+1)*B) against the current table -> composed maps.
 * A BLOCK is the set of Decisions of one infoset produced by one chunk of RP_COMPOSE_CHUNK consecutive trees of the
 * rank, composed sequentially in tree order from the identity; the blocks of an infoset are folded in chunk order,
 * RP_FOLD_GROUP chunks to a group, groups in order (include/rp_mi355x.h).  An infoset that a chunk did not visit
 * contributes the identity map and a zero payoff sum, which change nothing. */
ORA_API int ora_mccfr_step_local(ora_mccfr* h, uint32_t rank, void* blob) {
    float dr, dw;
    if (composed_discount(h, &dr, &dw)) return -1;
    uint32_t A = h->g.max_actions, NI = h->g.n_infos;
    size_t cells = (size_t)NI * A;
    float floor_r = regret_floor(h);
    ora_cell* cell = (ora_cell*)blob;
    ora_isum* sums = (ora_isum*)((unsigned char*)blob + cells * sizeof(ora_cell));
    ora_map* blk_r = (ora_map*)malloc(cells * sizeof(ora_map));   /* open chunk */
    ora_map* blk_w = (ora_map*)malloc(cells * sizeof(ora_map));
    ora_map* sup_r = (ora_map*)malloc(cells * sizeof(ora_map));   /* open group of RP_FOLD_GROUP chunks */
    ora_map* sup_w = (ora_map*)malloc(cells * sizeof(ora_map));
    ora_map* tot_r = (ora_map*)malloc(cells * sizeof(ora_map));
    ora_map* tot_w = (ora_map*)malloc(cells * sizeof(ora_map));
    float* blk_p = (float*)calloc(NI, 4);
    float* sup_p = (float*)calloc(NI, 4);
    float* tot_p = (float*)calloc(NI, 4);
    uint32_t* tot_pn = (uint32_t*)calloc(NI, 4);
    for (size_t c = 0; c < cells; ++c) blk_r[c] = blk_w[c] = sup_r[c] = sup_w[c] = tot_r[c] = tot_w[c] = MAP_ID;
    h->ndec = 0;
    uint64_t first = (uint64_t)rank * h->batch;
    batch_range(h, first, h->batch);
    uint32_t n_chunks = (h->batch + RP_COMPOSE_CHUNK - 1) / RP_COMPOSE_CHUNK;
    uint32_t open = 0; /* chunk being filled */
    for (uint64_t i = 0; i <= h->ndec; ++i) {
        uint32_t c = i < h->ndec ? (uint32_t)((h->dec[i].tree - first) / RP_COMPOSE_CHUNK) : n_chunks;
        while (open < c) { /* close chunk `open` (left folds from the identity / 0.0f, like the device) */
            for (uint32_t info = 0; info < NI; ++info) {
                for (uint32_t a = 0; a < A; ++a) {
                    size_t k = (size_t)info * A + a;
                    sup_r[k] = map_compose(sup_r[k], blk_r[k]);
                    sup_w[k] = map_compose(sup_w[k], blk_w[k]);
                    blk_r[k] = blk_w[k] = MAP_ID;
                }
                sup_p[info] += blk_p[info];
                blk_p[info] = 0.0f;
            }
            open += 1;
            if (open % RP_FOLD_GROUP == 0 || open == n_chunks) { /* close the group */
                for (uint32_t info = 0; info < NI; ++info) {
                    for (uint32_t a = 0; a < A; ++a) {
                        size_t k = (size_t)info * A + a;
                        tot_r[k] = map_compose(tot_r[k], sup_r[k]);
                        tot_w[k] = map_compose(tot_w[k], sup_w[k]);
                        sup_r[k] = sup_w[k] = MAP_ID;
                    }
                    tot_p[info] += sup_p[info];
                    sup_p[info] = 0.0f;
                }
            }
        }
        if (i == h->ndec) break;
        const ora_decision* d = &h->dec[i];
        for (uint32_t a = 0; a < d->n_actions; ++a) {
            size_t k = (size_t)d->info * A + a;
            if (d->expanded >> a & 1u) map_touch(&blk_r[k], dr, d->regret[a], floor_r);
            map_touch(&blk_w[k], dw, composed_wdelta(h, d->policy[a]), RP_EPSILON);
        }
        blk_p[d->info] += d->payoff;
        tot_pn[d->info] += 1;
    }
    for (size_t c = 0; c < cells; ++c) {
        cell[c].ra = tot_r[c].a; cell[c].rb = tot_r[c].b; cell[c].rm = tot_r[c].m; cell[c].rn = tot_r[c].n;
        cell[c].wa = tot_w[c].a; cell[c].wb = tot_w[c].b; cell[c].wm = tot_w[c].m; cell[c].wn = tot_w[c].n;
    }
    for (uint32_t info = 0; info < NI; ++info) {
        sums[info].count = tot_pn[info];
        sums[info].psum = tot_p[info];
    }
    free(blk_r); free(blk_w); free(sup_r); free(sup_w); free(tot_r); free(tot_w);
    free(blk_p); free(sup_p); free(tot_p); free(tot_pn);
    return 0;
}

/* rp_mccfr_step_apply: fold `world` blobs (back to back) in rank order, then epoch += 1 */
ORA_API void ora_mccfr_step_apply(ora_mccfr* h, const void* gathered, uint32_t world) {
    uint32_t A = h->g.max_actions;
    size_t cells = (size_t)h->g.n_infos * A;
    size_t stride = ora_mccfr_summary_bytes(h);
    for (uint32_t r = 0; r < world; ++r) {
        const unsigned char* b = (const unsigned char*)gathered + (size_t)r * stride;
        const ora_cell* cell = (const ora_cell*)b;
        const ora_isum* sums = (const ora_isum*)(b + cells * sizeof(ora_cell));
        for (uint32_t info = 0; info < h->g.n_infos; ++info) {
            for (uint32_t a = 0; a < h->info_actions[info]; ++a) {
                size_t k = (size_t)info * A + a;
                if (cell[k].rn) h->regret[k] = rp_maxf(cell[k].ra * h->regret[k] + cell[k].rb, cell[k].rm);
                if (cell[k].wn) h->weight[k] = rp_maxf(cell[k].wa * h->weight[k] + cell[k].wb, cell[k].wm);
                if (sums[info].count) {
                    uint32_t n2 = h->visits[k] + sums[info].count;
                    h->payoff[k] = h->payoff[k] + (sums[info].psum - (float)sums[info].count * h->payoff[k]) / (float)n2;
                    h->visits[k] = n2;
                }
            }
        }
    }
    h->epoch += 1;
}

/* single-process model of a `world`-rank step: every rank traverses against the SAME start-of-epoch table
 * (Solver::batch is pure w.r.t. the profile, solver.rs:225), then the maps are folded in rank order */
ORA_API int ora_mccfr_step_world(ora_mccfr* h, uint32_t world) {
    size_t stride = ora_mccfr_summary_bytes(h);
    unsigned char* all = (unsigned char*)malloc(stride * world);
    for (uint32_t r = 0; r < world; ++r) {
        if (ora_mccfr_step_local(h, r, all + (size_t)r * stride)) {
            free(all);
            return -1;
        }
    }
    ora_mccfr_step_apply(h, all, world);
    free(all);
    return 0;
}

/* ---- the periodic exchange (rp_mccfr_window_local / window_apply): `window` local steps per all-gather ------------- */
/* acc <- step o acc per cell (maps composed in step order), counts and payoff sums added; first != 0 starts a window */
ORA_API void ora_mccfr_window_accumulate(const ora_mccfr* h, void* acc, const void* step, int first) {
    uint32_t A = h->g.max_actions, NI = h->g.n_infos;
    size_t cells = (size_t)NI * A;
    ora_cell* ac = (ora_cell*)acc;
    const ora_cell* sc = (const ora_cell*)step;
    ora_isum* as = (ora_isum*)((unsigned char*)acc + cells * sizeof(ora_cell));
    const ora_isum* ss = (const ora_isum*)((const unsigned char*)step + cells * sizeof(ora_cell));
    if (first) {
        memcpy(acc, step, ora_mccfr_summary_bytes(h));
        return;
    }
    for (size_t k = 0; k < cells; ++k) {
        ora_map ar = {ac[k].ra, ac[k].rb, ac[k].rm, ac[k].rn}, sr = {sc[k].ra, sc[k].rb, sc[k].rm, sc[k].rn};
        ora_map aw = {ac[k].wa, ac[k].wb, ac[k].wm, ac[k].wn}, sw = {sc[k].wa, sc[k].wb, sc[k].wm, sc[k].wn};
        ora_map r = map_compose(ar, sr), w = map_compose(aw, sw);
        ac[k].ra = r.a; ac[k].rb = r.b; ac[k].rm = r.m; ac[k].rn = r.n;
        ac[k].wa = w.a; ac[k].wb = w.b; ac[k].wm = w.m; ac[k].wn = w.n;
    }
    for (uint32_t i = 0; i < NI; ++i) {
        as[i].count += ss[i].count;
        as[i].psum = as[i].psum + ss[i].psum;
    }
}
/* one local step of a window on rank `rank`: maps of this step folded into `window`, epoch += 1, table untouched */
ORA_API int ora_mccfr_window_local(ora_mccfr* h, uint32_t rank, void* window, int first) {
    void* step = malloc(ora_mccfr_summary_bytes(h));
    int rc = ora_mccfr_step_local(h, rank, step);
    if (!rc) {
        ora_mccfr_window_accumulate(h, window, step, first);
        h->epoch += 1;
    }
    free(step);
    return rc;
}
/* fold `world` window summaries in rank order; the epoch was advanced by the local steps */
ORA_API void ora_mccfr_window_apply(ora_mccfr* h, const void* gathered, uint32_t world) {
    uint64_t e = h->epoch;
    ora_mccfr_step_apply(h, gathered, world);
    h->epoch = e;
}
/* single-process model of one window on `world` ranks: every rank runs its `window` local steps from the same
 * start-of-window table and epoch, then the summaries are applied in rank order */
ORA_API int ora_mccfr_window_world(ora_mccfr* h, uint32_t world, uint32_t window) {
    size_t stride = ora_mccfr_summary_bytes(h);
    unsigned char* all = (unsigned char*)malloc(stride * world);
    uint64_t e0 = h->epoch;
    for (uint32_t r = 0; r < world; ++r) {
        h->epoch = e0;
        for (uint32_t s = 0; s < window; ++s)
            if (ora_mccfr_window_local(h, r, all + (size_t)r * stride, s == 0)) {
                free(all);
                h->epoch = e0;
                return -1;
            }
    }
    ora_mccfr_window_apply(h, all, world);
    free(all);
    return 0;
}

ORA_API uint64_t ora_mccfr_epoch(const ora_mccfr* h) { return h->epoch; }
ORA_API void ora_mccfr_counters(const ora_mccfr* h, uint64_t* nodes, uint64_t* infos) {
    if (nodes) *nodes = h->nodes;
    if (infos) *infos = h->infos;
}
ORA_API void ora_mccfr_set_batch(ora_mccfr* h, uint32_t batch) { h->batch = batch ? batch : 1; }

ORA_API void ora_mccfr_export(const ora_mccfr* h, rp_encounter* rows) {
    size_t cells = (size_t)h->g.n_infos * h->g.max_actions;
    for (size_t c = 0; c < cells; ++c) {
        rows[c].weight = h->weight[c];
        rows[c].regret = h->regret[c];
        rows[c].payoff = h->payoff[c];
        rows[c].visits = h->visits[c];
    }
}
ORA_API void ora_mccfr_import(ora_mccfr* h, const rp_encounter* rows, uint64_t epoch) {
    size_t cells = (size_t)h->g.n_infos * h->g.max_actions;
    for (size_t c = 0; c < cells; ++c) {
        h->weight[c] = rows[c].weight;
        h->regret[c] = rows[c].regret;
        h->payoff[c] = rows[c].payoff;
        h->visits[c] = rows[c].visits;
    }
    h->epoch = epoch;
}

/* RefProf::{iterated,averaged}_distribution, CfrFlow::sampling_distribution */
ORA_API void ora_mccfr_policy(const ora_mccfr* h, uint32_t info, int kind, float* out) {
    uint32_t n = h->info_actions[info];
    if (kind == RP_DIST_ITERATED) {
        float denom = 0.0f;
        for (uint32_t a = 0; a < n; ++a) denom += p_regret(h, info, a);
        for (uint32_t a = 0; a < n; ++a) out[a] = p_regret(h, info, a) / denom;
    } else if (kind == RP_DIST_AVERAGED) {
        float sum = 0.0f;
        for (uint32_t a = 0; a < n; ++a) sum += p_weight(h, info, a);
        for (uint32_t a = 0; a < n; ++a) out[a] = p_weight(h, info, a) / sum;
    } else {
        float denom = weight_denom(h, info);
        float z = sampling_z(h, info, denom);
        for (uint32_t a = 0; a < n; ++a) out[a] = sampling_weight(h, info, a, denom) / z;
    }
}

/* RefProf::sum_regret (book.rs:124-131); HashMap order in the reference, (info, edge) order here */
ORA_API float ora_mccfr_sum_regret(const ora_mccfr* h) {
    float s = 0.0f;
    for (uint32_t info = 0; info < h->g.n_infos; ++info)
        for (uint32_t a = 0; a < h->info_actions[info]; ++a)
            s += rp_maxf(h->regret[info * h->g.max_actions + a], 0.0f);
    uint64_t e = h->epoch > 1 ? h->epoch : 1;
    return s / (float)e;
}

/* ------------------------------------------------------------------ exploitability (nash.rs) */
/* CfrNash::subgamed_payoff (nash.rs:103-139); br == NULL -> average strategy everywhere */
static float subgamed_payoff(const ora_mccfr* h, int32_t node, uint32_t hero, const int32_t* br) {
    const ora_node* nd = &h->tree.nodes[node];
    const rp_state* st = &h->states[nd->state];
    int n = node_width(nd);
    if (n == 0) return terminal_value(h, nd->state, hero);
    if (st->turn == RP_TURN_CHANCE) {
        float s = 0.0f;
        for (uint32_t k = 0; k < st->n_children; ++k)
            if (nd->kids[k] >= 0) s += subgamed_payoff(h, nd->kids[k], hero, br);
        return s / (float)n;
    }
    if (st->turn == hero && br) return subgamed_payoff(h, nd->kids[br[st->info]], hero, br);
    float s = 0.0f;
    for (uint32_t k = 0; k < st->n_children; ++k)
        if (nd->kids[k] >= 0) s += averaged_policy(h, st->info, k) * subgamed_payoff(h, nd->kids[k], hero, br);
    return s;
}
/* CfrNash::external_reach (nash.rs:146-151) */
static float external_reach(const ora_mccfr* h, int32_t node, uint32_t hero) {
    float p = 1.0f;
    const ora_node* nd = &h->tree.nodes[node];
    while (nd->parent >= 0) {
        const ora_node* par = &h->tree.nodes[nd->parent];
        const rp_state* ps = &h->states[par->state];
        if (ps->turn != RP_TURN_CHANCE && ps->turn != hero) p = p * averaged_policy(h, ps->info, (uint32_t)nd->edge);
        nd = par;
    }
    return p;
}
/* Solver::exploitability (solver.rs:327-337) -> CfrNash::exploitability (nash.rs:31-38) with
 * optimal_response_payoff / optimal_cfactual_{payoff,choice} (nash.rs:155-194) */
ORA_API float ora_mccfr_exploitability(ora_mccfr* h) {
    build_tree(h, 0, h->g.exploit_root, 0xffffffffu, 1);
    const ora_tree* t = &h->tree;
    int32_t* br = (int32_t*)malloc(sizeof(int32_t) * h->g.n_infos);
    float* cfv = (float*)malloc(sizeof(float) * (size_t)h->g.n_infos * ORA_MAXA);
    float total = 0.0f;
    for (uint32_t hero = 0; hero < h->g.n_players; ++hero) {
        memset(cfv, 0, sizeof(float) * (size_t)h->g.n_infos * ORA_MAXA);
        /* optimal_cfactual_payoff: sum over the span in ascending node index (nash.rs:170-181) */
        for (uint32_t i = 0; i < t->n; ++i) {
            const ora_node* nd = &t->nodes[i];
            const rp_state* st = &h->states[nd->state];
            if (st->turn != hero || node_width(nd) == 0) continue;
            for (uint32_t a = 0; a < st->n_children; ++a) {
                if (nd->kids[a] < 0) continue;
                int32_t c = nd->kids[a];
                cfv[st->info * ORA_MAXA + a] += external_reach(h, c, hero) * subgamed_payoff(h, c, hero, NULL);
            }
        }
        /* optimal_cfactual_choice: max_by keeps the last maximum (nash.rs:184-193) */
        for (uint32_t info = 0; info < h->g.n_infos; ++info) {
            br[info] = 0;
            if (h->info_player[info] != hero) continue;
            float best = cfv[info * ORA_MAXA];
            for (uint32_t a = 1; a < h->info_actions[info]; ++a) {
                if (cfv[info * ORA_MAXA + a] >= best) {
                    best = cfv[info * ORA_MAXA + a];
                    br[info] = (int32_t)a;
                }
            }
        }
        total += subgamed_payoff(h, 0, hero, br);
    }
    free(br);
    free(cfv);
    return total / (float)h->g.n_players;
}

/* host evaluation of the arithmetic contract, same layout as rp_math_selftest (include/rp_mi355x.h) */
ORA_API void ora_math_selftest(uint64_t n, const float* x, const float* y, float* out) {
    for (uint64_t i = 0; i < n; ++i) {
        float a = x[i], b = y[i];
        out[0 * n + i] = rp_expf(a);
        out[1 * n + i] = rp_logf(rp_absf(a));
        out[2 * n + i] = a / b;
        out[3 * n + i] = sqrtf(rp_absf(a));
        out[4 * n + i] = fmaf(a, b, a);
        out[5 * n + i] = (float)rp_f2u(b);
    }
}

/* brute-force check of rp_div_by_recip against IEEE division (tests/test_oracle_mccfr.py):
 * random and adversarial (quotient next to a rounding midpoint) numerators over integer divisors */
static uint64_t g_div_unproven = 0;
ORA_API uint64_t ora_div_unproven(void) { return g_div_unproven; }
ORA_API uint64_t ora_div_by_recip_mismatches(uint64_t n, uint64_t seed) {
    uint64_t st = seed | 1ull, bad = 0;
    g_div_unproven = 0;
    for (uint64_t it = 0; it < n; ++it) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        uint64_t r1 = st;
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        uint64_t r2 = st;
        uint32_t k;
        switch (r1 & 7) {
            case 0: case 1: case 2: k = 1u + (uint32_t)((r1 >> 8) % 16777216u); break;
            case 3: k = 16777215u - (uint32_t)((r1 >> 8) % 64u); break;
            case 4: k = (1u << (1 + ((r1 >> 8) % 24))) - 1u; break;
            case 5: k = 1u + (uint32_t)((r1 >> 8) % 4096u); break;
            default: k = (uint32_t)(r1 >> 20) | 1u; break;
        }
        float b = (float)k, a;
        if (r2 & 1) { /* adversarial: a ~= b * (odd 25-bit integer) * 2^-j */
            uint32_t m = (1u << 24) | (uint32_t)((r2 >> 8) & 0xffffffu) | 1u;
            a = (float)((double)b * ldexp((double)m, -(int)((r2 >> 40) % 60)));
            if ((r2 >> 60) & 1) a = rp_u2f(rp_f2u(a) + (((r2 >> 61) & 1) ? 1u : 0xffffffffu));
        } else {
            int e = (int)((r2 >> 40) % 60) - 50;
            a = rp_u2f(((uint32_t)(e + 127) << 23) | (uint32_t)((r2 >> 8) & 0x7fffffu));
        }
        if ((r2 >> 62) & 1) a = -a;
        if (!rp_div_by_recip_ok(a)) continue;
        if (rp_f2u(rp_div_by_recip(a, b, 1.0f / b)) != rp_f2u(a / b)) bad += 1;
        if (rp_f2u(rp_div_by_recip64(a, 1.0 / (double)b)) != rp_f2u(a / b)) bad += 1;
        int proven = 0;
        float q1 = rp_div_by_recip1(a, b, 1.0f / b, &proven);
        if (proven && rp_f2u(q1) != rp_f2u(a / b)) bad += 1; /* a proven quotient must be THE quotient */
        if (!proven) g_div_unproven += 1;
    }
    return bad;
}

/* ======================================================================================================
 * Sparse profile (include/rp_mi355x.h rp_profile_*): the update half of Solver::step (solver.rs:96-105,143-192)
 * on a table addressed by row index — the NLHE-scale shape, where the producer of the
 * Decisions is not the built-in traversal.  `apply` is the reference's loop verbatim (batch order, same epoch for
 * the whole batch, epoch += 1 afterwards); `summarize` / `fold` restate the composed exchange semantics with
 * blocks of RP_SPARSE_BLOCK consecutive touches of a row.
 * ====================================================================================================== */
ORA_API ora_mccfr* ora_profile_create(uint64_t n_rows, uint32_t max_actions, int R, int W, const rp_hyper* hp,
                                      const float* default_regret) {
    ora_mccfr* h = (ora_mccfr*)calloc(1, sizeof(ora_mccfr));
    h->g.n_infos = (uint32_t)n_rows;
    h->g.max_actions = max_actions;
    h->R = R;
    h->W = W;
    h->hp = *hp; /* the caller passes the schedule constants (tests/oracle.py default_hyper) */
    size_t cells = (size_t)n_rows * max_actions;
    h->regret = (float*)calloc(cells, 4);
    h->weight = (float*)calloc(cells, 4);
    h->payoff = (float*)calloc(cells, 4);
    h->visits = (uint32_t*)calloc(cells, 4);
    if (default_regret)
        for (size_t r = 0; r < n_rows; ++r)
            for (uint32_t a = 0; a < max_actions; ++a) h->regret[r * max_actions + a] = default_regret[a];
    return h;
}
ORA_API void ora_profile_destroy(ora_mccfr* h) {
    if (!h) return;
    free(h->regret); free(h->weight); free(h->payoff); free(h->visits);
    free(h);
}
static ora_decision sparse_decision(const ora_mccfr* h, uint64_t i, const uint32_t* row, const uint8_t* nact,
                                    const uint16_t* expanded, const float* regret, const float* policy,
                                    const float* payoff) {
    ora_decision d;
    uint32_t A = h->g.max_actions;
    memset(&d, 0, sizeof(d));
    d.info = row[i];
    d.n_actions = nact[i];
    d.expanded = expanded[i];
    for (uint32_t a = 0; a < d.n_actions; ++a) {
        d.regret[a] = regret[i * A + a];
        d.policy[a] = policy[i * A + a];
    }
    d.payoff = payoff[i];
    return d;
}
ORA_API void ora_profile_apply(ora_mccfr* h, uint64_t n, const uint32_t* row, const uint8_t* nact,
                               const uint16_t* expanded, const float* regret, const float* policy, const float* payoff) {
    for (uint64_t i = 0; i < n; ++i) {
        ora_decision d = sparse_decision(h, i, row, nact, expanded, regret, policy, payoff);
        apply_decision(h, &d);
    }
    h->infos += n;
    h->epoch += 1;
}
ORA_API void ora_profile_get(const ora_mccfr* h, uint32_t row, rp_encounter* out) {
    uint32_t A = h->g.max_actions;
    for (uint32_t a = 0; a < A; ++a) {
        size_t k = (size_t)row * A + a;
        out[a].weight = h->weight[k];
        out[a].regret = h->regret[k];
        out[a].payoff = h->payoff[k];
        out[a].visits = h->visits[k];
    }
}
ORA_API void ora_profile_set_epoch(ora_mccfr* h, uint64_t e) { h->epoch = e; }
/* overwrite one row (rp_oracle_nlmc.c: a freshly met infoset starts at its edge-wise default regrets; resynchronisation) */
void ora_profile_set_row(ora_mccfr* h, uint32_t row, const rp_encounter* in) {
    uint32_t A = h->g.max_actions;
    for (uint32_t a = 0; a < A; ++a) {
        size_t k = (size_t)row * A + a;
        h->weight[k] = in[a].weight;
        h->regret[k] = in[a].regret;
        h->payoff[k] = in[a].payoff;
        h->visits[k] = in[a].visits;
    }
}

/* summary entry: [row u32][count u32][psum f32][n_actions u32][regret maps A x {a,b,m,n}][weight maps A x {a,b,m,n}] */
ORA_API size_t ora_profile_entry_bytes(const ora_mccfr* h) { return 16 + (size_t)2 * h->g.max_actions * sizeof(ora_map); }

typedef struct ora_touch_ref {
    uint32_t row;
    uint64_t idx;
} ora_touch_ref;
static int touch_cmp(const void* x, const void* y) { /* stable: row, then batch position */
    const ora_touch_ref* a = (const ora_touch_ref*)x;
    const ora_touch_ref* b = (const ora_touch_ref*)y;
    if (a->row != b->row) return a->row < b->row ? -1 : 1;
    return a->idx < b->idx ? -1 : (a->idx > b->idx ? 1 : 0);
}
/* one rank's batch -> entries sorted by row; returns the entry count, or -1 for a sign-dependent schedule */
ORA_API int64_t ora_profile_summarize(ora_mccfr* h, uint64_t n, const uint32_t* row, const uint8_t* nact,
                                      const uint16_t* expanded, const float* regret, const float* policy,
                                      const float* payoff, void* blob) {
    float dr, dw;
    if (composed_discount(h, &dr, &dw)) return -1;
    uint32_t A = h->g.max_actions;
    float floor_r = regret_floor(h);
    size_t eb = ora_profile_entry_bytes(h);
    ora_touch_ref* ord = (ora_touch_ref*)malloc((n ? n : 1) * sizeof(ora_touch_ref));
    for (uint64_t i = 0; i < n; ++i) { ord[i].row = row[i]; ord[i].idx = i; }
    qsort(ord, n, sizeof(ora_touch_ref), touch_cmp);
    int64_t ne = 0;
    ora_map blk_r[ORA_MAXA], blk_w[ORA_MAXA], sup_r[ORA_MAXA], sup_w[ORA_MAXA], tot_r[ORA_MAXA], tot_w[ORA_MAXA];
    for (uint64_t s = 0; s < n;) {
        uint64_t e = s;
        while (e < n && ord[e].row == ord[s].row) ++e;
        for (uint32_t a = 0; a < A; ++a) tot_r[a] = tot_w[a] = sup_r[a] = sup_w[a] = MAP_ID;
        float tot_p = 0.0f, sup_p = 0.0f;
        uint32_t na = 0, sup_nb = 0;
        for (uint64_t b0 = s; b0 < e; b0 += RP_SPARSE_BLOCK) {
            uint64_t b1 = b0 + RP_SPARSE_BLOCK < e ? b0 + RP_SPARSE_BLOCK : e;
            for (uint32_t a = 0; a < A; ++a) blk_r[a] = blk_w[a] = MAP_ID;
            float blk_p = 0.0f;
            for (uint64_t t = b0; t < b1; ++t) {
                ora_decision d = sparse_decision(h, ord[t].idx, row, nact, expanded, regret, policy, payoff);
                na = d.n_actions;
                for (uint32_t a = 0; a < d.n_actions; ++a) {
                    if (d.expanded >> a & 1u) map_touch(&blk_r[a], dr, d.regret[a], floor_r);
                    map_touch(&blk_w[a], dw, composed_wdelta(h, d.policy[a]), RP_EPSILON);
                }
                blk_p += d.payoff;
            }
            /* blocks fold sequentially into a group of RP_FOLD_GROUP blocks, groups into the total */
            for (uint32_t a = 0; a < A; ++a) {
                sup_r[a] = map_compose(sup_r[a], blk_r[a]);
                sup_w[a] = map_compose(sup_w[a], blk_w[a]);
            }
            sup_p += blk_p;
            sup_nb += 1;
            if (sup_nb == RP_FOLD_GROUP || b1 == e) {
                for (uint32_t a = 0; a < A; ++a) {
                    tot_r[a] = map_compose(tot_r[a], sup_r[a]);
                    tot_w[a] = map_compose(tot_w[a], sup_w[a]);
                    sup_r[a] = sup_w[a] = MAP_ID;
                }
                tot_p += sup_p;
                sup_p = 0.0f;
                sup_nb = 0;
            }
        }
        unsigned char* ent = (unsigned char*)blob + (size_t)ne * eb;
        uint32_t hdr[4] = {ord[s].row, (uint32_t)(e - s), rp_f2u(tot_p), na};
        memcpy(ent, hdr, 16);
        memcpy(ent + 16, tot_r, A * sizeof(ora_map));
        memcpy(ent + 16 + A * sizeof(ora_map), tot_w, A * sizeof(ora_map));
        ne += 1;
        s = e;
    }
    free(ord);
    h->infos += n;
    return ne;
}
/* fold entries (any number of ranks' lists back to back, rank-major) into the table in the given order — rows are
 * independent, so this equals the device's stable sort by row followed by a per-row fold in rank order — epoch += 1 */
ORA_API void ora_profile_fold(ora_mccfr* h, const void* blob, uint64_t n_entries) {
    uint32_t A = h->g.max_actions;
    size_t eb = ora_profile_entry_bytes(h);
    for (uint64_t i = 0; i < n_entries; ++i) {
        const unsigned char* ent = (const unsigned char*)blob + i * eb;
        uint32_t hdr[4];
        memcpy(hdr, ent, 16);
        const ora_map* mr = (const ora_map*)(ent + 16);
        const ora_map* mw = mr + A;
        uint32_t r = hdr[0], count = hdr[1], na = hdr[3];
        float psum = rp_u2f(hdr[2]);
        for (uint32_t a = 0; a < na; ++a) {
            size_t k = (size_t)r * A + a;
            if (mr[a].n) h->regret[k] = rp_maxf(mr[a].a * h->regret[k] + mr[a].b, mr[a].m);
            if (mw[a].n) h->weight[k] = rp_maxf(mw[a].a * h->weight[k] + mw[a].b, mw[a].m);
            if (count) {
                uint32_t n2 = h->visits[k] + count;
                h->payoff[k] = h->payoff[k] + (psum - (float)count * h->payoff[k]) / (float)n2;
                h->visits[k] = n2;
            }
        }
    }
    h->epoch += 1;
}

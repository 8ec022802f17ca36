/* rp_oracle_nlmc.c — CPU oracle of the NLHE blueprint trainer's hot path: external-sampling MCCFR over the no-limit
 * hold'em game with the abstraction encoder (BASELINE configs[3]).            TEST INFRASTRUCTURE ONLY.
 *
 * What it follows (paths under /root/reference/crates, cited per function):
 *   nlhe/src/solver.rs:11          mccfr!(Nlhe, NlheEncoder, NlheTurn, NlheEdge, NlheGame, NlheInfo, 128)
 *   nlhe/src/encoder.rs:30-68      NlheEncoder::{abstraction, root, info, resume}
 *   nlhe/src/info.rs:72-103,145-160  NlheInfo = (subgame Path, Abstraction, choices Path)
 *   nlhe/src/game.rs:33-65         NlheGame::{apply, payoff}
 *   nlhe/src/edge.rs:40-47, kicker/src/edge.rs:61-72, bias.rs:47-70   CfrEdge::default_regret = the warm-start bias
 *   mccfr/src/solver/{solver.rs:96-105,225-275, builder.rs:74-161}, strategy/flow.rs, sample/external.rs: the generic loop
 *   (restated for table games in rp_oracle_mccfr.c; here over a game that is GENERATED, not tabulated).
 *
 * The rules (GameN, action abstraction, showdown) are rp_oracle_nlhe.c's, pinned to the reference's own unit tests; the
 * table update is rp_oracle_mccfr.c's row-addressed profile (ora_profile_apply = Solver::step's update loop).
 *
 * PARITY STATUS: the reference deals hole cards and boards from the unseeded thread RNG (not reproducible in the reference
 * itself): here they are rp_node_hash of (seed, epoch, tree, path).  The sampled opponent edge and Pluribus' coin come from
 * rng(node) = DefaultHasher(epoch, info, tree id) -> SmallRng (flow.rs:285-295): ora_nlmc_set_rng(RP_RNG_REFERENCE) draws them
 * through include/rp_refrng.h's restatement of that chain (pinned to the published vectors, tests/test_refrng.py); the default
 * keys the counter hash by the infoset the same way (the same infoset samples the same edge within a tree).
 * The encoder's isomorphism -> abstraction map is an input: a table (sorted canonical observations + buckets, the
 * artifact of the clustering pipeline) or, for tests, a hash of the canonical observation.
 */
#include <stdlib.h>
#include <string.h>

#include "../include/rp_math.h"
#include "../include/rp_mi355x.h"
#include "../include/rp_refrng.h"
#include "rp_oracle_nlhe.h"

#define NLMC_A 9u /* widest infoset: 5 raise sizes + shove + call + fold (pokerkit/src/lib.rs:130-133: A <= 9) */

/* rp_oracle_mccfr.c: the row-addressed profile */
typedef struct ora_mccfr ora_mccfr;
ora_mccfr* ora_profile_create(uint64_t n_rows, uint32_t max_actions, int R, int W, const rp_hyper* hp, const float* default_regret);
void ora_profile_destroy(ora_mccfr* h);
void ora_profile_apply(ora_mccfr* h, uint64_t n, const uint32_t* row, const uint8_t* nact, const uint16_t* expanded,
                       const float* regret, const float* policy, const float* payoff);
void ora_profile_get(const ora_mccfr* h, uint32_t row, rp_encounter* out);
void ora_profile_set_row(ora_mccfr* h, uint32_t row, const rp_encounter* in);
uint64_t ora_mccfr_epoch(const ora_mccfr* h);

typedef struct nl_key {
    uint64_t past, choices;
    uint32_t present;
} nl_key;

typedef struct nl_node {
    ora_game g;
    int32_t parent;
    uint8_t edge;      /* edge taken at the parent (E_*) */
    uint8_t slot;      /* its index in the parent's choices */
    uint8_t n_choices; /* 0: terminal */
    int8_t turn;       /* T_TERMINAL, T_CHANCE or the acting seat */
    uint64_t past;     /* choice edges since the last chance edge (NlheInfo::subgame) */
    uint64_t hkey;     /* identity of the node inside its tree: hash chain of the edges from the root */
    uint32_t row;      /* decision nodes: the infoset's table row */
    uint8_t choice[NLMC_A];
    int32_t kid[NLMC_A]; /* node index per choice slot, -1 = not expanded */
} nl_node;

typedef struct ora_nlmc {
    ora_mccfr* prof;
    nl_key* keys;
    uint8_t* used;
    uint32_t cap_log2, n_keys;
    int R, W, encoder;
    int rng; /* rp_rng_kind: which generator draws the opponent's edge and Pluribus' coin */
    int S; /* rp_sampling_kind: External (the mccfr! default), Prunable, Pluribus (nlhe/src/lib.rs:86-90 Flagship) */
    rp_hyper hp;
    uint64_t seed;
    uint32_t batch;
    uint64_t nodes, infos;
    uint32_t max_tree, max_tree_decisions;
    uint32_t rank, world; /* tree ids [rank * batch, (rank + 1) * batch) of a world * batch-tree epoch */
    const int64_t* tab_obs[4]; /* encoder table per street: sorted canonical observations */
    const uint16_t* tab_abs[4];
    uint64_t tab_n[4];
    /* scratch */
    nl_node* nd;
    uint32_t n, cap;
    /* the batch's Decisions (rp_decisions layout) */
    uint32_t* d_row;
    uint8_t* d_nact;
    uint16_t* d_exp;
    float *d_regret, *d_policy, *d_payoff;
    uint64_t* d_tree;
    uint64_t nd_dec, cap_dec;
} ora_nlmc;

/* kicker/src/edge.rs:61-72 with BiasHyperParams::default (bias.rs:47-70): folds 100, raise 10, shove 0, other 50 */
static float default_regret(uint8_t e) {
    switch (e) {
        case E_FOLD: return 100.0f;
        case E_SHOVE: return 0.0f;
        case E_CHECK: case E_CALL: return 50.0f;
    }
    return 10.0f; /* Open / Raise */
}

static uint64_t key_hash(const nl_key* k) {
    return rp_mix64(rp_mix64(k->past ^ 0x9e3779b97f4a7c15ull) ^ rp_mix64(k->choices + 0xd1342543de82ef95ull) ^
                    ((uint64_t)k->present * 0xaf251af3b0f025b5ull));
}
/* key -> row: open addressing, linear probing; a new key's row starts at the edge-wise default regrets, everything
 * else 0 (mccfr/src/strategy/book.rs:93-122: a missing Encounter reads as (0, default_regret, 0, 0)) */
static uint32_t row_of(ora_nlmc* h, const nl_key* k, const uint8_t* choice, int n_choices) {
    const uint32_t mask = (1u << h->cap_log2) - 1u;
    uint32_t s = (uint32_t)key_hash(k) & mask;
    for (;;) {
        if (!h->used[s]) {
            h->used[s] = 1;
            h->keys[s] = *k;
            h->n_keys += 1;
            rp_encounter e[NLMC_A];
            memset(e, 0, sizeof(e));
            for (int a = 0; a < n_choices; ++a) e[a].regret = default_regret(choice[a]);
            ora_profile_set_row(h->prof, s, e);
            return s;
        }
        if (h->keys[s].past == k->past && h->keys[s].choices == k->choices && h->keys[s].present == k->present) return s;
        s = (s + 1) & mask;
    }
}

/* NlheEncoder::abstraction (encoder.rs:30-36): the bucket of the canonical isomorphism of an observation */
static const uint32_t HASH_BUCKETS[4] = {169, 256, 256, 101};
ORA_API uint32_t ora_nlmc_hash_bucket(int street, int64_t canonical_obs) {
    return (uint32_t)(rp_mix64((uint64_t)canonical_obs ^ (0x51ed270b5ull * (uint64_t)(street + 1))) % HASH_BUCKETS[street]);
}
static uint32_t abstraction(const ora_nlmc* h, const ora_game* g, int seat) {
    uint64_t cp, cb;
    const int street = ora_nlhe_street(g);
    ora_isomorphism(g->seats[seat].cards, g->board, &cp, &cb);
    /* Abstraction = [8 bits street][8 bits index] (kicker/src/abstraction.rs:14-24,70-76): the `present SMALLINT` column */
    if (h->encoder == 0) return ((uint32_t)street << 8) | ora_nlmc_hash_bucket(street, ora_obs_to_i64(cp, cb));
    const int64_t at = ora_lookup_index(h->tab_obs[street], h->tab_n[street], cp, cb);
    return at < 0 ? 0xffffu : (((uint32_t)street << 8) | (h->tab_abs[street][at] & 0xffu)); /* the reference panics on a miss */
}

static uint64_t draw_cards(uint64_t deck, int k, const ora_nlmc* h, uint64_t epoch, uint64_t tree, uint64_t key) {
    uint64_t out = 0;
    for (int c = 0; c < k; ++c) {
        const uint32_t pick = rp_pick_uniform(rp_node_hash(h->seed, epoch, tree, key + (uint64_t)c), (uint32_t)__builtin_popcountll(deck));
        uint64_t d = deck;
        for (uint32_t s = 0; s < pick; ++s) d &= d - 1;
        const uint64_t card = d & (~d + 1);
        out |= card;
        deck &= ~card;
    }
    return out;
}
static uint64_t deck_of(const ora_game* g) {
    uint64_t gone = g->board;
    for (int i = 0; i < g->n; ++i) gone |= g->seats[i].cards;
    return 0x000FFFFFFFFFFFFFull & ~gone;
}

/* ------------------------------------------------------------------ profile reads (strategy/profile.rs:31-51, flow.rs:18-59) */
typedef struct nl_view {
    float regret[NLMC_A], weight[NLMC_A];
} nl_view;
static void view_row(const ora_nlmc* h, uint32_t row, int n, nl_view* v) {
    rp_encounter e[NLMC_A];
    ora_profile_get(h->prof, row, e);
    for (int a = 0; a < n; ++a) {
        v->regret[a] = rp_maxf(e[a].regret, RP_EPSILON);
        v->weight[a] = rp_maxf(e[a].weight, RP_EPSILON);
    }
}
static float regret_denom(const nl_view* v, int n) {
    float s = 0.0f;
    for (int a = 0; a < n; ++a) s += v->regret[a];
    return s;
}
static float weight_denom(const ora_nlmc* h, const nl_view* v, int n) {
    float s = 0.0f;
    for (int a = 0; a < n; ++a) s += v->weight[a];
    return s + h->hp.smoothing;
}
static float sampling_weight(const ora_nlmc* h, const nl_view* v, int a, float denom) {
    return rp_maxf((v->weight[a] / h->hp.temperature + h->hp.smoothing) / denom, h->hp.curiosity);
}
static float sampling_z(const ora_nlmc* h, const nl_view* v, int n, float denom) {
    float z = 0.0f;
    for (int a = 0; a < n; ++a) z += sampling_weight(h, v, a, denom);
    return z;
}

/* ------------------------------------------------------------------ tree (builder.rs:74-161) */
static uint32_t push_node(ora_nlmc* h) {
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 4096;
        h->nd = (nl_node*)realloc(h->nd, (size_t)h->cap * sizeof(nl_node));
    }
    nl_node* x = &h->nd[h->n];
    memset(x, 0, sizeof(*x));
    for (unsigned a = 0; a < NLMC_A; ++a) x->kid[a] = -1;
    return h->n++;
}
/* encoder.info + branches' own part: turn, choices, infoset row of a freshly grown node */
static void describe(ora_nlmc* h, uint32_t idx) {
    nl_node* x = &h->nd[idx];
    x->turn = (int8_t)ora_nlhe_turn(&x->g);
    if (x->turn == T_TERMINAL) return;
    if (x->turn == T_CHANCE) { /* legal() = [reveal()] -> choices = [Draw] (game.rs:253-260) */
        x->n_choices = 1;
        x->choice[0] = E_DRAW;
        return;
    }
    const uint64_t ch = ora_nlhe_choices(&x->g, ora_path_aggression(x->past));
    uint8_t e[MAX_PATH_EDGES + 1];
    x->n_choices = (uint8_t)ora_path_unpack(ch, e);
    for (int a = 0; a < x->n_choices; ++a) x->choice[a] = e[a];
    nl_key k = {x->past, ch, abstraction(h, &x->g, x->turn)};
    x->row = row_of(h, &k, x->choice, x->n_choices);
}
/* CfrFlow::rng (flow.rs:285-295) in reference-seed mode: DefaultHasher over t, NlheInfo { subgame: Path(u64), choices: Path(u64),
 * Abstraction(u16) } (nlhe/src/{info.rs:41-42, public.rs:19-23, secret.rs:10-11}), the tree's index */
static uint64_t ref_seed(uint64_t epoch, uint64_t tree, const nl_key* k) {
    rp_sip s;
    rp_defaulthasher_new(&s);
    rp_defaulthasher_write_u64(&s, epoch);
    rp_defaulthasher_write_u64(&s, k->past);
    rp_defaulthasher_write_u64(&s, k->choices);
    rp_defaulthasher_write_u16(&s, (uint16_t)k->present);
    rp_defaulthasher_write_u64(&s, tree);
    return rp_defaulthasher_finish(&s);
}
static float draw_coin(const ora_nlmc* h, uint64_t epoch, uint64_t tree, const nl_key* k) {
    if (h->rng == RP_RNG_REFERENCE) return rp_ref_draw_f32(ref_seed(epoch, tree, k)); /* rng.random::<f32>() */
    return rp_u01(rp_node_hash(h->seed, epoch, tree, key_hash(k)));
}
static float draw_weight(const ora_nlmc* h, uint64_t epoch, uint64_t tree, const nl_key* k, float total) {
    if (h->rng == RP_RNG_REFERENCE) return rp_ref_draw_weight(ref_seed(epoch, tree, k), total); /* WeightedIndex's Uniform(0, total) */
    return rp_u01(rp_node_hash(h->seed, epoch, tree, key_hash(k))) * total;
}
/* ExternalSampling::sample (sample/external.rs:17-64): the choice slots to expand */
static uint32_t sample_mask(const ora_nlmc* h, uint64_t epoch, uint64_t tree, const nl_node* x, int walker) {
    const uint32_t all = (1u << x->n_choices) - 1u;
    if (x->n_choices == 0) return 0;
    if (x->turn == T_CHANCE) return all; /* the chance node's single Draw edge; its cards are drawn in grow() */
    if (x->turn == walker) {
        /* PrunableSampling (sample/pruning.rs:44-66) / PluribusSampling (sample/pluribus.rs:72-101) at a walker node:
         * cum_regret is the RAW accumulated regret (profile.rs:26-30), not the clamped one regret matching reads */
        if (h->S == RP_SAMPLING_EXTERNAL) return all;
        const nl_key k = {h->keys[x->row].past, h->keys[x->row].choices, h->keys[x->row].present};
        if (h->S == RP_SAMPLING_PLURIBUS) {
            if (epoch < h->hp.prune_warmup) return all;
            /* profile.rng(node): one stream per (epoch, infoset, tree) (flow.rs:285-295), first draw random::<f32>() */
            if (draw_coin(h, epoch, tree, &k) < h->hp.prune_explore) return all;
        }
        rp_encounter e[NLMC_A];
        ora_profile_get(h->prof, x->row, e);
        uint32_t mask = 0;
        for (int a = 0; a < x->n_choices; ++a) {
            int keep = e[a].regret > h->hp.prune_threshold;
            if (!keep && h->S == RP_SAMPLING_PLURIBUS) { /* game.turn().is_terminal() of the BRANCH's game (pluribus.rs:96) */
                ora_game c = x->g;
                uint64_t none = 0;
                ora_nlhe_apply_edge(&c, x->choice[a], &none); /* a walker node's choices are never Draw */
                keep = ora_nlhe_turn(&c) == T_TERMINAL;
            }
            if (keep) mask |= 1u << a;
        }
        return mask ? mask : all; /* pruning.rs:64, pluribus.rs:99 */
    }
    nl_view v;
    view_row(h, x->row, x->n_choices, &v);
    const float denom = weight_denom(h, &v, x->n_choices), z = sampling_z(h, &v, x->n_choices, denom);
    float cum[NLMC_A], total = 0.0f;
    for (int a = 0; a < x->n_choices; ++a) {
        total += rp_maxf(sampling_weight(h, &v, a, denom) / z, RP_EPSILON);
        cum[a] = total;
    }
    const nl_key k = {h->keys[x->row].past, h->keys[x->row].choices, h->keys[x->row].present};
    const float u = draw_weight(h, epoch, tree, &k, total);
    uint32_t idx = 0;
    while (idx + 1 < x->n_choices && cum[idx] <= u) ++idx;
    return 1u << idx;
}

typedef struct nl_leaf {
    int32_t parent;
    uint8_t slot;
} nl_leaf;

/* Solver::tree (solver.rs:252-262): root = Game::root() with hole cards dealt (P0 on the button, game.rs:66-78) */
static void build_tree(ora_nlmc* h, uint64_t epoch, uint64_t tree, int walker) {
    h->n = 0;
    const int16_t stacks[2] = {200, 200};
    uint64_t holes[2], deck = 0x000FFFFFFFFFFFFFull;
    for (int i = 0; i < 2; ++i) {
        holes[i] = draw_cards(deck, 2, h, epoch, tree, 0xD0C0000000000000ull + 8u * (uint64_t)i);
        deck &= ~holes[i];
    }
    uint32_t r = push_node(h);
    ora_nlhe_from_start(&h->nd[r].g, 2, 0, stacks, holes);
    h->nd[r].parent = -1;
    h->nd[r].hkey = rp_mix64(0x726f6f74ull);
    describe(h, r);
    size_t cap = 1024, top = 0;
    nl_leaf* todo = (nl_leaf*)malloc(cap * sizeof(nl_leaf));
    uint32_t cur = r;
    for (;;) {
        const uint32_t mask = sample_mask(h, epoch, tree, &h->nd[cur], walker);
        for (uint32_t a = 0; a < h->nd[cur].n_choices; ++a) {
            if (!(mask >> a & 1u)) continue;
            if (top == cap) {
                cap *= 2;
                todo = (nl_leaf*)realloc(todo, cap * sizeof(nl_leaf));
            }
            todo[top].parent = (int32_t)cur;
            todo[top].slot = (uint8_t)a;
            ++top;
        }
        if (top == 0) break;
        const nl_leaf lf = todo[--top]; /* pop-last (builder.rs:143) */
        const uint32_t c = push_node(h); /* may move h->nd */
        nl_node* p = &h->nd[lf.parent];
        nl_node* x = &h->nd[c];
        x->g = p->g;
        x->parent = lf.parent;
        x->slot = lf.slot;
        x->edge = p->choice[lf.slot];
        x->hkey = rp_mix64(p->hkey ^ ((uint64_t)(x->edge + 1u) * 0x9fb21c651e98df25ull));
        uint64_t draw = 0;
        if (x->edge == E_DRAW) { /* NlheGame::apply(Draw): reveal() deals the street (nlhe/src/game.rs:33-53) */
            draw = draw_cards(deck_of(&p->g), ora_nlhe_street(&p->g) == 0 ? 3 : 1, h, epoch, tree, x->hkey);
            x->past = 0;
        } else {
            uint8_t e[MAX_PATH_EDGES + 2];
            int n = ora_path_unpack(p->past, e);
            e[n++] = x->edge;
            x->past = ora_path_pack(e, n);
        }
        ora_nlhe_apply_edge(&x->g, x->edge, &draw);
        p->kid[lf.slot] = (int32_t)c;
        describe(h, c);
        cur = c;
    }
    free(todo);
}

/* ------------------------------------------------------------------ counterfactual values (flow.rs:64-216) */
static int width(const nl_node* x) {
    int w = 0;
    for (unsigned a = 0; a < NLMC_A; ++a) w += x->kid[a] >= 0;
    return w;
}
static float recursed_value(const ora_nlmc* h, int walker, int32_t node, float rel, float smp) {
    const nl_node* x = &h->nd[node];
    if (width(x) == 0) { /* terminal_value = game.payoff(walker) (nash.rs:66-79, nlhe/src/game.rs:59-65) */
        float pay = 0.0f;
        ora_nlhe_payoff(&x->g, walker, &pay);
        return rel / smp * pay;
    }
    const int chance = x->turn == T_CHANCE, is_walker = x->turn == walker;
    nl_view v;
    float rd = 0.0f, denom = 0.0f, z = 0.0f;
    if (!chance) {
        view_row(h, x->row, x->n_choices, &v);
        rd = regret_denom(&v, x->n_choices);
        if (!is_walker) {
            denom = weight_denom(h, &v, x->n_choices);
            z = sampling_z(h, &v, x->n_choices, denom);
        }
    }
    float sum = 0.0f;
    for (int a = 0; a < x->n_choices; ++a) {
        if (x->kid[a] < 0) continue;
        const float r2 = rel * (chance ? 1.0f : v.regret[a] / rd);
        const float s2 = smp * ((!chance && !is_walker) ? sampling_weight(h, &v, a, denom) / z : 1.0f);
        sum += recursed_value(h, walker, x->kid[a], r2, s2);
    }
    return sum;
}
static float ancestor_reach(const ora_nlmc* h, int walker, int32_t node) {
    float cf = 1.0f, sm = 1.0f;
    const nl_node* x = &h->nd[node];
    while (x->parent >= 0) {
        const nl_node* p = &h->nd[x->parent];
        if (p->turn != T_CHANCE && p->turn != walker) {
            nl_view v;
            view_row(h, p->row, p->n_choices, &v);
            const float denom = weight_denom(h, &v, p->n_choices);
            cf = cf * (v.regret[x->slot] / regret_denom(&v, p->n_choices));
            sm = sm * (sampling_weight(h, &v, x->slot, denom) / sampling_z(h, &v, p->n_choices, denom));
        }
        x = p;
    }
    return cf / sm;
}

static uint64_t push_decision(ora_nlmc* h) {
    if (h->nd_dec == h->cap_dec) {
        h->cap_dec = h->cap_dec ? h->cap_dec * 2 : 4096;
        h->d_row = (uint32_t*)realloc(h->d_row, h->cap_dec * 4);
        h->d_nact = (uint8_t*)realloc(h->d_nact, h->cap_dec);
        h->d_exp = (uint16_t*)realloc(h->d_exp, h->cap_dec * 2);
        h->d_regret = (float*)realloc(h->d_regret, h->cap_dec * NLMC_A * 4);
        h->d_policy = (float*)realloc(h->d_policy, h->cap_dec * NLMC_A * 4);
        h->d_payoff = (float*)realloc(h->d_payoff, h->cap_dec * 4);
        h->d_tree = (uint64_t*)realloc(h->d_tree, h->cap_dec * 8);
    }
    const uint64_t i = h->nd_dec++;
    memset(h->d_regret + i * NLMC_A, 0, NLMC_A * 4);
    memset(h->d_policy + i * NLMC_A, 0, NLMC_A * 4);
    h->d_exp[i] = 0;
    return i;
}
/* Tree::partition + record_infosets + update_vector (tree.rs:88-98, solver.rs:263-305): walker infosets in the order of
 * their first node, span in ascending node index */
static void tree_decisions(ora_nlmc* h, uint64_t tree, int walker) {
    for (uint32_t i = 0; i < h->n; ++i) {
        const nl_node* x = &h->nd[i];
        if (x->turn != walker || width(x) == 0) continue;
        int head = 1;
        for (uint32_t j = 0; j < i && head; ++j) {
            const nl_node* o = &h->nd[j];
            if (o->turn >= 0 && width(o) > 0 && o->row == x->row) head = 0;
        }
        if (!head) continue;
        const uint64_t d = push_decision(h);
        h->d_row[d] = x->row;
        h->d_nact[d] = x->n_choices;
        h->d_tree[d] = tree;
        nl_view v;
        view_row(h, x->row, x->n_choices, &v);
        const float rd = regret_denom(&v, x->n_choices);
        for (int a = 0; a < x->n_choices; ++a) h->d_policy[d * NLMC_A + a] = v.regret[a] / rd;
        float payoff = 0.0f;
        for (uint32_t j = i; j < h->n; ++j) {
            const nl_node* root = &h->nd[j];
            if (root->turn != walker || width(root) == 0 || root->row != x->row) continue;
            const float reach = ancestor_reach(h, walker, (int32_t)j);
            float val[NLMC_A], ev = 0.0f;
            for (int a = 0; a < root->n_choices; ++a)
                if (root->kid[a] >= 0) val[a] = reach * recursed_value(h, walker, root->kid[a], 1.0f, 1.0f);
            for (int a = 0; a < root->n_choices; ++a)
                if (root->kid[a] >= 0) ev += v.regret[a] / rd * val[a];
            payoff += ev;
            for (int a = 0; a < root->n_choices; ++a)
                if (root->kid[a] >= 0) {
                    h->d_exp[d] |= (uint16_t)(1u << a);
                    h->d_regret[d * NLMC_A + a] += val[a] - ev;
                }
        }
        h->d_payoff[d] = payoff;
    }
}

/* ------------------------------------------------------------------ API */
ORA_API ora_nlmc* ora_nlmc_create(uint32_t cap_log2, int R, int W, const rp_hyper* hp, uint64_t seed, uint32_t batch) {
    ora_nlmc* h = (ora_nlmc*)calloc(1, sizeof(ora_nlmc));
    h->cap_log2 = cap_log2;
    h->R = R;
    h->W = W;
    h->hp = *hp;
    h->seed = seed;
    h->batch = batch ? batch : 128; /* nlhe/src/solver.rs:11 */
    h->world = 1;
    h->prof = ora_profile_create(1ull << cap_log2, NLMC_A, R, W, hp, NULL);
    h->keys = (nl_key*)calloc(1ull << cap_log2, sizeof(nl_key));
    h->used = (uint8_t*)calloc(1ull << cap_log2, 1);
    return h;
}
/* SamplingScheme of the solver type (the macro's default is ExternalSampling; Flagship uses PluribusSampling) */
ORA_API void ora_nlmc_set_sampling(ora_nlmc* h, int sampling) { h->S = sampling; }
ORA_API void ora_nlmc_set_rng(ora_nlmc* h, int kind) { h->rng = kind; }
ORA_API void ora_nlmc_destroy(ora_nlmc* h) {
    if (!h) return;
    ora_profile_destroy(h->prof);
    free(h->keys); free(h->used); free(h->nd);
    free(h->d_row); free(h->d_nact); free(h->d_exp); free(h->d_regret); free(h->d_policy); free(h->d_payoff); free(h->d_tree);
    free(h);
}
/* encoder table of one street: sorted canonical observations (i64, the Lookup's obs column) and their buckets; the arrays
 * stay the caller's.  Without tables the hash encoder is used. */
ORA_API void ora_nlmc_set_table(ora_nlmc* h, int street, const int64_t* obs, const uint16_t* abs_, uint64_t n) {
    h->tab_obs[street] = obs;
    h->tab_abs[street] = abs_;
    h->tab_n[street] = n;
    h->encoder = 1;
}
/* Solver::batch (solver.rs:225-250): the Decisions of the current epoch, tree-id major; no table change except that
 * infosets met for the first time get their (default) row */
static void run_batch(ora_nlmc* h) {
    const uint64_t epoch = ora_mccfr_epoch(h->prof);
    const int walker = (int)(epoch % 2); /* CfrSampling::walker (book.rs:142-144) */
    h->nd_dec = 0;
    for (uint32_t i = 0; i < h->batch; ++i) {
        const uint64_t t = (uint64_t)h->rank * h->batch + i;
        build_tree(h, epoch, t, walker);
        h->nodes += h->n;
        if (h->n > h->max_tree) h->max_tree = h->n;
        const uint64_t before = h->nd_dec;
        tree_decisions(h, t, walker);
        h->infos += h->nd_dec - before;
        if (h->nd_dec - before > h->max_tree_decisions) h->max_tree_decisions = (uint32_t)(h->nd_dec - before);
    }
}
ORA_API void ora_nlmc_step(ora_nlmc* h) { /* Solver::step (solver.rs:96-105) */
    run_batch(h);
    ora_profile_apply(h->prof, h->nd_dec, h->d_row, h->d_nact, h->d_exp, h->d_regret, h->d_policy, h->d_payoff);
}
/* the batch without the update (kernel debugging); counters are not advanced */
ORA_API uint64_t ora_nlmc_batch(ora_nlmc* h, const uint32_t** row, const uint8_t** nact, const uint16_t** expanded, const float** regret,
                                const float** policy, const float** payoff, const uint64_t** tree) {
    const uint64_t nodes = h->nodes, infos = h->infos;
    run_batch(h);
    h->nodes = nodes;
    h->infos = infos;
    *row = h->d_row; *nact = h->d_nact; *expanded = h->d_exp; *regret = h->d_regret; *policy = h->d_policy; *payoff = h->d_payoff;
    *tree = h->d_tree;
    return h->nd_dec;
}
/* ---- the multi-GPU exchange (rp_nlhe_set_shard / step_local / step_apply): trees sharded by rank, each rank's Decisions
 * reduced to per-infoset composed entries (rp_oracle_mccfr.c ora_profile_summarize), the entries exchanged BY KEY — every
 * rank's table assigns rows in its own insertion order — and folded in rank order (ora_profile_fold). */
int64_t ora_profile_summarize(ora_mccfr* h, uint64_t n, const uint32_t* row, const uint8_t* nact, const uint16_t* expanded,
                              const float* regret, const float* policy, const float* payoff, void* blob);
void ora_profile_fold(ora_mccfr* h, const void* blob, uint64_t n_entries);
size_t ora_profile_entry_bytes(const ora_mccfr* h);
ORA_API void ora_nlmc_set_shard(ora_nlmc* h, uint32_t rank, uint32_t world) {
    h->rank = rank;
    h->world = world ? world : 1;
}
ORA_API size_t ora_nlmc_entry_bytes(const ora_nlmc* h) { return ora_profile_entry_bytes(h->prof); }
/* this rank's batch -> entries (sorted by its rows) with the infoset key of each; returns the count (-1: schedule unsupported) */
ORA_API int64_t ora_nlmc_step_local(ora_nlmc* h, void* entries, uint64_t* past, uint32_t* present, uint64_t* choices) {
    run_batch(h);
    const int64_t ne = ora_profile_summarize(h->prof, h->nd_dec, h->d_row, h->d_nact, h->d_exp, h->d_regret, h->d_policy, h->d_payoff, entries);
    const size_t eb = ora_profile_entry_bytes(h->prof);
    for (int64_t i = 0; i < ne; ++i) {
        uint32_t row;
        memcpy(&row, (const unsigned char*)entries + (size_t)i * eb, 4);
        past[i] = h->keys[row].past;
        present[i] = h->keys[row].present;
        choices[i] = h->keys[row].choices;
    }
    return ne;
}
/* all ranks' entries back to back (rank-major) with their keys: rows rewritten to THIS table's, then folded; epoch += 1 */
ORA_API void ora_nlmc_step_apply(ora_nlmc* h, void* entries, const uint64_t* past, const uint32_t* present, const uint64_t* choices,
                                 uint64_t n) {
    const size_t eb = ora_profile_entry_bytes(h->prof);
    for (uint64_t i = 0; i < n; ++i) {
        nl_key k = {past[i], choices[i], present[i]};
        uint8_t e[MAX_PATH_EDGES + 1];
        const int m = ora_path_unpack(choices[i], e);
        const uint32_t row = row_of(h, &k, e, m);
        memcpy((unsigned char*)entries + (size_t)i * eb, &row, 4);
    }
    ora_profile_fold(h->prof, entries, n);
}
/* single-process model of a `world`-rank step: every rank traverses against the same start-of-epoch table */
ORA_API int ora_nlmc_step_world(ora_nlmc* h, uint32_t world) {
    const size_t eb = ora_profile_entry_bytes(h->prof);
    size_t cap = 1 << 16, n = 0;
    unsigned char* ent = (unsigned char*)malloc(cap * eb);
    uint64_t* kp = (uint64_t*)malloc(cap * 8);
    uint64_t* kc = (uint64_t*)malloc(cap * 8);
    uint32_t* kb = (uint32_t*)malloc(cap * 4);
    const uint32_t rank0 = h->rank, world0 = h->world;
    int rc = 0;
    for (uint32_t r = 0; r < world && !rc; ++r) {
        ora_nlmc_set_shard(h, r, world);
        run_batch(h);
        while (n + h->nd_dec > cap) {
            cap *= 2;
            ent = (unsigned char*)realloc(ent, cap * eb);
            kp = (uint64_t*)realloc(kp, cap * 8);
            kc = (uint64_t*)realloc(kc, cap * 8);
            kb = (uint32_t*)realloc(kb, cap * 4);
        }
        const int64_t ne = ora_profile_summarize(h->prof, h->nd_dec, h->d_row, h->d_nact, h->d_exp, h->d_regret, h->d_policy, h->d_payoff,
                                                 ent + n * eb);
        if (ne < 0) {
            rc = -1;
            break;
        }
        for (int64_t i = 0; i < ne; ++i) {
            uint32_t row;
            memcpy(&row, ent + (n + (size_t)i) * eb, 4);
            kp[n + i] = h->keys[row].past;
            kb[n + i] = h->keys[row].present;
            kc[n + i] = h->keys[row].choices;
        }
        n += (size_t)ne;
    }
    if (!rc) ora_nlmc_step_apply(h, ent, kp, kb, kc, n);
    ora_nlmc_set_shard(h, rank0, world0);
    free(ent); free(kp); free(kc); free(kb);
    return rc;
}
ORA_API uint64_t ora_nlmc_epoch(const ora_nlmc* h) { return ora_mccfr_epoch(h->prof); }
ORA_API void ora_nlmc_counters(const ora_nlmc* h, uint64_t* nodes, uint64_t* infos, uint64_t* keys) {
    if (nodes) *nodes = h->nodes;
    if (infos) *infos = h->infos;
    if (keys) *keys = h->n_keys;
}
ORA_API uint32_t ora_nlmc_last_tree_nodes(const ora_nlmc* h) { return h->n; }
ORA_API void ora_nlmc_max_tree(const ora_nlmc* h, uint32_t* nodes, uint32_t* decisions) {
    *nodes = h->max_tree;
    *decisions = h->max_tree_decisions;
}
/* the infoset behind a row (for comparisons across implementations whose rows differ) */
ORA_API int ora_nlmc_row_key(const ora_nlmc* h, uint32_t row, uint64_t* past, uint32_t* present, uint64_t* choices) {
    if (row >= (1u << h->cap_log2) || !h->used[row]) return 1;
    *past = h->keys[row].past;
    *present = h->keys[row].present;
    *choices = h->keys[row].choices;
    return 0;
}
/* every infoset of the table: keys and Encounters, in slot order; returns the count (cap = capacity of the outputs) */
ORA_API uint64_t ora_nlmc_export(const ora_nlmc* h, uint64_t cap, uint64_t* past, uint32_t* present, uint64_t* choices, rp_encounter* enc) {
    uint64_t n = 0;
    for (uint32_t s = 0; s < (1u << h->cap_log2); ++s) {
        if (!h->used[s]) continue;
        if (n < cap) {
            past[n] = h->keys[s].past;
            present[n] = h->keys[s].present;
            choices[n] = h->keys[s].choices;
            ora_profile_get(h->prof, s, enc + n * NLMC_A);
        }
        n += 1;
    }
    return n;
}
/* load Encounters by key (resynchronising another implementation's table): unknown keys are inserted */
ORA_API void ora_nlmc_import(ora_nlmc* h, uint64_t n, const uint64_t* past, const uint32_t* present, const uint64_t* choices,
                             const rp_encounter* enc) {
    for (uint64_t i = 0; i < n; ++i) {
        nl_key k = {past[i], choices[i], present[i]};
        uint8_t e[MAX_PATH_EDGES + 1];
        const int m = ora_path_unpack(choices[i], e);
        const uint32_t r = row_of(h, &k, e, m);
        ora_profile_set_row(h->prof, r, enc + i * NLMC_A);
    }
}
/* shape of the LAST tree built (sizing the device's level-synchronous expansion): nodes per depth into levels[0..cap),
 * kinds[4] = terminal, chance, walker, opponent node counts; returns the tree's depth (levels used) */
ORA_API uint32_t ora_nlmc_last_tree_shape(const ora_nlmc* h, int walker, uint32_t* levels, uint32_t cap, uint32_t* kinds) {
    uint32_t depth = 0;
    uint32_t* d = (uint32_t*)calloc(h->n ? h->n : 1, sizeof(uint32_t));
    for (uint32_t l = 0; l < cap; ++l) levels[l] = 0;
    for (int k = 0; k < 4; ++k) kinds[k] = 0;
    for (uint32_t i = 0; i < h->n; ++i) {
        const nl_node* x = &h->nd[i];
        d[i] = x->parent < 0 ? 0 : d[x->parent] + 1; /* a parent's index is below its children's */
        if (d[i] < cap) levels[d[i]] += 1;
        if (d[i] + 1 > depth) depth = d[i] + 1;
        kinds[x->turn == T_TERMINAL ? 0 : (x->turn == T_CHANCE ? 1 : (x->turn == walker ? 2 : 3))] += 1;
    }
    free(d);
    return depth;
}

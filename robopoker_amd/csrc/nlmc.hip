// nlmc.hip — external-sampling MCCFR over no-limit hold'em on the device: the producer of the blueprint trainer's
// Decisions (BASELINE configs[3]).  rp_nlhe_*.
//
// Reference path: mccfr!(Nlhe, NlheEncoder, NlheTurn, NlheEdge, NlheGame, NlheInfo, 128) (crates/nlhe/src/solver.rs:11) =
//   Solver::step (mccfr/src/solver/solver.rs:96-105): batch() -> per tree TreeBuilder (builder.rs:74-161) over NlheGame
//   (nlhe/src/game.rs:33-65) with NlheEncoder::info (encoder.rs:30-68, info.rs:145-160: key = (subgame Path, bucket,
//   choices Path)), ExternalSampling (sample/external.rs:17-64), CfrFlow::dfs per walker infoset (strategy/flow.rs:64-216);
//   then the sequential update (solver.rs:143-192) = rp_profile_apply on the row-addressed table (sparse.hip).
// Oracle: oracle/rp_oracle_nlmc.c (same RNG contract: rp_node_hash of (seed, epoch, tree, key)).
//
// MAPPING (first device version: correctness and the data path).  One LANE per sampled tree — the trees of a
// batch are independent and 10^2-10^3 nodes each; a lane runs the reference's pop-last DFS with its stack and node list in
// its own HBM scratch region.  Node order = the reference's creation order, which makes the bottom-up evaluation a single
// descending sweep (every node's index exceeds its parent's; descending index adds a node's children in choices() order)
// and Tree::partition's "first node of an infoset, span in ascending index" a single ascending sweep.
//   per node:  turn -> choices (nl_choices) -> NlheInfo key -> row (open addressing in HBM: keys beside the rows, a new
//              infoset's row starts at the edge-wise default regrets, kicker/src/edge.rs:61-72) -> regret matching / sampling
//              distribution from the row -> children (walker: all, opponent: one, chance: one Draw with hashed cards)
//   per tree:  D(node) = sum_children f(edge) D(child) with f = sigma (walker), sigma / q (opponent), 1 (chance), leaves =
//              payoff(walker); reach(node) = prod of sigma / q over opponent ancestors; per walker node cfv_a = reach * D(child_a),
//              ev = sum sigma_a cfv_a; regret_a += cfv_a - ev and payoff += ev summed over the nodes that share an infoset.
// Integer state (tree shapes, keys, rows' identities, expanded masks, counters) is the oracle's bit for bit; D is the
// factorised form of recursed_value (flow.rs:182-216 multiplies the reach products at the leaves), so regret vectors agree to
// f32 re-association (tests state the tolerance), like the composed update.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "../../include/rp_math.h"
#include "nlhe_engine.hpp"
#include "obs.hpp"
#include "rp_internal.h"

namespace rp {

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define NLMC_A 9u
typedef NlGameT<2, 2> G2;  // heads-up: two seats at compile time
enum : uint32_t { NK_TERMINAL = 0, NK_CHANCE = 1, NK_WALKER = 2, NK_OPP = 3 };
enum : uint32_t { NERR_NODES = 1u, NERR_STACK = 2u, NERR_WALKERS = 4u, NERR_DECISIONS = 8u, NERR_ILLEGAL = 16u, NERR_TABLE_FULL = 32u };

struct NlTable {  // NlheInfo -> row: open addressing, linear probing; slot index = row of the profile
    uint64_t* past;
    uint64_t* choices;
    uint32_t* present;
    uint32_t* state;  // 0 empty, 1 being written, 2 ready
    uint32_t mask;
    unsigned int* n_keys;
    float* rows;  // the profile's table: row r = rows + r * 4A: regret[A] weight[A] payoff[A] visits[A]
};

struct NlParams {
    uint64_t seed, epoch;
    uint32_t batch, walker;
    uint64_t tree_base;  // first tree id of this rank's shard (rank * batch)
    float temperature, smoothing, curiosity;
    int encoder;  // 0: hash of the canonical observation, 1: lookup tables
    const uint64_t* tkeys[4];
    const uint8_t* tabs[4];
    uint64_t tn[4];
    uint32_t ncap, scap, wcap, dcap;  // per-tree capacities: nodes, stack entries, walker nodes, Decisions
    uint32_t check_legal;             // evaluate Game::is_allowed on every applied action (RP_NLHE_CHECK_LEGAL=1)
};

// per-tree scratch, tree-major (a lane walks its own region sequentially)
struct NlScratch {
    uint32_t* meta;    // [batch][ncap]  parent (13) | slot (4) << 13 | kind (2) << 17 | n_choices (4) << 19
    float* fac;        // [batch][ncap]  factor of the edge into the node
    float* val;        // [batch][ncap]  D(node), later reach(node)
    uint32_t* aux;     // [batch][ncap]  walker nodes: ordinal among the tree's walker nodes
    uint32_t* wrow;    // [batch][wcap]
    uint32_t* wnode;   // [batch][wcap]
    float* kidd;       // [batch][wcap][A]  D of the walker node's children by choice slot
    uint32_t* stack;   // [batch][scap][12]
    // per-tree Decisions
    uint32_t* drow;    // [batch][dcap]
    uint32_t* dmeta;   // [batch][dcap]  n_actions | expanded << 8
    float* dreg;       // [batch][dcap][A]
    float* dpol;       // [batch][dcap][A]
    float* dpay;       // [batch][dcap]
    uint32_t* dmap;    // [batch][2 * dcap]  row -> decision index + 1
    uint32_t* dcount;  // [batch]
    uint32_t* ncount;  // [batch] nodes of the tree
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
__device__ __forceinline__ uint64_t nl_key_hash(uint64_t past, uint64_t choices, uint32_t present) {
    return rp_mix64(rp_mix64(past ^ 0x9e3779b97f4a7c15ull) ^ rp_mix64(choices + 0xd1342543de82ef95ull) ^ ((uint64_t)present * 0xaf251af3b0f025b5ull));
}
// find or insert; every lane makes progress in every iteration (the winner of a slot writes it inside the same iteration),
// so lanes of one wavefront that meet on a slot cannot deadlock
__device__ uint32_t nl_row_of(const NlTable& t, uint64_t past, uint64_t choices, uint32_t present, uint64_t key_hash, const uint32_t* edges,
                              uint32_t nch, uint32_t* err) {
    uint32_t s = (uint32_t)key_hash & t.mask;
    for (uint32_t probes = 0; probes <= t.mask; ) {
        const uint32_t st = atomicCAS(&t.state[s], 0u, 1u);
        if (st == 0u) {
            t.past[s] = past;
            t.choices[s] = choices;
            t.present[s] = present;
            float* row = t.rows + (size_t)s * 4u * NLMC_A;
            for (uint32_t a = 0; a < nch; ++a) row[a] = nl_default_regret(edges[a]);
            __threadfence();
            atomicExch(&t.state[s], 2u);
            atomicAdd(t.n_keys, 1u);
            return s;
        }
        if (st == 1u) continue;  // being written by another lane or wavefront: look again
        __threadfence();
        if (t.past[s] == past && t.choices[s] == choices && t.present[s] == present) return s;
        s = (s + 1u) & t.mask;
        probes += 1;
    }
    *err |= NERR_TABLE_FULL;
    return 0;
}

__device__ __forceinline__ uint64_t nl_draw(uint64_t deck, int k, const NlParams& p, uint64_t tree, uint64_t key) {  // tree = its id in the epoch
    uint64_t out = 0;
    for (int c = 0; c < k; ++c) {
        const uint32_t pick = rp_pick_uniform(rp_node_hash(p.seed, p.epoch, tree, key + (uint64_t)c), (uint32_t)__popcll(deck));
        // the pick-th lowest card of the deck: a popcount search (a loop clearing `pick` bits runs up to 51 rounds at the few
        // lanes of a wavefront that sit at a chance node)
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
// NlheEncoder::abstraction (nlhe/src/encoder.rs:30-36); Abstraction = [8 bits street][8 bits index] (kicker/src/abstraction.rs:14-24)
__device__ __forceinline__ uint32_t nl_bucket(const NlParams& p, int street, uint64_t pocket, uint64_t board) {
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
    return at < 0 ? 0xffffu : (((uint32_t)street << 8) | (uint32_t)p.tabs[street][at]);
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
    v.to_call = ms - g.stake[me];
    v.to_shove = g.stack[me];
    v.to_raise = g.to_raise();
    v.pot = g.pot;
    v.street = g.street();
    v.may_fold = v.to_call > 0;
    v.may_call = v.may_fold && v.to_call < v.to_shove;
    v.may_check = ms == g.stake[me];
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

// stack entry: [0..4] packed game, [5] parent | slot << 13 | edge << 17 | depth << 22 | plen << 25, [6,7] past, [8,9] hkey, [10] fac
#define NL_SENT 12u

__global__ __launch_bounds__(64) void k_nlhe_traverse(NlParams p, NlTable t, NlScratch sc, unsigned long long* counters) {
    const uint32_t tree = blockIdx.x * 64u + threadIdx.x;  // slot of the tree in this rank's scratch
    if (tree >= p.batch) return;
    const uint64_t tree_id = p.tree_base + tree;           // its id in the epoch: what the random draws are keyed by
    uint32_t err = 0;
    uint32_t* meta = sc.meta + (size_t)tree * p.ncap;
    float* fac = sc.fac + (size_t)tree * p.ncap;
    float* val = sc.val + (size_t)tree * p.ncap;
    uint32_t* aux = sc.aux + (size_t)tree * p.ncap;
    uint32_t* wrow = sc.wrow + (size_t)tree * p.wcap;
    uint32_t* wnode = sc.wnode + (size_t)tree * p.wcap;
    float* kidd = sc.kidd + (size_t)tree * p.wcap * NLMC_A;
    uint32_t* stack = sc.stack + (size_t)tree * p.scap * NL_SENT;
    const int walker = (int)p.walker;
    // ---- Solver::tree: Game::root() with the hole cards dealt (P0 on the button, game.rs:66-78)
    G2 g;
    g.n = 2;
    g.dealer = 0;
    g.ticker = 0;  // n == 2: the dealer posts the small blind
    g.pot = 0;
    g.board = 0;
    uint64_t deck = HAND_MASK;
    for (int i = 0; i < 2; ++i) {
        g.state[i] = NL_BETTING;
        g.stack[i] = 200;
        g.stake[i] = g.spent[i] = 0;
        g.cards[i] = nl_draw(deck, 2, p, tree_id, 0xD0C0000000000000ull + 8u * (uint64_t)i);
        deck &= ~g.cards[i];
    }
    for (int b = 0; b < 2; ++b) g.force_act(NlAction{NA_BLIND, g.to_post(), 0});
    const uint64_t hole0 = g.cards[0], hole1 = g.cards[1];
    uint32_t n = 0, top = 0, nw = 0;
    // the node being grown
    uint32_t cur_parent = 0, cur_slot = 0, cur_depth = 0, cur_plen = 0;
    uint64_t cur_past = 0, cur_hkey = rp_mix64(0x726f6f74ull);
    float cur_fac = 1.0f;
    for (;;) {
        // ---- grow: node n = (g, info)
        if (n >= p.ncap) {
            err |= NERR_NODES;
            break;
        }
        const uint32_t me = n++;
        const int turn = g.turn();
        uint32_t kind, nch = 0;
        uint32_t edges[12];
        float nodeval = 0.0f;
        float sigma[NLMC_A];
        float childfac_opp = 1.0f;
        uint32_t pick = 0;
        NlView view{};
        if (turn == NT_TERMINAL) {
            kind = NK_TERMINAL;
            int reward[2];
            nl_settle(g, reward);
            nodeval = (float)(reward[walker] - g.spent[walker]);  // NlheGame::payoff (nlhe/src/game.rs:59-65)
        } else if (turn == NT_CHANCE) {
            kind = NK_CHANCE;
            nch = 1;
            edges[0] = NE_DRAW;
        } else {
            view = nl_view(g);
            nch = (uint32_t)nl_choices_v(view, (int)cur_depth, edges);
            uint64_t chpath = 0;
            for (uint32_t a = 0; a < nch; ++a) chpath |= (uint64_t)edges[a] << (5u * a);
            const uint32_t bucket = nl_bucket(p, g.street(), turn == 0 ? hole0 : hole1, g.board);
            const uint64_t khash = nl_key_hash(cur_past, chpath, bucket);  // the table slot and, at an opponent node, the draw's key
            const uint32_t row = nl_row_of(t, cur_past, chpath, bucket, khash, edges, nch, &err);
            const float* r = t.rows + (size_t)row * 4u * NLMC_A;
            float rd = 0.0f;
            for (uint32_t a = 0; a < nch; ++a) {
                sigma[a] = rp_maxf(r[a], RP_EPSILON);  // RefProf::regret (profile.rs:31-33)
                rd += sigma[a];
            }
            for (uint32_t a = 0; a < nch; ++a) sigma[a] = sigma[a] / rd;  // instant_policy (flow.rs:46-48)
            if (turn == walker) {
                kind = NK_WALKER;
                if (nw >= p.wcap) {
                    err |= NERR_WALKERS;
                    break;
                }
                aux[me] = nw;
                wrow[nw] = row;
                wnode[nw] = me;
                nw += 1;
            } else {
                // weighted (sample/external.rs:41-64) over sampling_distribution (flow.rs:24-42), one draw per (epoch, infoset, tree)
                kind = NK_OPP;
                float wsum = 0.0f, wv[NLMC_A];
                for (uint32_t a = 0; a < nch; ++a) {
                    wv[a] = rp_maxf(r[NLMC_A + a], RP_EPSILON);
                    wsum += wv[a];
                }
                const float denom = wsum + p.smoothing;
                float z = 0.0f, sw[NLMC_A];
                for (uint32_t a = 0; a < nch; ++a) {
                    sw[a] = rp_maxf((wv[a] / p.temperature + p.smoothing) / denom, p.curiosity);
                    z += sw[a];
                }
                float cum[NLMC_A], total = 0.0f;
                for (uint32_t a = 0; a < nch; ++a) {
                    total += rp_maxf(sw[a] / z, RP_EPSILON);
                    cum[a] = total;
                }
                const float u = rp_u01(rp_node_hash(p.seed, p.epoch, tree_id, khash)) * total;
                while (pick + 1 < nch && cum[pick] <= u) ++pick;
                childfac_opp = sigma[pick] / (sw[pick] / z);
            }
        }
        meta[me] = cur_parent | (cur_slot << 13) | (kind << 17) | (nch << 19);
        fac[me] = cur_fac;
        val[me] = nodeval;
        // ---- branches + sample: push the children to expand, ascending choice slot (popped last-first)
        for (uint32_t a = 0; a < nch; ++a) {
            if (kind == NK_OPP && a != pick) continue;
            if (top >= p.scap) {
                err |= NERR_STACK;
                break;
            }
            const uint32_t e = edges[a];
            const uint64_t hk = rp_mix64(cur_hkey ^ ((uint64_t)(e + 1u) * 0x9fb21c651e98df25ull));
            G2 c = g;
            NlAction act;
            if (e == NE_DRAW) act = NlAction{NA_DRAW, 0, nl_draw(g.deck(), g.street() == 0 ? 3 : 1, p, tree_id, hk)};
            else act = nl_action_v(view, e);
            // Game::apply panics on an illegal action (kicker game.rs:234-247).  snap()'s output is legal by construction, so
            // the test can only catch an engine bug: it runs in the checking mode (RP_NLHE_CHECK_LEGAL=1, the tests), not per child
            if (p.check_legal && !c.allowed(act)) err |= NERR_ILLEGAL;
            c.force_act(act);
            const Packed pk = pack_game(c);
            uint32_t* se = stack + (size_t)top * NL_SENT;
            const bool raise = e == NE_SHOVE || e >= NE_OPEN0;
            const uint32_t cdepth = e == NE_DRAW ? 0u : cur_depth + (raise ? 1u : 0u);
            const uint32_t cplen = e == NE_DRAW ? 0u : cur_plen + 1u;
            const uint64_t cpast = e == NE_DRAW ? 0ull : (cur_plen < 12u ? cur_past | ((uint64_t)e << (5u * cur_plen)) : cur_past);
            se[0] = pk.w0; se[1] = pk.w1; se[2] = pk.w2; se[3] = pk.blo; se[4] = pk.bhi;
            se[5] = me | (a << 13) | (cdepth << 22) | (cplen << 25);
            se[6] = (uint32_t)cpast; se[7] = (uint32_t)(cpast >> 32);
            se[8] = (uint32_t)hk; se[9] = (uint32_t)(hk >> 32);
            const float f = kind == NK_WALKER ? sigma[a] : (kind == NK_OPP ? childfac_opp : 1.0f);
            se[10] = __float_as_uint(f);
            top += 1;
        }
        if (err & (NERR_STACK | NERR_NODES | NERR_WALKERS)) break;
        if (top == 0) break;
        // ---- pop-last (builder.rs:143)
        top -= 1;
        const uint32_t* se = stack + (size_t)top * NL_SENT;
        unpack_game(Packed{se[0], se[1], se[2], se[3], se[4]}, g);
        cur_parent = se[5] & 0x1fffu;
        cur_slot = (se[5] >> 13) & 15u;
        cur_depth = (se[5] >> 22) & 7u;
        cur_plen = (se[5] >> 25) & 15u;
        cur_past = (uint64_t)se[6] | ((uint64_t)se[7] << 32);
        cur_hkey = (uint64_t)se[8] | ((uint64_t)se[9] << 32);
        cur_fac = __uint_as_float(se[10]);
    }
    sc.ncount[tree] = n;
    uint32_t nd = 0;
    if (!err) {
        // ---- D(node): descending index adds a node's children in choices() order (the reference's newest-edge-first walk)
        for (uint32_t w = 0; w < nw * NLMC_A; ++w) kidd[w] = 0.0f;
        for (uint32_t i = n - 1; i >= 1; --i) {
            const uint32_t m = meta[i], par = m & 0x1fffu;
            const float d = val[i];
            val[par] += fac[i] * d;
            if (((meta[par] >> 17) & 3u) == NK_WALKER) kidd[(size_t)aux[par] * NLMC_A + ((m >> 13) & 15u)] = d;
        }
        // ---- reach(node): sigma / q over the opponent ancestors (ancestor_reach, flow.rs:166-174); chance and walker edges carry 1
        val[0] = 1.0f;
        for (uint32_t i = 1; i < n; ++i) {
            const uint32_t par = meta[i] & 0x1fffu;
            val[i] = ((meta[par] >> 17) & 3u) == NK_OPP ? val[par] * fac[i] : val[par];
        }
        // ---- Decisions: infosets in the order of their first walker node, span in ascending node index (tree.rs:88-98)
        uint32_t* drow = sc.drow + (size_t)tree * p.dcap;
        uint32_t* dmeta = sc.dmeta + (size_t)tree * p.dcap;
        float* dreg = sc.dreg + (size_t)tree * p.dcap * NLMC_A;
        float* dpol = sc.dpol + (size_t)tree * p.dcap * NLMC_A;
        float* dpay = sc.dpay + (size_t)tree * p.dcap;
        uint32_t* dmap = sc.dmap + (size_t)tree * 2u * p.dcap;
        const uint32_t dmask = 2u * p.dcap - 1u;
        for (uint32_t q = 0; q <= dmask; ++q) dmap[q] = 0u;
        for (uint32_t w = 0; w < nw; ++w) {
            const uint32_t node = wnode[w], row = wrow[w];
            const uint32_t nch = (meta[node] >> 19) & 15u;
            if (nch == 0) continue;
            uint32_t s = (row * 2654435761u) & dmask, d = 0;
            for (;;) {
                if (dmap[s] == 0u) {
                    if (nd >= p.dcap) {
                        err |= NERR_DECISIONS;
                        break;
                    }
                    d = nd++;
                    dmap[s] = d + 1u;
                    drow[d] = row;
                    dmeta[d] = nch;
                    dpay[d] = 0.0f;
                    for (uint32_t a = 0; a < NLMC_A; ++a) dreg[(size_t)d * NLMC_A + a] = 0.0f, dpol[(size_t)d * NLMC_A + a] = 0.0f;
                    break;
                }
                if (drow[dmap[s] - 1u] == row) {
                    d = dmap[s] - 1u;
                    break;
                }
                s = (s + 1u) & dmask;
            }
            if (err) break;
            const float* r = t.rows + (size_t)row * 4u * NLMC_A;
            float sg[NLMC_A], rd = 0.0f;
            for (uint32_t a = 0; a < nch; ++a) {
                sg[a] = rp_maxf(r[a], RP_EPSILON);
                rd += sg[a];
            }
            const float reach = val[node];
            float cfv[NLMC_A], ev = 0.0f;
            for (uint32_t a = 0; a < nch; ++a) cfv[a] = reach * kidd[(size_t)w * NLMC_A + a];
            for (uint32_t a = 0; a < nch; ++a) ev += sg[a] / rd * cfv[a];
            dpay[d] += ev;
            for (uint32_t a = 0; a < nch; ++a) {
                dreg[(size_t)d * NLMC_A + a] += cfv[a] - ev;
                dpol[(size_t)d * NLMC_A + a] = sg[a] / rd;  // policy_vector = iterated_distribution (flow.rs:118-120)
            }
            dmeta[d] = nch | (((1u << nch) - 1u) << 8);  // external sampling expands every walker edge
        }
    }
    sc.dcount[tree] = err ? 0u : nd;
    // metrics: nodes, infos; error flags
    atomicAdd(counters + 0, (unsigned long long)n);
    atomicAdd(counters + 1, (unsigned long long)(err ? 0u : nd));
    if (err) atomicOr(counters + 2, (unsigned long long)err);
}

// exclusive scan of the per-tree Decisions counts (one workgroup; a batch has at most a few 10^5 trees)
__global__ __launch_bounds__(1024) void k_nlhe_scan(const uint32_t* dcount, uint32_t batch, uint32_t* offset, uint32_t* total) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x, per = (batch + 1023u) / 1024u;
    uint32_t s = 0;
    for (uint32_t i = tid * per; i < min(batch, (tid + 1) * per); ++i) s += dcount[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < 1024; ++i) {
            const uint32_t v = part[i];
            part[i] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    uint32_t run = part[tid];
    for (uint32_t i = tid * per; i < min(batch, (tid + 1) * per); ++i) {
        offset[i] = run;
        run += dcount[i];
    }
}

struct NlBatch {  // rp_decisions layout
    uint32_t* row;
    uint8_t* nact;
    uint16_t* expanded;
    float* regret;
    float* policy;
    float* payoff;
    uint32_t* tree;
};
__global__ __launch_bounds__(256) void k_nlhe_pack(NlScratch sc, uint32_t batch, uint32_t dcap, const uint32_t* offset, NlBatch out) {
    const uint32_t tree = blockIdx.x;
    const uint32_t nd = sc.dcount[tree], base = offset[tree];
    for (uint32_t e = threadIdx.x; e < nd * NLMC_A; e += 256) {
        const uint32_t d = e / NLMC_A, a = e % NLMC_A;
        const size_t src = ((size_t)tree * dcap + d) * NLMC_A + a, dst = (size_t)(base + d) * NLMC_A + a;
        out.regret[dst] = sc.dreg[src];
        out.policy[dst] = sc.dpol[src];
    }
    for (uint32_t d = threadIdx.x; d < nd; d += 256) {
        const size_t src = (size_t)tree * dcap + d;
        out.row[base + d] = sc.drow[src];
        out.nact[base + d] = (uint8_t)(sc.dmeta[src] & 0xffu);
        out.expanded[base + d] = (uint16_t)(sc.dmeta[src] >> 8);
        out.payoff[base + d] = sc.dpay[src];
        out.tree[base + d] = tree;
    }
}

// ---- the multi-GPU exchange by infoset key (every rank's table assigns rows in its own insertion order) ----
// entries: rp_profile_summarize's records, [row u32][count u32][psum f32][n_actions u32][maps]; keys beside them
__global__ __launch_bounds__(256) void k_nlhe_entry_keys(NlTable t, const unsigned char* entries, uint32_t eb, uint32_t n, uint64_t* past,
                                                         uint32_t* present, uint64_t* choices) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t row = *reinterpret_cast<const uint32_t*>(entries + (size_t)i * eb);
    past[i] = t.past[row];
    present[i] = t.present[row];
    choices[i] = t.choices[row];
}
__global__ __launch_bounds__(256) void k_nlhe_entry_remap(NlTable t, unsigned char* entries, uint32_t eb, uint32_t n, const uint64_t* past,
                                                          const uint32_t* present, const uint64_t* choices, unsigned long long* counters) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t edges[12], nch = 0, err = 0;
    for (uint64_t c = choices[i]; nch < 12u && (c & 0x1full) != 0; c >>= 5) edges[nch++] = (uint32_t)(c & 0x1full);
    const uint32_t row = nl_row_of(t, past[i], choices[i], present[i], nl_key_hash(past[i], choices[i], present[i]), edges, nch, &err);
    *reinterpret_cast<uint32_t*>(entries + (size_t)i * eb) = row;
    if (err) atomicOr(counters + 2, (unsigned long long)err);
}

}  // namespace rp

using namespace rp;

struct rp_nlhe {
    int device = 0;
    rp_profile* prof = nullptr;
    uint32_t cap_log2 = 0, batch = 128;
    rp_hyper hp{};
    uint64_t seed = 0;
    NlTable tab{};
    NlParams prm{};
    NlScratch sc{};
    NlBatch out{};
    uint32_t out_cap = 0;
    uint32_t* d_offset = nullptr;
    uint32_t* d_total = nullptr;
    unsigned long long* d_counters = nullptr;  // nodes, infos, error flags
    std::vector<void*> allocs;
    uint32_t last_n = 0;
};

namespace {
template <typename T>
int nl_alloc(rp_nlhe* h, T** out, size_t count) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    HIP_TRY(hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)));
    h->allocs.push_back(p);
    *out = reinterpret_cast<T*>(p);
    return RP_OK;
}
// the traversal of the current epoch: Decisions in h->out, their count in h->last_n
int nl_traverse(rp_nlhe* h) {
    hipStream_t st = rp::profile_stream(h->prof);
    h->prm.epoch = rp::profile_epoch(h->prof);
    h->prm.walker = (uint32_t)(h->prm.epoch % 2u);  // CfrSampling::walker (book.rs:142-144)
    hipLaunchKernelGGL(k_nlhe_traverse, dim3((h->batch + 63u) / 64u), dim3(64), 0, st, h->prm, h->tab, h->sc, h->d_counters);
    hipLaunchKernelGGL(k_nlhe_scan, dim3(1), dim3(1024), 0, st, h->sc.dcount, h->batch, h->d_offset, h->d_total);
    HIP_TRY(hipGetLastError());
    uint32_t total = 0;
    unsigned long long c[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(&total, h->d_total, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(c, h->d_counters, 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (c[2]) return rp::fail(RP_ERR_CAPACITY, "rp_nlhe: traversal capacity exceeded (flags %llu: 1 nodes, 2 stack, 4 walker nodes, 8 decisions, "
                                                "16 illegal action, 32 infoset table full)", c[2]);
    if (total > h->out_cap) return rp::fail(RP_ERR_CAPACITY, "rp_nlhe: %u Decisions in one batch exceed the buffer (%u)", total, h->out_cap);
    hipLaunchKernelGGL(k_nlhe_pack, dim3(h->batch), dim3(256), 0, st, h->sc, h->batch, h->prm.dcap, h->d_offset, h->out);
    HIP_TRY(hipGetLastError());
    h->last_n = total;
    return RP_OK;
}
}  // namespace

extern "C" {

int rp_nlhe_create(int device, uint32_t cap_log2, rp_regret_kind regret, rp_weight_kind weight, const rp_hyper* hp, uint64_t seed,
                   uint32_t batch, const rp_lookup* const* tables, rp_nlhe** out) {
    if (!out || !hp || cap_log2 < 8 || cap_log2 > 30) return rp::fail(RP_ERR_INVALID, "rp_nlhe_create: bad argument");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_nlhe_create: no HIP device visible; the MI355X path has no CPU fallback");
    if (batch == 0) batch = 128;  // nlhe/src/solver.rs:11
    rp_nlhe* h = new rp_nlhe();
    h->device = device;
    h->cap_log2 = cap_log2;
    h->batch = batch;
    h->hp = *hp;
    h->seed = seed;
#define NL_TRY(expr)                    \
    do {                                \
        int _rc = (expr);               \
        if (_rc) {                      \
            rp_nlhe_destroy(h);         \
            return _rc;                 \
        }                               \
    } while (0)
    if (hipSetDevice(device) != hipSuccess) {
        delete h;
        return rp::fail(RP_ERR_HIP, "rp_nlhe_create: hipSetDevice(%d) failed", device);
    }
    const uint64_t rows = 1ull << cap_log2;
    const uint32_t ncap = 4096, scap = 256, wcap = 2048, dcap = 1024;
    // Decisions per tree: ~60 on average, 483 the largest seen in 20 000 oracle trees; the batch buffer holds 160 per tree
    const uint64_t dec_cap64 = std::max<uint64_t>((uint64_t)batch * 160u, 4096u);
    if (dec_cap64 >= (1ull << 31)) {
        delete h;
        return rp::fail(RP_ERR_INVALID, "rp_nlhe_create: batch too large");
    }
    NL_TRY(rp_profile_create(device, rows, NLMC_A, regret, weight, hp, nullptr, (uint32_t)dec_cap64, &h->prof));
    NL_TRY(nl_alloc(h, &h->tab.past, rows));
    NL_TRY(nl_alloc(h, &h->tab.choices, rows));
    NL_TRY(nl_alloc(h, &h->tab.present, rows));
    NL_TRY(nl_alloc(h, &h->tab.state, rows));
    NL_TRY(nl_alloc(h, &h->tab.n_keys, 1));
    h->tab.mask = (uint32_t)(rows - 1);
    h->tab.rows = rp::profile_table(h->prof);
    h->prm.seed = seed;
    h->prm.batch = batch;
    h->prm.temperature = hp->temperature;
    h->prm.smoothing = hp->smoothing;
    h->prm.curiosity = hp->curiosity;
    h->prm.ncap = ncap; h->prm.scap = scap; h->prm.wcap = wcap; h->prm.dcap = dcap;
    h->prm.check_legal = getenv("RP_NLHE_CHECK_LEGAL") ? 1u : 0u;
    h->prm.encoder = 0;
    if (tables) {
        for (int s = 0; s < 4; ++s) {
            int street = -1;
            const uint8_t* abs_ = nullptr;
            NL_TRY(rp::lookup_view(tables[s], &h->prm.tkeys[s], &abs_, &h->prm.tn[s], &street));
            h->prm.tabs[s] = abs_;
            if (street != s) NL_TRY(rp::fail(RP_ERR_INVALID, "rp_nlhe_create: tables[%d] is a lookup of street %d", s, street));
        }
        h->prm.encoder = 1;
    }
    const size_t B = batch;
    NL_TRY(nl_alloc(h, &h->sc.meta, B * ncap));
    NL_TRY(nl_alloc(h, &h->sc.fac, B * ncap));
    NL_TRY(nl_alloc(h, &h->sc.val, B * ncap));
    NL_TRY(nl_alloc(h, &h->sc.aux, B * ncap));
    NL_TRY(nl_alloc(h, &h->sc.wrow, B * wcap));
    NL_TRY(nl_alloc(h, &h->sc.wnode, B * wcap));
    NL_TRY(nl_alloc(h, &h->sc.kidd, B * wcap * NLMC_A));
    NL_TRY(nl_alloc(h, &h->sc.stack, B * scap * NL_SENT));
    NL_TRY(nl_alloc(h, &h->sc.drow, B * dcap));
    NL_TRY(nl_alloc(h, &h->sc.dmeta, B * dcap));
    NL_TRY(nl_alloc(h, &h->sc.dreg, B * dcap * NLMC_A));
    NL_TRY(nl_alloc(h, &h->sc.dpol, B * dcap * NLMC_A));
    NL_TRY(nl_alloc(h, &h->sc.dpay, B * dcap));
    NL_TRY(nl_alloc(h, &h->sc.dmap, B * 2 * dcap));
    NL_TRY(nl_alloc(h, &h->sc.dcount, B));
    NL_TRY(nl_alloc(h, &h->sc.ncount, B));
    h->out_cap = (uint32_t)dec_cap64;
    NL_TRY(nl_alloc(h, &h->out.row, h->out_cap));
    NL_TRY(nl_alloc(h, &h->out.nact, h->out_cap));
    NL_TRY(nl_alloc(h, &h->out.expanded, h->out_cap));
    NL_TRY(nl_alloc(h, &h->out.regret, (size_t)h->out_cap * NLMC_A));
    NL_TRY(nl_alloc(h, &h->out.policy, (size_t)h->out_cap * NLMC_A));
    NL_TRY(nl_alloc(h, &h->out.payoff, h->out_cap));
    NL_TRY(nl_alloc(h, &h->out.tree, h->out_cap));
    NL_TRY(nl_alloc(h, &h->d_offset, B));
    NL_TRY(nl_alloc(h, &h->d_total, 1));
    NL_TRY(nl_alloc(h, &h->d_counters, 4));
#undef NL_TRY
    // hipMemset on device memory returns before it has run, and a non-blocking stream does not wait for the null stream: the
    // first launch on this handle's stream could otherwise overtake the initialisation above and be overwritten by it
    (void)hipDeviceSynchronize();
    *out = h;
    return RP_OK;
}

int rp_nlhe_destroy(rp_nlhe* h) {
    if (!h) return RP_OK;
    (void)hipSetDevice(h->device);
    if (h->prof) {
        (void)rp_profile_sync(h->prof);
        (void)rp_profile_destroy(h->prof);
    }
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
    return RP_OK;
}

int rp_nlhe_step(rp_nlhe* h, rp_update_mode mode) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_nlhe_step: NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    int rc = nl_traverse(h);
    if (rc) return rc;
    rp_decisions b{h->last_n, h->out.row, h->out.nact, h->out.expanded, h->out.regret, h->out.policy, h->out.payoff};
    return rp_profile_apply(h->prof, &b, mode);
}

int rp_nlhe_batch(rp_nlhe* h, uint32_t cap, uint32_t* n, uint32_t* tree, uint64_t* past, uint32_t* present, uint64_t* choices,
                  uint8_t* n_actions, uint16_t* expanded, float* regret, float* policy, float* payoff) {
    if (!h || !n) return rp::fail(RP_ERR_INVALID, "rp_nlhe_batch: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    unsigned long long before[3];
    hipStream_t st = rp::profile_stream(h->prof);
    HIP_TRY(hipMemcpyAsync(before, h->d_counters, 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    int rc = nl_traverse(h);
    HIP_TRY(hipMemcpyAsync(h->d_counters, before, 16, hipMemcpyHostToDevice, st));  // a debugging view: counters not advanced
    if (rc) return rc;
    *n = h->last_n;
    const uint32_t m = std::min(cap, h->last_n);
    if (m == 0) return RP_OK;
    std::vector<uint32_t> rows(m);
    HIP_TRY(hipMemcpyAsync(rows.data(), h->out.row, (size_t)m * 4, hipMemcpyDeviceToHost, st));
    if (tree) HIP_TRY(hipMemcpyAsync(tree, h->out.tree, (size_t)m * 4, hipMemcpyDeviceToHost, st));
    if (n_actions) HIP_TRY(hipMemcpyAsync(n_actions, h->out.nact, m, hipMemcpyDeviceToHost, st));
    if (expanded) HIP_TRY(hipMemcpyAsync(expanded, h->out.expanded, (size_t)m * 2, hipMemcpyDeviceToHost, st));
    if (regret) HIP_TRY(hipMemcpyAsync(regret, h->out.regret, (size_t)m * NLMC_A * 4, hipMemcpyDeviceToHost, st));
    if (policy) HIP_TRY(hipMemcpyAsync(policy, h->out.policy, (size_t)m * NLMC_A * 4, hipMemcpyDeviceToHost, st));
    if (payoff) HIP_TRY(hipMemcpyAsync(payoff, h->out.payoff, (size_t)m * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (past || present || choices) {  // the infoset behind each row (rows differ between implementations; keys do not)
        const size_t rowsn = (size_t)1 << h->cap_log2;
        std::vector<uint64_t> kp(rowsn), kc(rowsn);
        std::vector<uint32_t> kb(rowsn);
        HIP_TRY(hipMemcpy(kp.data(), h->tab.past, rowsn * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(kc.data(), h->tab.choices, rowsn * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(kb.data(), h->tab.present, rowsn * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < m; ++i) {
            if (past) past[i] = kp[rows[i]];
            if (choices) choices[i] = kc[rows[i]];
            if (present) present[i] = kb[rows[i]];
        }
    }
    return RP_OK;
}

int rp_nlhe_epoch(rp_nlhe* h, uint64_t* epoch) {
    if (!h || !epoch) return rp::fail(RP_ERR_INVALID, "rp_nlhe_epoch: NULL argument");
    *epoch = rp::profile_epoch(h->prof);
    return RP_OK;
}

int rp_nlhe_counters(rp_nlhe* h, uint64_t* nodes, uint64_t* infos, uint64_t* keys) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_nlhe_counters: NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = rp::profile_stream(h->prof);
    unsigned long long c[3];
    unsigned int k = 0;
    HIP_TRY(hipMemcpyAsync(c, h->d_counters, 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&k, h->tab.n_keys, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (nodes) *nodes = c[0];
    if (infos) *infos = c[1];
    if (keys) *keys = k;
    return RP_OK;
}

// every infoset of the table with its Encounters, in slot order
int rp_nlhe_export(rp_nlhe* h, uint64_t cap, uint64_t* n, uint64_t* past, uint32_t* present, uint64_t* choices, rp_encounter* enc) {
    if (!h || !n) return rp::fail(RP_ERR_INVALID, "rp_nlhe_export: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    int rc = rp_profile_sync(h->prof);
    if (rc) return rc;
    const size_t rowsn = (size_t)1 << h->cap_log2;
    std::vector<uint32_t> state(rowsn), rows;
    HIP_TRY(hipMemcpy(state.data(), h->tab.state, rowsn * 4, hipMemcpyDeviceToHost));
    for (size_t s = 0; s < rowsn; ++s)
        if (state[s] == 2u) rows.push_back((uint32_t)s);
    *n = rows.size();
    const size_t m = std::min<size_t>(cap, rows.size());
    if (m == 0 || !past || !present || !choices || !enc) return RP_OK;
    std::vector<uint64_t> kp(rowsn), kc(rowsn);
    std::vector<uint32_t> kb(rowsn);
    HIP_TRY(hipMemcpy(kp.data(), h->tab.past, rowsn * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(kc.data(), h->tab.choices, rowsn * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(kb.data(), h->tab.present, rowsn * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < m; ++i) {
        past[i] = kp[rows[i]];
        choices[i] = kc[rows[i]];
        present[i] = kb[rows[i]];
    }
    return rp_profile_get_rows(h->prof, m, rows.data(), enc);
}

// load Encounters by key (hydrate / resynchronisation): unknown keys are inserted (host-side probing, same hash)
int rp_nlhe_import(rp_nlhe* h, uint64_t n, const uint64_t* past, const uint32_t* present, const uint64_t* choices, const rp_encounter* enc,
                   uint64_t epoch) {
    if (!h || (n && (!past || !present || !choices || !enc))) return rp::fail(RP_ERR_INVALID, "rp_nlhe_import: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    int rc = rp_profile_sync(h->prof);
    if (rc) return rc;
    const size_t rowsn = (size_t)1 << h->cap_log2;
    std::vector<uint32_t> state(rowsn), kb(rowsn), rows(n);
    std::vector<uint64_t> kp(rowsn), kc(rowsn);
    HIP_TRY(hipMemcpy(state.data(), h->tab.state, rowsn * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(kp.data(), h->tab.past, rowsn * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(kc.data(), h->tab.choices, rowsn * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(kb.data(), h->tab.present, rowsn * 4, hipMemcpyDeviceToHost));
    unsigned int added = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t hk = rp_mix64(rp_mix64(past[i] ^ 0x9e3779b97f4a7c15ull) ^ rp_mix64(choices[i] + 0xd1342543de82ef95ull) ^
                                     ((uint64_t)present[i] * 0xaf251af3b0f025b5ull));
        uint32_t s = (uint32_t)hk & h->tab.mask;
        for (size_t probes = 0;; ++probes) {
            if (probes > rowsn) return rp::fail(RP_ERR_CAPACITY, "rp_nlhe_import: infoset table full");
            if (state[s] != 2u) {
                state[s] = 2u;
                kp[s] = past[i];
                kc[s] = choices[i];
                kb[s] = present[i];
                added += 1;
                break;
            }
            if (kp[s] == past[i] && kc[s] == choices[i] && kb[s] == present[i]) break;
            s = (s + 1u) & h->tab.mask;
        }
        rows[i] = s;
    }
    HIP_TRY(hipMemcpy(h->tab.state, state.data(), rowsn * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->tab.past, kp.data(), rowsn * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->tab.choices, kc.data(), rowsn * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->tab.present, kb.data(), rowsn * 4, hipMemcpyHostToDevice));
    unsigned int k = 0;
    HIP_TRY(hipMemcpy(&k, h->tab.n_keys, 4, hipMemcpyDeviceToHost));
    k += added;
    HIP_TRY(hipMemcpy(h->tab.n_keys, &k, 4, hipMemcpyHostToDevice));
    if ((rc = rp_profile_set_rows(h->prof, n, rows.data(), enc))) return rc;
    return rp_profile_set_epoch(h->prof, epoch);
}

int rp_nlhe_set_shard(rp_nlhe* h, uint32_t rank, uint32_t world) {
    if (!h || world == 0 || rank >= world) return rp::fail(RP_ERR_INVALID, "rp_nlhe_set_shard: bad rank/world");
    h->prm.tree_base = (uint64_t)rank * h->batch;
    return RP_OK;
}

int rp_nlhe_entry_bytes(rp_nlhe* h, size_t* bytes, uint32_t* max_entries) {
    if (!h || !bytes) return rp::fail(RP_ERR_INVALID, "rp_nlhe_entry_bytes: NULL argument");
    int rc = rp_profile_entry_bytes(h->prof, bytes);
    if (max_entries) *max_entries = h->out_cap;
    return rc;
}

int rp_nlhe_step_local(rp_nlhe* h, void* entries_dev, uint64_t* past_dev, uint32_t* present_dev, uint64_t* choices_dev, uint32_t* n_entries) {
    if (!h || !entries_dev || !past_dev || !present_dev || !choices_dev || !n_entries)
        return rp::fail(RP_ERR_INVALID, "rp_nlhe_step_local: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    int rc = nl_traverse(h);
    if (rc) return rc;
    rp_decisions b{h->last_n, h->out.row, h->out.nact, h->out.expanded, h->out.regret, h->out.policy, h->out.payoff};
    if ((rc = rp_profile_summarize(h->prof, &b, entries_dev, n_entries))) return rc;  // synchronises: *n_entries is valid
    if (*n_entries) {
        size_t eb = 0;
        (void)rp_profile_entry_bytes(h->prof, &eb);
        hipLaunchKernelGGL(k_nlhe_entry_keys, dim3((*n_entries + 255u) / 256u), dim3(256), 0, rp::profile_stream(h->prof), h->tab,
                           reinterpret_cast<const unsigned char*>(entries_dev), (uint32_t)eb, *n_entries, past_dev, present_dev, choices_dev);
        HIP_TRY(hipGetLastError());
    }
    return RP_OK;
}

int rp_nlhe_step_apply(rp_nlhe* h, void* entries_dev, const uint64_t* past_dev, const uint32_t* present_dev, const uint64_t* choices_dev,
                       uint32_t n_entries) {
    if (!h || (n_entries && (!entries_dev || !past_dev || !present_dev || !choices_dev)))
        return rp::fail(RP_ERR_INVALID, "rp_nlhe_step_apply: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    if (n_entries) {
        size_t eb = 0;
        (void)rp_profile_entry_bytes(h->prof, &eb);
        hipLaunchKernelGGL(k_nlhe_entry_remap, dim3((n_entries + 255u) / 256u), dim3(256), 0, rp::profile_stream(h->prof), h->tab,
                           reinterpret_cast<unsigned char*>(entries_dev), (uint32_t)eb, n_entries, past_dev, present_dev, choices_dev,
                           h->d_counters);
        HIP_TRY(hipGetLastError());
    }
    return rp_profile_fold(h->prof, entries_dev, n_entries);
}

int rp_nlhe_set_stream(rp_nlhe* h, void* hip_stream) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_nlhe_set_stream: null handle");
    return rp_profile_set_stream(h->prof, hip_stream);
}

int rp_nlhe_sync(rp_nlhe* h) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_nlhe_sync: null handle");
    return rp_profile_sync(h->prof);
}

}  // extern "C"

// nlhe.hip — the no-limit hold'em rules engine on the device (first device step): the betting
// state machine, legal actions, the action abstraction and side-pot settlement as device functions, exercised by a
// playout kernel (`rp_nlhe_playouts`) whose every intermediate state is digested and compared with the CPU oracle.
// Reference: crates/kicker/src/{game,seat,action,turn,showdown,edge,size,path}.rs, crates/nlhe/src/game.rs.
// Oracle: oracle/rp_oracle_nlhe.c (ora_nlhe_playout).  All integer work; payoffs are i16 chip counts as f32.
//
// One LANE per game here: this kernel exists to pin the rules on the device.  The MCCFR traversal over this engine
// (one workgroup per tree) is the next step and will reuse these functions.
#include <hip/hip_runtime.h>

#include "../../include/rp_math.h"
#include "cards.hpp"
#include "rp_internal.h"

namespace rp {

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define NL_MAXP 10
#define NL_SBLIND 1
#define NL_BBLIND 2
enum : int { NL_BETTING = 0, NL_SHOVING = 1, NL_FOLDING = 2 };                           // seat.rs:79-84
enum : int { NA_DRAW = 0, NA_FOLD, NA_CALL, NA_CHECK, NA_RAISE, NA_SHOVE, NA_BLIND };    // action.rs:8-16
enum : int { NT_TERMINAL = -2, NT_CHANCE = -1 };                                         // turn.rs:2-6
enum : uint32_t { NE_DRAW = 1, NE_FOLD = 2, NE_CHECK = 3, NE_CALL = 4, NE_SHOVE = 5, NE_OPEN0 = 6, NE_RAISE0 = 10 };  // edge.rs:101-120

struct NlAction {
    int kind;
    int chips;
    uint64_t cards;
};
// GameN<P> (game.rs:30-36) with the seats as parallel arrays
struct NlGame {
    int n, dealer, ticker, pot;
    uint64_t board;
    int state[NL_MAXP], stack[NL_MAXP], stake[NL_MAXP], spent[NL_MAXP];
    uint64_t cards[NL_MAXP];

    __device__ int street() const {  // Board::street
        const int c = __popcll(board);
        return c == 0 ? 0 : (c == 3 ? 1 : (c == 4 ? 2 : 3));
    }
    __device__ int actor() const { return (dealer + ticker) % n; }  // game.rs:656-658
    __device__ int max_stake() const {                              // :693-695
        int m = stake[0];
        for (int i = 1; i < n; ++i) m = max(m, stake[i]);
        return m;
    }
    // ---- the closing predicates (game.rs:463-511) ----
    __device__ bool touched() const { return ticker > n + (street() == 0 ? (n == 2 ? 1 : 2) : 0); }
    __device__ bool matched() const {
        const int top = max_stake();
        bool ok = true;
        for (int i = 0; i < n; ++i) ok = ok && !(state[i] == NL_BETTING && stake[i] != top);
        return ok;
    }
    __device__ int alive() const {
        int a = 0;
        for (int i = 0; i < n; ++i) a += state[i] != NL_FOLDING;
        return a;
    }
    __device__ bool all_shoving() const {
        bool ok = true;
        for (int i = 0; i < n; ++i) ok = ok && (state[i] == NL_FOLDING || state[i] == NL_SHOVING);
        return ok;
    }
    __device__ bool all_folding() const { return alive() == 1; }
    __device__ bool alright() const { return (touched() && matched()) || all_folding() || all_shoving(); }
    __device__ bool must_stop() const { return street() == 3 ? alright() : all_folding(); }
    __device__ bool must_deal() const { return street() != 3 && alright(); }
    __device__ bool must_post() const { return street() == 0 && pot < NL_SBLIND + NL_BBLIND; }
    __device__ int turn() const { return must_stop() ? NT_TERMINAL : (must_deal() ? NT_CHANCE : actor()); }  // :166-174
    // ---- amounts (game.rs:537-576) ----
    __device__ int to_call() const { return max_stake() - stake[actor()]; }
    __device__ int to_post() const { return min(pot < NL_SBLIND ? NL_SBLIND : NL_BBLIND, stack[actor()]); }
    __device__ int to_shove() const { return stack[actor()]; }
    __device__ int to_raise() const {
        int most = 0, next = 0;
        for (int i = 0; i < n; ++i) {
            if (state[i] == NL_FOLDING) continue;
            if (stake[i] > most) {
                next = most;
                most = stake[i];
            } else if (stake[i] > next) {
                next = stake[i];
            }
        }
        return (most - stake[actor()]) + max(most - next, NL_BBLIND);
    }
    // ---- permissions (game.rs:513-531) ----
    __device__ bool choosing() const { return turn() >= 0; }
    __device__ bool may_fold() const { return choosing() && to_call() > 0; }
    __device__ bool may_call() const { return may_fold() && to_call() < to_shove(); }
    __device__ bool may_check() const { return choosing() && max_stake() == stake[actor()]; }
    __device__ bool may_raise() const { return choosing() && to_raise() < to_shove(); }
    __device__ bool may_shove() const { return choosing() && to_shove() > 0; }
    __device__ uint64_t deck() const {  // :644-650
        uint64_t gone = board;
        for (int i = 0; i < n; ++i) gone |= cards[i];
        return ~gone & HAND_MASK;
    }
    // ---- act (game.rs:395-460) ----
    __device__ void next_player() {
        if (alright()) return;
        do ticker += 1;
        while (state[actor()] != NL_BETTING);
    }
    __device__ void force_act(const NlAction& a) {
        const int me = actor();
        if (a.kind == NA_FOLD) state[me] = NL_FOLDING;
        if (a.kind == NA_CALL || a.kind == NA_BLIND || a.kind == NA_RAISE || a.kind == NA_SHOVE) {
            pot += a.chips;
            stack[me] -= a.chips;
            stake[me] += a.chips;
            spent[me] += a.chips;
            if (stack[me] == 0) state[me] = NL_SHOVING;
        }
        if (a.kind == NA_DRAW) {
            ticker = 0;
            board |= a.cards;
        }
        next_player();
        if (a.kind == NA_DRAW)
            for (int i = 0; i < n; ++i) stake[i] = 0;  // next_street
    }
    // is_allowed (game.rs:297-319) for the kinds a solver produces
    __device__ bool allowed(const NlAction& a) const {
        switch (a.kind) {
            case NA_RAISE: return may_raise() && !must_stop() && !must_deal() && a.chips >= to_raise() && a.chips < to_shove();
            case NA_DRAW: return must_deal() && !must_stop() && (a.cards & ~deck()) == 0 && __popcll(a.cards) == (street() == 0 ? 3 : 1);
            case NA_SHOVE: return !must_stop() && !must_deal() && !must_post() && may_shove() && a.chips == to_shove();
            case NA_CALL: return !must_stop() && !must_deal() && !must_post() && may_call() && a.chips == to_call();
            case NA_FOLD: return !must_stop() && !must_deal() && !must_post() && may_fold();
            case NA_CHECK: return !must_stop() && !must_deal() && !must_post() && may_check();
            case NA_BLIND: return !must_stop() && !must_deal() && must_post() && a.chips == to_post();
        }
        return false;
    }
    // snap (game.rs:835-854)
    __device__ NlAction passive() const { return NlAction{may_check() ? NA_CHECK : NA_FOLD, 0, 0}; }
    __device__ NlAction snap(NlAction a) const {
        const NlAction shove{NA_SHOVE, to_shove(), 0}, calls{NA_CALL, to_call(), 0};
        for (int guard = 0; guard < 2; ++guard) {  // Raise may turn into Shove, which is then snapped once more
            if (a.kind == NA_RAISE) {
                if (a.chips >= to_shove() || !may_raise()) {
                    a = shove;
                    continue;
                }
                if (a.chips < to_raise()) return NlAction{NA_RAISE, to_raise(), 0};
                return a;
            }
            break;
        }
        switch (a.kind) {
            case NA_SHOVE: return may_shove() ? shove : (may_call() ? calls : passive());
            case NA_CALL: return may_call() ? calls : (may_shove() ? shove : passive());
            case NA_CHECK: return may_check() ? a : (may_call() ? calls : NlAction{NA_FOLD, 0, 0});
            case NA_FOLD: return may_fold() ? a : NlAction{NA_CHECK, 0, 0};
        }
        return a;
    }
};

// ---- the action abstraction (edge.rs:77-92, size.rs:95-138, pokerkit/src/lib.rs:81-151; Pluribus regime) ----
__device__ __constant__ int8_t NL_OPENS[4] = {2, 3, 4, 5};
__device__ __constant__ int8_t NL_RAISES[10][2] = {{1, 4}, {1, 3}, {1, 2}, {2, 3}, {3, 4}, {1, 1}, {5, 4}, {3, 2}, {2, 1}, {3, 1}};
__device__ __constant__ int8_t NL_GRID[12][6] = {{-1}, {5, 8, -1}, {5, -1}, {0, 2, 4, 5, 8, -1}, {2, 5, -1}, {5, -1}, {1, 2, 5, 8, -1},
                                                  {5, 8, -1}, {5, -1}, {1, 2, 5, 8, -1}, {5, 8, -1}, {5, -1}};
__device__ int nl_raise_edges(int street, int depth, uint32_t* out) {
    int k = 0;
    if (depth > 3) return 0;  // MAX_RAISE_REPEATS
    if (street == 0 && depth == 0) {
        for (int i = 0; i < 4; ++i) out[k++] = NE_OPEN0 + i;
        return k;
    }
    const int8_t* row = NL_GRID[street * 3 + min(depth, 2)];
    for (int i = 0; row[i] >= 0; ++i) out[k++] = NE_RAISE0 + row[i];
    return k;
}
__device__ int nl_edge_chips(uint32_t e, int pot) {  // Edge::into_chips
    if (e >= NE_OPEN0 && e < NE_RAISE0) return NL_OPENS[e - NE_OPEN0] * NL_BBLIND;
    if (e >= NE_RAISE0 && e < NE_RAISE0 + 10) {
        const float odds = (float)NL_RAISES[e - NE_RAISE0][0] / (float)NL_RAISES[e - NE_RAISE0][1];
        return (int)(int16_t)((float)pot * odds);
    }
    return 0;
}
// GameN::choices (game.rs:724-739): legal()'s order — raise grid, shove, call, fold, check
__device__ int nl_choices(const NlGame& g, int depth, uint32_t* out) {
    int k = 0;
    if (g.must_stop() || g.must_deal() || g.must_post()) return 0;
    if (g.may_raise()) k += nl_raise_edges(g.street(), depth, out + k);
    if (g.may_shove()) out[k++] = NE_SHOVE;
    if (g.may_call()) out[k++] = NE_CALL;
    if (g.may_fold()) out[k++] = NE_FOLD;
    if (g.may_check()) out[k++] = NE_CHECK;
    return k;
}
__device__ NlAction nl_actionize(const NlGame& g, uint32_t e, uint64_t draw) {  // game.rs:741-753
    switch (e) {
        case NE_FOLD: return NlAction{NA_FOLD, 0, 0};
        case NE_DRAW: return NlAction{NA_DRAW, 0, draw};
        case NE_CALL: return NlAction{NA_CALL, g.to_call(), 0};
        case NE_CHECK: return NlAction{NA_CHECK, 0, 0};
        case NE_SHOVE: return NlAction{NA_SHOVE, g.to_shove(), 0};
    }
    return NlAction{NA_RAISE, nl_edge_chips(e, g.pot), 0};
}
// Showdown::settle (showdown.rs:36-109) on the seats of a terminal game; reward[i] = chips received
__device__ void nl_settle(const NlGame& g, int* reward) {
    uint32_t strength[NL_MAXP];
    for (int i = 0; i < g.n; ++i) {
        reward[i] = 0;
        strength[i] = strength_key(sw_of_hand(g.cards[i] | g.board));
    }
    uint32_t best = 0xffffffffu;
    int distributing = 0, distributed = 0;
    for (;;) {
        bool found = false;
        uint32_t top = 0;
        for (int i = 0; i < g.n; ++i)
            if (strength[i] < best && g.state[i] != NL_FOLDING && (!found || strength[i] > top)) {
                found = true;
                top = strength[i];
            }
        if (!found) return;
        best = top;
        for (;;) {
            distributed = distributing;
            int amount = -1;
            for (int i = 0; i < g.n; ++i)
                if (strength[i] == best && g.spent[i] > distributed && g.state[i] != NL_FOLDING && (amount < 0 || g.spent[i] < amount))
                    amount = g.spent[i];
            if (amount < 0) break;
            distributing = amount;
            int chips = 0, nw = 0;
            for (int i = 0; i < g.n; ++i) {
                chips += max(min(g.spent[i], distributing) - distributed, 0);
                nw += g.state[i] != NL_FOLDING && strength[i] == best && g.spent[i] > distributed;
            }
            const int share = chips / nw, bonus = chips % nw;
            int w = 0, staked = 0, paid = 0;
            for (int i = 0; i < g.n; ++i) {
                if (g.state[i] != NL_FOLDING && strength[i] == best && g.spent[i] > distributed) {
                    reward[i] += share + (w < bonus ? 1 : 0);
                    w += 1;
                }
                staked += g.spent[i];
                paid += reward[i];
            }
            if (staked == paid) return;
        }
    }
}

// ---- the playout (the oracle's ora_nlhe_playout restates this driver; the RULES are restated independently) ----
// Random numbers: rp_node_hash(seed, 0, game, counter) (include/rp_math.h), one counter per draw.
__device__ uint64_t nl_draw_cards(uint64_t deck, int k, uint64_t seed, uint64_t game, uint32_t* counter) {
    uint64_t out = 0;
    for (int c = 0; c < k; ++c) {
        const uint32_t pick = rp_pick_uniform(rp_node_hash(seed, 0, game, (*counter)++), (uint32_t)__popcll(deck));
        uint64_t d = deck;
        for (uint32_t s = 0; s < pick; ++s) d &= d - 1;  // drop the `pick` lowest cards
        const uint64_t card = d & (~d + 1);
        out |= card;
        deck &= ~card;
    }
    return out;
}
__device__ uint64_t nl_digest(uint64_t h, const NlGame& g) {
    h = rp_mix64(h ^ ((uint64_t)(uint32_t)g.pot | (uint64_t)(uint32_t)g.ticker << 20 | (uint64_t)(uint32_t)g.dealer << 40));
    h = rp_mix64(h ^ g.board);
    for (int i = 0; i < g.n; ++i)
        h = rp_mix64(h ^ ((uint64_t)(uint32_t)g.stack[i] | (uint64_t)(uint32_t)g.stake[i] << 16 | (uint64_t)(uint32_t)g.spent[i] << 32 |
                          (uint64_t)(uint32_t)g.state[i] << 48));
    return h;
}
__global__ void k_nlhe_playouts(int n, uint64_t n_games, uint64_t seed, uint32_t max_steps, float* payoffs, uint64_t* digests,
                                uint32_t* steps_out) {
    const uint64_t game = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (game >= n_games) return;
    uint32_t counter = 0;
    NlGame g;
    g.n = n;
    g.dealer = (int)(game % (uint64_t)n);
    g.ticker = n != 2;
    g.pot = 0;
    g.board = 0;
    uint64_t deck = HAND_MASK;
    for (int i = 0; i < n; ++i) {  // preblind (game.rs:59-71): holes in seat order
        g.state[i] = NL_BETTING;
        g.stack[i] = 200;
        g.stake[i] = g.spent[i] = 0;
        g.cards[i] = nl_draw_cards(deck, 2, seed, game, &counter);
        deck &= ~g.cards[i];
    }
    for (int b = 0; b < 2; ++b) g.force_act(NlAction{NA_BLIND, g.to_post(), 0});  // from_start (game.rs:80-85)
    uint64_t h = nl_digest(seed ^ game, g);
    int depth = 0;  // raises on this street (Path::aggression of the current street's edges)
    uint32_t steps = 0;
    for (; steps < max_steps; ++steps) {
        const int t = g.turn();
        if (t == NT_TERMINAL) break;
        if (t == NT_CHANCE) {  // NlheGame::apply(Edge::Draw) (nlhe/src/game.rs:33-53)
            const NlAction d{NA_DRAW, 0, nl_draw_cards(g.deck(), g.street() == 0 ? 3 : 1, seed, game, &counter)};
            if (!g.allowed(d)) break;
            g.force_act(d);
            depth = 0;
        } else {
            uint32_t edges[12];
            const int k = nl_choices(g, depth, edges);
            if (k == 0) break;
            const uint32_t e = edges[rp_pick_uniform(rp_node_hash(seed, 0, game, counter++), (uint32_t)k)];
            const NlAction a = g.snap(nl_actionize(g, e, 0));
            if (!g.allowed(a)) {
                h = rp_mix64(h ^ 0xbadbadbadull);
                break;
            }
            g.force_act(a);
            depth += e == NE_SHOVE || e >= NE_OPEN0;
        }
        h = nl_digest(h, g);
    }
    int reward[NL_MAXP];
    const bool done = g.turn() == NT_TERMINAL;
    if (done) nl_settle(g, reward);
    for (int i = 0; i < n; ++i) payoffs[game * n + i] = done ? (float)(reward[i] - g.spent[i]) : 0.0f;  // NlheGame::payoff
    digests[game] = h;
    steps_out[game] = done ? steps : 0xffffffffu;
}

}  // namespace rp

using namespace rp;

extern "C" int rp_nlhe_playouts(int device, uint32_t n_players, uint64_t n_games, uint64_t seed, uint32_t max_steps, float* payoffs_dev,
                                uint64_t* digests_dev, uint32_t* steps_dev) {
    if (n_players < 2 || n_players > NL_MAXP) return rp::fail(RP_ERR_INVALID, "rp_nlhe_playouts: 2..%d players", NL_MAXP);
    if ((!payoffs_dev || !digests_dev || !steps_dev) && n_games) return rp::fail(RP_ERR_INVALID, "rp_nlhe_playouts: null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return rp::fail(RP_ERR_NO_DEVICE, "no HIP device: the library has no CPU path");
    if (device < 0 || device >= count) return rp::fail(RP_ERR_INVALID, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    if (!n_games) return RP_OK;
    hipLaunchKernelGGL(k_nlhe_playouts, dim3((unsigned)((n_games + 63) / 64)), dim3(64), 0, nullptr, (int)n_players, n_games, seed, max_steps,
                       payoffs_dev, digests_dev, steps_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return RP_OK;
}

// nlhe_engine.hpp — the no-limit hold'em rules engine as device functions (crates/kicker/src/{game,seat,action,turn,
// showdown,edge,size,path}.rs, crates/nlhe/src/game.rs): the betting state machine, legal actions, the action abstraction
// and side-pot settlement.  Shared by the playout kernel that pins the rules on the device (nlhe.hip, against
// oracle/rp_oracle_nlhe.c) and by the MCCFR traversal over the game (nlmc.hip).  NlGameT<MAXP>: seats as parallel arrays
// sized for the caller (10 for the rules tests, 2 for the heads-up blueprint: everything stays in registers).
#ifndef RP_NLHE_ENGINE_HPP
#define RP_NLHE_ENGINE_HPP

#include <hip/hip_runtime.h>

#include "../../include/rp_math.h"
#include "cards.hpp"

namespace rp {

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
// FIXN > 0: the number of seats is that compile-time constant (the heads-up MCCFR traversal: every seat loop unrolls, the
// seat arrays stay in registers, `% n` is a mask); FIXN = 0: the run-time field `n` (the playout kernels, 2..MAXP seats)
template <int MAXP, int FIXN = 0>
struct NlGameT {
    static constexpr int CAP = MAXP;
    int n, dealer, ticker, pot;
    __device__ __forceinline__ int N() const { return FIXN > 0 ? FIXN : n; }
    uint64_t board;
    int state[MAXP], stack[MAXP], stake[MAXP], spent[MAXP];
    uint64_t cards[MAXP];

    // seat arrays are indexed by the actor: with two compile-time seats a select keeps them in registers (a run-time index
    // would send every game of the traversal kernels through scratch memory)
    __device__ __forceinline__ int at(const int* a, int i) const {
        if constexpr (FIXN == 2) {
            const int x0 = a[0], x1 = a[1];  // both loads first: a conditional LOAD would be folded back into an indexed one
            return i ? x1 : x0;
        } else {
            return a[i];
        }
    }
    __device__ int street() const {  // Board::street
        const int c = __popcll(board);
        return c == 0 ? 0 : (c == 3 ? 1 : (c == 4 ? 2 : 3));
    }
    __device__ int actor() const { return (dealer + ticker) % N(); }  // game.rs:656-658
    __device__ int max_stake() const {                              // :693-695
        int m = stake[0];
        _Pragma("unroll")
        for (int i = 1; i < N(); ++i) m = max(m, stake[i]);
        return m;
    }
    // ---- the closing predicates (game.rs:463-511) ----
    __device__ bool touched() const { return ticker > N() + (street() == 0 ? (N() == 2 ? 1 : 2) : 0); }
    __device__ bool matched() const {
        const int top = max_stake();
        bool ok = true;
        _Pragma("unroll")
        for (int i = 0; i < N(); ++i) ok = ok && !(state[i] == NL_BETTING && stake[i] != top);
        return ok;
    }
    __device__ int alive() const {
        int a = 0;
        _Pragma("unroll")
        for (int i = 0; i < N(); ++i) a += state[i] != NL_FOLDING;
        return a;
    }
    __device__ bool all_shoving() const {
        bool ok = true;
        _Pragma("unroll")
        for (int i = 0; i < N(); ++i) ok = ok && (state[i] == NL_FOLDING || state[i] == NL_SHOVING);
        return ok;
    }
    __device__ bool all_folding() const { return alive() == 1; }
    __device__ bool alright() const { return (touched() && matched()) || all_folding() || all_shoving(); }
    __device__ bool must_stop() const { return street() == 3 ? alright() : all_folding(); }
    __device__ bool must_deal() const { return street() != 3 && alright(); }
    __device__ bool must_post() const { return street() == 0 && pot < NL_SBLIND + NL_BBLIND; }
    __device__ int turn() const { return must_stop() ? NT_TERMINAL : (must_deal() ? NT_CHANCE : actor()); }  // :166-174
    // ---- amounts (game.rs:537-576) ----
    __device__ int to_call() const { return max_stake() - at(stake, actor()); }
    __device__ int to_post() const { return min(pot < NL_SBLIND ? NL_SBLIND : NL_BBLIND, at(stack, actor())); }
    __device__ int to_shove() const { return at(stack, actor()); }
    __device__ int to_raise() const {
        int most = 0, next = 0;
        _Pragma("unroll")
        for (int i = 0; i < N(); ++i) {
            if (state[i] == NL_FOLDING) continue;
            if (stake[i] > most) {
                next = most;
                most = stake[i];
            } else if (stake[i] > next) {
                next = stake[i];
            }
        }
        return (most - at(stake, actor())) + max(most - next, NL_BBLIND);
    }
    // ---- permissions (game.rs:513-531) ----
    __device__ bool choosing() const { return turn() >= 0; }
    __device__ bool may_fold() const { return choosing() && to_call() > 0; }
    __device__ bool may_call() const { return may_fold() && to_call() < to_shove(); }
    __device__ bool may_check() const { return choosing() && max_stake() == at(stake, actor()); }
    __device__ bool may_raise() const { return choosing() && to_raise() < to_shove(); }
    __device__ bool may_shove() const { return choosing() && to_shove() > 0; }
    __device__ uint64_t deck() const {  // :644-650
        uint64_t gone = board;
        _Pragma("unroll")
        for (int i = 0; i < N(); ++i) gone |= cards[i];
        return ~gone & HAND_MASK;
    }
    // ---- act (game.rs:395-460) ----
    __device__ void next_player() {
        if (alright()) return;
        do ticker += 1;
        while (at(state, actor()) != NL_BETTING);
    }
    __device__ void force_act(const NlAction& a) {
        const int me = actor();
        const bool pays = a.kind == NA_CALL || a.kind == NA_BLIND || a.kind == NA_RAISE || a.kind == NA_SHOVE;
        if constexpr (FIXN == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {  // the same updates as below, the seat selected instead of indexed
                const bool mine = i == me;
                if (mine && a.kind == NA_FOLD) state[i] = NL_FOLDING;
                if (mine && pays) {
                    stack[i] -= a.chips;
                    stake[i] += a.chips;
                    spent[i] += a.chips;
                    if (stack[i] == 0) state[i] = NL_SHOVING;
                }
            }
            if (pays) pot += a.chips;
        } else {
            if (a.kind == NA_FOLD) state[me] = NL_FOLDING;
            if (pays) {
                pot += a.chips;
                stack[me] -= a.chips;
                stake[me] += a.chips;
                spent[me] += a.chips;
                if (stack[me] == 0) state[me] = NL_SHOVING;
            }
        }
        if (a.kind == NA_DRAW) {
            ticker = 0;
            board |= a.cards;
        }
        next_player();
        if (a.kind == NA_DRAW)
            _Pragma("unroll")
            for (int i = 0; i < N(); ++i) stake[i] = 0;  // next_street
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
#define NL_GRID_INIT                                                                                                          \
    {{-1}, {5, 8, -1}, {5, -1}, {0, 2, 4, 5, 8, -1}, {2, 5, -1}, {5, -1}, {1, 2, 5, 8, -1}, {5, 8, -1}, {5, -1}, {1, 2, 5, 8, -1}, \
     {5, 8, -1}, {5, -1}}
__device__ __constant__ int8_t NL_GRID[12][6] = NL_GRID_INIT;
// the same grid as a compile-time table, and every row as a ready-made piece of a choices Path: the row's Raise edges
// in 5-bit fields (path.rs), their count in bits 27..29 — nl_choices_path (nlmc_common.hpp) ORs it in instead of looping
constexpr int8_t NL_GRID_CE[12][6] = NL_GRID_INIT;
constexpr uint32_t nl_grid_packed(int r) {
    uint32_t p = 0, k = 0;
    for (int i = 0; i < 6 && NL_GRID_CE[r][i] >= 0; ++i, ++k) p |= (uint32_t)(NE_RAISE0 + NL_GRID_CE[r][i]) << (5u * k);
    return p | (k << 27);
}
__device__ __constant__ uint32_t NL_GRIDP[12] = {nl_grid_packed(0), nl_grid_packed(1), nl_grid_packed(2),  nl_grid_packed(3),
                                                 nl_grid_packed(4), nl_grid_packed(5), nl_grid_packed(6),  nl_grid_packed(7),
                                                 nl_grid_packed(8), nl_grid_packed(9), nl_grid_packed(10), nl_grid_packed(11)};
constexpr uint32_t NL_OPENSP = (NE_OPEN0 + 0u) | ((NE_OPEN0 + 1u) << 5) | ((NE_OPEN0 + 2u) << 10) | ((NE_OPEN0 + 3u) << 15) | (4u << 27);
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
template <class G>
__device__ int nl_choices(const G& g, int depth, uint32_t* out) {
    int k = 0;
    if (g.must_stop() || g.must_deal() || g.must_post()) return 0;
    if (g.may_raise()) k += nl_raise_edges(g.street(), depth, out + k);
    if (g.may_shove()) out[k++] = NE_SHOVE;
    if (g.may_call()) out[k++] = NE_CALL;
    if (g.may_fold()) out[k++] = NE_FOLD;
    if (g.may_check()) out[k++] = NE_CHECK;
    return k;
}
template <class G>
__device__ NlAction nl_actionize(const G& g, uint32_t e, uint64_t draw) {  // game.rs:741-753
    switch (e) {
        case NE_FOLD: return NlAction{NA_FOLD, 0, 0};
        case NE_DRAW: return NlAction{NA_DRAW, 0, draw};
        case NE_CALL: return NlAction{NA_CALL, g.to_call(), 0};
        case NE_CHECK: return NlAction{NA_CHECK, 0, 0};
        case NE_SHOVE: return NlAction{NA_SHOVE, g.to_shove(), 0};
    }
    return NlAction{NA_RAISE, nl_edge_chips(e, g.pot), 0};
}
// Showdown::settle (showdown.rs:36-109) on the seats of a terminal game; reward[i] = chips received.  The strengths enter only
// through comparisons, so any keys with the showdown's order give the showdown's rewards (nl_settle_ranked: the level-
// synchronous traversal ranks the two hands once per river card, not once per terminal node).
template <class G>
__device__ void nl_settle_ranked(const G& g, const uint32_t* strength, int* reward);
template <class G>
__device__ void nl_settle(const G& g, int* reward) {
    uint32_t strength[G::CAP];
    // a hand's strength only ranks the seats still in the pot against each other: when everybody else has folded there is
    // nothing to rank (and most terminal nodes of a betting tree are folds), so the seven-card evaluation is skipped
    const bool showdown = g.alive() > 1;
    _Pragma("unroll")
    for (int i = 0; i < g.N(); ++i) strength[i] = showdown ? strength_key(sw_of_hand(g.cards[i] | g.board)) : 1u;
    nl_settle_ranked(g, strength, reward);
}
template <class G>
__device__ void nl_settle_ranked(const G& g, const uint32_t* strength, int* reward) {
    _Pragma("unroll")
    for (int i = 0; i < g.N(); ++i) reward[i] = 0;
    uint32_t best = 0xffffffffu;
    int distributing = 0, distributed = 0;
    for (;;) {
        bool found = false;
        uint32_t top = 0;
        _Pragma("unroll")
        for (int i = 0; i < g.N(); ++i)
            if (strength[i] < best && g.state[i] != NL_FOLDING && (!found || strength[i] > top)) {
                found = true;
                top = strength[i];
            }
        if (!found) return;
        best = top;
        for (;;) {
            distributed = distributing;
            int amount = -1;
            _Pragma("unroll")
            for (int i = 0; i < g.N(); ++i)
                if (strength[i] == best && g.spent[i] > distributed && g.state[i] != NL_FOLDING && (amount < 0 || g.spent[i] < amount))
                    amount = g.spent[i];
            if (amount < 0) break;
            distributing = amount;
            int chips = 0, nw = 0;
            _Pragma("unroll")
            for (int i = 0; i < g.N(); ++i) {
                chips += max(min(g.spent[i], distributing) - distributed, 0);
                nw += g.state[i] != NL_FOLDING && strength[i] == best && g.spent[i] > distributed;
            }
            const int share = chips / nw, bonus = chips % nw;
            int w = 0, staked = 0, paid = 0;
            _Pragma("unroll")
            for (int i = 0; i < g.N(); ++i) {
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


}  // namespace rp

#endif

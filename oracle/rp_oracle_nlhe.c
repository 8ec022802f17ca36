/* CPU oracle for the no-limit hold'em rules engine (first step: the oracle).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (robopoker_amd/, include/) includes, links or calls this file.
 * There is no device-side NLHE engine yet: this restatement and its known-answer tests (tests/test_oracle_nlhe.py,
 * the data of the reference's own unit tests in crates/kicker/src/{game,showdown}.rs) are what the engine will be
 * built against.
 *
 * A plain-C restatement of crates/kicker/src/game.rs (GameN<P>: the betting state machine, legal actions, street
 * advance, hand rotation), seat.rs, action.rs, turn.rs, pnl.rs / settlement.rs / showdown.rs (side-pot settlement),
 * written from their behaviour; every function cites the reference file:line it follows.  Chips are i16 as in the
 * reference (pokerkit/src/lib.rs:28); blinds 1 / 2, default stack 200 (lib.rs:62-66).  Hole cards and dealt boards
 * are INPUTS here (the reference draws them from its thread RNG).
 */
#include <stdint.h>
#include <string.h>

#include "rp_oracle_nlhe.h"

static int popc(uint64_t x) { return __builtin_popcountll(x); }
static int street_of(const ora_game* g) { /* Board::street: 0, 3, 4, 5 cards */
    switch (popc(g->board)) {
        case 0: return 0;
        case 3: return 1;
        case 4: return 2;
        default: return 3;
    }
}
static int actor_idx(const ora_game* g) { return (g->dealer + g->ticker) % g->n; } /* game.rs:656-658 */
static const ora_seat* actor(const ora_game* g) { return &g->seats[actor_idx(g)]; }
static int16_t max_stake(const ora_game* g) { /* game.rs:693-695 */
    int16_t m = g->seats[0].stake;
    for (int i = 1; i < g->n; ++i)
        if (g->seats[i].stake > m) m = g->seats[i].stake;
    return m;
}

/* ---- predicates (game.rs:463-532) ---------------------------------------------------------------------------- */
static int everyone_touched(const ora_game* g) { /* :489-492 */
    const int offset = g->n == 2 ? 1 : 2;
    return g->ticker > g->n + (street_of(g) == 0 ? offset : 0);
}
static int everyone_matched(const ora_game* g) { /* :494-500 */
    const int16_t stake = max_stake(g);
    for (int i = 0; i < g->n; ++i)
        if (g->seats[i].state == BETTING && g->seats[i].stake != stake) return 0;
    return 1;
}
static int everyone_shoving(const ora_game* g) { /* :502-507 */
    for (int i = 0; i < g->n; ++i)
        if (g->seats[i].state != FOLDING && g->seats[i].state != SHOVING) return 0;
    return 1;
}
static int everyone_folding(const ora_game* g) { /* :509-511 */
    int alive = 0;
    for (int i = 0; i < g->n; ++i) alive += g->seats[i].state != FOLDING;
    return alive == 1;
}
static int everyone_calling(const ora_game* g) { return everyone_touched(g) && everyone_matched(g); }
static int everyone_alright(const ora_game* g) { return everyone_calling(g) || everyone_folding(g) || everyone_shoving(g); }
static int must_stop(const ora_game* g) { return street_of(g) == 3 ? everyone_alright(g) : everyone_folding(g); } /* :465-471 */
static int must_deal(const ora_game* g) { return street_of(g) != 3 && everyone_alright(g); }                      /* :473-475 */
static int must_post(const ora_game* g) { return street_of(g) == 0 && g->pot < S_BLIND + B_BLIND; }               /* :477-479 */
static int turn_of(const ora_game* g) { /* game.rs:166-174 */
    if (must_stop(g)) return T_TERMINAL;
    if (must_deal(g)) return T_CHANCE;
    return actor_idx(g);
}
static int16_t to_call(const ora_game* g) { return (int16_t)(max_stake(g) - actor(g)->stake); } /* :537-539 */
static int16_t to_post(const ora_game* g) {                                                     /* :541-548 */
    const int16_t want = g->pot < S_BLIND ? S_BLIND : B_BLIND;
    return want < actor(g)->stack ? want : actor(g)->stack;
}
static int16_t to_shove(const ora_game* g) { return actor(g)->stack; } /* :550-552 */
static int16_t to_raise(const ora_game* g) {                           /* :556-576 */
    int16_t most = 0, next = 0;
    for (int i = 0; i < g->n; ++i) {
        if (g->seats[i].state == FOLDING) continue;
        const int16_t s = g->seats[i].stake;
        if (s > most) next = most, most = s;
        else if (s > next) next = s;
    }
    const int16_t relative = (int16_t)(most - actor(g)->stake), marginal = (int16_t)(most - next);
    return (int16_t)(relative + (marginal > B_BLIND ? marginal : B_BLIND));
}
static int is_choice(const ora_game* g) { return turn_of(g) >= 0; }
static int may_fold(const ora_game* g) { return is_choice(g) && to_call(g) > 0; }                             /* :513-515 */
static int may_call(const ora_game* g) { return is_choice(g) && may_fold(g) && to_call(g) < to_shove(g); }    /* :517-519 */
static int may_check(const ora_game* g) { return is_choice(g) && max_stake(g) == actor(g)->stake; }           /* :521-523 */
static int may_raise(const ora_game* g) { return is_choice(g) && to_raise(g) < to_shove(g); }                 /* :525-527 */
static int may_shove(const ora_game* g) { return is_choice(g) && to_shove(g) > 0; }                           /* :529-531 */

static uint64_t deck_of(const ora_game* g) { /* game.rs:644-650 */
    uint64_t removed = g->board;
    for (int i = 0; i < g->n; ++i) removed |= g->seats[i].cards;
    return ~removed & 0x000FFFFFFFFFFFFFull;
}
static int n_revealed_next(int street) { return street == 0 ? 3 : 1; } /* deuce/src/street.rs:75-82 */

/* ---- legal / is_allowed (game.rs:253-319) ------------------------------------------------------------------- */
/* Options in the reference's order: raise, shove, call, fold, check; a chance node has no enumerable action here (the
 * reference returns one random draw), a node that must post returns the blind. */
ORA_API int ora_nlhe_legal(const ora_game* g, ora_action* out) {
    int n = 0;
    if (must_stop(g)) return 0;
    if (must_deal(g)) return 0;
    if (must_post(g)) {
        out[n++] = (ora_action){A_BLIND, to_post(g), 0};
        return n;
    }
    if (may_raise(g)) out[n++] = (ora_action){A_RAISE, to_raise(g), 0};
    if (may_shove(g)) out[n++] = (ora_action){A_SHOVE, to_shove(g), 0};
    if (may_call(g)) out[n++] = (ora_action){A_CALL, to_call(g), 0};
    if (may_fold(g)) out[n++] = (ora_action){A_FOLD, 0, 0};
    if (may_check(g)) out[n++] = (ora_action){A_CHECK, 0, 0};
    return n;
}
ORA_API int ora_nlhe_is_allowed(const ora_game* g, const ora_action* a) {
    if (a->kind == A_RAISE)
        return may_raise(g) && !must_stop(g) && !must_deal(g) && a->chips >= to_raise(g) && a->chips < to_shove(g);
    if (a->kind == A_DRAW)
        return must_deal(g) && !must_stop(g) && (a->cards & ~deck_of(g)) == 0 && popc(a->cards) == n_revealed_next(street_of(g));
    ora_action opts[8];
    const int n = ora_nlhe_legal(g, opts);
    for (int i = 0; i < n; ++i)
        if (opts[i].kind == a->kind && (a->kind == A_FOLD || a->kind == A_CHECK || opts[i].chips == a->chips)) return 1;
    return 0;
}

/* ---- act (game.rs:387-461) ---------------------------------------------------------------------------------- */
static void next_player(ora_game* g) { /* :448-460 */
    if (everyone_alright(g)) return;
    for (;;) {
        g->ticker += 1;
        if (actor(g)->state == BETTING) break;
    }
}
static void force_act(ora_game* g, const ora_action* a) { /* :395-414 */
    ora_seat* s = &g->seats[actor_idx(g)];
    switch (a->kind) {
        case A_CHECK: next_player(g); break;
        case A_FOLD:
            s->state = FOLDING;
            next_player(g);
            break;
        case A_CALL: case A_BLIND: case A_RAISE: case A_SHOVE: /* bet (:416-423), allin (:425-427) */
            g->pot = (int16_t)(g->pot + a->chips);
            s->stack = (int16_t)(s->stack - a->chips);
            s->stake = (int16_t)(s->stake + a->chips);
            s->spent = (int16_t)(s->spent + a->chips);
            if (s->stack == 0) s->state = SHOVING;
            next_player(g);
            break;
        case A_DRAW: /* show (:433-436), next_player, next_street (:442-446) */
            g->ticker = 0;
            g->board |= a->cards;
            next_player(g);
            for (int i = 0; i < g->n; ++i) g->seats[i].stake = 0;
            break;
    }
}
ORA_API int ora_nlhe_apply(ora_game* g, const ora_action* a) { /* try_apply (:241-250): 0 ok, 1 illegal */
    if (!ora_nlhe_is_allowed(g, a)) return 1;
    force_act(g, a);
    return 0;
}
ORA_API void ora_nlhe_force_apply(ora_game* g, const ora_action* a) { force_act(g, a); }

/* ---- construction (game.rs:59-85) --------------------------------------------------------------------------- */
ORA_API void ora_nlhe_preblind(ora_game* g, int n, int dealer, const int16_t* stacks, const uint64_t* holes) {
    memset(g, 0, sizeof *g);
    g->n = n;
    g->dealer = dealer;
    g->ticker = n != 2;
    for (int i = 0; i < n; ++i) g->seats[i] = (ora_seat){BETTING, stacks[i], 0, 0, holes[i]};
}
static void post(ora_game* g) {
    const ora_action b = {A_BLIND, to_post(g), 0};
    force_act(g, &b);
}
ORA_API void ora_nlhe_from_start(ora_game* g, int n, int dealer, const int16_t* stacks, const uint64_t* holes) {
    ora_nlhe_preblind(g, n, dealer, stacks, holes);
    post(g);
    post(g);
}

/* ---- queries -------------------------------------------------------------------------------------------------- */
ORA_API int ora_nlhe_turn(const ora_game* g) { return turn_of(g); }
ORA_API int ora_nlhe_street(const ora_game* g) { return street_of(g); }
/* which: 0 must_stop 1 must_deal 2 must_post 3 alright 4 calling 5 touched 6 matched 7 shoving 8 folding
 *        9 may_fold 10 may_call 11 may_check 12 may_raise 13 may_shove 14 is_showdown (game.rs:618-620) */
ORA_API int ora_nlhe_predicate(const ora_game* g, int which) {
    switch (which) {
        case 0: return must_stop(g);
        case 1: return must_deal(g);
        case 2: return must_post(g);
        case 3: return everyone_alright(g);
        case 4: return everyone_calling(g);
        case 5: return everyone_touched(g);
        case 6: return everyone_matched(g);
        case 7: return everyone_shoving(g);
        case 8: return everyone_folding(g);
        case 9: return may_fold(g);
        case 10: return may_call(g);
        case 11: return may_check(g);
        case 12: return may_raise(g);
        case 13: return may_shove(g);
        case 14: {
            int active = 0;
            for (int i = 0; i < g->n; ++i) active += g->seats[i].state != FOLDING;
            return active > 1;
        }
    }
    return -1;
}
/* which: 0 to_call 1 to_post 2 to_shove 3 to_raise 4 total (:675-677) 5 effective (:682-684) */
ORA_API int ora_nlhe_amount(const ora_game* g, int which) {
    switch (which) {
        case 0: return to_call(g);
        case 1: return to_post(g);
        case 2: return to_shove(g);
        case 3: return to_raise(g);
        case 4: {
            int t = g->pot;
            for (int i = 0; i < g->n; ++i) t += g->seats[i].stack;
            return t;
        }
        case 5: {
            int e = g->seats[0].stack;
            for (int i = 1; i < g->n; ++i)
                if (g->seats[i].stack < e) e = g->seats[i].stack;
            return e;
        }
    }
    return -1;
}

/* ---- showdown (showdown.rs:36-109, settlement.rs, pnl.rs) ----------------------------------------------------- */
typedef struct {
    int32_t reward, risked, status;
    uint32_t strength;
} payout_t;
/* Showdown::settle over (risked, status, strength) triples; reward[] receives each seat's winnings */
ORA_API void ora_showdown_settle(int n, const int16_t* risked, const int32_t* status, const uint32_t* strength, int32_t* reward) {
    payout_t p[MAXP];
    for (int i = 0; i < n; ++i) p[i] = (payout_t){0, risked[i], status[i], strength[i]};
    uint32_t best = 0xffffffffu; /* Ranking::MAX */
    int32_t distributing = 0, distributed = 0;
    for (;;) {
        /* strongest (:54-62): the best hand below `best` among the players still in */
        int found = 0;
        uint32_t top = 0;
        for (int i = 0; i < n; ++i)
            if (p[i].strength < best && p[i].status != FOLDING && (!found || p[i].strength > top)) found = 1, top = p[i].strength;
        if (!found) break;
        best = top;
        for (;;) {
            /* remaining (:64-73) */
            distributed = distributing;
            int any = 0;
            int32_t amount = 0;
            for (int i = 0; i < n; ++i)
                if (p[i].strength == best && p[i].risked > distributed && p[i].status != FOLDING && (!any || p[i].risked < amount))
                    any = 1, amount = p[i].risked;
            if (!any) break;
            distributing = amount;
            /* winnings (:75-82) + distribute (:84-102) */
            int32_t chips = 0;
            for (int i = 0; i < n; ++i) {
                const int32_t s = p[i].risked < distributing ? p[i].risked : distributing;
                chips += s - distributed > 0 ? s - distributed : 0;
            }
            int winners[MAXP], nw = 0;
            for (int i = 0; i < n; ++i)
                if (p[i].status != FOLDING && p[i].strength == best && p[i].risked > distributed) winners[nw++] = i;
            const int32_t share = chips / nw, bonus = chips % nw;
            for (int w = 0; w < nw; ++w) p[winners[w]].reward += share;
            for (int w = 0; w < nw && w < bonus; ++w) p[winners[w]].reward += 1;
            /* is_complete (:104-108) */
            int32_t staked = 0, paid = 0;
            for (int i = 0; i < n; ++i) staked += p[i].risked, paid += p[i].reward;
            if (staked == paid) goto done;
        }
    }
done:
    for (int i = 0; i < n; ++i) reward[i] = p[i].reward;
}
/* GameN::settlements (game.rs:613-634): reward per seat at a terminal state; returns 1 if the state is not terminal */
ORA_API int ora_nlhe_settlements(const ora_game* g, int32_t* reward) {
    if (!must_stop(g)) return 1;
    int16_t risked[MAXP];
    int32_t status[MAXP];
    uint32_t strength[MAXP];
    for (int i = 0; i < g->n; ++i) {
        risked[i] = g->seats[i].spent;
        status[i] = g->seats[i].state;
        strength[i] = ora_strength_key(g->seats[i].cards | g->board);
    }
    ora_showdown_settle(g->n, risked, status, strength, reward);
    return 0;
}
/* GameN::continuation (game.rs:327-342 with give_chips / wipe_* / move_button :344-381): the next hand, or 0 when a
 * player could no longer post the big blind; new hole cards are inputs */
ORA_API int ora_nlhe_continuation(ora_game* g, const uint64_t* holes) {
    int32_t reward[MAXP];
    if (ora_nlhe_settlements(g, reward)) return 0;
    for (int i = 0; i < g->n; ++i)
        if (g->seats[i].stack + reward[i] < B_BLIND) return 0;
    for (int i = 0; i < g->n; ++i) {
        g->seats[i].stack = (int16_t)(g->seats[i].stack + reward[i]);
        g->seats[i].state = BETTING;
        g->seats[i].cards = holes[i];
        g->seats[i].stake = 0;
        g->seats[i].spent = 0;
    }
    g->pot = 0;
    g->board = 0;
    g->dealer = (g->dealer + 1) % g->n;
    g->ticker = g->n != 2;
    post(g);
    post(g);
    return 1;
}

/* GameN::snap (game.rs:835-854): the nearest legal action */
static ora_action passive(const ora_game* g) { return (ora_action){may_check(g) ? A_CHECK : A_FOLD, 0, 0}; }
ORA_API ora_action ora_nlhe_snap(const ora_game* g, ora_action a) {
    const ora_action shove = {A_SHOVE, to_shove(g), 0}, calls = {A_CALL, to_call(g), 0}, raise = {A_RAISE, to_raise(g), 0};
    switch (a.kind) {
        case A_RAISE:
            if (a.chips >= to_shove(g)) return ora_nlhe_snap(g, shove);
            if (!may_raise(g)) return ora_nlhe_snap(g, shove);
            if (a.chips < to_raise(g)) return raise;
            return a;
        case A_SHOVE:
            if (may_shove(g)) return shove;
            if (may_call(g)) return calls;
            return passive(g);
        case A_CALL:
            if (may_call(g)) return calls;
            if (may_shove(g)) return shove;
            return passive(g);
        case A_CHECK:
            if (may_check(g)) return a;
            if (may_call(g)) return calls;
            return (ora_action){A_FOLD, 0, 0};
        case A_FOLD:
            if (may_fold(g)) return a;
            return (ora_action){A_CHECK, 0, 0};
        default: return a;
    }
}

/* ================================================================================================================
 * The action abstraction (crates/kicker/src/edge.rs, size.rs, odds.rs, path.rs; grids in pokerkit/src/lib.rs:81-151).
 * An Edge travels as its u8 code (edge.rs:101-120): 1 Draw, 2 Fold, 3 Check, 4 Call, 5 Shove, 6..9 Open(OPENS[c-6] big
 * blinds), 10..19 Raise(RAISES[c-10] of the pot).  Pluribus regime (pokerkit/src/regime.rs:20-24, the default).
 * ================================================================================================================ */
#define MAX_RAISE_REPEATS 3 /* lib.rs:68 */
static const int16_t OPENS[4] = {2, 3, 4, 5};                                                                /* lib.rs:81 */
static const int16_t RAISES[10][2] = {{1, 4}, {1, 3}, {1, 2}, {2, 3}, {3, 4}, {1, 1}, {5, 4}, {3, 2}, {2, 1}, {3, 1}}; /* :86-97 */
static const int8_t PLURIBUS[12][6] = { /* lib.rs:138-151: indices into RAISES, -1 terminated; row = street*3 + min(depth, 2) */
    {-1}, {5, 8, -1}, {5, -1}, {0, 2, 4, 5, 8, -1}, {2, 5, -1}, {5, -1}, {1, 2, 5, 8, -1}, {5, 8, -1}, {5, -1}, {1, 2, 5, 8, -1},
    {5, 8, -1}, {5, -1}};

/* Edge::raises = Size::raises (edge.rs:77-85, size.rs:113-138): the raise edges offered at (street, depth) */
ORA_API int ora_edge_raises(int street, int depth, uint8_t* out) {
    int n = 0;
    if (depth > MAX_RAISE_REPEATS) return 0;
    if (street == 0 && depth == 0) {
        for (int i = 0; i < 4; ++i) out[n++] = (uint8_t)(E_OPEN0 + i);
        return n;
    }
    const int8_t* row = PLURIBUS[street * 3 + (depth > 2 ? 2 : depth)];
    for (int i = 0; row[i] >= 0; ++i) out[n++] = (uint8_t)(E_RAISE0 + row[i]);
    return n;
}
/* Edge::into_chips (edge.rs:86-92): Open(n) -> n big blinds; Raise(odds) -> (pot as f32 * (n as f32 / d as f32)) as i16 */
ORA_API int16_t ora_edge_into_chips(uint8_t edge, int16_t pot) {
    if (edge >= E_OPEN0 && edge < E_RAISE0) return (int16_t)(OPENS[edge - E_OPEN0] * B_BLIND);
    if (edge >= E_RAISE0 && edge < E_RAISE0 + 10) {
        const float odds = (float)RAISES[edge - E_RAISE0][0] / (float)RAISES[edge - E_RAISE0][1];
        return (int16_t)((float)pot * odds);
    }
    return 0;
}
/* From<Edge> for u64 / From<u64> for Edge (edge.rs:122-160): the `edge BIGINT` column of the blueprint table */
ORA_API uint64_t ora_edge_to_u64(uint8_t e) {
    switch (e) {
        case E_DRAW: return 0;
        case E_FOLD: return 1;
        case E_CHECK: return 2;
        case E_CALL: return 3;
        case E_SHOVE: return 5;
    }
    if (e >= E_OPEN0 && e < E_RAISE0) return 6ull | ((uint64_t)OPENS[e - E_OPEN0] << 3);
    return 4ull | ((uint64_t)RAISES[e - E_RAISE0][0] << 3) | ((uint64_t)RAISES[e - E_RAISE0][1] << 11);
}
ORA_API uint8_t ora_edge_from_u64(uint64_t v) { /* 0 = not an edge of the current grids */
    const uint32_t n = (uint32_t)(v >> 3) & 0xff, d = (uint32_t)(v >> 11) & 0xff;
    switch (v & 7) {
        case 0: return E_DRAW;
        case 1: return E_FOLD;
        case 2: return E_CHECK;
        case 3: return E_CALL;
        case 5: return E_SHOVE;
        case 6:
            for (int i = 0; i < 4; ++i)
                if (OPENS[i] == (int16_t)n) return (uint8_t)(E_OPEN0 + i);
            return 0;
        case 4:
            if (v & (1ull << 19)) { /* the old big-blind encoding (edge.rs:131-134) */
                for (int i = 0; i < 4; ++i)
                    if (OPENS[i] == (int16_t)n) return (uint8_t)(E_OPEN0 + i);
                return 0;
            }
            for (int i = 0; i < 10; ++i)
                if (RAISES[i][0] == (int16_t)n && RAISES[i][1] == (int16_t)d) return (uint8_t)(E_RAISE0 + i);
            return 0;
    }
    return 0;
}
/* Path: up to 12 edges, 5 bits each, first edge in the low bits (path.rs:146-175) */
ORA_API uint64_t ora_path_pack(const uint8_t* edges, int n) {
    uint64_t p = 0;
    for (int i = 0; i < n && i < MAX_PATH_EDGES; ++i) p |= (uint64_t)edges[i] << (5 * i);
    return p;
}
ORA_API int ora_path_unpack(uint64_t p, uint8_t* edges) { /* Iterator for Path (path.rs:128-139) */
    int n = 0;
    while (p != 0 && (p & 0x1f) != 0) {
        edges[n++] = (uint8_t)(p & 0x1f);
        p >>= 5;
    }
    return n;
}
ORA_API int ora_path_length(uint64_t p) { return (68 - (p ? __builtin_clzll(p) : 64)) / 5; } /* path.rs:9-11 */
/* Path::aggression (path.rs:12-18): raises and shoves since the last chance edge */
ORA_API int ora_path_aggression(uint64_t p) {
    uint8_t e[MAX_PATH_EDGES + 1];
    const int n = ora_path_unpack(p, e);
    int a = 0;
    for (int i = n - 1; i >= 0 && e[i] != E_DRAW; --i) a += e[i] == E_SHOVE || e[i] >= E_OPEN0;
    return a;
}
/* GameN::choices (game.rs:724-739): legal() with the raise unfolded into the grid of (street, depth), as a Path */
ORA_API uint64_t ora_nlhe_choices(const ora_game* g, int depth) {
    ora_action opts[8];
    uint8_t edges[32];
    int n = 0;
    const int k = ora_nlhe_legal(g, opts);
    for (int i = 0; i < k; ++i) {
        switch (opts[i].kind) {
            case A_RAISE: n += ora_edge_raises(street_of(g), depth, edges + n); break;
            case A_FOLD: edges[n++] = E_FOLD; break;
            case A_CHECK: edges[n++] = E_CHECK; break;
            case A_CALL: edges[n++] = E_CALL; break;
            case A_SHOVE: edges[n++] = E_SHOVE; break;
            default: break; /* blinds are not in any MCCFR tree (edge.rs:66) */
        }
    }
    return ora_path_pack(edges, n);
}
/* GameN::actionize (game.rs:741-753); a Draw edge needs the cards from the caller */
ORA_API ora_action ora_nlhe_actionize(const ora_game* g, uint8_t edge, uint64_t draw_cards) {
    switch (edge) {
        case E_FOLD: return (ora_action){A_FOLD, 0, 0};
        case E_DRAW: return (ora_action){A_DRAW, 0, draw_cards};
        case E_CALL: return (ora_action){A_CALL, to_call(g), 0};
        case E_CHECK: return (ora_action){A_CHECK, 0, 0};
        case E_SHOVE: return (ora_action){A_SHOVE, to_shove(g), 0};
    }
    return (ora_action){A_RAISE, ora_edge_into_chips(edge, g->pot), 0};
}
/* GameN::edgify with snap_to_edge (game.rs:754-766,826-833): the grid edge nearest in chips, first one on ties */
ORA_API uint8_t ora_nlhe_edgify(const ora_game* g, const ora_action* a, int depth) {
    switch (a->kind) {
        case A_FOLD: return E_FOLD;
        case A_CHECK: return E_CHECK;
        case A_DRAW: return E_DRAW;
        case A_CALL: case A_BLIND: return E_CALL;
        case A_SHOVE: return E_SHOVE;
    }
    uint8_t grid[8];
    const int n = ora_edge_raises(street_of(g), depth, grid);
    if (n == 0) return E_SHOVE;
    int best = 0, best_gap = 1 << 30;
    for (int i = 0; i < n; ++i) {
        int gap = (int)ora_edge_into_chips(grid[i], g->pot) - (int)a->chips;
        if (gap < 0) gap = -gap;
        if (gap < best_gap) best = i, best_gap = gap;
    }
    return grid[best];
}

/* ================================================================================================================
 * The NLHE instance of the solver's game / info interfaces (crates/nlhe/src/game.rs, info.rs, encoder.rs).
 * ================================================================================================================ */
/* NlheGame::apply (nlhe/src/game.rs:33-53): an abstract edge on the concrete game.  A choice edge met at a chance
 * node first deals the pending streets (the reference draws them at random: `draws` supplies them here, one entry
 * per street dealt, consumed in order); a Draw edge off a chance node is a no-op; the action is snapped to the
 * rules.  Returns the number of entries of `draws` consumed, or -1 when the snapped action is not allowed. */
ORA_API int ora_nlhe_apply_edge(ora_game* g, uint8_t edge, const uint64_t* draws) {
    int used = 0;
    if (turn_of(g) == T_TERMINAL) return 0;
    if (edge != E_DRAW) {
        while (turn_of(g) == T_CHANCE) {
            const ora_action d = {A_DRAW, 0, draws[used++]};
            force_act(g, &d);
        }
        if (turn_of(g) == T_TERMINAL) return used;
    }
    if (edge == E_DRAW && turn_of(g) != T_CHANCE) return used;
    ora_action a = ora_nlhe_actionize(g, edge, edge == E_DRAW ? draws[used] : 0);
    if (edge == E_DRAW) used += 1;
    a = ora_nlhe_snap(g, a);
    if (!ora_nlhe_is_allowed(g, &a)) return -1;
    force_act(g, &a);
    return used;
}
/* NlheGame::payoff (nlhe/src/game.rs:59-65): settlement.won() of a seat at a terminal state */
ORA_API int ora_nlhe_payoff(const ora_game* g, int seat, float* out) {
    int32_t reward[MAXP];
    if (ora_nlhe_settlements(g, reward)) return 1;
    *out = (float)(reward[seat] - g->seats[seat].spent);
    return 0;
}
/* NlheInfo::from((Path, Abstraction, Path)) (nlhe/src/info.rs:72-86) with the choices of encoder.rs:44-52 /
 * info.rs:88-103: the key is (past = the choice edges since the last chance edge, present = the bucket of the actor's
 * observation, choices = GameN::choices(past.aggression())); columns past BIGINT, present SMALLINT, choices BIGINT
 * (nlhe/src/profile.rs:20-31).  `history`: every edge since the root, chance edges included. */
ORA_API void ora_nlhe_info(const ora_game* g, uint64_t history, uint64_t* past, uint64_t* choices) {
    uint8_t e[MAX_PATH_EDGES + 1], tail[MAX_PATH_EDGES + 1];
    const int n = ora_path_unpack(history, e);
    int k = n;
    while (k > 0 && e[k - 1] != E_DRAW) --k; /* rev().take_while(is_choice) ... rev() */
    int m = 0;
    for (int i = k; i < n; ++i) tail[m++] = e[i];
    *past = ora_path_pack(tail, m);
    *choices = ora_nlhe_choices(g, ora_path_aggression(*past));
}

/* ================================================================================================================
 * The playout driver of robopoker_amd/csrc/nlhe.hip (rp_nlhe_playouts), restated over THIS file's rules: deals and
 * choices come from the shared random-number contract (include/rp_math.h: rp_node_hash(seed, 0, game, counter),
 * rp_pick_uniform), every intermediate state is folded into a digest.  Used by tests/test_gpu_nlhe.py.
 * ================================================================================================================ */
#include "../include/rp_math.h"

static uint64_t playout_draw(uint64_t deck, int k, uint64_t seed, uint64_t game, uint32_t* counter) {
    uint64_t out = 0;
    for (int c = 0; c < k; ++c) {
        const uint32_t pick = rp_pick_uniform(rp_node_hash(seed, 0, game, (*counter)++), (uint32_t)popc(deck));
        uint64_t d = deck;
        for (uint32_t s = 0; s < pick; ++s) d &= d - 1;
        const uint64_t card = d & (~d + 1);
        out |= card;
        deck &= ~card;
    }
    return out;
}
static uint64_t playout_digest(uint64_t h, const ora_game* g) {
    h = rp_mix64(h ^ ((uint64_t)(uint32_t)g->pot | (uint64_t)(uint32_t)g->ticker << 20 | (uint64_t)(uint32_t)g->dealer << 40));
    h = rp_mix64(h ^ g->board);
    for (int i = 0; i < g->n; ++i)
        h = rp_mix64(h ^ ((uint64_t)(uint32_t)g->seats[i].stack | (uint64_t)(uint32_t)g->seats[i].stake << 16 |
                          (uint64_t)(uint32_t)g->seats[i].spent << 32 | (uint64_t)(uint32_t)g->seats[i].state << 48));
    return h;
}
ORA_API void ora_nlhe_playout(int n, uint64_t game, uint64_t seed, uint32_t max_steps, float* payoffs, uint64_t* digest, uint32_t* steps_out) {
    uint32_t counter = 0;
    int16_t stacks[MAXP];
    uint64_t holes[MAXP], deck = 0x000FFFFFFFFFFFFFull;
    for (int i = 0; i < n; ++i) {
        stacks[i] = 200;
        holes[i] = playout_draw(deck, 2, seed, game, &counter);
        deck &= ~holes[i];
    }
    ora_game g;
    ora_nlhe_from_start(&g, n, (int)(game % (uint64_t)n), stacks, holes);
    uint64_t h = playout_digest(seed ^ game, &g);
    int depth = 0;
    uint32_t steps = 0;
    for (; steps < max_steps; ++steps) {
        const int t = turn_of(&g);
        if (t == T_TERMINAL) break;
        if (t == T_CHANCE) {
            const ora_action d = {A_DRAW, 0, playout_draw(deck_of(&g), street_of(&g) == 0 ? 3 : 1, seed, game, &counter)};
            if (ora_nlhe_apply(&g, &d)) break;
            depth = 0;
        } else {
            uint8_t edges[MAX_PATH_EDGES + 1];
            const int k = ora_path_unpack(ora_nlhe_choices(&g, depth), edges);
            if (k == 0) break;
            const uint8_t e = edges[rp_pick_uniform(rp_node_hash(seed, 0, game, counter++), (uint32_t)k)];
            ora_action a = ora_nlhe_snap(&g, ora_nlhe_actionize(&g, e, 0));
            if (ora_nlhe_apply(&g, &a)) {
                h = rp_mix64(h ^ 0xbadbadbadull);
                break;
            }
            depth += e == E_SHOVE || e >= E_OPEN0;
        }
        h = playout_digest(h, &g);
    }
    const int done = turn_of(&g) == T_TERMINAL;
    for (int i = 0; i < n; ++i) {
        float v = 0.0f;
        if (done) ora_nlhe_payoff(&g, i, &v);
        payoffs[i] = v;
    }
    *digest = h;
    *steps_out = done ? steps : 0xffffffffu;
}

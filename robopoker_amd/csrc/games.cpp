// games.cpp — flat rp_game_table builders for the reference's validation games.
//
// The reference expresses games as Rust generics (CfrGame/CfrTurn/CfrEdge/CfrInfo/CfrEncoder);
// the C-ABI boundary expresses them as a flat full-game-tree table (include/rp_mi355x.h,
// rp_game_table).  This file enumerates the three table-driven games the reference validates
// the mccfr crate with:
//   Kuhn   crates/kuhn/src/{game.rs:7-168, info.rs:4-76, card.rs}    6 cards, 12 infosets
//   Leduc  crates/leduc/src/{game.rs:7-245, info.rs:4-94, card.rs}   6 cards, 120 infosets
//   RPS    crates/roshambo/src/game.rs:7-78                          2 infosets, asymmetric x2
// Children of a player state are in `choices()` order; children of a chance state are in
// `deals()` order (Card::ALL order minus dealt cards).  Host-only code: no HIP here.
#include "rp_internal.h"

#include <algorithm>
#include <functional>
#include <map>
#include <string>
#include <vector>

namespace {

struct Builder {
    std::vector<rp_state> states;
    std::vector<uint32_t> children;
    std::vector<float> payoffs;
    std::vector<uint8_t> info_actions, info_player;
    std::vector<std::string> info_names;
    std::map<std::string, uint32_t> info_by_name;
    // reference-seed mode (rp_rng_kind): what `impl Hash for I` writes for each infoset / in-tree chance node's info
    std::vector<rp_hash_stream> info_streams, chance_streams;
    std::map<std::string, uint16_t> chance_by_bytes;
    uint32_t n_players = 2;

    uint32_t info(const std::string& name, uint8_t actions, uint8_t player, const rp_hash_stream& stream) {
        auto it = info_by_name.find(name);
        if (it != info_by_name.end()) return it->second;
        uint32_t id = (uint32_t)info_names.size();
        info_by_name[name] = id;
        info_names.push_back(name);
        info_actions.push_back(actions);
        info_player.push_back(player);
        info_streams.push_back(stream);
        return id;
    }
    // 1 + index of a chance node's info among the distinct ones (rp_state.chance_info)
    uint16_t chance_info(const rp_hash_stream& stream) {
        const std::string key((const char*)stream.bytes, stream.len);
        auto it = chance_by_bytes.find(key);
        if (it != chance_by_bytes.end()) return it->second;
        chance_streams.push_back(stream);
        return chance_by_bytes[key] = (uint16_t)chance_streams.size();
    }
    uint32_t terminal(float p0, float p1) {
        rp_state s{};
        s.turn = RP_TURN_TERMINAL;
        s.n_children = 0;
        s.info = RP_NO_INFO;
        s.offset = (uint32_t)(payoffs.size() / n_players);
        payoffs.push_back(p0);
        payoffs.push_back(p1);
        states.push_back(s);
        return (uint32_t)states.size() - 1;
    }
    // reserve a non-terminal state; children are filled by the caller through set_child
    uint32_t inner(uint8_t turn, uint8_t n, uint32_t info_id) {
        rp_state s{};
        s.turn = turn;
        s.n_children = n;
        s.info = info_id;
        s.offset = (uint32_t)children.size();
        children.resize(children.size() + n, 0xffffffffu);
        states.push_back(s);
        return (uint32_t)states.size() - 1;
    }
    void set_child(uint32_t parent, uint32_t k, uint32_t child) { children[states[parent].offset + k] = child; }
};

// #[derive(Hash)] streams (include/rp_refrng.h): bool = one byte; an enum's discriminant = isize, 8 bytes little-endian, before
// the variant's fields; Option<T> = discriminant 0 / 1 then T
struct HashWriter {
    rp_hash_stream s{};
    HashWriter& u8(uint8_t v) {
        s.bytes[s.len++] = v;
        return *this;
    }
    HashWriter& isize(uint64_t v) {
        for (int i = 0; i < 8; ++i) s.bytes[s.len++] = (uint8_t)(v >> (8 * i));
        return *this;
    }
    HashWriter& some(uint64_t v) { return isize(1).isize(v); }
    HashWriter& none() { return isize(0); }
};

// Card::ALL = J0 J1 Q0 Q1 K0 K1 (card.rs): two suits per rank.  The reference games have three ranks; the synthetic
// "wide" Leduc (RP_GAME_LEDUC_WIDE, same rules) has seven, which pushes the infoset count past 256.
thread_local int g_cards = 6;
thread_local const char* RANKS = "JQK";
inline int rank_of(int card) { return card / 2; }

// ------------------------------------------------------------------ Kuhn
// Node::{Open, Check, Bet, CheckBet, Over} kuhn/src/game.rs:7-15; apply :131-151; payoff :35-64
enum KuhnNode { K_OPEN, K_CHECK, K_BET, K_CHECKBET };

float kuhn_showdown(int p, int c0, int c1, bool raised) {
    float stake = raised ? 2.0f : 1.0f;
    int r0 = rank_of(c0), r1 = rank_of(c1);
    if (r0 > r1) return p == 0 ? stake : -stake;
    if (r0 < r1) return p == 1 ? stake : -stake;
    return 0.0f;
}
float kuhn_fold(int p, int who) { return who == p ? -1.0f : 1.0f; }

uint32_t kuhn_node(Builder& b, int c0, int c1, KuhnNode node) {
    static const char* HIST[] = {"", "X", "B", "XB"};
    int actor = (node == K_OPEN || node == K_CHECKBET) ? 0 : 1;
    int rank = rank_of(actor == 0 ? c0 : c1);
    std::string name = std::string(1, RANKS[rank]) + "|" + HIST[node];
    // KuhnInfo = Composite { public: KuhnPublic { acting: bool, node: History }, secret: Rank } (kuhn/src/info.rs:20-24,71;
    // mccfr/src/state/composite.rs:19-22); History::{Open, Check, Bet, CheckBet} = 0..3 (info.rs:7-12)
    uint32_t id = b.info(name, 2, (uint8_t)actor, HashWriter().u8(1).isize((uint64_t)node).isize((uint64_t)rank).s);
    uint32_t s = b.inner((uint8_t)actor, 2, id);
    switch (node) {
        case K_OPEN:  // [Check, Bet]
            b.set_child(s, 0, kuhn_node(b, c0, c1, K_CHECK));
            b.set_child(s, 1, kuhn_node(b, c0, c1, K_BET));
            break;
        case K_CHECK:  // [Check, Bet]
            b.set_child(s, 0, b.terminal(kuhn_showdown(0, c0, c1, false), kuhn_showdown(1, c0, c1, false)));
            b.set_child(s, 1, kuhn_node(b, c0, c1, K_CHECKBET));
            break;
        case K_BET:  // [Fold, Call]
            b.set_child(s, 0, b.terminal(kuhn_fold(0, 1), kuhn_fold(1, 1)));
            b.set_child(s, 1, b.terminal(kuhn_showdown(0, c0, c1, true), kuhn_showdown(1, c0, c1, true)));
            break;
        case K_CHECKBET:  // [Fold, Call]
            b.set_child(s, 0, b.terminal(kuhn_fold(0, 0), kuhn_fold(1, 0)));
            b.set_child(s, 1, b.terminal(kuhn_showdown(0, c0, c1, true), kuhn_showdown(1, c0, c1, true)));
            break;
    }
    return s;
}

// Start -> Dealt -> first decision; exploitability_root() = Start (kuhn/src/game.rs:162-167),
// root() = uniform ordered pair of distinct cards (game.rs:115-123) = the same two chance draws.
template <class F>
uint32_t deal_prefix(Builder& b, F first_decision) {
    uint32_t start = b.inner(RP_TURN_CHANCE, (uint8_t)g_cards, RP_NO_INFO);
    for (int c0 = 0; c0 < g_cards; ++c0) {
        uint32_t dealt = b.inner(RP_TURN_CHANCE, (uint8_t)(g_cards - 1), RP_NO_INFO);
        b.set_child(start, (uint32_t)c0, dealt);
        uint32_t k = 0;
        for (int c1 = 0; c1 < g_cards; ++c1) {
            if (c1 == c0) continue;
            b.set_child(dealt, k++, first_decision(c0, c1));
        }
    }
    return start;
}

// ------------------------------------------------------------------ Leduc
// Spot leduc/src/game.rs:7-12; Outcome::pot/payoff :57-110; apply :187-223; info leduc/src/info.rs:12-68
enum Spot { S_OPEN, S_CHECKED, S_RAISED, S_CHECKRAISED };
inline bool spot_raised(Spot s) { return s == S_RAISED || s == S_CHECKRAISED; }
inline int spot_actor(Spot s) { return (s == S_OPEN || s == S_CHECKRAISED) ? 0 : 1; }

void leduc_fold1(int who, float out[2]) {
    int pot[2] = {1, 1};
    pot[1 - who] += 2;
    for (int p = 0; p < 2; ++p) out[p] = (who == p) ? -(float)pot[p] : (float)pot[who];
}
void leduc_fold2(Spot r1, int who, float out[2]) {
    int base = spot_raised(r1) ? 3 : 1;
    int pot[2] = {base, base};
    pot[1 - who] += 4;
    for (int p = 0; p < 2; ++p) out[p] = (who == p) ? -(float)pot[p] : (float)pot[who];
}
void leduc_showdown(int c0, int c1, int board, Spot r1, Spot r2, float out[2]) {
    int base = spot_raised(r1) ? 3 : 1;
    int extra = spot_raised(r2) ? 4 : 0;
    int pot[2] = {base + extra, base + extra};
    int rank = rank_of(board), r0 = rank_of(c0), r1k = rank_of(c1);
    bool pair0 = r0 == rank, pair1 = r1k == rank;
    int winner;
    if (pair0 && !pair1) winner = 0;
    else if (!pair0 && pair1) winner = 1;
    else winner = r0 > r1k ? 0 : (r0 < r1k ? 1 : -1);
    for (int p = 0; p < 2; ++p) {
        if (winner < 0) out[p] = 0.0f;
        else if (winner == p) out[p] = (float)pot[1 - p];
        else out[p] = -(float)pot[p];
    }
}
// LeducPublic::subgame edge string (info.rs:37-68)
std::string leduc_hist(Spot r1, bool closed, int r2 /* -1 = none */) {
    static const char* SP[] = {"", "X", "R", "XR"};
    std::string h = SP[r1];
    if (closed) {
        if (r1 == S_CHECKED) h += "X";
        else if (r1 == S_RAISED || r1 == S_CHECKRAISED) h += "C";
    }
    if (r2 >= 0) h += SP[r2];
    return h;
}
uint32_t leduc_r2(Builder& b, int c0, int c1, int board, Spot r1, Spot r2) {
    int actor = spot_actor(r2);
    int rank = rank_of(actor == 0 ? c0 : c1);
    std::string name = std::string(1, RANKS[rank]) + "|" + std::string(1, RANKS[rank_of(board)]) + "|" +
                       leduc_hist(r1, true, (int)r2);
    // LeducInfo = Composite { public: LeducPublic { acting: bool, board: Option<Rank>, r1: Spot, r2: Option<Spot> }, secret: Rank }
    // (leduc/src/info.rs:11-17,85; encoder.rs:21-32)
    uint32_t id = b.info(name, 2, (uint8_t)actor,
                         HashWriter().u8(1).some((uint64_t)rank_of(board)).isize((uint64_t)r1).some((uint64_t)r2).isize((uint64_t)rank).s);
    uint32_t s = b.inner((uint8_t)actor, 2, id);
    float pay[2];
    switch (r2) {
        case S_OPEN:  // [Check, Raise]
            b.set_child(s, 0, leduc_r2(b, c0, c1, board, r1, S_CHECKED));
            b.set_child(s, 1, leduc_r2(b, c0, c1, board, r1, S_RAISED));
            break;
        case S_CHECKED:  // [Check, Raise]
            leduc_showdown(c0, c1, board, r1, S_CHECKED, pay);
            b.set_child(s, 0, b.terminal(pay[0], pay[1]));
            b.set_child(s, 1, leduc_r2(b, c0, c1, board, r1, S_CHECKRAISED));
            break;
        case S_RAISED:  // [Fold, Call]
            leduc_fold2(r1, 1, pay);
            b.set_child(s, 0, b.terminal(pay[0], pay[1]));
            leduc_showdown(c0, c1, board, r1, S_RAISED, pay);
            b.set_child(s, 1, b.terminal(pay[0], pay[1]));
            break;
        case S_CHECKRAISED:  // [Fold, Call]
            leduc_fold2(r1, 0, pay);
            b.set_child(s, 0, b.terminal(pay[0], pay[1]));
            leduc_showdown(c0, c1, board, r1, S_CHECKRAISED, pay);
            b.set_child(s, 1, b.terminal(pay[0], pay[1]));
            break;
    }
    return s;
}
// Node::Deal(spot): chance over the 4 undealt cards in Card::ALL order (game.rs:152-161)
uint32_t leduc_deal(Builder& b, int c0, int c1, Spot r1) {
    uint32_t s = b.inner(RP_TURN_CHANCE, (uint8_t)(g_cards - 2), RP_NO_INFO);
    // the chance node's info (encoder.rs:21-32): acting = false, actor 0's rank, no board yet, spots() = (r1, Some(Open)) (game.rs:118)
    b.states[s].chance_info = b.chance_info(HashWriter().u8(0).none().isize((uint64_t)r1).some(0).isize((uint64_t)rank_of(c0)).s);
    uint32_t k = 0;
    for (int c = 0; c < g_cards; ++c) {
        if (c == c0 || c == c1) continue;
        b.set_child(s, k++, leduc_r2(b, c0, c1, c, r1, S_OPEN));
    }
    return s;
}
uint32_t leduc_r1(Builder& b, int c0, int c1, Spot spot) {
    int actor = spot_actor(spot);
    int rank = rank_of(actor == 0 ? c0 : c1);
    std::string name = std::string(1, RANKS[rank]) + "|" + leduc_hist(spot, false, -1);
    uint32_t id = b.info(name, 2, (uint8_t)actor, HashWriter().u8(1).none().isize((uint64_t)spot).none().isize((uint64_t)rank).s);
    uint32_t s = b.inner((uint8_t)actor, 2, id);
    float pay[2];
    switch (spot) {
        case S_OPEN:
            b.set_child(s, 0, leduc_r1(b, c0, c1, S_CHECKED));
            b.set_child(s, 1, leduc_r1(b, c0, c1, S_RAISED));
            break;
        case S_CHECKED:
            b.set_child(s, 0, leduc_deal(b, c0, c1, S_CHECKED));
            b.set_child(s, 1, leduc_r1(b, c0, c1, S_CHECKRAISED));
            break;
        case S_RAISED:
            leduc_fold1(1, pay);
            b.set_child(s, 0, b.terminal(pay[0], pay[1]));
            b.set_child(s, 1, leduc_deal(b, c0, c1, S_RAISED));
            break;
        case S_CHECKRAISED:
            leduc_fold1(0, pay);
            b.set_child(s, 0, b.terminal(pay[0], pay[1]));
            b.set_child(s, 1, leduc_deal(b, c0, c1, S_CHECKRAISED));
            break;
    }
    return s;
}

// ------------------------------------------------------------------ RPS
// roshambo/src/game.rs:7-78: P1 then P2 (P2 does not observe), info = turn; scissors wins count x2
uint32_t rps_build(Builder& b) {
    // the info IS the turn: RpsTurn::{P1, P2, Terminal} (roshambo/src/turn.rs:10-18, encoder.rs:15-24)
    uint32_t i1 = b.info("P1", 3, 0, HashWriter().isize(0).s), i2 = b.info("P2", 3, 1, HashWriter().isize(1).s);
    uint32_t root = b.inner(0, 3, i1);
    const float S_WIN = 2.0f, P_WIN = 1.0f;
    // payoff to P1 for (a1, a2), R=0 P=1 S=2  (game.rs:63-73)
    float pay[3][3] = {{0.0f, -P_WIN, +S_WIN}, {+P_WIN, 0.0f, -S_WIN}, {-S_WIN, +S_WIN, 0.0f}};
    for (uint32_t a = 0; a < 3; ++a) {
        uint32_t n = b.inner(1, 3, i2);
        b.set_child(root, a, n);
        for (uint32_t c = 0; c < 3; ++c) b.set_child(n, c, b.terminal(pay[a][c], -pay[a][c]));
    }
    return root;
}

}  // namespace

struct rp_game {
    Builder b;
    rp_game_table view{};
    rp_game_kind kind;
};

namespace rp {

// longest path and the largest externally-sampled tree (walker nodes expand all children,
// other nodes exactly one), used to size per-tree scratch on the device
static void tree_bounds(const rp_game_table& t, uint32_t root, uint32_t* depth, uint32_t* nodes) {
    std::function<uint32_t(uint32_t)> dep = [&](uint32_t s) -> uint32_t {
        const rp_state& st = t.states[s];
        uint32_t d = 0;
        for (uint32_t k = 0; k < st.n_children; ++k) d = std::max(d, dep(t.children[st.offset + k]));
        return d + 1;
    };
    std::function<uint32_t(uint32_t, uint32_t)> cnt = [&](uint32_t s, uint32_t walker) -> uint32_t {
        const rp_state& st = t.states[s];
        uint32_t acc = 0;
        for (uint32_t k = 0; k < st.n_children; ++k) {
            uint32_t c = cnt(t.children[st.offset + k], walker);
            acc = (st.turn == walker) ? acc + c : std::max(acc, c);
        }
        return acc + 1;
    };
    *depth = dep(root);
    uint32_t n = 0;
    for (uint32_t w = 0; w < t.n_players; ++w) n = std::max(n, cnt(root, w));
    *nodes = n;
}

void finalize_view(rp_game* g, uint32_t train_root, uint32_t exploit_root) {
    Builder& b = g->b;
    rp_game_table& v = g->view;
    v.n_states = (uint32_t)b.states.size();
    v.n_infos = (uint32_t)b.info_names.size();
    v.n_players = b.n_players;
    uint8_t amax = 0;
    for (uint8_t a : b.info_actions) amax = std::max(amax, a);
    v.max_actions = amax;
    v.n_children = (uint32_t)b.children.size();
    v.n_terminals = (uint32_t)(b.payoffs.size() / b.n_players);
    v.train_root = train_root;
    v.exploit_root = exploit_root;
    v.states = b.states.data();
    v.children = b.children.data();
    v.payoffs = b.payoffs.data();
    v.info_actions = b.info_actions.data();
    v.info_player = b.info_player.data();
    v.default_regret = nullptr;
    uint32_t d1, n1, d2, n2;
    tree_bounds(v, train_root, &d1, &n1);
    tree_bounds(v, exploit_root, &d2, &n2);
    v.max_depth = std::max(d1, d2);
    v.max_tree_nodes = n1;
}

}  // namespace rp

extern "C" {

int rp_game_create(rp_game_kind kind, rp_game** out) {
    if (!out) return rp::fail(RP_ERR_INVALID, "rp_game_create: out is NULL");
    rp_game* g = new rp_game();
    g->kind = kind;
    uint32_t root;
    switch (kind) {
        case RP_GAME_KUHN:
            root = deal_prefix(g->b, [&](int c0, int c1) { return kuhn_node(g->b, c0, c1, K_OPEN); });
            break;
        case RP_GAME_LEDUC:
            root = deal_prefix(g->b, [&](int c0, int c1) { return leduc_r1(g->b, c0, c1, S_OPEN); });
            break;
        case RP_GAME_LEDUC_WIDE:  // synthetic: Leduc's rules over seven ranks (14 cards): 616 infosets
            g_cards = 14;
            RANKS = "789TJQK";
            root = deal_prefix(g->b, [&](int c0, int c1) { return leduc_r1(g->b, c0, c1, S_OPEN); });
            g_cards = 6;
            RANKS = "JQK";
            break;
        case RP_GAME_RPS:
            root = rps_build(g->b);
            break;
        default:
            delete g;
            return rp::fail(RP_ERR_INVALID, "rp_game_create: unknown game kind");
    }
    rp::finalize_view(g, root, root);
    *out = g;
    return RP_OK;
}

int rp_game_view(const rp_game* g, rp_game_table* out) {
    if (!g || !out) return rp::fail(RP_ERR_INVALID, "rp_game_view: NULL argument");
    *out = g->view;
    return RP_OK;
}

int rp_game_hash_streams(const rp_game* g, rp_hash_streams* out) {
    if (!g || !out) return rp::fail(RP_ERR_INVALID, "rp_game_hash_streams: NULL argument");
    out->n_infos = (uint32_t)g->b.info_streams.size();
    out->n_chance = (uint32_t)g->b.chance_streams.size();
    out->infos = g->b.info_streams.data();
    out->chance = g->b.chance_streams.data();
    return RP_OK;
}

int rp_game_destroy(rp_game* g) {
    delete g;
    return RP_OK;
}

int rp_game_info_id(const rp_game* g, const char* name, uint32_t* out) {
    if (!g || !name || !out) return rp::fail(RP_ERR_INVALID, "rp_game_info_id: NULL argument");
    auto it = g->b.info_by_name.find(name);
    if (it == g->b.info_by_name.end()) return rp::fail(RP_ERR_INVALID, "rp_game_info_id: unknown infoset name");
    *out = it->second;
    return RP_OK;
}

int rp_game_info_name(const rp_game* g, uint32_t info, char* buf, size_t cap) {
    if (!g || !buf || cap == 0) return rp::fail(RP_ERR_INVALID, "rp_game_info_name: NULL argument");
    if (info >= g->b.info_names.size()) return rp::fail(RP_ERR_INVALID, "rp_game_info_name: info out of range");
    const std::string& s = g->b.info_names[info];
    size_t n = std::min(cap - 1, s.size());
    memcpy(buf, s.data(), n);
    buf[n] = 0;
    return RP_OK;
}

int rp_game_table_check(const rp_game_table* t) {
    if (!t || !t->states || !t->children || !t->payoffs || !t->info_actions || !t->info_player)
        return rp::fail(RP_ERR_INVALID, "rp_game_table_check: NULL table field");
    if (t->n_players < 1 || t->n_players > 8) return rp::fail(RP_ERR_INVALID, "table: n_players out of range");
    if (t->max_actions < 1 || t->max_actions > RP_MAX_ACTIONS)
        return rp::fail(RP_ERR_INVALID, "table: max_actions out of range");
    if (t->train_root >= t->n_states || t->exploit_root >= t->n_states)
        return rp::fail(RP_ERR_INVALID, "table: root out of range");
    for (uint32_t s = 0; s < t->n_states; ++s) {
        const rp_state& st = t->states[s];
        if (st.turn == RP_TURN_TERMINAL) {
            if (st.n_children != 0 || st.offset >= t->n_terminals)
                return rp::fail(RP_ERR_INVALID, "table: bad terminal state");
            continue;
        }
        if (st.n_children == 0 || (uint64_t)st.offset + st.n_children > t->n_children)
            return rp::fail(RP_ERR_INVALID, "table: bad child range");
        for (uint32_t k = 0; k < st.n_children; ++k) {
            uint32_t c = t->children[st.offset + k];
            if (c >= t->n_states || c <= s) return rp::fail(RP_ERR_INVALID, "table: children must follow parents");
        }
        if (st.turn == RP_TURN_CHANCE) continue;
        if (st.turn >= t->n_players) return rp::fail(RP_ERR_INVALID, "table: bad turn");
        if (st.info >= t->n_infos) return rp::fail(RP_ERR_INVALID, "table: info out of range");
        if (t->info_actions[st.info] != st.n_children || t->info_player[st.info] != st.turn)
            return rp::fail(RP_ERR_INVALID, "table: infoset shape mismatch");
    }
    return RP_OK;
}

}  // extern "C"

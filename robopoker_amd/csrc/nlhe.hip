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
#include "nlhe_engine.hpp"
#include "rp_internal.h"

namespace rp {

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

using NlGame = NlGameT<NL_MAXP>;

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

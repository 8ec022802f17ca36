// nlmc_level.hpp — the NLHE MCCFR traversal of a whole batch, LEVEL-SYNCHRONOUS (BASELINE configs[3]; the product path).
//
// Reference: Solver::batch (mccfr/src/solver/solver.rs:225-250) = per tree TreeBuilder (builder.rs:74-161) over NlheGame
// (nlhe/src/game.rs:33-65) with NlheEncoder::info (encoder.rs:30-68), ExternalSampling / PrunableSampling / PluribusSampling
// (sample/{external.rs:17-64, pruning.rs:44-66, pluribus.rs:72-101}), Tree::partition (tree.rs:88-98) and CfrFlow::dfs per
// walker infoset (strategy/flow.rs:64-216).  Oracle: oracle/rp_oracle_nlmc.c.
//
// WHY NOT A LANE PER TREE (the first device version, round 2; deleted in round 4): a wavefront then runs as long as the
// largest of its 64 trees and every iteration splits by node kind — 5.3 of 64 lanes did work.  Here the trees of a batch grow
// together, one level of ALL trees per pair of launches, and every kernel works on nodes of ONE kind:
//
//   k_nl_roots      lane = tree      hole cards, blinds, preflop buckets -> level 0
//   k_nl_expand(L)  lane = node      a workgroup takes a tile of 512 consecutive nodes of level L and sorts it by kind in LDS
//                                    (walker | opponent | chance, each padded to whole wavefronts; terminals drop out): a
//                                    wavefront holds one kind.  Choices, NlheInfo key -> row (one 32-B slot + one row per probe),
//                                    regret matching; opponent: the sampled edge; walker: the pruning scheme's mask; children
//                                    allocated as one contiguous block per node (ONE cursor bump per tile: bumps of one
//                                    address serialise at ~8 ns, the first version's per-wavefront work lists spent 2/3 of the
//                                    step there), per child its (parent, slot) and edge factor
//   k_nl_children(L) lane = child    apply(edge) on the parent's game -> the child's game, kind, reach; chance children draw
//                                    their cards, compute BOTH seats' buckets for the new street and, on the river, rank the two
//                                    hands once — so a decision node never canonicalises cards and a terminal node settles
//                                    with integer compares; terminals get their payoff here.  No atomics, no lists
//   k_nl_up(L)      lane = node      D(node) = sum f(edge) D(child) in choices() order, subtree sizes     (L descending)
//   k_nl_down(L)    lane = node      pre-order index of every child = the reference's creation index       (L ascending)
//   k_nl_fill       lane = node      walker nodes bucketed by tree; the batch's node census
//   k_nl_group(_big) wave = tree     the tree's walker nodes sorted by (row, creation index) in LDS -> Tree::partition's spans,
//                                    in the order of their first node (trees with more than 256 walker nodes: a listed launch)
//   k_nl_emit       lane = Decisions regret vector / policy / payoff of one walker infoset of one tree -> rp_decisions
//
// Node placement (which index a child block gets) depends on timing; nothing else does: a node's children are contiguous and in
// slot order, sums run over slots, spans over creation indices, draws are hashes of (seed, epoch, tree, path).  The integer
// state equals the oracle's (tests/test_gpu_nlmc.py); the float results are the factorised evaluation's (within rtol 2e-4 of the
// oracle) or, with the chain rows (rp_nlhe_set_exact; always in k_nl_tree below), the reference's own order: bit for bit.
//
// HBM per node: 44 B of tree structure + 48 B of game state (SoA, coalesced by node index) instead of 260 KB of worst-case
// scratch per tree: ~92 B x 1 536 nodes per tree of capacity (a batch whose trees outgrow it is traversed in several passes:
// nlmc.hip nl_traverse_levels).
#ifndef RP_NLMC_LEVEL_HPP
#define RP_NLMC_LEVEL_HPP

#include "nlmc_common.hpp"
#include "sortscan.hpp"

namespace rp {

#define NL_MAXL 48u        // levels of a tree: <= 8 decisions per street, 3 draws, the terminal (observed: <= 20)
#define NL_WMAX 2048u      // walker nodes of one tree (observed: <= 455)
#define NL_LINK_NONE 0xffffffffu
// link[c] = the parent's node index (the root: NL_LINK_NONE)
// meta: kind [0,2) | n_choices [2,6) | n_kids [6,10) | depth [10,13) | path length [13,17) | showdown order [17,19) | parent kind [19,21)
#define NL_META_KIND(m) ((m) & 3u)
#define NL_META_NCH(m) (((m) >> 2) & 15u)
#define NL_META_NKIDS(m) (((m) >> 6) & 15u)
#define NL_META_DEPTH(m) (((m) >> 10) & 7u)
#define NL_META_PLEN(m) (((m) >> 13) & 15u)
#define NL_META_CMP(m) (((m) >> 17) & 3u)
#define NL_META_PKIND(m) (((m) >> 19) & 3u)  // the PARENT's kind (the root: 0), so that a walk towards the root needs one load round per step

struct NlCtl {
    uint32_t n_nodes;  // allocation cursor
    uint32_t err;
    uint32_t pad[2];
    uint32_t lvl_node[NL_MAXL + 2];  // first node of each level
    uint32_t kinds[4];               // nodes of the batch by kind (terminal, chance, walker, opponent) and ...
    uint32_t walker_kids;            // ... children of its walker nodes: what k_nl_expand's algorithmic bytes are counted from
    uint32_t n_big;                  // trees with more than 256 walker nodes (k_nl_group -> k_nl_group_big)
};
struct NlNodes {
    // tree structure, by node
    uint32_t *link, *tree, *meta, *kid0, *row, *size, *dfs, *aux;
    float *fac, *val, *reach;
    // game state, by node (decision and chance nodes only), as three 16-byte records — a node's record is read and written whole, and
    // a lane's 16-byte access is one request where five to eight 4-byte arrays were five to eight (round 6; the kernels are bound by
    // the rate of memory requests):  ga = {w0, w1, w2, board lo}   gb = {board hi, buckets of the two seats, past lo, past hi}
    // gc = {path hash lo, hi, choices path lo, hi}
    uint4 *ga, *gb, *gc;
    // by tree
    uint64_t *hole0, *hole1;
    uint32_t *t_nw, *t_woff, *t_dcount, *t_doff;
    // walker nodes by tree (unsorted / sorted), span descriptors
    uint32_t *wl, *ws, *gdesc;
    uint32_t* big;  // [batch] trees for k_nl_group_big
    NlCtl* ctl;
    uint32_t ncap, lcap;
    // k_nl_tree's evaluation in the REFERENCE'S OWN ORDER (NULL on the batch-wide path): per child sigma and q of its edge apart
    // (fsig, fq); per node and walker ancestor ("chain") the reach products from that ancestor's child down and the value summed
    // back up (ex_r, ex_s, ex_v: [node][NL_EX_K]; ex_k: chains at the node); per walker node the nine action values (wval)
    float *fsig, *fq, *ex_r, *ex_s, *ex_v, *wval;
    uint32_t* ex_k;
};
#define NL_EX_K 16u  // walker decision nodes on one root-to-leaf path (observed: <= 9; MAX_RAISE_REPEATS = 3 allows about four per street)
struct NlBatch {  // rp_decisions layout
    uint32_t* row;
    uint8_t* nact;
    uint16_t* expanded;
    float* regret;
    float* policy;
    float* payoff;
    uint32_t* tree;
};

__device__ __forceinline__ uint32_t nl_lane() { return __lane_id(); }
__device__ __forceinline__ uint32_t nl_rank_in(unsigned long long mask) {  // set bits of mask below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// with_bucket == false (a child of a chance node in k_nl_tree): the bucket word belongs to the seats' lanes, which store its halves
__device__ __forceinline__ void nl_store_game(const NlNodes& nd, uint32_t i, const G2& g, uint32_t bucket, uint64_t past, bool with_bucket = true) {
    const Packed pk = pack_game(g);
    nd.ga[i] = make_uint4(pk.w0, pk.w1, pk.w2, pk.blo);
    if (with_bucket) {
        nd.gb[i] = make_uint4(pk.bhi, bucket, (uint32_t)past, (uint32_t)(past >> 32));
    } else {
        uint32_t* w = reinterpret_cast<uint32_t*>(nd.gb + i);
        w[0] = pk.bhi;
        *reinterpret_cast<uint64_t*>(w + 2) = past;
    }
}
__device__ __forceinline__ void nl_load_game(const NlNodes& nd, uint32_t i, G2& g, uint32_t& bucket, uint64_t& past) {
    const uint4 a = nd.ga[i], b = nd.gb[i];
    unpack_game(Packed{a.x, a.y, a.z, a.w, b.x}, g);
    g.cards[0] = g.cards[1] = 0;
    bucket = b.y;
    past = (uint64_t)b.z | ((uint64_t)b.w << 32);
}
__device__ __forceinline__ uint64_t* nl_hkey_ptr(const NlNodes& nd, uint32_t i) { return reinterpret_cast<uint64_t*>(nd.gc + i); }
__device__ __forceinline__ uint64_t* nl_chpath_ptr(const NlNodes& nd, uint32_t i) { return reinterpret_cast<uint64_t*>(nd.gc + i) + 1; }
// the order of the two river hands: 1 = seat 0 stronger, 2 = equal, 3 = seat 1 stronger (0: the board is not complete)
__device__ __forceinline__ uint32_t nl_showdown_order(uint64_t hole0, uint64_t hole1, uint64_t board) {
    const uint32_t s0 = strength_key(sw_of_hand(hole0 | board)), s1 = strength_key(sw_of_hand(hole1 | board));
    return s0 > s1 ? 1u : (s0 == s1 ? 2u : 3u);
}

// ---------------------------------------------------------------------------------------------------------------
// level 0: Solver::tree — Game::root() with the hole cards dealt (P0 on the button, kicker game.rs:66-78)
// ---------------------------------------------------------------------------------------------------------------
// the root of one tree at node index `node`: hole cards, blinds, preflop buckets
__device__ __forceinline__ uint32_t nl_make_root(const NlParams& p, const NlNodes& nd, uint32_t tree, uint32_t node) {
    uint32_t err = 0;
    const uint64_t tree_id = p.tree_base + tree;
    G2 g;
    g.n = 2;
    g.dealer = 0;
    g.ticker = 0;  // n == 2: the dealer posts the small blind
    g.pot = 0;
    g.board = 0;
    uint64_t deck = HAND_MASK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        g.state[i] = NL_BETTING;
        g.stack[i] = 200;
        g.stake[i] = g.spent[i] = 0;
        g.cards[i] = nl_draw(deck, 2, p, tree_id, 0xD0C0000000000000ull + 8u * (uint64_t)i);
        deck &= ~g.cards[i];
    }
    for (int b = 0; b < 2; ++b) g.force_act(NlAction{NA_BLIND, g.to_post(), 0});
    nd.hole0[tree] = g.cards[0];
    nd.hole1[tree] = g.cards[1];
    const uint32_t b0 = nl_bucket(p, 0, g.cards[0], 0ull, &err), b1 = nl_bucket(p, 0, g.cards[1], 0ull, &err);
    const int turn = g.turn();  // a player: nobody is all-in after the blinds of a 200-chip stack
    const uint32_t kind = turn == (int)p.walker ? NK_WALKER : NK_OPP;
    nl_store_game(nd, node, g, b0 | (b1 << 16), 0ull);
    *nl_hkey_ptr(nd, node) = rp_mix64(0x726f6f74ull);
    nd.link[node] = NL_LINK_NONE;
    nd.tree[node] = tree;
    nd.meta[node] = kind;
    nd.fac[node] = 1.0f;
    nd.reach[node] = 1.0f;
    nd.val[node] = 0.0f;
    nd.dfs[node] = 0u;
    return err;
}
__global__ __launch_bounds__(256) void k_nl_roots(NlParams p, NlNodes nd) {
    const uint32_t tree = blockIdx.x * 256u + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        nd.ctl->n_nodes = p.batch;  // the roots are nodes [0, batch): children are allocated behind them
        nd.ctl->lvl_node[0] = 0;
        nd.ctl->lvl_node[1] = p.batch;
    }
    const uint32_t err = tree < p.batch ? nl_make_root(p, nd, tree, tree) : 0u;
    if (tree < p.batch && nd.ex_k) nd.ex_k[tree] = 0u;
    if (err) atomicOr(&nd.ctl->err, err);
}

// ---------------------------------------------------------------------------------------------------------------
// k_nl_expand: the nodes of one level, by kind.  encoder.info + branches + sample (builder.rs:98-139).
// ---------------------------------------------------------------------------------------------------------------
#define NL_TILE 512u  // nodes a workgroup sorts by kind at a time (measured: 512 and 1024 equal at 262 144 trees, 2048 -7 %, 4096 -13 % in
                      // k_nl_expand; at 128 trees per step 8.0 / 7.8 / 6.6 / 5.3 M updates per second: a tile is a serial chain of phases)
// One decision or chance node of a level (what a lane of k_nl_expand / k_nl_tree does for its node): choices, NlheInfo key -> row,
// regret matching, the opponent's sampled edge or the walker's surviving edges (builder.rs:98-139).  seg: 0 walker, 1 opponent,
// 2 chance.  walker_count: the tree's counter of walker nodes.  Returns n_kids | expanded mask << 4 | sampled slot << 13; aux = the
// infoset's row (walker) or the bits of sigma / q of the sampled edge (opponent).
__device__ __forceinline__ uint32_t nl_expand_item(const NlParams& p, const NlTable& t, const NlNodes& nd, uint32_t node, uint32_t seg,
                                                   bool pruning, uint32_t* walker_count, uint32_t& err, uint32_t& aux, float& osig, float& oq,
                                                   float* rf_out = nullptr, uint32_t* nch_out = nullptr /* k_nl_tree: the row's nine regrets and the
                                                   number of choices, for nl_place_children */) {
    const uint32_t m = nd.meta[node];
    if (seg == 2) {  // chance: legal() = [reveal()] -> choices = [Draw] (kicker game.rs:253-260)
        nd.meta[node] = m | (1u << 2) | (1u << 6);
        return 1u | (1u << 4);
    }
    const uint32_t tree = nd.tree[node];
    G2 g;
    uint32_t bk;
    uint64_t past;
    nl_load_game(nd, node, g, bk, past);
    const int turn = g.actor();
    const NlView view = nl_view(g);
    uint64_t chpath;
    const uint32_t nch = nl_choices_path(view, (int)NL_META_DEPTH(m), &chpath);
    const uint32_t present = turn == 0 ? (bk & 0xffffu) : (bk >> 16);
    const uint64_t khash = nl_key_hash(past, chpath, present);  // the table slot and the key of the node's random draws
    // the row is read from the key's home slot WHILE the key is probed: one memory round trip instead of two whenever the
    // infoset sits at its home slot and is older than this launch (the usual case); otherwise it is read again
    float rf[20];
    nl_load_row(t.rows, (uint32_t)khash & t.mask, seg == 1, rf);
    bool settled = true;
    const uint32_t row = nl_row_of(t, past, chpath, present, khash, nch, p.tag, &err, &settled);
    if (!settled) nl_load_row(t.rows, row, seg == 1, rf);
    const uint32_t all = (1u << nch) - 1u;
    uint32_t mask, pick = 0, nkids;
    float oppfac = 1.0f;
    if (seg == 0) {
        // walker: every edge (ExternalSampling) or the pruning scheme's survivors
        mask = all;
        if (pruning) {
            bool prune = true;
            if (p.sampling == RP_SAMPLING_PLURIBUS)  // profile.rng(node).random::<f32>() < explore (pluribus.rs:89-91)
                prune = !(nl_draw_coin(p, p.tree_base + tree, khash, past, chpath, present) < p.prune_explore);
            if (prune) {
                uint32_t keep = 0;
#pragma unroll
                for (uint32_t a = 0; a < NLMC_A; ++a) {
                    if (a >= nch) continue;
                    bool k = rf[a] > p.prune_threshold;  // cum_regret: the RAW accumulated regret (book.rs:101-106)
                    if (!k && p.sampling == RP_SAMPLING_PLURIBUS) {  // never prune an edge into a terminal node (pluribus.rs:96)
                        G2 c = g;
                        c.force_act(nl_action_v(view, (uint32_t)(chpath >> (5u * a)) & 31u));
                        k = c.turn() == NT_TERMINAL;
                    }
                    keep |= k ? (1u << a) : 0u;
                }
                mask = keep ? keep : all;  // pruning.rs:64, pluribus.rs:99
            }
        }
        nkids = (uint32_t)__popc(mask);
        const uint32_t ord = atomicAdd(walker_count, 1u);
        if (ord >= NL_WMAX) err |= NERR_WALKERS;
        nd.aux[node] = ord | (mask << 16);
        aux = row;
    } else {
        // opponent: weighted (sample/external.rs:41-64) over sampling_distribution (flow.rs:24-42), one draw per
        // (epoch, infoset, tree)
        // (registers are what limits this kernel's occupancy: the policy sigma is needed for the sampled edge only, the
        // cumulative weights are a running sum — no arrays beyond the row itself and the sampling weights)
        float sw[NLMC_A], rd = 0.0f, wsum_ = 0.0f, z = 0.0f;
#pragma unroll
        for (uint32_t a = 0; a < NLMC_A; ++a)
            if (a < nch) {
                rd += rp_maxf(rf[a], RP_EPSILON);  // RefProf::regret (profile.rs:31-33), summed in choices() order
                wsum_ += rp_maxf(rf[NLMC_A + a], RP_EPSILON);
            }
        const float denom = wsum_ + p.smoothing;
#pragma unroll
        for (uint32_t a = 0; a < NLMC_A; ++a) {
            sw[a] = 0.0f;
            if (a < nch) {
                sw[a] = rp_maxf((rp_maxf(rf[NLMC_A + a], RP_EPSILON) / p.temperature + p.smoothing) / denom, p.curiosity);
                z += sw[a];
            }
        }
        float total_w = 0.0f;
#pragma unroll
        for (uint32_t a = 0; a < NLMC_A; ++a)
            if (a < nch) total_w += rp_maxf(sw[a] / z, RP_EPSILON);
        const float u = nl_draw_weight(p, p.tree_base + tree, khash, past, chpath, present, total_w);
        float swp = sw[0], rgp = rf[0], cum = 0.0f;
#pragma unroll
        for (uint32_t a = 0; a + 1 < NLMC_A; ++a) {  // while (pick + 1 < nch && cum[pick] <= u) ++pick
            if (a < nch) cum += rp_maxf(sw[a] / z, RP_EPSILON);  // cum[a]: the same running sum as total_w
            if (pick == a && a + 1 < nch && cum <= u) {
                pick = a + 1;
                swp = sw[a + 1];
                rgp = rf[a + 1];
            }
        }
        osig = rp_maxf(rgp, RP_EPSILON) / rd;  // instant_policy of the sampled edge (flow.rs:46-48)
        oq = swp / z;                          // its sampling probability (flow.rs:33-42)
        oppfac = osig / oq;                    // sigma / q of the sampled edge
        mask = 1u << pick;
        nkids = 1;
        nd.aux[node] = mask << 16;  // k_nl_children reads the child's slot from it, as at a walker node
        aux = __float_as_uint(oppfac);
    }
    nd.row[node] = row;
    *nl_chpath_ptr(nd, node) = chpath;
    nd.meta[node] = m | (nch << 2) | (nkids << 6);
    if (rf_out) {
#pragma unroll
        for (uint32_t a = 0; a < NLMC_A; ++a) rf_out[a] = rf[a];
        *nch_out = nch;
    }
    return nkids | (mask << 4) | (pick << 13);
}
// the children of one expanded node: a contiguous block of node indices from `run`, in slot order; per child its parent and the
// factor of its edge (walker: sigma of every surviving edge, recomputed from the row with the operations of nl_expand_item;
// opponent: sigma / q of the sampled edge; chance: 1).  info / aux: what nl_expand_item returned.
__device__ __forceinline__ void nl_place_children(const NlTable& t, const NlNodes& nd, uint32_t node, uint32_t seg, uint32_t info, uint32_t run,
                                                  uint32_t aux, float osig = 1.0f, float oq = 1.0f, const float* rf_in = nullptr, uint32_t nch_in = 0) {
    const uint32_t mask = (info >> 4) & 0x1ffu;
    nd.kid0[node] = run;
    if (seg == 0) {
        float rf[12], sg[NLMC_A], rd = 0.0f;
        if (rf_in) {  // k_nl_tree: the regrets nl_expand_item read (a row does not change during a traversal)
#pragma unroll
            for (uint32_t a = 0; a < NLMC_A; ++a) rf[a] = rf_in[a];
        } else {
            nl_load_row(t.rows, aux, false, rf);
        }
        const uint32_t nch = rf_in ? nch_in : NL_META_NCH(nd.meta[node]);
#pragma unroll
        for (uint32_t a = 0; a < NLMC_A; ++a) {
            sg[a] = 0.0f;
            if (a < nch) {
                sg[a] = rp_maxf(rf[a], RP_EPSILON);
                rd += sg[a];
            }
        }
#pragma unroll
        for (uint32_t a = 0; a < NLMC_A; ++a)
            if ((mask >> a) & 1u) {
                nd.link[run] = node;
                nd.fac[run] = sg[a] / rd;  // instant_policy (flow.rs:46-48)
                if (nd.fsig) {
                    nd.fsig[run] = sg[a] / rd;
                    nd.fq[run] = 1.0f;
                }
                run += 1;
            }
    } else {
        nd.link[run] = node;
        nd.fac[run] = seg == 1 ? __uint_as_float(aux) : 1.0f;
        if (nd.fsig) {
            nd.fsig[run] = seg == 1 ? osig : 1.0f;
            nd.fq[run] = seg == 1 ? oq : 1.0f;
        }
    }
}
template <int MINW, uint32_t BT>  // minimum wavefronts per SIMD the register allocation aims for; threads per workgroup
__global__ __launch_bounds__(BT, MINW) void k_nl_expand(NlParams p, NlTable t, NlNodes nd, uint32_t level) {
    constexpr uint32_t R = NL_TILE / BT;           // classification sub-rounds
    constexpr uint32_t NW = BT / 64u;              // wavefronts of the workgroup
    constexpr uint32_t SCAP = NL_TILE + 192u;      // the sorted tile: three kinds, each from a multiple of 64
    static_assert(SCAP * NLMC_A < (1u << 15), "a tile's children must fit the 15-bit prefix kept in s_info");
    static_assert(NL_TILE % BT == 0 && BT % 64u == 0, "whole sub-rounds of whole wavefronts");
    __shared__ uint32_t sorted[SCAP];  // node index
    __shared__ uint32_t s_info[SCAP];  // n_kids | expanded mask << 4 | sampled slot << 13   (what the write phase needs)
    __shared__ uint32_t s_aux[SCAP];   // walker items: the infoset's row; opponent items: the bits of sigma / q of the sampled edge
    __shared__ float s_sig[SCAP], s_q[SCAP];  // opponent items: sigma and q apart (read only when the handle evaluates in the reference's order)
    __shared__ uint32_t wcnt[R][NW][3];  // per sub-round, wavefront, kind: count, then exclusive prefix
    __shared__ uint32_t segbase[3], segcnt[3], wsum[NW], blockbase, tiletotal;
    NlCtl* ctl = nd.ctl;
    const uint32_t lo = ctl->lvl_node[level], hi = ctl->lvl_node[level + 1];
    // An error anywhere ends the batch (the host fails the step).  The word can be raised by another workgroup of THIS launch
    // (the node budget): ONE work-item reads it for the whole workgroup — were every work-item to look for itself, part of a
    // workgroup could leave before the barriers below and the rest would sort a tile with holes (found by tests/emul)
    __shared__ uint32_t s_stop;
    if (threadIdx.x == 0) s_stop = ctl->err;
    __syncthreads();
    if (level + 1u >= NL_MAXL) {  // a tree deeper than the level table: the step fails (never observed; the rules bound the depth)
        if (threadIdx.x == 0 && hi > lo && !s_stop) atomicOr(&ctl->err, NERR_LEVELS);
        return;
    }
    if (hi <= lo || s_stop) return;
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    // pruning is live for this launch? (sample/pluribus.rs:86-88: the warm-up is on the profile's epoch)
    const bool pruning = p.sampling == RP_SAMPLING_PRUNABLE || (p.sampling == RP_SAMPLING_PLURIBUS && p.epoch >= p.prune_warmup);
    for (uint32_t t0 = lo + blockIdx.x * NL_TILE; t0 < hi; t0 += gridDim.x * NL_TILE) {
        // ---- 1. the tile sorted by kind (walker 0 | opponent 1 | chance 2; terminals have nothing to expand)
        uint32_t kd[R], rk[R];
#pragma unroll
        for (uint32_t r = 0; r < R; ++r) {
            const uint32_t i = t0 + r * BT + tid;
            const uint32_t kind = i < hi ? NL_META_KIND(nd.meta[i]) : (uint32_t)NK_TERMINAL;
            kd[r] = kind == NK_WALKER ? 0u : (kind == NK_OPP ? 1u : (kind == NK_CHANCE ? 2u : 3u));
            rk[r] = 0;
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k) {
                const unsigned long long m = __ballot(kd[r] == k);
                if (kd[r] == k) rk[r] = nl_rank_in(m);
                if ((tid & 63u) == 0) wcnt[r][wave][k] = (uint32_t)__popcll(m);
            }
        }
        __syncthreads();
        if (tid < 3) {
            uint32_t run = 0;
            for (uint32_t r = 0; r < R; ++r)
                for (uint32_t w = 0; w < NW; ++w) {
                    const uint32_t v = wcnt[r][w][tid];
                    wcnt[r][w][tid] = run;
                    run += v;
                }
            segcnt[tid] = run;
        }
        __syncthreads();
        if (tid == 0) {
            segbase[0] = 0;
            segbase[1] = (segcnt[0] + 63u) & ~63u;
            segbase[2] = segbase[1] + ((segcnt[1] + 63u) & ~63u);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < R; ++r)
            if (kd[r] < 3u) sorted[segbase[kd[r]] + wcnt[r][wave][kd[r]] + rk[r]] = t0 + r * BT + tid;
        __syncthreads();
        const uint32_t total = segbase[2] + ((segcnt[2] + 63u) & ~63u);
        // ---- 2. every item: choices, key -> row, policy, the sampled / surviving edges.  No synchronisation between rounds:
        //         the children are placed afterwards, with ONE bump of the node cursor for the whole tile
        uint32_t mykids = 0, err = 0;
        for (uint32_t jb = 0; jb < total; jb += BT) {
            const uint32_t j = jb + tid;
            const uint32_t seg = j < segbase[1] ? 0u : (j < segbase[2] ? 1u : 2u);  // wave-uniform: the segments start at multiples of 64
            const bool valid = j < total && j - segbase[seg] < segcnt[seg];
            if (!valid) {
                if (j < SCAP) s_info[j] = 0u;
                continue;
            }
            const uint32_t node = sorted[j];
            uint32_t aux = 0;
            float osig = 1.0f, oq = 1.0f;
            const uint32_t info = nl_expand_item(p, t, nd, node, seg, pruning, &nd.t_nw[nd.tree[node]], err, aux, osig, oq);
            s_info[j] = info;
            if (seg != 2) s_aux[j] = aux;
            if (seg == 1) {
                s_sig[j] = osig;
                s_q[j] = oq;
            }
            mykids += info & 15u;
        }
        // ---- 3. one contiguous run of node indices for all children of the tile, in the tile's sorted order (neighbouring
        //         parents get neighbouring child blocks: the next kernels read both): block prefix sum over s_info, ONE cursor bump
        __syncthreads();  // every s_info / s_aux of the tile is written
        constexpr uint32_t CH = (SCAP + BT - 1u) / BT;  // consecutive items per thread in the scan
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t q = 0; q < CH; ++q) {
            const uint32_t j = tid * CH + q;
            mine += j < total ? (s_info[j] & 15u) : 0u;
        }
        uint32_t incl = mine;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if (nl_lane() >= d) incl += up;
        }
        if ((tid & 63u) == 63u) wsum[wave] = incl;
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0;
            for (uint32_t w = 0; w < NW; ++w) {  // exclusive prefix over the wavefronts
                const uint32_t v = wsum[w];
                wsum[w] = tot;
                tot += v;
            }
            blockbase = tot ? atomicAdd(&ctl->n_nodes, tot) : 0u;
            tiletotal = tot;
        }
        __syncthreads();
        {
            uint32_t pre = wsum[wave] + incl - mine;  // exclusive prefix of the thread's first item: < (NL_TILE + 192) * 9 < 2^15, kept in bits 17..31
#pragma unroll
            for (uint32_t q = 0; q < CH; ++q) {
                const uint32_t j = tid * CH + q;
                if (j < total) {
                    const uint32_t info = s_info[j];
                    s_info[j] = info | (pre << 17);
                    pre += info & 15u;
                }
            }
        }
        __syncthreads();
        const bool fits = blockbase + tiletotal <= nd.ncap;  // else the batch's node budget is spent: the step fails, nothing is written
        if (!fits && mykids) err |= NERR_NODES;
        // ---- 4. per child its (parent, slot) and edge factor
        for (uint32_t jb = 0; jb < total && fits; jb += BT) {
            const uint32_t j = jb + tid;
            if (j >= total) continue;
            const uint32_t info = s_info[j], nk = info & 15u;
            if (!nk) continue;
            const uint32_t sg_ = j < segbase[1] ? 0u : (j < segbase[2] ? 1u : 2u);
            nl_place_children(t, nd, sorted[j], sg_, info, blockbase + (info >> 17), s_aux[j], sg_ == 1u ? s_sig[j] : 1.0f, sg_ == 1u ? s_q[j] : 1.0f);
        }
        if (err) atomicOr(&ctl->err, err);
        __syncthreads();  // the LDS arrays are rewritten for the next tile
    }
}

// ---- the values in the REFERENCE'S OWN ORDER (CfrFlow::recursed_value / ancestor_reach, flow.rs:166-216) — k_nl_tree always, the level kernels when
// the handle asked for it (rp_nlhe_set_exact: the chain rows are 192 B per node on top of the 92) —: for every
// walker decision node j and expanded edge a, reach(j) * recursed_value(kid_a, 1, 1), where recursed_value carries the products
// rel = ((1 * sigma_1) * sigma_2) ... and smp = ((1 * q_1) * q_2) ... from j's child DOWN to each leaf, values a leaf at
// rel / smp * payoff and sums children in choices() order.  nl_up_node multiplies the same factors bottom-up (the batch-wide path's
// form: within 2e-4).  Here every node carries one (rel, smp) pair per walker ancestor ("chain") — extended when the node is made,
// nl_ex_child — and one value per chain on the way back up, nl_ex_up_kids: the reference's float operations in the reference's order, so
// the Decisions equal the oracle's bit for bit.
// (a node's NL_EX_K slots are one 64-byte row: moved as four float4 whatever the number of live chains — the loads of a row are
// independent of each other, so a node costs one memory round trip, not one per chain; slots past ex_k hold junk nobody reads)
// (called by nl_make_child with what it already holds: the parent, its kind, whether the child is a leaf and its payoff)
__device__ __forceinline__ void nl_ex_child(const NlNodes& nd, uint32_t i, uint32_t par, uint32_t pk, bool leaf, float pay) {
    const float4* pr = reinterpret_cast<const float4*>(nd.ex_r + (size_t)par * NL_EX_K);
    const float4* ps = reinterpret_cast<const float4*>(nd.ex_s + (size_t)par * NL_EX_K);
    const uint32_t kp = nd.ex_k[par];
    const float sg = nd.fsig[i], q = nd.fq[i];  // chance: (1, 1); walker: (sigma, 1); opponent: (sigma, q)
    float r[NL_EX_K], sm[NL_EX_K];
#pragma unroll
    for (uint32_t v = 0; v < NL_EX_K / 4u; ++v) {
        const float4 a = pr[v], b = ps[v];
        r[4 * v + 0] = a.x * sg; r[4 * v + 1] = a.y * sg; r[4 * v + 2] = a.z * sg; r[4 * v + 3] = a.w * sg;
        sm[4 * v + 0] = b.x * q; sm[4 * v + 1] = b.y * q; sm[4 * v + 2] = b.z * q; sm[4 * v + 3] = b.w * q;
    }
    uint32_t k = kp;
    if (pk == NK_WALKER) {  // recursed_value(kid, 1.0, 1.0): the chain of this walker node starts at its children
        if (kp < NL_EX_K) {
#pragma unroll
            for (uint32_t c = 0; c < NL_EX_K; ++c)
                if (c == kp) {
                    r[c] = 1.0f;
                    sm[c] = 1.0f;
                }
            k = kp + 1u;
        } else {
            atomicOr(&nd.ctl->err, NERR_CHAINS);  // a 17th walker decision on one path (the rules allow ~4 per street): the step fails loudly
        }
    }
    nd.ex_k[i] = k;
    float4* wr = reinterpret_cast<float4*>(nd.ex_r + (size_t)i * NL_EX_K);
    float4* ws = reinterpret_cast<float4*>(nd.ex_s + (size_t)i * NL_EX_K);
    if (!leaf) {
#pragma unroll
        for (uint32_t v = 0; v < NL_EX_K / 4u; ++v) {
            wr[v] = make_float4(r[4 * v], r[4 * v + 1], r[4 * v + 2], r[4 * v + 3]);
            ws[v] = make_float4(sm[4 * v], sm[4 * v + 1], sm[4 * v + 2], sm[4 * v + 3]);
        }
    } else {  // terminal_value: rel / smp * payoff per chain
        float4* wv = reinterpret_cast<float4*>(nd.ex_v + (size_t)i * NL_EX_K);
#pragma unroll
        for (uint32_t v = 0; v < NL_EX_K / 4u; ++v)
            wv[v] = make_float4(r[4 * v] / sm[4 * v] * pay, r[4 * v + 1] / sm[4 * v + 1] * pay, r[4 * v + 2] / sm[4 * v + 2] * pay,
                                r[4 * v + 3] / sm[4 * v + 3] * pay);
    }
}
__device__ __forceinline__ void nl_ex_up_kids(const NlNodes& nd, uint32_t i, uint32_t nk, uint32_t k0) {
    float sum[NL_EX_K];
#pragma unroll
    for (uint32_t c = 0; c < NL_EX_K; ++c) sum[c] = 0.0f;
    // children in choices() order, every chain's sum in that order; three children's rows are in flight at a time (a loop that waits
    // for each child's row in turn costs a memory round trip per child: up to nine per level)
    for (uint32_t ch0 = 0; ch0 < nk; ch0 += 3u) {
        float4 a[3][NL_EX_K / 4u];
#pragma unroll
        for (uint32_t u = 0; u < 3u; ++u)
            if (ch0 + u < nk) {
                const float4* cv = reinterpret_cast<const float4*>(nd.ex_v + (size_t)(k0 + ch0 + u) * NL_EX_K);
#pragma unroll
                for (uint32_t v = 0; v < NL_EX_K / 4u; ++v) a[u][v] = cv[v];
            }
#pragma unroll
        for (uint32_t u = 0; u < 3u; ++u)
            if (ch0 + u < nk) {
#pragma unroll
                for (uint32_t v = 0; v < NL_EX_K / 4u; ++v) {
                    sum[4 * v] += a[u][v].x; sum[4 * v + 1] += a[u][v].y; sum[4 * v + 2] += a[u][v].z; sum[4 * v + 3] += a[u][v].w;
                }
            }
    }
    float4* wv = reinterpret_cast<float4*>(nd.ex_v + (size_t)i * NL_EX_K);
#pragma unroll
    for (uint32_t v = 0; v < NL_EX_K / 4u; ++v) wv[v] = make_float4(sum[4 * v], sum[4 * v + 1], sum[4 * v + 2], sum[4 * v + 3]);
}
// the nine action values of walker node i: ancestor_reach upward over the opponent's decisions, nearest first, times the chain
// this node's children start
__device__ __forceinline__ void nl_ex_walker(const NlNodes& nd, uint32_t i, uint32_t slot0 /* the tree's first walker slot */, uint32_t WC) {
    float cf = 1.0f, sm = 1.0f;
    for (uint32_t x = i;;) {
        const uint32_t par = nd.link[x], mx = nd.meta[x];
        const float sg = nd.fsig[x], q = nd.fq[x];  // all four loads depend on x only: one round trip per step
        if (par == NL_LINK_NONE) break;
        if (NL_META_PKIND(mx) == NK_OPP) {
            cf = cf * sg;
            sm = sm * q;
        }
        x = par;
    }
    const float reach = cf / sm;
    const uint32_t ord = nd.aux[i] & 0xffffu, em = nd.aux[i] >> 16, k0 = nd.kid0[i], kc = nd.ex_k[i];
    if (ord >= WC || kc >= NL_EX_K) return;
    uint32_t rank = 0;
    for (uint32_t a = 0; a < NLMC_A; ++a) {
        float v = 0.0f;
        if ((em >> a) & 1u) {
            v = reach * nd.ex_v[(size_t)(k0 + rank) * NL_EX_K + kc];
            rank += 1;
        }
        nd.wval[((size_t)slot0 + ord) * NLMC_A + a] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_nl_children: NlheGame::apply(edge) (nlhe/src/game.rs:33-53) for every child of the level, the child's node
// ---------------------------------------------------------------------------------------------------------------
// one child: NlheGame::apply(edge) on its parent's state, the child's node record.  Returns error bits.
// PART (k_nl_tree splits the child of a chance node over several lanes: the deal's two canonicalisations and the hand ranking are the
// longest lane-serial stretch of a level otherwise):
//   NL_CHILD_ALL     everything (k_nl_children)
//   NL_CHILD_PLAIN   everything, but a child of a chance node is left to the three parts below
//   NL_CHILD_DEAL    child of a chance node: everything but the buckets (the draw, apply, the showdown order on the river, the record)
//   NL_CHILD_SEAT0/1 child of a chance node: that seat's bucket only (the draw again — a hash — and one canonicalisation), stored as its
//                    half of the bucket word
enum : uint32_t { NL_CHILD_ALL = 0, NL_CHILD_PLAIN = 1, NL_CHILD_DEAL = 2, NL_CHILD_SEAT0 = 3, NL_CHILD_SEAT1 = 4 };
__device__ __forceinline__ uint32_t nl_make_child(const NlParams& p, const NlNodes& nd, uint32_t c, int walker, uint32_t part = NL_CHILD_ALL) {
    uint32_t err = 0;
    const uint32_t par = nd.link[c];
    const uint32_t pm = nd.meta[par], pkind = NL_META_KIND(pm);
    if (part == NL_CHILD_PLAIN && pkind == NK_CHANCE) return 0u;
    // the child's slot among the parent's choices: children are stored in slot order, so it is the (c - kid0)-th expanded
    // edge of the parent's mask
    uint32_t slot = 0;
    if (pkind != NK_CHANCE) {
        uint32_t mbits = nd.aux[par] >> 16;
        for (uint32_t r = c - nd.kid0[par]; r > 0; --r) mbits &= mbits - 1u;
        slot = (uint32_t)__builtin_ctz(mbits);
    }
    const uint32_t tree = nd.tree[par];
    G2 g;
    uint32_t bucket;
    uint64_t ppast;
    nl_load_game(nd, par, g, bucket, ppast);
    const uint4 pc = nd.gc[par];
    const uint64_t phk = (uint64_t)pc.x | ((uint64_t)pc.y << 32), pch = (uint64_t)pc.z | ((uint64_t)pc.w << 32);
    const uint32_t e = pkind == NK_CHANCE ? (uint32_t)NE_DRAW : (uint32_t)(pch >> (5u * slot)) & 31u;
    const uint64_t hk = rp_mix64(phk ^ ((uint64_t)(e + 1u) * 0x9fb21c651e98df25ull));
    uint32_t cmp = NL_META_CMP(pm), cdepth, cplen;
    uint64_t cpast;
    if (e == NE_DRAW) {
        const uint64_t h0 = nd.hole0[tree], h1 = nd.hole1[tree];
        g.cards[0] = h0;
        g.cards[1] = h1;
        const NlAction act{NA_DRAW, 0, nl_draw(g.deck(), g.street() == 0 ? 3 : 1, p, p.tree_base + tree, hk)};
        if (p.check_legal && !g.allowed(act)) err |= NERR_ILLEGAL;
        g.force_act(act);
        const int st = g.street();
        if (part == NL_CHILD_SEAT0 || part == NL_CHILD_SEAT1) {
            const uint32_t seat = part - NL_CHILD_SEAT0;
            (reinterpret_cast<uint16_t*>(nd.gb + c) + 2)[seat] = (uint16_t)nl_bucket(p, st, seat ? h1 : h0, g.board, &err);
            return err;
        }
        if (part == NL_CHILD_ALL) bucket = nl_bucket(p, st, h0, g.board, &err) | (nl_bucket(p, st, h1, g.board, &err) << 16);
        if (st == 3) cmp = nl_showdown_order(h0, h1, g.board);
        cdepth = 0;
        cplen = 0;
        cpast = 0ull;
    } else {
        const NlView view = nl_view(g);
        const NlAction act = nl_action_v(view, e);
        // Game::apply panics on an illegal action (kicker game.rs:234-247).  snap()'s output is legal by construction, so the
        // test can only catch an engine bug: it runs in the checking mode (RP_NLHE_CHECK_LEGAL=1, the tests)
        if (p.check_legal && !g.allowed(act)) err |= NERR_ILLEGAL;
        g.force_act(act);
        const uint32_t pdepth = NL_META_DEPTH(pm), pplen = NL_META_PLEN(pm);
        const bool raise = e == NE_SHOVE || e >= NE_OPEN0;
        cdepth = pdepth + (raise ? 1u : 0u);
        cplen = pplen + 1u;
        cpast = pplen < 12u ? ppast | ((uint64_t)e << (5u * pplen)) : ppast;
    }
    const int turn = g.turn();
    nd.tree[c] = tree;
    const float f = nd.fac[c], pr = nd.reach[par];
    nd.reach[c] = pkind == NK_OPP ? pr * f : pr;  // ancestor_reach (flow.rs:166-174): sigma / q over the opponent's edges
    if (turn == NT_TERMINAL) {
        // NlheGame::payoff (nlhe/src/game.rs:59-65): settlement minus what the walker put in; the showdown order was
        // fixed when the river card fell
        int reward[2];
        const uint32_t strength[2] = {cmp == 1u ? 2u : 1u, cmp == 3u ? 2u : 1u};
        nl_settle_ranked(g, strength, reward);
        const float pay = (float)(walker ? reward[1] - g.spent[1] : reward[0] - g.spent[0]);
        nd.val[c] = pay;
        nd.meta[c] = NK_TERMINAL | (pkind << 19);
        if (nd.ex_k) nl_ex_child(nd, c, par, pkind, true, pay);
    } else {
        const uint32_t kind = turn == NT_CHANCE ? NK_CHANCE : (turn == walker ? NK_WALKER : NK_OPP);
        nl_store_game(nd, c, g, bucket, cpast, part != NL_CHILD_DEAL);  // NL_CHILD_DEAL: the bucket's two halves come from the seats' lanes
        *nl_hkey_ptr(nd, c) = hk;
        nd.meta[c] = kind | ((cdepth & 7u) << 10) | ((cplen & 15u) << 13) | (cmp << 17) | (pkind << 19);
        nd.val[c] = 0.0f;
        if (nd.ex_k) nl_ex_child(nd, c, par, pkind, false, 0.0f);  // the parent's chains are a launch (a level) old
    }
    return err;
}
__global__ __launch_bounds__(256) void k_nl_children(NlParams p, NlNodes nd, uint32_t level) {
    NlCtl* ctl = nd.ctl;
    const uint32_t lo = ctl->lvl_node[level + 1], hi = min(ctl->n_nodes, nd.ncap);
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->lvl_node[level + 2] = hi;
    if (hi <= lo || ctl->err) return;
    const uint32_t total = (hi - lo + 63u) & ~63u;
    const int walker = (int)p.walker;
    for (uint32_t base = blockIdx.x * 256u; base < total; base += gridDim.x * 256u) {
        const uint32_t c = lo + base + threadIdx.x;
        const uint32_t err = c < hi ? nl_make_child(p, nd, c, walker) : 0u;
        if (err) atomicOr(&ctl->err, err);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// evaluation sweeps
// ---------------------------------------------------------------------------------------------------------------
// D(node) = sum over the expanded children, in choices() order, of f(edge) D(child): the factorised form of
// CfrFlow::recursed_value (flow.rs:182-216), which multiplies the same factors at the leaves; subtree sizes beside it
__device__ __forceinline__ void nl_ex_up_kids(const NlNodes& nd, uint32_t i, uint32_t nk, uint32_t k0);
__device__ __forceinline__ uint32_t nl_up_node(const NlNodes& nd, uint32_t i) {  // returns the node's subtree size
    const uint32_t nk = NL_META_NKIDS(nd.meta[i]);
    uint32_t sz = 1;
    if (nk) {
        const uint32_t k0 = nd.kid0[i];
        float sum = 0.0f;
        for (uint32_t c = 0; c < nk; ++c) {
            sum += nd.fac[k0 + c] * nd.val[k0 + c];
            sz += nd.size[k0 + c];
        }
        if (nd.ex_k) nl_ex_up_kids(nd, i, nk, k0);  // the chains' sums beside it (the reference's order), before any store of this node
        nd.val[i] = sum;
    }
    nd.size[i] = sz;
    return sz;
}
__global__ __launch_bounds__(256) void k_nl_up(NlNodes nd, uint32_t level) {
    const uint32_t lo = nd.ctl->lvl_node[level], hi = nd.ctl->lvl_node[level + 1];
    for (uint32_t i = lo + blockIdx.x * 256u + threadIdx.x; i < hi; i += gridDim.x * 256u) {
        const uint32_t sz = nl_up_node(nd, i);
        if (level == 0 && sz >= 65536u) atomicOr(&nd.ctl->err, NERR_NODES);  // creation indices are sort keys of 16 bits
    }
}
// the reference's creation index: pop-last DFS (builder.rs:141-161) visits the children of a node from the LAST choice to the
// first, each with its whole subtree
__device__ __forceinline__ void nl_down_node(const NlNodes& nd, uint32_t i) {
    const uint32_t nk = NL_META_NKIDS(nd.meta[i]);
    if (!nk) return;
    const uint32_t k0 = nd.kid0[i];
    uint32_t run = nd.dfs[i] + 1u;
    for (uint32_t c = nk; c-- > 0;) {
        nd.dfs[k0 + c] = run;
        run += nd.size[k0 + c];
    }
}
__global__ __launch_bounds__(256) void k_nl_down(NlNodes nd, uint32_t level) {
    const uint32_t lo = nd.ctl->lvl_node[level], hi = nd.ctl->lvl_node[level + 1];
    for (uint32_t i = lo + blockIdx.x * 256u + threadIdx.x; i < hi; i += gridDim.x * 256u) nl_down_node(nd, i);
}


// exclusive scan of per-tree counts (one workgroup; a batch has at most a few 10^5 trees)
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

// walker nodes by tree: wl[t_woff[tree] + ordinal]; the batch's node census on the way (one pass over every node's meta)
__global__ __launch_bounds__(256) void k_nl_fill(NlNodes nd, uint32_t n_nodes) {
    __shared__ uint32_t census[5];
    if (threadIdx.x < 5) census[threadIdx.x] = 0;
    __syncthreads();
    uint32_t c[4] = {0, 0, 0, 0}, wk = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_nodes; i += gridDim.x * 256u) {
        const uint32_t m = nd.meta[i], kind = NL_META_KIND(m);
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) c[k] += kind == k ? 1u : 0u;
        if (kind == NK_WALKER) {
            wk += NL_META_NKIDS(m);
            const uint32_t woff = nd.t_woff[nd.tree[i]], at = woff + (nd.aux[i] & 0xffffu);
            if (at < nd.lcap) nd.wl[at] = i;  // more walker nodes than the arrays hold: the host refuses the batch (their total)
            if (nd.ex_k && NL_META_NKIDS(m) != 0u && woff < nd.lcap) nl_ex_walker(nd, i, woff, nd.lcap - woff);  // every chain value is in place
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (c[k]) atomicAdd(&census[k], c[k]);
    if (wk) atomicAdd(&census[4], wk);
    __syncthreads();
    if (threadIdx.x < 4 && census[threadIdx.x]) atomicAdd(&nd.ctl->kinds[threadIdx.x], census[threadIdx.x]);
    if (threadIdx.x == 4 && census[4]) atomicAdd(&nd.ctl->walker_kids, census[4]);
}

// Tree::partition (tree.rs:88-98) for one tree per wavefront: walker nodes sorted by (row, creation index) -> spans; spans
// ordered by their first node.
// the partition of one tree by one wavefront (LDS arrays of capacity CAP >= n)
// NT work-items (one wavefront in k_nl_group / k_nl_group_big, the whole workgroup of k_nl_tree) — `t` = the work-item's index
template <uint32_t CAP, uint32_t NT>
__device__ __forceinline__ void nl_group_tree(const NlNodes& nd, uint32_t tree, uint32_t n, uint32_t t, uint64_t* key, uint32_t* key2,
                                              uint16_t* hp, uint32_t* sG) {
    const uint32_t off = nd.t_woff[tree];
    // wl / ws / gdesc hold lcap entries: a pass with more walker nodes than that is a spent node budget (the host retries it in
    // more, smaller passes); the whole workgroup leaves together, before the first barrier, and nothing past the arrays is touched
    if (off + n > nd.lcap) {
        if (t == 0) {
            nd.t_dcount[tree] = 0;
            atomicOr(&nd.ctl->err, NERR_NODES);
        }
        return;
    }
    uint32_t P = 64;
    while (P < n) P <<= 1;
    for (uint32_t i = t; i < P; i += NT) {
        uint64_t k = ~0ull;
        if (i < n) {
            const uint32_t node = nd.wl[off + i];
            k = ((uint64_t)nd.row[node] << 32) | ((uint64_t)(nd.dfs[node] & 0xffffu) << 16) | (uint64_t)i;
        }
        key[i] = k;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t tix = t; tix < (P >> 1); tix += NT) {
                const uint32_t i = ((tix & ~(j - 1u)) << 1) | (tix & (j - 1u)), x = i | j;
                const uint64_t a = key[i], b = key[x];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    key[i] = b;
                    key[x] = a;
                }
            }
            __syncthreads();
        }
    // span heads in row order (the first wavefront: the running count is a wavefront ballot)
    if (t < 64u) {
        uint32_t G = 0;
        for (uint32_t base = 0; base < P; base += 64) {
            const uint32_t i = base + t;
            const bool head = i < n && (i == 0 || (uint32_t)(key[i] >> 32) != (uint32_t)(key[i - 1] >> 32));
            const unsigned long long m = __ballot(head);
            if (head) {
                const uint32_t g = G + nl_rank_in(m);
                hp[g] = (uint16_t)i;
                key2[g] = (((uint32_t)(key[i] >> 16) & 0xffffu) << 16) | g;  // first creation index of the span | its rank by row
            }
            G += (uint32_t)__popcll(m);
        }
        if (t == 0) {
            hp[G] = (uint16_t)n;
            *sG = G;
        }
    }
    __syncthreads();
    const uint32_t G = *sG;
    uint32_t P2 = 64;
    while (P2 < G) P2 <<= 1;
    for (uint32_t i = G + t; i < P2; i += NT) key2[i] = 0xffffffffu;
    __syncthreads();
    for (uint32_t k = 2; k <= P2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t tix = t; tix < (P2 >> 1); tix += NT) {
                const uint32_t i = ((tix & ~(j - 1u)) << 1) | (tix & (j - 1u)), x = i | j;
                const uint32_t a = key2[i], b = key2[x];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    key2[i] = b;
                    key2[x] = a;
                }
            }
            __syncthreads();
        }
    for (uint32_t i = t; i < n; i += NT) nd.ws[off + i] = nd.wl[off + (uint32_t)(key[i] & 0xffffu)];
    for (uint32_t g = t; g < G; g += NT) {
        const uint32_t r = key2[g] & 0xffffu;
        const uint32_t start = hp[r], len = (uint32_t)hp[r + 1] - start;
        nd.gdesc[off + g] = start | (len << 12);
    }
    if (t == 0) nd.t_dcount[tree] = G;
}
// every tree of the batch, one per workgroup of one wavefront: trees with at most CAP walker nodes are partitioned here (the
// usual case: 40 to 110 per tree), the larger ones are put on a list for k_nl_group_big (whose 28 KB of LDS per workgroup would
// otherwise gate the launch of 262 144 mostly idle workgroups: 0.67 ms per step)
template <uint32_t CAP>
__global__ __launch_bounds__(64) void k_nl_group(NlNodes nd, uint32_t batch) {
    __shared__ uint64_t key[CAP];
    __shared__ uint32_t key2[CAP];
    __shared__ uint16_t hp[CAP + 2];
    __shared__ uint32_t sG;
    const uint32_t tree = blockIdx.x, lane = threadIdx.x;
    const uint32_t n = nd.t_nw[tree];
    if (n == 0 || n > CAP) {
        if (lane == 0) {
            if (n == 0 || n > NL_WMAX) nd.t_dcount[tree] = 0;  // nothing to do for an empty tree or one k_nl_expand has flagged
            else nd.big[atomicAdd(&nd.ctl->n_big, 1u)] = tree;
        }
        return;
    }
    nl_group_tree<CAP, 64>(nd, tree, n, lane, key, key2, hp, &sG);
}
__global__ __launch_bounds__(64) void k_nl_group_big(NlNodes nd) {
    __shared__ uint64_t key[NL_WMAX];
    __shared__ uint32_t key2[NL_WMAX];
    __shared__ uint16_t hp[NL_WMAX + 2];
    __shared__ uint32_t sG;
    const uint32_t n_big = nd.ctl->n_big;
    for (uint32_t b = blockIdx.x; b < n_big; b += gridDim.x) {
        const uint32_t tree = nd.big[b];
        nl_group_tree<NL_WMAX, 64>(nd, tree, nd.t_nw[tree], threadIdx.x, key, key2, hp, &sG);
        __syncthreads();
    }
}

// one Decisions per lane: record_infosets + update_vector (solver.rs:263-305) for one walker infoset of one tree.  Walks the
// slots of the walker-node array; the first t_dcount[tree] slots of a tree stand for its spans.  Slots and Decisions are both
// numbered tree-major, so the active lanes of a wavefront own a CONTIGUOUS run of Decisions: the [n][9] regret / policy rows are
// staged in LDS by active rank and stored as whole 64-lane lines instead of 18 stores of stride 36 B.
__global__ __launch_bounds__(256) void k_nl_emit(NlNodes nd, NlTable t, uint32_t n, uint32_t d_base, uint32_t tree_off, uint32_t out_cap,
                                                 NlBatch out, uint32_t wc, const float* wval) {  // d_base / tree_off: this pass' first Decisions slot and first tree of the batch
                                                                                // wc != 0: every tree owns wc walker slots (k_nl_tree); unused ones hold nothing
    __shared__ float tile[4][2][64 * NLMC_A];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t padded = (n + 255u) & ~255u;
    // wc != 0 (k_nl_tree's fixed regions of wc slots per tree, of which a tree's first t_dcount stand for its spans — ~70 of 2 048): a
    // workgroup walks the trees, not the slots
    const uint32_t step = wc ? gridDim.x * wc : gridDim.x * 256u;
    for (uint32_t j0 = wc ? blockIdx.x * wc : blockIdx.x * 256u; j0 < padded; j0 += step)
    for (uint32_t sub = 0, nsub = wc && j0 < n ? (nd.t_dcount[j0 / wc] + 255u) & ~255u : 256u; sub < nsub; sub += 256u) {
        const uint32_t j = j0 + sub + threadIdx.x;
        bool active = false;
        uint32_t d = 0, tr = 0, off = 0, g = 0;
        if (j < n) {
            tr = wc ? j / wc : nd.tree[nd.ws[j]];
            off = nd.t_woff[tr];
            g = j - off;
            if (g < nd.t_dcount[tr]) {
                d = d_base + nd.t_doff[tr] + g;
                active = d < out_cap;  // beyond: the host has already refused the batch
            }
        }
        float acc[NLMC_A], pol[NLMC_A];
        if (active) {
            const uint32_t desc = nd.gdesc[j], start = desc & 0xfffu, len = desc >> 12;
            const uint32_t first = nd.ws[off + start];
            const uint32_t row = nd.row[first], nch = NL_META_NCH(nd.meta[first]);
            float rf[12], sg[NLMC_A], rd = 0.0f;
            nl_load_row(t.rows, row, false, rf);
#pragma unroll
            for (uint32_t a = 0; a < NLMC_A; ++a) {
                acc[a] = 0.0f;
                sg[a] = 0.0f;
                if (a < nch) {
                    sg[a] = rp_maxf(rf[a], RP_EPSILON);
                    rd += sg[a];
                }
            }
            float pay = 0.0f;
            uint32_t expanded = 0;
            for (uint32_t mb = 0; mb < len; ++mb) {  // the span in ascending creation index
                const uint32_t node = nd.ws[off + start + mb];
                const uint32_t em = nd.aux[node] >> 16, k0 = nd.kid0[node];
                const float reach = nd.reach[node];
                const float* wv = wval ? wval + ((size_t)off + (nd.aux[node] & 0xffffu)) * NLMC_A : nullptr;  // k_nl_tree: reach * recursed_value
                float cfv[NLMC_A], ev = 0.0f;
#pragma unroll
                for (uint32_t a = 0; a < NLMC_A; ++a) {
                    cfv[a] = 0.0f;
                    if ((em >> a) & 1u) {
                        cfv[a] = wv ? wv[a] : reach * nd.val[k0 + (uint32_t)__popc(em & ((1u << a) - 1u))];
                        ev += sg[a] / rd * cfv[a];
                    }
                }
                pay += ev;
#pragma unroll
                for (uint32_t a = 0; a < NLMC_A; ++a)
                    if ((em >> a) & 1u) acc[a] += cfv[a] - ev;
                expanded |= em;
            }
#pragma unroll
            for (uint32_t a = 0; a < NLMC_A; ++a) pol[a] = a < nch ? sg[a] / rd : 0.0f;  // policy_vector = iterated_distribution (flow.rs:118-120)
            out.row[d] = row;
            out.nact[d] = (uint8_t)nch;
            out.expanded[d] = (uint16_t)expanded;
            out.payoff[d] = pay;
            out.tree[d] = tr + tree_off;
        }
        const unsigned long long am = __ballot(active);
        if (am) {
            const uint32_t nact = (uint32_t)__popcll(am), rank = nl_rank_in(am);
            const uint32_t d0 = (uint32_t)__shfl((int)d, __builtin_ctzll(am));  // Decisions d0 .. d0 + nact - 1, in lane order
            if (active) {
#pragma unroll
                for (uint32_t a = 0; a < NLMC_A; ++a) {
                    tile[wave][0][rank * NLMC_A + a] = acc[a];
                    tile[wave][1][rank * NLMC_A + a] = pol[a];
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t e = lane; e < nact * NLMC_A; e += 64u) {
                out.regret[(size_t)d0 * NLMC_A + e] = tile[wave][0][e];
                out.policy[(size_t)d0 * NLMC_A + e] = tile[wave][1][e];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_nl_tree: ONE TREE PER WORKGROUP — the traversal of a SMALL batch (the reference's own: 128 trees per step,
// nlhe/src/solver.rs:11) in one launch.  The level-synchronous kernels above pay two launches per tree level and one per sweep
// level whatever the batch: ~85 launches of 3 - 14 us each for a step whose levels hold a few thousand nodes.  Here a workgroup
// grows its tree in its own region of the node arrays [tree * C, (tree + 1) * C), level by level, with workgroup barriers where the
// batch-wide path has kernel boundaries: the same per-node functions (nl_make_root, nl_expand_item, nl_place_children,
// nl_make_child, nl_up_node, nl_down_node), children of a level allocated in node order from the region's cursor; then both sweeps
// and the tree's walker list (a fixed region of WC slots per tree: no scan across trees).  Placement differs from the batch-wide
// path, nothing else does (children contiguous and in slot order, sums over slots, spans over creation indices), so the Decisions
// are the same bit for bit.  A tree that outgrows its region raises NERR_NODES / NERR_WALKERS and the host repeats the step on the
// batch-wide path.  ctl: err, n_nodes (sum over trees), pad[0] (deepest tree), the census.
// ---------------------------------------------------------------------------------------------------------------
#define NL_TREE_BATCH 2048u  // batches up to this many trees take k_nl_tree
#define NL_TREE_CAP 4096u  // nodes of a tree's region (observed: <= 2 900 on fresh tables; a blueprint's trees are pruned smaller); a tree that
                           // outgrows it raises NERR_NODES and the handle moves to the batch-wide kernels.  (8 192 until round 5: 4.8 GB at 2 048 trees)
#ifdef NL_TREE_PROF  // a diagnostic build (scripts/r6_nltree_prof.sh): 10 ns ticks per phase, summed over the trees of every launch
__device__ unsigned long long g_nl_prof[16];
__device__ unsigned int g_nl_rec[2048 * 16];  // the last launch, per tree: ticks by phase [0..8], levels, nodes, walker nodes
#define NLP(k)                                          \
    do {                                                \
        if (threadIdx.x == 0) {                         \
            const unsigned long long _t = wall_clock64(); \
            atomicAdd(&g_nl_prof[k], _t - nlp_t0);      \
            if (blockIdx.x < 2048u) g_nl_rec[blockIdx.x * 16u + (k)] += (unsigned int)(_t - nlp_t0); \
            nlp_t0 = _t;                                \
        }                                               \
    } while (0)
#else
#define NLP(k)
#endif
// What the host must know between a small batch's traversal and its update (the number of Decisions sizes the update's launches):
// written into pinned host memory by the last workgroup of k_nl_tree, `seq` last; the host waits on that word.
struct NlPost {
    uint32_t seq, total, err, n_nodes, levels, kinds[4], walker_kids;
};
// Every exit of k_nl_tree, by the whole workgroup: the workgroup that counts in last (ctl->pad[1]) scans the trees' Decisions counts
// into their offsets, posts the batch's total and control block to the host and clears the block for the next step.
template <uint32_t BT>
__device__ __forceinline__ void nl_tree_exit(const NlNodes& nd, uint32_t B, uint32_t* total_out, NlPost* post, uint32_t seq) {
    __shared__ uint64_t x_wt[BT / 64u];
    __shared__ uint32_t x_last;
    __syncthreads();  // this workgroup's stores to the per-tree words are issued
    if (threadIdx.x == 0) {
        __threadfence();
        x_last = atomicAdd(&nd.ctl->pad[1], 1u) == B - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!x_last) return;
    __threadfence();  // the other workgroups' words
    const uint32_t per = (B + BT - 1u) / BT, lo = min(B, threadIdx.x * per), hi = min(B, lo + per);
    uint64_t s = 0;
    for (uint32_t t = lo; t < hi; ++t) s += __hip_atomic_load(&nd.t_dcount[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t tot;
    uint32_t run = (uint32_t)ss::block_exscan64(s, x_wt, &tot);
    for (uint32_t t = lo; t < hi; ++t) {
        nd.t_doff[t] = run;
        run += __hip_atomic_load(&nd.t_dcount[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
        NlCtl* ctl = nd.ctl;
        *total_out = (uint32_t)tot;
        post->total = (uint32_t)tot;
        post->err = __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        post->n_nodes = __hip_atomic_load(&ctl->n_nodes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        post->levels = __hip_atomic_load(&ctl->pad[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = 0; k < 4; ++k) post->kinds[k] = __hip_atomic_load(&ctl->kinds[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        post->walker_kids = __hip_atomic_load(&ctl->walker_kids, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ctl->err = 0;
        ctl->n_nodes = 0;
        ctl->pad[0] = 0;
        ctl->pad[1] = 0;
        for (int k = 0; k < 4; ++k) ctl->kinds[k] = 0;
        ctl->walker_kids = 0;
        __threadfence_system();
        __hip_atomic_store(&post->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#define NL_TREE_CC 1024u  // children of chance nodes in one level of one tree (observed: a few dozen); more raise NERR_NODES like a full region
// BT: threads of the workgroup (one workgroup per CU at this batch, so a wide one costs nothing and shortens wide levels).
// (Measured and not kept, round 6: a variant with the sampling scheme, the draw and the legality check as compile-time constants —
// 14 008 -> 11 025 instructions, 137 -> 117 registers — is 2 % SLOWER in an A/B on one box: the level loop is not bound by the
// instruction cache.)
template <uint32_t BT>
__global__ __launch_bounds__(BT) void k_nl_tree(NlParams p, NlTable t, NlNodes nd, uint32_t C, uint32_t WC, uint32_t* total_out, NlPost* post,
                                                uint32_t seq) {
    constexpr uint32_t NW = BT / 64u;
#ifdef NL_TREE_PROF
    unsigned long long nlp_t0 = wall_clock64();
    if (threadIdx.x < 16 && blockIdx.x < 2048u) g_nl_rec[blockIdx.x * 16u + threadIdx.x] = 0u;
    __syncthreads();
#endif
    __shared__ uint32_t lvl[NL_MAXL + 2];
    __shared__ uint32_t s_cursor, s_err, s_nw, s_wsum[NW], s_census[5];
    __shared__ uint32_t s_cc[NL_TREE_CC], s_ncc;  // the children of this level's chance nodes: dealt by three lanes each (nl_make_child's parts)
    const uint32_t tree = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t base = tree * C;
    if (tid == 0) {
        s_err = nl_make_root(p, nd, tree, base);
        if (nd.ex_k) nd.ex_k[base] = 0u;
        s_cursor = base + 1u;
        s_nw = 0;
        s_ncc = 0;
        lvl[0] = base;
        lvl[1] = base + 1u;
    }
    if (tid < 5) s_census[tid] = 0;
    __syncthreads();
    NLP(0);
    const bool pruning = p.sampling == RP_SAMPLING_PRUNABLE || (p.sampling == RP_SAMPLING_PLURIBUS && p.epoch >= p.prune_warmup);
    uint32_t levels = 0, err = 0;
    for (uint32_t L = 0;; ++L) {
        const uint32_t lo = lvl[L], hi = lvl[L + 1];
        if (hi == lo) break;  // an empty level ends the tree
        levels = L + 1u;
        if (L + 2u >= NL_MAXL) {  // deeper than the level table (never observed: the rules bound the depth)
            err |= NERR_LEVELS;
            break;
        }
        // ---- encoder.info + branches + sample for the level's nodes, BT at a time; children in node order from the cursor
        for (uint32_t c0 = lo; c0 < hi; c0 += BT) {
            const uint32_t i = c0 + tid;
            uint32_t info = 0, aux = 0, seg = 3, nch = 0;
            float osig = 1.0f, oq = 1.0f, rf[NLMC_A];
            if (i < hi) {
                const uint32_t kind = NL_META_KIND(nd.meta[i]);
                seg = kind == NK_WALKER ? 0u : (kind == NK_OPP ? 1u : (kind == NK_CHANCE ? 2u : 3u));
                if (seg < 3u) info = nl_expand_item(p, t, nd, i, seg, pruning, &s_nw, err, aux, osig, oq, rf, &nch);
            }
            const uint32_t nk = info & 15u;
            uint32_t incl = nk;
#pragma unroll
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                if (nl_lane() >= d) incl += up;
            }
            if (lane == 63u) s_wsum[wave] = incl;
            __syncthreads();
            NLP(1);
            const uint32_t cur = s_cursor;
            uint32_t wpre = 0, tot = 0;
#pragma unroll
            for (uint32_t w = 0; w < NW; ++w) {
                const uint32_t v = s_wsum[w];
                wpre += w < wave ? v : 0u;
                tot += v;
            }
            const bool fits = cur + tot <= base + C;  // workgroup uniform
            if (!fits) err |= NERR_NODES;
            const uint32_t run = cur + wpre + incl - nk;
            if (nk && fits) nl_place_children(t, nd, i, seg, info, run, aux, osig, oq, rf, nch);
            {   // the level's chance nodes hand their (one) child to the deal lanes
                const bool cc = seg == 2u && nk && fits;
                const unsigned long long cm = __ballot(cc);
                if (cm) {
                    uint32_t at = 0;
                    if (lane == 0) at = atomicAdd(&s_ncc, (uint32_t)__popcll(cm));
                    at = (uint32_t)__shfl((int)at, 0, 64) + nl_rank_in(cm);
                    if (cc) {
                        if (at < NL_TREE_CC) s_cc[at] = run;
                        else err |= NERR_NODES;
                    }
                }
            }
            __syncthreads();  // every work-item has read the cursor and the wavefront sums
            NLP(2);
            if (tid == 0 && fits) s_cursor = cur + tot;
            if (!fits) break;
        }
        if (err) atomicOr(&s_err, err);
        __syncthreads();  // the cursor, the children's (parent, factor), the error word, the deal list
        if (s_err) break;
        // ---- NlheGame::apply(edge) for the level's children: the next level's nodes.  Wavefront-sized tasks, the long ones first: per
        // 64 children of chance nodes one task for each seat's bucket and one for the rest of the deal; per 64 children one for the others.
        // Typical level: one task of each kind, one per wavefront.
        const uint32_t chi = s_cursor, ncc = min(s_ncc, NL_TREE_CC);
        if (tid == 0) lvl[L + 2] = chi;
        {
            const uint32_t tb = (ncc + 63u) >> 6, ta = (chi - hi + 63u) >> 6;
            for (uint32_t wt = wave; wt < 3u * tb + ta; wt += NW) {
                uint32_t c = NL_LINK_NONE, part = NL_CHILD_PLAIN;  // (one call site: the function is a third of the kernel's code)
                if (wt < 3u * tb) {
                    const uint32_t q = wt / tb, j = (wt - q * tb) * 64u + lane;  // 0 / 1: the seats, 2: the rest
                    if (j < ncc) c = s_cc[j];
                    part = q == 2u ? (uint32_t)NL_CHILD_DEAL : (uint32_t)NL_CHILD_SEAT0 + q;
                } else {
                    const uint32_t cc = hi + (wt - 3u * tb) * 64u + lane;
                    if (cc < chi) c = cc;
                }
                if (c != NL_LINK_NONE) err |= nl_make_child(p, nd, c, (int)p.walker, part);
            }
        }
        if (err) atomicOr(&s_err, err);
        __syncthreads();  // the node records and the level table are read by other work-items from here on
        if (tid == 0) s_ncc = 0;  // (read again only after the next level's barriers)
        NLP(3);
        if (s_err) break;
    }
    if (err) atomicOr(&s_err, err);
    __syncthreads();
    if (s_err) {
        if (tid == 0) {
            atomicOr(&nd.ctl->err, s_err);
            nd.t_nw[tree] = 0;
            nd.t_woff[tree] = tree * WC;
            nd.t_dcount[tree] = 0;
        }
        nl_tree_exit<BT>(nd, gridDim.x, total_out, post, seq);
        return;
    }
    // ---- the sweeps: D(node) and subtree sizes bottom-up, creation indices top-down
    for (uint32_t l = levels; l-- > 0;) {
        for (uint32_t i = lvl[l] + tid; i < lvl[l + 1]; i += BT) {
            const uint32_t sz = nl_up_node(nd, i);
            if (l == 0 && sz >= 65536u) atomicOr(&nd.ctl->err, NERR_NODES);
        }
        __syncthreads();
    }
    NLP(4);
    // The creation indices go top-down on the FIRST wavefront alone (a level is a store -> load dependency inside one wavefront: a fence,
    // no workgroup barrier) while the other wavefronts compute the walker nodes' action values (every chain value is in place; read by
    // k_nl_emit, the next launch), list the walker nodes by ordinal (k_nl_fill's part) and count the census: none of these reads dfs.
    const uint32_t end = s_cursor, woff = tree * WC;
    uint32_t c4[4] = {0, 0, 0, 0}, wk = 0;
    if (wave == 0) {
        for (uint32_t l = 0; l + 1 < levels; ++l) {
            for (uint32_t i = lvl[l] + lane; i < lvl[l + 1]; i += 64u) nl_down_node(nd, i);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        for (uint32_t i = base + tid - 64u; i < end; i += BT - 64u) {
            const uint32_t m = nd.meta[i], kind = NL_META_KIND(m);
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) c4[k] += kind == k ? 1u : 0u;
            if (kind == NK_WALKER) {
                wk += NL_META_NKIDS(m);
                const uint32_t ord = nd.aux[i] & 0xffffu;
                if (ord < WC) nd.wl[woff + ord] = i;
                if (nd.ex_k && NL_META_NKIDS(m) != 0u) nl_ex_walker(nd, i, woff, WC);
            }
        }
    }
    NLP(6);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (c4[k]) atomicAdd(&s_census[k], c4[k]);
    if (wk) atomicAdd(&s_census[4], wk);
    __syncthreads();
    if (tid < 4 && s_census[tid]) atomicAdd(&nd.ctl->kinds[tid], s_census[tid]);
    if (tid == 4 && s_census[4]) atomicAdd(&nd.ctl->walker_kids, s_census[4]);
    NLP(7);
    if (tid == 0) {
        nd.t_nw[tree] = s_nw;
        nd.t_woff[tree] = woff;
        atomicAdd(&nd.ctl->n_nodes, end - base);
        atomicMax(&nd.ctl->pad[0], levels);
        if (s_nw > WC) atomicOr(&nd.ctl->err, NERR_WALKERS);
    }
    // ---- Tree::partition for this tree (k_nl_group's part; the key arrays overlay nothing: 28 KB of LDS for NL_WMAX walker nodes)
    __shared__ uint64_t g_key[NL_WMAX];
    __shared__ uint32_t g_key2[NL_WMAX];
    __shared__ uint16_t g_hp[NL_WMAX + 2];
    __shared__ uint32_t g_count;
    const uint32_t nw = s_nw;  // workgroup uniform (the barrier above)
    if (nw == 0 || nw > WC) {
        if (tid == 0) nd.t_dcount[tree] = 0;
        nl_tree_exit<BT>(nd, gridDim.x, total_out, post, seq);
        return;
    }
    __syncthreads();  // t_woff[tree] and the walker list are read by other work-items below
    nl_group_tree<NL_WMAX, BT>(nd, tree, nw, tid, g_key, g_key2, g_hp, &g_count);
    __syncthreads();
    NLP(8);
#ifdef NL_TREE_PROF
    if (tid == 0) {
        atomicAdd(&g_nl_prof[9], 1ull);
        atomicAdd(&g_nl_prof[10], (unsigned long long)levels);
        atomicAdd(&g_nl_prof[11], (unsigned long long)(end - base));
        if (blockIdx.x < 2048u) {
            g_nl_rec[blockIdx.x * 16u + 9] = levels;
            g_nl_rec[blockIdx.x * 16u + 10] = end - base;
            g_nl_rec[blockIdx.x * 16u + 11] = nw;
        }
    }
#endif
    nl_tree_exit<BT>(nd, gridDim.x, total_out, post, seq);
}

}  // namespace rp

#endif

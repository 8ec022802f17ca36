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
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "nlmc_level.hpp"
#include "rp_internal.h"
#include "sortscan.hpp"

namespace rp {

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ---- the multi-GPU exchange by infoset key (every rank's table assigns rows in its own insertion order) ----
// entries: rp_profile_summarize's records, [row u32][count u32][psum f32][n_actions u32][maps]; keys beside them
__global__ __launch_bounds__(256) void k_nlhe_entry_keys(NlTable t, const unsigned char* entries, uint32_t eb, uint32_t n, uint64_t* past,
                                                         uint32_t* present, uint64_t* choices) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t row = *reinterpret_cast<const uint32_t*>(entries + (size_t)i * eb);
    past[i] = t.slots[row].past;
    present[i] = t.slots[row].present;
    choices[i] = t.slots[row].choices;
}
__global__ __launch_bounds__(256) void k_nlhe_entry_remap(NlTable t, unsigned char* entries, uint32_t eb, uint32_t n, const uint64_t* past,
                                                          const uint32_t* present, const uint64_t* choices, uint32_t tag, uint32_t* errflags) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t nch = 0, err = 0;
    for (uint64_t c = choices[i]; nch < 12u && (c & 0x1full) != 0; c >>= 5) nch += 1;
    const uint32_t row = nl_row_of(t, past[i], choices[i], present[i], nl_key_hash(past[i], choices[i], present[i]), nch, tag, &err);
    *reinterpret_cast<uint32_t*>(entries + (size_t)i * eb) = row;
    if (err) atomicOr(errflags, err);
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
    NlNodes lv{};                // level-synchronous traversal
    NlBatch out{};
    uint32_t out_cap = 0;
    uint32_t* d_offset = nullptr;
    uint32_t* d_total = nullptr;   // [0] Decisions of the batch, [1] walker nodes of the batch
    void* d_scan = nullptr;                    // scratch of the per-tree scans
    void *x_keys = nullptr, *x_counts = nullptr, *x_all = nullptr, *x_packed = nullptr;  // rp_nlhe_step_comm: grow-only exchange buffers
    size_t x_keys_bytes = 0, x_counts_bytes = 0, x_all_bytes = 0, x_packed_bytes = 0;
    uint32_t* d_remap_err = nullptr;           // rp_nlhe_step_apply: table full while inserting exchanged keys
    std::vector<void*> allocs;
    uint32_t last_n = 0;
    uint32_t tag = 0;              // launch tags handed to nl_row_of (never 0)
    uint64_t nodes = 0, infos = 0;  // Metrics (mccfr/src/metrics/mod.rs): nodes grown, Decisions recorded
    uint32_t last_levels = 0, last_nodes = 0;
    // profiling (rp_nlhe_profile): HIP event pairs around the launches of each kernel group, on the launch stream
    struct Clock {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
        double total_ms = 0.0;
        uint64_t launches = 0;
    };
    bool profiling = false;
    Clock clk[5];  // expand, children, sweeps (up + down), decide (scan + fill + group + emit), apply
    uint64_t census[5] = {0, 0, 0, 0, 0};  // nodes by kind + walker children, summed over the profiled steps
    // small batches (the reference's 128): one tree per workgroup, the whole traversal in one launch (k_nl_tree); tree_cap = nodes of a
    // tree's region (0: the node arrays were not sized for it), level_ncap = the batch-wide path's own node budget (its launch sizes)
    uint32_t tree_cap = 0, level_ncap = 0;
    NlPost* post = nullptr;      // pinned, mapped host memory (k_nl_finish)
    NlPost* post_dev = nullptr;  // the same words as the device addresses them
    uint32_t post_seq = 0;
    bool ctl_clean = false;      // the control block is zero (k_nl_finish cleared it; the batch-wide path leaves it dirty)
    uint32_t tree_bt = 512;  // k_nl_tree's workgroup (RP_NL_TREE_BT = 256 / 1024: the experiment; measured round 6: 256 / 512 / 1024 = 0.483 / 0.428 / 0.443 ms per step)
    bool tree_mode_off = false;  // a tree outgrew its region once: the handle stays on the batch-wide path
    uint32_t* ex_k_parked = nullptr;  // rp_nlhe_set_exact(h, 0) on a large-batch handle: lv.ex_k while the exact evaluation is off
    uint32_t chunks = 1;  // passes per batch (RP_NLHE_CHUNKS; doubled when a pass runs out of nodes)
    uint32_t grid_cap = 16384;  // workgroups of the grid-stride kernels (measured: 1024 -14 %, 4096 -4 %)
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
// the arrays of the evaluation in the reference's own order (nlmc_level.hpp nl_ex_*): per node sigma and q of its edge, a 64-byte row of
// chains for each of rel / smp / value, per walker slot its nine action values.  192 B per node on top of the 92.
int nl_alloc_exact(rp_nlhe* h) {
    NlNodes& lv = h->lv;
    if (lv.ex_k) return RP_OK;
    const size_t N = lv.ncap;
    int rc;
    if ((rc = nl_alloc(h, &lv.fsig, N)) || (rc = nl_alloc(h, &lv.fq, N)) || (rc = nl_alloc(h, &lv.ex_r, N * NL_EX_K)) ||
        (rc = nl_alloc(h, &lv.ex_s, N * NL_EX_K)) || (rc = nl_alloc(h, &lv.ex_v, N * NL_EX_K)) ||
        (rc = nl_alloc(h, &lv.wval, (size_t)lv.lcap * NLMC_A)))
        return rc;
    return nl_alloc(h, &lv.ex_k, N);  // last: the kernels take a non-NULL ex_k as "all of them are there"
}
uint32_t nl_next_tag(rp_nlhe* h) {
    h->tag += 1;
    if (h->tag == 0) h->tag = 1;
    return h->tag;
}
int nl_capacity_error(unsigned long long flags) {
    return rp::fail(RP_ERR_CAPACITY, "rp_nlhe: traversal failed (flags %llu: 1 node budget, 2 stack, 4 walker nodes of a tree, 8 decisions, "
                                     "16 illegal action, 32 infoset table full, 64 isomorphism not found in the encoder table, 128 tree deeper "
                                     "than the level table, 256 work list full, 512 more than 16 walker decisions on one path with the exact evaluation on)", flags);
}
void nl_clock_begin(rp_nlhe* h, int k) {
    if (!h->profiling) return;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, rp::profile_stream(h->prof));
    h->clk[k].pending.emplace_back(a, b);
}
void nl_clock_end(rp_nlhe* h, int k) {
    if (!h->profiling) return;
    (void)hipEventRecord(h->clk[k].pending.back().second, rp::profile_stream(h->prof));
    h->clk[k].launches += 1;
}
void nl_clock_drain(rp_nlhe* h) {
    for (auto& c : h->clk) {
        for (auto& pr : c.pending) {
            float ms = 0.0f;
            (void)hipEventSynchronize(pr.second);
            (void)hipEventElapsedTime(&ms, pr.first, pr.second);
            c.total_ms += ms;
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        c.pending.clear();
    }
}
void nl_begin_step(rp_nlhe* h) {
    h->prm.epoch = rp::profile_epoch(h->prof);
    h->prm.walker = (uint32_t)(h->prm.epoch % 2u);  // CfrSampling::walker (book.rs:142-144)
    h->prm.step_hash = rp_node_hash_step(h->prm.seed, h->prm.epoch);
    if (h->prm.ref_rng) {  // DefaultHasher::new(); self.t().hash(hasher)  (flow.rs:289-291)
        rp_sip s;
        rp_defaulthasher_new(&s);
        rp_defaulthasher_write_u64(&s, h->prm.epoch);
        h->prm.ref_v[0] = s.v0; h->prm.ref_v[1] = s.v1; h->prm.ref_v[2] = s.v2; h->prm.ref_v[3] = s.v3;
    }
}

// waits for k_nl_finish's post of this step (h->post_seq).  Looks at the stream now and then: a failed launch never posts.
int nl_wait_post(rp_nlhe* h, hipStream_t st) {
    const volatile uint32_t* seq = &h->post->seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 1;; ++spins) {
        if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == h->post_seq) return RP_OK;
        if ((spins & 0x3fffu) == 0) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) {  // everything launched has run: the word is there, or the kernel did not run
                if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == h->post_seq) return RP_OK;
                return rp::fail(RP_ERR_HIP, "rp_nlhe: the traversal finished without posting its counts");
            }
            if (q != hipErrorNotReady) return rp::fail(RP_ERR_HIP, "rp_nlhe: %s while waiting for the traversal", hipGetErrorString(q));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0)
                return rp::fail(RP_ERR_HIP, "rp_nlhe: no post from the traversal after 30 s");
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

// ---- the level-synchronous traversal (nlmc_level.hpp)
// the trees [lo, lo + B) of the batch, grown and evaluated together; their Decisions go behind the `d_base` already emitted.
// *flags: the traversal's error flags when the return code is RP_ERR_CAPACITY (the caller retries a spent node budget in chunks)
int nl_traverse_chunk(rp_nlhe* h, uint32_t lo, uint32_t B, uint32_t d_base, uint32_t* n_dec, uint32_t* n_nod, uint32_t* n_lev, uint32_t* flags) {
    hipStream_t st = rp::profile_stream(h->prof);
    NlNodes& lv = h->lv;
    NlParams prm = h->prm;
    prm.batch = B;
    prm.tree_base = h->prm.tree_base + lo;
    *flags = 0;
    const dim3 wide(std::min<uint32_t>(h->grid_cap, std::max<uint32_t>(1u, (h->level_ncap / 4u + 255u) / 256u))), blk(256);
    const dim3 wide_x(std::min<uint32_t>(h->grid_cap * 2u, std::max<uint32_t>(1u, (h->level_ncap / 4u + NL_TILE - 1u) / NL_TILE)));  // one tile per workgroup
    if (h->tree_cap && !h->tree_mode_off && lo == 0 && B == h->batch) {
        // ---- a small batch: one tree per workgroup, one launch (k_nl_tree), then the partition and the Decisions as below
        const uint32_t WC = NL_WMAX;
        if (!h->ctl_clean) HIP_TRY(hipMemsetAsync(lv.ctl, 0, sizeof(NlCtl), st));
        h->ctl_clean = false;
        prm.tag = nl_next_tag(h);
        nl_clock_begin(h, 0);
        // the workgroup that finishes last scans the trees' Decisions counts and posts the batch's total and control block to the host
        // through pinned memory: the host spins on the sequence word (a scan launch + two copies + hipStreamSynchronize until round 6:
        // three launches and an interrupt round trip per step)
        h->post_seq += 1;
        if (h->post_seq == 0) h->post_seq = 1;
        if (h->tree_bt == 256u) hipLaunchKernelGGL(k_nl_tree<256>, dim3(B), blk, 0, st, prm, h->tab, lv, h->tree_cap, WC, h->d_total, h->post_dev, h->post_seq);
        else if (h->tree_bt == 1024u) hipLaunchKernelGGL(k_nl_tree<1024>, dim3(B), dim3(1024), 0, st, prm, h->tab, lv, h->tree_cap, WC, h->d_total, h->post_dev, h->post_seq);
        else hipLaunchKernelGGL(k_nl_tree<512>, dim3(B), dim3(512), 0, st, prm, h->tab, lv, h->tree_cap, WC, h->d_total, h->post_dev, h->post_seq);
        nl_clock_end(h, 0);
        nl_clock_begin(h, 3);
        // the Decisions are emitted behind the traversal without waiting for the host (nothing of the launch depends on the counts; a
        // tree that failed has no Decisions, and the host discards the buffer of a failed step): the post's way to the host and the
        // update's launches hide behind this kernel
        hipLaunchKernelGGL(k_nl_emit, dim3(B), blk, 0, st, lv, h->tab, B * WC, d_base, lo, h->out_cap, h->out, WC, (const float*)lv.wval);
        HIP_TRY(hipGetLastError());
        {
            int rcw = nl_wait_post(h, st);
            if (rcw) return rcw;
        }
        h->ctl_clean = true;
        const uint32_t total0 = h->post->total;
        NlCtl ctl{};
        ctl.err = h->post->err;
        ctl.n_nodes = h->post->n_nodes;
        ctl.pad[0] = h->post->levels;
        for (int k = 0; k < 4; ++k) ctl.kinds[k] = h->post->kinds[k];
        ctl.walker_kids = h->post->walker_kids;
        if (ctl.err & (NERR_NODES | NERR_WALKERS)) {
            h->tree_mode_off = true;  // a tree outgrew its region: this step and the following ones on the batch-wide path
            nl_clock_end(h, 3);
        } else {
            if (ctl.err) {
                *flags = ctl.err;
                return nl_capacity_error(ctl.err);
            }
            if (h->profiling) {
                for (int k = 0; k < 4; ++k) h->census[k] += ctl.kinds[k];
                h->census[4] += ctl.walker_kids;
            }
            if ((uint64_t)d_base + total0 > h->out_cap)
                return rp::fail(RP_ERR_CAPACITY, "rp_nlhe: %llu Decisions in one batch exceed the buffer (%u)", (unsigned long long)d_base + total0, h->out_cap);
            nl_clock_end(h, 3);
            HIP_TRY(hipGetLastError());
            *n_dec = total0;
            *n_nod = ctl.n_nodes;
            *n_lev = ctl.pad[0];
            return RP_OK;
        }
    }
    h->ctl_clean = false;
    HIP_TRY(hipMemsetAsync(lv.ctl, 0, sizeof(NlCtl), st));
    HIP_TRY(hipMemsetAsync(lv.t_nw, 0, (size_t)B * 4, st));
    prm.tag = nl_next_tag(h);
    hipLaunchKernelGGL(k_nl_roots, dim3((B + 255u) / 256u), blk, 0, st, prm, lv);
    NlCtl ctl;
    uint32_t L = 0;
    for (;;) {  // grow all trees one level per pair of launches; look at the frontier every few levels
        const uint32_t stop = std::min<uint32_t>(NL_MAXL - 1u, L + (L == 0 ? 22u : 6u));
        for (; L < stop; ++L) {
            prm.tag = nl_next_tag(h);
            nl_clock_begin(h, 0);
            hipLaunchKernelGGL((k_nl_expand<4, 256>), wide_x, blk, 0, st, prm, h->tab, lv, L);
            nl_clock_end(h, 0);
            nl_clock_begin(h, 1);
            hipLaunchKernelGGL(k_nl_children, wide, blk, 0, st, prm, lv, L);
            nl_clock_end(h, 1);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&ctl, lv.ctl, sizeof(NlCtl), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (ctl.err) {
            *flags = ctl.err;
            return nl_capacity_error(ctl.err);
        }
        // level L = the children the last launch pair created; an empty level ends every tree
        const bool alive = ctl.lvl_node[L + 1] > ctl.lvl_node[L];
        if (!alive) break;
        if (L >= NL_MAXL - 1u) return nl_capacity_error(NERR_LEVELS);
    }
    uint32_t levels = 0;
    while (levels < NL_MAXL && ctl.lvl_node[levels + 1] > ctl.lvl_node[levels]) levels += 1;
    const uint32_t n_nodes = ctl.lvl_node[levels];
    nl_clock_begin(h, 2);
    for (uint32_t l = levels; l-- > 0;) hipLaunchKernelGGL(k_nl_up, wide, blk, 0, st, lv, l);
    for (uint32_t l = 0; l + 1 < levels; ++l) hipLaunchKernelGGL(k_nl_down, wide, blk, 0, st, lv, l);
    nl_clock_end(h, 2);
    nl_clock_begin(h, 3);
    HIP_TRY(rp::ss::exclusive_scan<uint32_t>(lv.t_nw, lv.t_woff, B, h->d_scan, st, h->d_total + 1));
    hipLaunchKernelGGL(k_nl_fill, dim3(std::min<uint32_t>(wide.x, 2048u)), blk, 0, st, lv, n_nodes);
    hipLaunchKernelGGL((k_nl_group<256>), dim3(B), dim3(64), 0, st, lv, B);
    hipLaunchKernelGGL(k_nl_group_big, dim3(std::min<uint32_t>(B, 1280u)), dim3(64), 0, st, lv);
    HIP_TRY(rp::ss::exclusive_scan<uint32_t>(lv.t_dcount, lv.t_doff, B, h->d_scan, st, h->d_total));
    HIP_TRY(hipGetLastError());
    uint32_t total[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(total, h->d_total, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&ctl, lv.ctl, sizeof(NlCtl), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (ctl.err) {
        *flags = ctl.err;
        return nl_capacity_error(ctl.err);
    }
    if (h->profiling) {
        for (int k = 0; k < 4; ++k) h->census[k] += ctl.kinds[k];
        h->census[4] += ctl.walker_kids;
    }
    if ((uint64_t)d_base + total[0] > h->out_cap)
        return rp::fail(RP_ERR_CAPACITY, "rp_nlhe: %llu Decisions in one batch exceed the buffer (%u)", (unsigned long long)d_base + total[0], h->out_cap);
    if (total[1] > lv.lcap) {  // as many walker nodes as this is a spent node budget too
        *flags = NERR_NODES;
        return rp::fail(RP_ERR_CAPACITY, "rp_nlhe: %u walker nodes in one pass exceed the buffer (%u)", total[1], lv.lcap);
    }
    hipLaunchKernelGGL(k_nl_emit, wide, blk, 0, st, lv, h->tab, total[1], d_base, lo, h->out_cap, h->out, 0u, (const float*)(lv.ex_k ? lv.wval : nullptr));
    nl_clock_end(h, 3);
    HIP_TRY(hipGetLastError());
    *n_dec = total[0];
    *n_nod = n_nodes;
    *n_lev = levels;
    return RP_OK;
}
// Solver::batch for the whole batch.  The node arrays hold 1 536 nodes per tree of the batch; trees grow with training, and when a
// pass runs out of nodes the batch is traversed in twice as many passes from then on (same trees, same Decisions in the same
// order: a pass is a contiguous range of tree ids) instead of failing the step.
int nl_traverse_levels(rp_nlhe* h) {
    nl_begin_step(h);
    const uint32_t B = h->batch;
    for (;;) {
        const uint32_t K = std::min<uint32_t>(h->chunks, B);
        uint32_t dec = 0, nodes = 0, levels = 0;
        bool again = false;
        for (uint32_t c = 0; c < K; ++c) {
            const uint32_t lo = (uint32_t)((uint64_t)B * c / K), hi = (uint32_t)((uint64_t)B * (c + 1) / K);
            if (hi == lo) continue;
            uint32_t d = 0, nn = 0, nl = 0, flags = 0;
            const int rc = nl_traverse_chunk(h, lo, hi - lo, dec, &d, &nn, &nl, &flags);
            if (rc) {
                if (flags == NERR_NODES && K < B && K < 4096u) {  // only the node budget: more, smaller passes
                    h->chunks = K * 2u;
                    again = true;
                    break;
                }
                return rc;
            }
            dec += d;
            nodes += nn;
            levels = std::max(levels, nl);
        }
        if (again) continue;
        h->last_n = dec;
        h->nodes += nodes;
        h->infos += dec;
        h->last_levels = levels;
        h->last_nodes = nodes;
        return RP_OK;
    }
}
int nl_traverse(rp_nlhe* h) { return nl_traverse_levels(h); }
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
    if (getenv("RP_NLHE_CHUNKS")) h->chunks = (uint32_t)std::max(1, atoi(getenv("RP_NLHE_CHUNKS")));
    if (getenv("RP_NL_TREE_BT")) h->tree_bt = (uint32_t)atoi(getenv("RP_NL_TREE_BT"));
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
    // Decisions per tree: ~40 / ~90 on average by walker, 483 the largest seen in 20 000 oracle trees; the batch buffer holds 160 per
    // tree.  Nodes per tree: 280 / 660 on average by walker on a fresh table, 3 300 the largest — and they GROW with training (the
    // opponent's average strategy calls and raises more than the warm-start bias: 760 per tree after a few steps on the trained
    // abstraction): 1 536 per tree of budget, 92 B each (the batch's total is what counts)
    // (round 5: 224 — after a few steps of external sampling in the reference's float order a 262 144-tree batch emitted 175 per tree)
    const uint64_t dec_cap64 = std::max<uint64_t>((uint64_t)batch * 224u, 4096u);
    // RP_NLHE_NODE_BUDGET = "<nodes per tree>[,<walker divisor>]" (tests of the chunked retry): the node budget per tree, and the
    // share of it the walker-node arrays hold (a quarter by default: ",16" makes a pass overflow THOSE arrays with nodes to spare)
    const char* budget_env = getenv("RP_NLHE_NODE_BUDGET");
    const uint64_t per_tree = budget_env ? (uint64_t)std::max(64, atoi(budget_env)) : 1536u;
    uint64_t walker_div = 4;
    if (budget_env && strchr(budget_env, ',')) walker_div = (uint64_t)std::min(4096, std::max(1, atoi(strchr(budget_env, ',') + 1)));
    const uint64_t ncap64 = budget_env ? (uint64_t)batch * per_tree : std::max<uint64_t>((uint64_t)batch * per_tree, 1u << 17);
    if (dec_cap64 >= (1ull << 31) || ncap64 >= (1ull << 32)) {
        delete h;
        return rp::fail(RP_ERR_INVALID, "rp_nlhe_create: batch too large (at most %u trees per step)", (uint32_t)((1ull << 32) / 1536u));
    }
    NL_TRY(rp_profile_create(device, rows, NLMC_A, regret, weight, hp, nullptr, (uint32_t)dec_cap64, &h->prof));
    NL_TRY(nl_alloc(h, &h->tab.slots, rows));
    NL_TRY(nl_alloc(h, &h->tab.n_keys, 1));
    h->tab.mask = (uint32_t)(rows - 1);
    h->tab.rows = rp::profile_table(h->prof);
    h->prm.seed = seed;
    h->prm.batch = batch;
    h->prm.temperature = hp->temperature;
    h->prm.smoothing = hp->smoothing;
    h->prm.curiosity = hp->curiosity;
    h->prm.sampling = RP_SAMPLING_EXTERNAL;  // the mccfr! macro's default scheme; rp_nlhe_set_sampling selects Flagship's
    h->prm.prune_threshold = hp->prune_threshold;
    h->prm.prune_explore = hp->prune_explore;
    h->prm.prune_warmup = hp->prune_warmup;
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
    {
        NlNodes& lv = h->lv;
        // a batch of at most NL_TREE_BATCH trees also gets a region of NL_TREE_CAP nodes and NL_WMAX walker slots per tree (k_nl_tree)
        const bool tree_mode = !budget_env && batch <= NL_TREE_BATCH;
        h->level_ncap = (uint32_t)ncap64;
        h->tree_cap = tree_mode ? NL_TREE_CAP : 0u;
        const size_t N = std::max<size_t>((size_t)ncap64, tree_mode ? (size_t)batch * NL_TREE_CAP : 0);
        const size_t LC = std::max<size_t>(std::max<size_t>((size_t)ncap64 / walker_div, 64), tree_mode ? (size_t)batch * NL_WMAX : 0);  // walker nodes of a batch (a seventh of its nodes): a quarter of the node budget
        lv.ncap = (uint32_t)N;
        lv.lcap = (uint32_t)LC;
        NL_TRY(nl_alloc(h, &lv.link, N)); NL_TRY(nl_alloc(h, &lv.tree, N)); NL_TRY(nl_alloc(h, &lv.meta, N));
        NL_TRY(nl_alloc(h, &lv.kid0, N)); NL_TRY(nl_alloc(h, &lv.row, N)); NL_TRY(nl_alloc(h, &lv.size, N));
        NL_TRY(nl_alloc(h, &lv.dfs, N)); NL_TRY(nl_alloc(h, &lv.aux, N));
        NL_TRY(nl_alloc(h, &lv.fac, N)); NL_TRY(nl_alloc(h, &lv.val, N)); NL_TRY(nl_alloc(h, &lv.reach, N));
        NL_TRY(nl_alloc(h, &lv.ga, N)); NL_TRY(nl_alloc(h, &lv.gb, N)); NL_TRY(nl_alloc(h, &lv.gc, N));
        NL_TRY(nl_alloc(h, &lv.hole0, B)); NL_TRY(nl_alloc(h, &lv.hole1, B));
        NL_TRY(nl_alloc(h, &lv.t_nw, B)); NL_TRY(nl_alloc(h, &lv.t_woff, B));
        NL_TRY(nl_alloc(h, &lv.t_dcount, B)); NL_TRY(nl_alloc(h, &lv.t_doff, B));
        NL_TRY(nl_alloc(h, &lv.wl, LC)); NL_TRY(nl_alloc(h, &lv.ws, LC)); NL_TRY(nl_alloc(h, &lv.gdesc, LC));
        NL_TRY(nl_alloc(h, &lv.big, B));
        NL_TRY(nl_alloc(h, &lv.ctl, 1));
        if (tree_mode) NL_TRY(nl_alloc_exact(h));  // a small batch is always evaluated in the reference's own order (k_nl_tree)
        if (tree_mode) {
            void* hp_ = nullptr;
            void* dp_ = nullptr;
            if (hipHostMalloc(&hp_, sizeof(NlPost), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
                hipHostGetDevicePointer(&dp_, hp_, 0) != hipSuccess) {
                if (hp_) (void)hipHostFree(hp_);
                NL_TRY(rp::fail(RP_ERR_HIP, "rp_nlhe_create: no pinned host memory for the traversal's post"));
            }
            memset(hp_, 0, sizeof(NlPost));
            h->post = reinterpret_cast<NlPost*>(hp_);
            h->post_dev = reinterpret_cast<NlPost*>(dp_);
        }
    }
    h->out_cap = (uint32_t)dec_cap64;
    NL_TRY(nl_alloc(h, &h->out.row, h->out_cap));
    NL_TRY(nl_alloc(h, &h->out.nact, h->out_cap));
    NL_TRY(nl_alloc(h, &h->out.expanded, h->out_cap));
    NL_TRY(nl_alloc(h, &h->out.regret, (size_t)h->out_cap * NLMC_A));
    NL_TRY(nl_alloc(h, &h->out.policy, (size_t)h->out_cap * NLMC_A));
    NL_TRY(nl_alloc(h, &h->out.payoff, h->out_cap));
    NL_TRY(nl_alloc(h, &h->out.tree, h->out_cap));
    NL_TRY(nl_alloc(h, &h->d_total, 2));
    NL_TRY(nl_alloc(h, &h->d_remap_err, 1));
    {
        unsigned char* scratch = nullptr;
        NL_TRY(nl_alloc(h, &scratch, rp::ss::scan_scratch_bytes(B)));
        h->d_scan = scratch;
    }
#undef NL_TRY
    // hipMemset on device memory returns before it has run, and a non-blocking stream does not wait for the null stream: the
    // first launch on this handle's stream could otherwise overtake the initialisation above and be overwritten by it
    (void)hipDeviceSynchronize();
    *out = h;
    return RP_OK;
}

// the SamplingScheme of the solver type at walker nodes: ExternalSampling (the mccfr! macro's default, nlhe/src/solver.rs:11),
// PrunableSampling, PluribusSampling (Flagship, nlhe/src/lib.rs:86-90); thresholds from the rp_hyper given at creation
int rp_nlhe_set_sampling(rp_nlhe* h, rp_sampling_kind sampling) {
    if (!h || (int)sampling < 0 || (int)sampling > (int)RP_SAMPLING_PLURIBUS) return rp::fail(RP_ERR_INVALID, "rp_nlhe_set_sampling: bad argument");
    h->prm.sampling = (int)sampling;
    return RP_OK;
}

// shape of the last traversed batch (level-synchronous traversal): levels grown and nodes
int rp_nlhe_set_rng(rp_nlhe* h, rp_rng_kind kind) {
    if (!h || (kind != RP_RNG_COUNTER && kind != RP_RNG_REFERENCE)) return rp::fail(RP_ERR_INVALID, "rp_nlhe_set_rng: bad argument");
    h->prm.ref_rng = kind == RP_RNG_REFERENCE ? 1u : 0u;
    return RP_OK;
}
int rp_nlhe_set_exact(rp_nlhe* h, int on) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_nlhe_set_exact: NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(rp::profile_stream(h->prof)));
    if (!on) {
        if (h->tree_cap) return rp::fail(RP_ERR_UNSUPPORTED, "rp_nlhe_set_exact: a batch of at most %u trees is always evaluated in the reference's order", NL_TREE_BATCH);
        if (h->lv.ex_k) h->ex_k_parked = h->lv.ex_k;
        h->lv.ex_k = nullptr;  // the kernels look at ex_k only; the arrays stay with the handle
        return RP_OK;
    }
    if (!h->lv.ex_k && h->ex_k_parked) {
        h->lv.ex_k = h->ex_k_parked;
        return RP_OK;
    }
    return nl_alloc_exact(h);
}
int rp_nlhe_last_shape(rp_nlhe* h, uint32_t* levels, uint32_t* nodes) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_nlhe_last_shape: NULL handle");
    if (levels) *levels = h->last_levels;
    if (nodes) *nodes = h->last_nodes;
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
    if (h->post) (void)hipHostFree(h->post);
    for (void* p : {h->x_keys, h->x_counts, h->x_all, h->x_packed})
        if (p) (void)hipFree(p);
    delete h;
    return RP_OK;
}

int rp_nlhe_step(rp_nlhe* h, rp_update_mode mode) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_nlhe_step: NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    int rc = nl_traverse(h);
    if (rc) return rc;
    rp_decisions b{h->last_n, h->out.row, h->out.nact, h->out.expanded, h->out.regret, h->out.policy, h->out.payoff};
    nl_clock_begin(h, 4);
    rc = rp_profile_apply(h->prof, &b, mode);
    nl_clock_end(h, 4);
    return rc;
}

// Trainer::train (crates/forge/src/trainer.rs:18-66) over the NLHE solver — what forge runs on the Flagship type: loop { step;
// checkpoint; flush; interrupt? } with Metrics::checkpoint's rate (mccfr/src/metrics/mod.rs:67-80), Checkpoint's display line
// (metrics/checkpoint.rs:39-50) and Progress::summary (progress.rs:24-26); the same contract as rp_mccfr_train.  The counters
// live on the host (the step synchronises once anyway), so a checkpoint costs nothing.
int rp_nlhe_train(rp_nlhe* h, rp_update_mode mode, uint64_t max_steps, double max_seconds, double log_interval, double flush_interval,
                  rp_train_event_fn on_event, void* user, const volatile int* interrupt, char* summary, size_t summary_cap) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_nlhe_train: NULL handle");
    using clock = std::chrono::steady_clock;
    const auto start = clock::now();
    auto prior = start, flushed = start;
    uint64_t prior_infos = 0, steps = 0;
    auto secs_since = [](clock::time_point t) { return std::chrono::duration<double>(clock::now() - t).count(); };
    for (;;) {
        int rc = rp_nlhe_step(h, mode);
        if (rc) return rc;
        steps += 1;
        const uint64_t epoch = rp::profile_epoch(h->prof);
        if (secs_since(prior) >= log_interval) {
            const double secs = std::max(1.0, std::floor(secs_since(prior)));  // elapsed().as_secs().max(1)
            rp_checkpoint cp{epoch, h->nodes, h->infos, (double)(h->infos - prior_infos) / secs};
            prior = clock::now();
            prior_infos = h->infos;
            if (on_event) {
                char line[96];
                rp::format_progress(line, sizeof line, cp.epoch, cp.nodes, cp.infos, cp.rate);
                on_event(RP_TRAIN_CHECKPOINT, &cp, line, user);
            }
        }
        if (secs_since(flushed) >= flush_interval) {  // FastSession::flush cadence (forge/src/fast.rs:97-125)
            flushed = clock::now();
            if (on_event) {
                rp_checkpoint cp{epoch, h->nodes, h->infos, 0.0};
                on_event(RP_TRAIN_FLUSH, &cp, "", user);
            }
        }
        const bool stop = (interrupt && *interrupt) || (max_steps && steps >= max_steps) || (max_seconds > 0.0 && secs_since(start) >= max_seconds);
        if (stop) break;
    }
    if (summary && summary_cap) {
        char line[96];
        const double secs = std::max(1.0, std::floor(secs_since(start)));
        rp::format_progress(line, sizeof line, rp::profile_epoch(h->prof), h->nodes, h->infos, (double)h->infos / secs);
        snprintf(summary, summary_cap, "training stopped\n%s", line);
    }
    return RP_OK;
}

// ---- profiling hooks used by bench.py: HIP events on the launch stream around each kernel group
int rp_nlhe_profile(rp_nlhe* h, int enable) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_nlhe_profile: NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(rp::profile_stream(h->prof)));
    nl_clock_drain(h);
    h->profiling = enable != 0;
    for (auto& c : h->clk) c.total_ms = 0.0, c.launches = 0;
    for (auto& c : h->census) c = 0;
    return RP_OK;
}
int rp_nlhe_kernel_time(rp_nlhe* h, const char* name, double* total_ms, uint64_t* launches) {
    if (!h || !name) return rp::fail(RP_ERR_INVALID, "rp_nlhe_kernel_time: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(rp::profile_stream(h->prof)));
    nl_clock_drain(h);
    static const char* names[5] = {"expand", "children", "sweeps", "decide", "apply"};
    for (int k = 0; k < 5; ++k)
        if (!strcmp(name, names[k])) {
            if (total_ms) *total_ms = h->clk[k].total_ms;
            if (launches) *launches = h->clk[k].launches;
            return RP_OK;
        }
    return rp::fail(RP_ERR_INVALID, "rp_nlhe_kernel_time: unknown kernel group '%s'", name);
}
// nodes of the profiled steps by kind {terminal, chance, walker, opponent} and the children of their walker nodes
int rp_nlhe_census(rp_nlhe* h, uint64_t* kinds4, uint64_t* walker_children) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_nlhe_census: NULL handle");
    if (kinds4)
        for (int k = 0; k < 4; ++k) kinds4[k] = h->census[k];
    if (walker_children) *walker_children = h->census[4];
    return RP_OK;
}

int rp_nlhe_batch(rp_nlhe* h, uint32_t cap, uint32_t* n, uint32_t* tree, uint64_t* past, uint32_t* present, uint64_t* choices,
                  uint8_t* n_actions, uint16_t* expanded, float* regret, float* policy, float* payoff) {
    if (!h || !n) return rp::fail(RP_ERR_INVALID, "rp_nlhe_batch: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = rp::profile_stream(h->prof);
    const uint64_t nodes0 = h->nodes, infos0 = h->infos;
    int rc = nl_traverse(h);
    h->nodes = nodes0;  // a debugging view: counters not advanced
    h->infos = infos0;
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
    HIP_TRY(hipStreamSynchronize(st));  // every copy above has landed before any return below
    if (past || present || choices) {  // the infoset behind each row (rows differ between implementations; keys do not)
        const size_t rowsn = (size_t)1 << h->cap_log2;
        std::vector<NlSlot> slots(rowsn);
        HIP_TRY(hipMemcpy(slots.data(), h->tab.slots, rowsn * sizeof(NlSlot), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < m; ++i) {
            if (past) past[i] = slots[rows[i]].past;
            if (choices) choices[i] = slots[rows[i]].choices;
            if (present) present[i] = slots[rows[i]].present;
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
    unsigned int k = 0;
    HIP_TRY(hipMemcpyAsync(&k, h->tab.n_keys, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (nodes) *nodes = h->nodes;
    if (infos) *infos = h->infos;
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
    std::vector<NlSlot> slots(rowsn);
    std::vector<uint32_t> rows;
    HIP_TRY(hipMemcpy(slots.data(), h->tab.slots, rowsn * sizeof(NlSlot), hipMemcpyDeviceToHost));
    for (size_t s = 0; s < rowsn; ++s)
        if (slots[s].state == 2u) rows.push_back((uint32_t)s);
    *n = rows.size();
    const size_t m = std::min<size_t>(cap, rows.size());
    if (m == 0 || !past || !present || !choices || !enc) return RP_OK;
    for (size_t i = 0; i < m; ++i) {
        past[i] = slots[rows[i]].past;
        choices[i] = slots[rows[i]].choices;
        present[i] = slots[rows[i]].present;
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
    std::vector<NlSlot> slots(rowsn);
    std::vector<uint32_t> rows(n);
    HIP_TRY(hipMemcpy(slots.data(), h->tab.slots, rowsn * sizeof(NlSlot), hipMemcpyDeviceToHost));
    unsigned int added = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t hk = rp_mix64(rp_mix64(past[i] ^ 0x9e3779b97f4a7c15ull) ^ rp_mix64(choices[i] + 0xd1342543de82ef95ull) ^
                                     ((uint64_t)present[i] * 0xaf251af3b0f025b5ull));
        uint32_t s = (uint32_t)hk & h->tab.mask;
        for (size_t probes = 0;; ++probes) {
            if (probes > rowsn) return rp::fail(RP_ERR_CAPACITY, "rp_nlhe_import: infoset table full");
            NlSlot& sl = slots[s];
            if (sl.state != 2u) {
                sl.state = 2u;
                sl.born = 0u;
                sl.past = past[i];
                sl.choices = choices[i];
                sl.present = present[i];
                added += 1;
                break;
            }
            if (sl.past == past[i] && sl.choices == choices[i] && sl.present == present[i]) break;
            s = (s + 1u) & h->tab.mask;
        }
        rows[i] = s;
    }
    HIP_TRY(hipMemcpy(h->tab.slots, slots.data(), rowsn * sizeof(NlSlot), hipMemcpyHostToDevice));
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
                           nl_next_tag(h), h->d_remap_err);
        HIP_TRY(hipGetLastError());
    }
    return rp_profile_fold(h->prof, entries_dev, n_entries);
}

// Solver::step across the ranks of an rp_comm (the library's own RCCL communicator; hosts without torch.distributed): step_local,
// the entry counts of all ranks (4 bytes each — ncclAllGather takes host-known sizes, so this is the one host read of a step),
// entries and keys all-gathered padded to the longest list, packed rank-major on the device, step_apply.  `steps` steps.
int rp_nlhe_step_comm(rp_nlhe* h, rp_comm* c, uint32_t steps) {
    if (!h || !c) return rp::fail(RP_ERR_INVALID, "rp_nlhe_step_comm: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    const uint32_t world = (uint32_t)rp::comm_world(c), rank = (uint32_t)rp::comm_rank(c);
    int rc = rp_nlhe_set_shard(h, rank, world);
    if (rc) return rc;
    size_t eb = 0;
    (void)rp_profile_entry_bytes(h->prof, &eb);
    hipStream_t st = rp::profile_stream(h->prof);
    unsigned char* mine_ent = rp::profile_entries(h->prof);
    if (!mine_ent) return rp::fail(RP_ERR_INVALID, "rp_nlhe_step_comm: the profile has no summary buffer");
    auto grow = [&](void** p, size_t* have, size_t need) -> int {
        if (*have >= need) return RP_OK;
        HIP_TRY(hipStreamSynchronize(st));
        if (*p) (void)hipFree(*p);
        *p = nullptr;
        *have = 0;
        HIP_TRY(hipMalloc(p, need + need / 4));
        *have = need + need / 4;
        return RP_OK;
    };
    for (uint32_t s = 0; s < steps; ++s) {
        // A rank that fails here (a full table, the Decisions buffer, the node budget after its retries) must not leave the others
        // waiting in the gather: its failure travels in the count exchange (NL_RANK_FAILED) and every rank leaves the step with an
        // error together.
        constexpr uint32_t NL_RANK_FAILED = 0xffffffffu;
        uint32_t n_mine = 0;
        int local_rc = nl_traverse(h);
        if (!local_rc) {
            rp_decisions b{h->last_n, h->out.row, h->out.nact, h->out.expanded, h->out.regret, h->out.policy, h->out.payoff};
            local_rc = rp_profile_summarize(h->prof, &b, mine_ent, &n_mine);  // synchronises: n_mine is valid
        }
        const std::string local_msg = local_rc ? rp_last_error() : "";
        if (local_rc) n_mine = NL_RANK_FAILED;
        // every rank's count
        if ((rc = grow(&h->x_counts, &h->x_counts_bytes, (size_t)(world + 1) * 4u))) return rc;
        uint32_t* d_counts = reinterpret_cast<uint32_t*>(h->x_counts);
        HIP_TRY(hipMemcpyAsync(d_counts + world, &n_mine, 4, hipMemcpyHostToDevice, st));
        if ((rc = rp::comm_all_gather(c, d_counts + world, d_counts, 4, st))) return rc;
        std::vector<uint32_t> counts(world);
        HIP_TRY(hipMemcpyAsync(counts.data(), d_counts, (size_t)world * 4u, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (local_rc) return rp::fail(local_rc, "%s", local_msg.c_str());
        for (uint32_t r = 0; r < world; ++r)
            if (counts[r] == NL_RANK_FAILED) return rp::fail(RP_ERR_CAPACITY, "rp_nlhe_step_comm: rank %u failed its part of the step", r);
        uint32_t width = 0, total = 0;
        for (uint32_t r = 0; r < world; ++r) width = std::max(width, counts[r]), total += counts[r];
        // keys of my entries: [past u64][choices u64][present u32] planes of `width` items (the gather sends `width` of each)
        if ((rc = grow(&h->x_keys, &h->x_keys_bytes, (size_t)std::max<uint32_t>(width, 1u) * 20u))) return rc;
        uint64_t* kp = reinterpret_cast<uint64_t*>(h->x_keys);
        uint64_t* kc = kp + width;
        uint32_t* kb = reinterpret_cast<uint32_t*>(kc + width);
        if (n_mine)
            hipLaunchKernelGGL(k_nlhe_entry_keys, dim3((n_mine + 255u) / 256u), dim3(256), 0, st, h->tab, mine_ent, (uint32_t)eb, n_mine, kp, kb, kc);
        // padded gathers: four planes (entries, past, choices, present), each world x width items
        const size_t unit[4] = {eb, 8, 8, 4};
        size_t all_bytes = 0, pk_bytes = 0;
        for (int q = 0; q < 4; ++q) all_bytes += (size_t)world * width * unit[q], pk_bytes += (size_t)std::max(total, 1u) * unit[q];
        if ((rc = grow(&h->x_all, &h->x_all_bytes, std::max<size_t>(all_bytes, 64)))) return rc;
        if ((rc = grow(&h->x_packed, &h->x_packed_bytes, std::max<size_t>(pk_bytes, 64)))) return rc;
        unsigned char* all = reinterpret_cast<unsigned char*>(h->x_all);
        unsigned char* pk = reinterpret_cast<unsigned char*>(h->x_packed);
        const void* src[4] = {mine_ent, kp, kc, kb};
        unsigned char* pk_plane[4];
        size_t all_off = 0, pk_off = 0;
        for (int q = 0; q < 4 && width; ++q) {
            unsigned char* plane = all + all_off;
            if ((rc = rp::comm_all_gather(c, src[q], plane, (size_t)width * unit[q], st))) return rc;  // reads `width` items of mine: within its buffers
            pk_plane[q] = pk + pk_off;
            size_t at = 0;
            for (uint32_t r = 0; r < world; ++r) {  // rank-major packing
                if (counts[r])
                    HIP_TRY(hipMemcpyAsync(pk_plane[q] + at, plane + (size_t)r * width * unit[q], (size_t)counts[r] * unit[q], hipMemcpyDeviceToDevice, st));
                at += (size_t)counts[r] * unit[q];
            }
            all_off += (size_t)world * width * unit[q];
            pk_off += (size_t)total * unit[q];
        }
        if (!width) pk_plane[0] = pk_plane[1] = pk_plane[2] = pk_plane[3] = pk;
        if ((rc = rp_nlhe_step_apply(h, pk_plane[0], reinterpret_cast<const uint64_t*>(pk_plane[1]), reinterpret_cast<const uint32_t*>(pk_plane[3]),
                                     reinterpret_cast<const uint64_t*>(pk_plane[2]), total)))
            return rc;
    }
    return RP_OK;
}

int rp_nlhe_set_stream(rp_nlhe* h, void* hip_stream) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_nlhe_set_stream: null handle");
    return rp_profile_set_stream(h->prof, hip_stream);
}

int rp_nlhe_sync(rp_nlhe* h) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_nlhe_sync: null handle");
    int rc = rp_profile_sync(h->prof);
    if (rc) return rc;
    uint32_t err = 0;  // keys of an exchange that did not fit the table: reported once, then cleared
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy(&err, h->d_remap_err, 4, hipMemcpyDeviceToHost));
    if (err) {
        HIP_TRY(hipMemset(h->d_remap_err, 0, 4));
        return nl_capacity_error(err);
    }
    return RP_OK;
}

}  // extern "C"

#ifdef NL_TREE_PROF  // diagnostic build only (scripts/r6_nltree_prof.sh): k_nl_tree's phase clocks
extern "C" RP_API int rp_nl_tree_prof(uint64_t* out16, int reset) {
    unsigned long long z[16] = {0};
    if (out16) HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(rp::g_nl_prof), sizeof(z)));
    if (reset) HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rp::g_nl_prof), z, sizeof(z)));
    return RP_OK;
}
extern "C" RP_API int rp_nl_tree_rec(uint32_t* out, uint32_t trees) {
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(rp::g_nl_rec), (size_t)std::min<uint32_t>(trees, 2048u) * 16u * 4u));
    return RP_OK;
}
#endif

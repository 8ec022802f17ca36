// lloyd.hip — Elkan k-means over Sinkhorn EMD / equity variation on MI355X (gfx950): kernels + rp_kmeans_* ABI.
//
// Reference path: crates/elkan (Elkan<K,N>::{init_bounds, neighbor, pairwises, step_elkan, step_naive}),
// crates/lloyd (Layer, Kmeans, Sinkhorn, Metric::emd, Equity::variation).  MI355X mapping:
//
//   one WAVEFRONT per (point) work item.  A Sinkhorn solve (sinkhorn.rs:77-92) is wave-cooperative:
//   lane i owns support row i and walks the other support sequentially, so every softmin sum is the
//   reference's left fold in ascending bin order — bit-exact with the CPU oracle — and needs no cross-lane
//   reduction.  Potentials, supports and log-densities live in 6 KB of LDS per wave; the ground cost C and
//   C/T (bins x bins f32, 2 x 256 KB) are L2-resident and read row-wise (coalesced or in-row gathers).
//   Elkan's candidate loop (elkan.rs:153-168) keeps its sequential semantics per point: the wave evaluates
//   has_shifted() for all K centroids at once (ballot), solves the first hit, re-tests with the updated (j,u).
//   Centroids are exact integer sums (bins.rs:75-82): order-free, so recompute() is a parallel reduction.
//
// Everything f32 is spelled with the primitives of include/rp_math.h and compiled -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/rp_math.h"
#include "../../include/rp_refrng.h"
#include "../../include/rp_libm_glibc.h"
#include "rp_internal.h"

#include "lloyd_shared.hpp"
#include "lm_glibc_dev.hpp"

namespace rp {

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// the device code, once per arithmetic (lloyd_kernels.hpp)
namespace lm_contract {
#define LM_GLIBC 0
#include "lloyd_kernels.hpp"
#undef LM_GLIBC
}  // namespace lm_contract
namespace lm_glibc {
#define LM_GLIBC 1
#include "lloyd_kernels.hpp"
#undef LM_GLIBC
}  // namespace lm_glibc
using namespace lm_contract;  // unqualified kernel names below are the contract's; KSEL picks by the handle's mode
#define KSEL_IF(glibc, name) ((glibc) ? &lm_glibc::name : &lm_contract::name)
#define KSEL(h, name) KSEL_IF((h)->libm != RP_LIBM_CONTRACT, name)

}  // namespace rp

// =================================================================================================
// host side
// =================================================================================================
using namespace rp;

namespace {
struct Clock {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    uint64_t launches = 0;
};
const char* const CLOCK_NAMES[] = {"pairwise", "step", "recompute", "bounds", "neighbor", "selfcost", "kpp", "drift", "mfma_bound", "kpp_bound", "refresh_bound"};
enum { CK_PAIRWISE, CK_STEP, CK_RECOMPUTE, CK_BOUNDS, CK_NEIGHBOR, CK_SELF, CK_KPP, CK_DRIFT, CK_BOUND, CK_KPP_BOUND, CK_REFRESH_BOUND, CK_COUNT };
}  // namespace

// the reference-seed k-means++ draw (kpp_refpick.hpp): four launches on `stream`; out = [picked index, total bits, chunks walked]
namespace {
uint32_t ref_pick_chunks(uint64_t N) { return (uint32_t)((N + KR_ELEMS - 1) / KR_ELEMS); }
size_t ref_pick_work_floats(uint64_t N) { return (size_t)5 * ref_pick_chunks(N) + 64; }
void ref_pick_launch(hipStream_t stream, float* pot, float* kpp_d, uint64_t N, float* work, float v01, unsigned long long* out) {
    const uint32_t nc = ref_pick_chunks(N);
    float *cum = work, *csum = work + nc;
    uint32_t* expo = reinterpret_cast<uint32_t*>(work + 2 * (size_t)nc);
    uint2* meta = reinterpret_cast<uint2*>(work + 3 * (size_t)nc + (nc & 1u));  // 8-byte aligned
    hipLaunchKernelGGL(k_kr_sums, dim3((nc + 3) / 4), dim3(256), 0, stream, pot, N, nc, csum);
    hipLaunchKernelGGL(k_kr_scan, dim3(1), dim3(1024), 0, stream, csum, nc, expo);
    hipLaunchKernelGGL(k_kr_chunks, dim3((nc + 3) / 4), dim3(256), 0, stream, pot, N, nc, expo, meta);
    hipLaunchKernelGGL(k_kr_pick, dim3(1), dim3(64), 0, stream, pot, kpp_d, N, nc, meta, cum, v01, out);
}
}  // namespace

#define SB_SAMPLE_STRIDE 521u  // the production self-check of the MFMA prune looks at every 521st point (0.2 % more exact solves)
struct rp_kmeans {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t K = 0, bins = 0, stride = 0;
    uint64_t N = 0;
    int kind = 0;
    rp_sinkhorn_hp hp{};
    uint64_t seed = 0;
    rp_rng_kind rng = RP_RNG_COUNTER;  // rp_kmeans_set_rng: RP_RNG_REFERENCE = the reference's generator and WeightedIndex<f32>
    int street = 0;                    // Street discriminant hashed into that generator's seed (layer.rs:156-158)
    float* kpp_cum = nullptr;          // reference-seed draw (kpp_refpick.hpp): [5 x chunks] chunk-end sums, approximate sums, exponents, summaries
    unsigned long long* kr_out = nullptr;  // [4] picked index, total bits, chunks walked term by term
    uint64_t kr_walked = 0, kr_chunks = 0;  // over the layer's picks
    std::vector<void*> allocs;
    bool owns_counts = true;
    Points P{};
    Metric M{};
    CentroidSet cs[2]{};
    int cur = 0;
    Bounds B{};
    uint8_t* prior = nullptr;
    float* pairw = nullptr;
    uint32_t* cver = nullptr;   // [K] content versions of the centroids (Bounds::cver)
    uint32_t* pver = nullptr;   // [K*K][2] versions pairw was computed from
    uint32_t* kpp_todo = nullptr;      // init_bounds after k-means++: the points whose neighbor the notes do not settle
    unsigned int* kpp_ntodo = nullptr;
    uint32_t memo_epoch = 0;
    bool memo_dirty = true;     // centroids were installed outside an Elkan step: forget everything at the next one
    bool memo_on = true;        // RP_LLOYD_NO_MEMO=1 switches the remembered refreshes off
    float* mid = nullptr;
    float* drift = nullptr;
    float* pot = nullptr;
    float* pdist = nullptr;
    uint32_t* quads = nullptr;    // [n_quads][4] points with <= QUAD_ROWS support bins, four per wavefront (Sinkhorn)
    uint32_t* pairs = nullptr;    // [n_pairs][2] points with <= PAIR_ROWS support bins, two per wavefront
    uint32_t* singles = nullptr;  // [n_singles] the other points
    uint64_t n_quads = 0, n_pairs = 0, n_singles = 0;
    Refresh refresh{};            // grouped stale-bound refresh (Sinkhorn; null nsup = off)
    // k-means++ with the column-marginal bound (k_kpp_filter): per round, only the points whose potential can still drop
    unsigned char* comm_partial = nullptr;  // rp_kmeans_step_comm's exchange buffer
    bool kpp_lb = false;
    uint8_t* d_nsup = nullptr;
    float* minc = nullptr;        // [bins]
    KppLists kpp{};
    uint64_t kpp_cap[3] = {0, 0, 0};  // points per support class (<= QUAD_ROWS, <= PAIR_ROWS, more): grid bounds
    // the second k-means++ filter (kpp_bound.hpp): a scaling-domain interval per (new centroid, point) pair the column bound let through
    bool kb_on = false;
    float kb_claim_scale = 1.0f;  // RP_KPP_CLAIM_TEST: the tripwire's claims scaled (a test proves that it trips)
    int kb_dual = 2;  // the dual exit's pair: 2 = (f, its c-transform), 1 = (f, -T ln K^T u) (RP_KPP_DUAL; measurements)
    KppLists kpp2{};                        // the pairs the solve is still needed for, per support class
    unsigned int* kb_cursor = nullptr;      // [3] work cursors of the three launches of a round
    unsigned long long* kb_stats = nullptr; // striped: pairs examined, kept, pair-iterations, cost passes
    // the interval-decided refresh (refresh_bound.hpp): B.ulo / B.uiv, what each point needs this step, the kernel's counters
    bool rb_on = false;
    uint8_t* rb_code = nullptr;             // [N]
    unsigned int* rb_cursor = nullptr;      // [2]
    unsigned long long* rb_stats = nullptr; // striped: pairs examined, settled, pair-iterations, cost passes
    std::vector<uint8_t> ns_host;           // support size of every point (saturated at 255)
    std::vector<uint32_t> cent_m;           // [K] support size of the centroids installed one at a time (0 = not known on the host)
    // the MFMA bound in front of the neighbor passes (sinkhorn_bound.hpp)
    bool sb_on = false, sb_audit = false;
    SbParams sb{};
    uint32_t* sb_list[4] = {nullptr, nullptr, nullptr, nullptr};  // points by ceil(support / 16) = 1..4
    uint32_t sb_count[4] = {0, 0, 0, 0};
    uint32_t* sb_big = nullptr;   // points with more than SB_MAXROWS bins: never pruned
    uint32_t sb_nbig = 0;
    unsigned int* sb_cursor = nullptr;      // [4]
    unsigned long long* sb_mask = nullptr;  // [N][4]
    unsigned long long* sb_stats = nullptr; // striped: survivors, points, column-block iterations, cost passes
    unsigned long long* sb_bad = nullptr;   // [0] audit disagreements, [1] disagreements of the production sample check
    uint32_t* sb_sample = nullptr;          // the sampled points (every SB_SAMPLE_STRIDE-th)
    uint32_t sb_nsample = 0;
    uint64_t sb_sampled = 0;
    bool sb_check_pending = false;
    uint8_t* audit_j = nullptr;
    float* audit_d = nullptr;
    float* sb_d = nullptr;
    uint8_t* nb_tmp_j = nullptr;  // [NB_LIST_MAX][NB_CHUNKS] partial nearest centroids of launch_neighbor_list
    float* nb_tmp_d = nullptr;
    uint8_t* sb_crank = nullptr;  // [256] rank of every centroid in the similarity order (centroid_order)
    bool sb_crank_set = false;
    bool pairw_seen = false;      // an Elkan step has filled pairw for (nearly) the current centroids
    float* sb_ub0 = nullptr;      // [N] upper bound handed to the bound kernel
    uint8_t* sb_hint_j = nullptr; // [N]
    unsigned long long* sb_hint_mask = nullptr;  // [N][4]
    bool pot_is_min_d2 = false;   // the k-means++ potentials describe the CURRENT centroids
    uint64_t sb_audited = 0;
    unsigned long long* bsum = nullptr;
    unsigned long long* scal = nullptr;  // [0] picked, [1] moved
    unsigned long long* sizes = nullptr; // [K]
    unsigned long long* stats = nullptr; // [2]
    uint8_t* tmp_j = nullptr;
    uint32_t* hist_stage = nullptr;
    bool bounds_ready = false, centroids_ready = false;
    int libm = RP_LIBM_CONTRACT;  // rp_kmeans_set_libm: which arithmetic pass of the kernels runs (KSEL)
    bool profiling = false;
    Clock clk[CK_COUNT];
};

namespace {

template <typename T>
int dev_alloc(rp_kmeans* h, T** out, size_t count) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    h->allocs.push_back(p);
    *out = reinterpret_cast<T*>(p);
    return RP_OK;
}

void ck_begin(rp_kmeans* h, int id) {
    if (!h->profiling) return;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, h->stream);
    h->clk[id].pending.emplace_back(a, b);
}
void ck_end(rp_kmeans* h, int id) {
    if (!h->profiling) return;
    (void)hipEventRecord(h->clk[id].pending.back().second, h->stream);
    h->clk[id].launches += 1;
}
void ck_drain(rp_kmeans* h) {
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

int alloc_centroid_set(rp_kmeans* h, CentroidSet* cs) {
    int rc;
    if ((rc = dev_alloc(h, &cs->counts, (size_t)h->K * h->bins))) return rc;
    if ((rc = dev_alloc(h, &cs->weight, h->K))) return rc;
    if ((rc = dev_alloc(h, &cs->n, h->K))) return rc;
    if ((rc = dev_alloc(h, &cs->sup, (size_t)h->K * MAXB))) return rc;
    if ((rc = dev_alloc(h, &cs->lnd, (size_t)h->K * MAXB))) return rc;
    if ((rc = dev_alloc(h, &cs->dens, (size_t)h->K * h->bins))) return rc;
    if ((rc = dev_alloc(h, &cs->densR, (size_t)h->K * MAXB))) return rc;
    HIP_TRY(hipMemset(cs->densR, 0, (size_t)h->K * MAXB * 4));
    if ((rc = dev_alloc(h, &cs->mincT, (size_t)MAXB * MAXB))) return rc;
    HIP_TRY(hipMemset(cs->mincT, 0, (size_t)MAXB * MAXB * 4));
    if ((rc = dev_alloc(h, &cs->self, h->K))) return rc;
    return RP_OK;
}

__global__ void k_fill_u32(uint32_t* p, uint32_t n, uint32_t v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// content versions of the centroids (Bounds::cver).  against = the set being replaced: centroid k keeps its version iff its
// integer sums and weight are unchanged; against = none: every centroid gets a new version
__global__ __launch_bounds__(64) void k_centroid_versions(CentroidSet now, CentroidSet was, bool compare, uint32_t bins, uint32_t* cver) {
    const uint32_t k = blockIdx.x, lane = lane_id();
    bool differs = !compare;
    if (compare) {
        for (uint32_t t = lane; t < bins; t += 64) differs |= now.counts[(size_t)k * bins + t] != was.counts[(size_t)k * bins + t];
        differs |= now.weight[k] != was.weight[k];
    }
    if (__ballot(differs) && lane == 0) cver[k] += 1u;
}

int prepare_centroids(rp_kmeans* h, int set, bool replaces_other = false) {
    // an Elkan step replaces the other set: compare contents.  Any other way of installing centroids (k-means++, set_*,
    // step_naive) invalidates everything remembered, lazily at the next Elkan step (memo_reset).
    if (h->cver && replaces_other)
        hipLaunchKernelGGL(k_centroid_versions, dim3(h->K), dim3(64), 0, h->stream, h->cs[set], h->cs[set ^ 1], true, h->bins, h->cver);
    else
        h->memo_dirty = true, h->pairw_seen = false;
    ck_begin(h, CK_SELF);
    if (h->kind == RP_METRIC_SINKHORN) {  // the tables by one wavefront per centroid, OT(c, c) by four (K solves of up to 256 x 256 bins)
        hipLaunchKernelGGL(KSEL(h, k_prepare_centroids), dim3(h->K), dim3(64), 0, h->stream, h->cs[set], h->K, h->M, h->kind, 0u,
                           (const float*)h->cs[set].self);
        hipLaunchKernelGGL(KSEL(h, k_self_block), dim3(h->K), dim3(256), 0, h->stream, h->cs[set], h->K, h->M);
    } else
        hipLaunchKernelGGL(KSEL(h, k_prepare_centroids), dim3(h->K), dim3(64), 0, h->stream, h->cs[set], h->K, h->M, h->kind, 0u, (const float*)nullptr);
    ck_end(h, CK_SELF);
    HIP_TRY(hipGetLastError());
    return RP_OK;
}

int create_common(uint32_t K, uint64_t N, uint32_t bins, const void* counts, bool counts_on_device, rp_metric_kind kind,
                  const float* tri_metric, const rp_sinkhorn_hp* hp, uint64_t seed, int device, rp_kmeans** out) {
    if (!out || !counts) return rp::fail(RP_ERR_INVALID, "rp_kmeans_create: NULL argument");
    if (K == 0 || K > MAXB || bins == 0 || bins > MAXB || N == 0)
        return rp::fail(RP_ERR_INVALID, "rp_kmeans_create: need 1 <= K <= 256, 1 <= bins <= 256, N >= 1");
    if (kind != RP_METRIC_SINKHORN && kind != RP_METRIC_VARIATION) return rp::fail(RP_ERR_INVALID, "rp_kmeans_create: unknown metric kind");
    if (kind == RP_METRIC_SINKHORN && !tri_metric && bins > 1) return rp::fail(RP_ERR_INVALID, "rp_kmeans_create: Sinkhorn needs tri_metric");
    if (rp_device_count() <= 0)
        return rp::fail(RP_ERR_NO_DEVICE, "rp_kmeans_create: no HIP device visible; the MI355X path has no CPU fallback");
    rp_kmeans* h = new rp_kmeans();
    h->device = device;
    h->K = K; h->N = N; h->bins = bins; h->kind = kind; h->seed = seed;
    if (hp) h->hp = *hp; else rp_sinkhorn_hp_default(&h->hp);
    h->stride = counts_on_device ? bins : ((bins + 15u) & ~15u);
#define KM_TRY(expr)                                                                          \
    do {                                                                                      \
        int _rc = (expr);                                                                     \
        if (_rc) { rp_kmeans_destroy(h); return _rc; }                                        \
    } while (0)
#define KM_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            int _rc = rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));    \
            rp_kmeans_destroy(h);                                                             \
            return _rc;                                                                       \
        }                                                                                     \
    } while (0)
    KM_HIP(hipSetDevice(device));
    KM_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    uint8_t* d_counts = nullptr;
    if (counts_on_device) {
        d_counts = const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(counts));
        h->owns_counts = false;
    } else {
        KM_TRY(dev_alloc(h, &d_counts, (size_t)N * h->stride));
        if (h->stride == bins) {
            KM_HIP(hipMemcpy(d_counts, counts, (size_t)N * bins, hipMemcpyHostToDevice));
        } else {  // pad rows to 16 B on the host: one large copy instead of N pitched ones
            std::vector<uint8_t> padded((size_t)N * h->stride, 0);
            const uint8_t* src = reinterpret_cast<const uint8_t*>(counts);
            for (uint64_t i = 0; i < N; ++i) memcpy(&padded[i * h->stride], src + i * bins, bins);
            KM_HIP(hipMemcpy(d_counts, padded.data(), padded.size(), hipMemcpyHostToDevice));
        }
    }
    uint32_t* d_w = nullptr;
    float* d_self = nullptr;
    KM_TRY(dev_alloc(h, &d_w, N));
    KM_TRY(dev_alloc(h, &d_self, N));
    h->P = Points{d_counts, d_w, d_self, h->stride, N};
    // metric matrices
    float *d_C = nullptr, *d_R = nullptr;
    KM_TRY(dev_alloc(h, &d_C, (size_t)bins * bins));
    KM_TRY(dev_alloc(h, &d_R, (size_t)bins * bins));
    {
        std::vector<float> C((size_t)bins * bins, 0.0f), R((size_t)bins * bins, 0.0f);
        if (kind == RP_METRIC_SINKHORN) {
            for (uint32_t x = 0; x < bins; ++x)
                for (uint32_t y = 0; y < bins; ++y) {
                    const float c = x == y ? 0.0f : tri_metric[rp_tri_index(x, y)];
                    C[(size_t)x * bins + y] = c;
                    R[(size_t)x * bins + y] = c / h->hp.temperature;  // regularization (sinkhorn.rs:129-131)
                }
        }
        KM_HIP(hipMemcpy(d_C, C.data(), C.size() * 4, hipMemcpyHostToDevice));
        KM_HIP(hipMemcpy(d_R, R.data(), R.size() * 4, hipMemcpyHostToDevice));
    }
    KM_TRY(dev_alloc(h, &h->stats, (size_t)KM_STAT_STRIPES * STAT_STRIDE));
    KM_HIP(hipMemset(h->stats, 0, (size_t)KM_STAT_STRIPES * STAT_STRIDE * 8));
    h->M = Metric{d_C, d_R, bins, h->hp.iterations, h->hp.tolerance, h->stats, KM_STAT_STRIPES, nullptr, nullptr};
    {
        KM_TRY(dev_alloc(h, &h->M.kpp_d, N));
        KM_TRY(dev_alloc(h, &h->M.kpp_j, N));
        KM_TRY(dev_alloc(h, &h->kpp_todo, N));
        KM_TRY(dev_alloc(h, &h->kpp_ntodo, 1));
    }
    KM_TRY(alloc_centroid_set(h, &h->cs[0]));
    KM_TRY(alloc_centroid_set(h, &h->cs[1]));
    KM_TRY(dev_alloc(h, &h->B.j, N));
    KM_TRY(dev_alloc(h, &h->B.u, N));
    KM_TRY(dev_alloc(h, &h->B.stale, N));
    KM_TRY(dev_alloc(h, &h->B.lower, (size_t)N * K));
    KM_TRY(dev_alloc(h, &h->prior, N));
    KM_TRY(dev_alloc(h, &h->tmp_j, N));
    KM_TRY(dev_alloc(h, &h->pairw, (size_t)K * K));
    h->memo_on = kind == RP_METRIC_SINKHORN;
    if (h->memo_on) {
        KM_TRY(dev_alloc(h, &h->cver, (size_t)K));
        KM_TRY(dev_alloc(h, &h->pver, (size_t)K * K * 2));
        KM_TRY(dev_alloc(h, &h->B.memo_d, N));
        KM_TRY(dev_alloc(h, &h->B.memo_ver, N));
        KM_TRY(dev_alloc(h, &h->B.memo_j, N));
        h->B.cver = h->cver;
    }
    KM_TRY(dev_alloc(h, &h->mid, K));
    KM_TRY(dev_alloc(h, &h->drift, K));
    KM_TRY(dev_alloc(h, &h->pot, N));
    KM_TRY(dev_alloc(h, &h->pdist, N));
    KM_TRY(dev_alloc(h, &h->bsum, (N + KPP_BLOCK - 1) / KPP_BLOCK));
    KM_TRY(dev_alloc(h, &h->scal, 2));
    KM_TRY(dev_alloc(h, &h->sizes, K));
    KM_TRY(dev_alloc(h, &h->hist_stage, MAXB));
    // point masses and memoised self costs OT(p,p) (sinkhorn.rs:175-191)
    hipLaunchKernelGGL(k_point_weights, dim3((unsigned)N), dim3(64), 0, h->stream, d_counts, h->stride, bins, N, d_w);
    KM_HIP(hipGetLastError());
    if (kind == RP_METRIC_SINKHORN) {
        hipLaunchKernelGGL(KSEL(h, k_point_self), dim3((unsigned)N), dim3(64), 0, h->stream, h->P, h->M, d_self);
        KM_HIP(hipGetLastError());
    } else {
        KM_HIP(hipMemsetAsync(d_self, 0, N * 4, h->stream));
    }
    KM_HIP(hipStreamSynchronize(h->stream));
    std::vector<uint8_t> ns;
    uint8_t* d_ns = nullptr;
    if (kind == RP_METRIC_SINKHORN) {
        KM_TRY(dev_alloc(h, &d_ns, N));
        hipLaunchKernelGGL(k_point_support, dim3((unsigned)N), dim3(64), 0, h->stream, h->P, bins, d_ns);
        KM_HIP(hipGetLastError());
        ns.resize(N);
        KM_HIP(hipMemcpyAsync(ns.data(), d_ns, N, hipMemcpyDeviceToHost, h->stream));  // same (non-blocking) stream as the kernel
        KM_HIP(hipStreamSynchronize(h->stream));
    }
    if (kind == RP_METRIC_SINKHORN && !getenv("RP_LLOYD_NO_KPP_BOUND")) {
        float cmax = 0.0f;
        for (size_t t = 0; t < (size_t)bins * (bins - 1) / 2; ++t) cmax = std::max(cmax, tri_metric[t]);
        if (cmax / h->hp.temperature <= 64.0f) {  // no softmin term near the MIN_POSITIVE clamp: column sums are nu to ~1e-5
            h->d_nsup = d_ns;
            KM_TRY(dev_alloc(h, &h->minc, MAXB));
            for (int c = 0; c < 3; ++c) KM_TRY(dev_alloc(h, &h->kpp.list[c], (size_t)N + 4));
            KM_TRY(dev_alloc(h, &h->kpp.count, 4));
            for (uint64_t i = 0; i < N; ++i) h->kpp_cap[ns[i] <= QUAD_ROWS ? 0 : (ns[i] <= PAIR_ROWS ? 1 : 2)] += 1;
            h->kpp_lb = true;
        }
    }
    if (kind == RP_METRIC_SINKHORN && !getenv("RP_LLOYD_NO_MFMA_BOUND")) {
        // the MFMA bound (sinkhorn_bound.hpp): K = exp(-C/T) padded to 256 x 256, point lists by ceil(support / 16)
        std::vector<float> Km((size_t)MAXB * MAXB, 1.0f);
        for (uint32_t x = 0; x < bins; ++x)
            for (uint32_t y = 0; y < bins; ++y) {
                const float c = x == y ? 0.0f : tri_metric[rp_tri_index(x, y)];
                Km[(size_t)x * MAXB + y] = (float)std::exp(-(double)c / (double)h->hp.temperature);
            }
        float* d_K = nullptr;
        KM_TRY(dev_alloc(h, &d_K, Km.size()));
        KM_HIP(hipMemcpy(d_K, Km.data(), Km.size() * 4, hipMemcpyHostToDevice));
        auto envf = [](const char* name, float dflt) {
            const char* v = getenv(name);
            return v ? (float)atof(v) : dflt;
        };
        h->sb.Kmat = d_K;
        h->sb.neg_t_ln2 = -h->hp.temperature * 0.6931472f;
        h->sb.tol = h->hp.tolerance;
        h->sb.iters = h->hp.iterations;
        // the margins can be WIDENED through the environment (more survivors, same results); values on the unsafe side of the
        // validated defaults (profiles/r03_mfma_audit.json) are clamped back to them
        h->sb.kappa = std::max(envf("RP_SB_KAPPA", 2.0f), 2.0f);
        h->sb.rho = std::max(envf("RP_SB_RHO", 1.25f), 1.25f);
        h->sb.dc_abs = std::max(envf("RP_SB_DC_ABS", 4e-6f), 4e-6f);
        h->sb.dc_rel = std::max(envf("RP_SB_DC_REL", 4e-5f), 4e-5f);
        h->sb.flat = std::min(envf("RP_SB_FLAT", 4.0f), 4.0f);
        h->sb.lip = getenv("RP_SB_LIP") ? atoi(getenv("RP_SB_LIP")) : 2;
        h->sb.tight = getenv("RP_SB_TIGHT") ? (uint32_t)std::max(0, atoi(getenv("RP_SB_TIGHT"))) : SB_TIGHT_EVERY;
        {
            float cmax = 0.0f;
            for (size_t t = 0; t < (size_t)bins * (bins - 1) / 2; ++t) cmax = std::max(cmax, tri_metric[t]);
            h->sb.use_lb0 = (cmax / h->hp.temperature <= 64.0f && !getenv("RP_SB_NO_LB0")) ? 1 : 0;
        }
        std::vector<uint32_t> cls[4], big;
        for (uint64_t i = 0; i < N; ++i) {
            // k_point_support saturates at 255 bins; 0 (an empty histogram) is left to the exact kernel as well
            const uint32_t t = ns[i] == 0 || ns[i] > SB_MAXROWS ? 4u : (uint32_t)(ns[i] - 1) / 16u;
            (t < 4 ? cls[t] : big).push_back((uint32_t)i);
        }
        for (int t = 0; t < 4; ++t) {
            h->sb_count[t] = (uint32_t)cls[t].size();
            KM_TRY(dev_alloc(h, &h->sb_list[t], cls[t].size()));
            if (!cls[t].empty()) KM_HIP(hipMemcpy(h->sb_list[t], cls[t].data(), cls[t].size() * 4, hipMemcpyHostToDevice));
        }
        h->sb_nbig = (uint32_t)big.size();
        KM_TRY(dev_alloc(h, &h->sb_big, big.size()));
        if (!big.empty()) KM_HIP(hipMemcpy(h->sb_big, big.data(), big.size() * 4, hipMemcpyHostToDevice));
        KM_TRY(dev_alloc(h, &h->sb_cursor, 4));
        KM_TRY(dev_alloc(h, &h->sb_mask, (size_t)N * 4));
        KM_TRY(dev_alloc(h, &h->sb_stats, (size_t)KM_STAT_STRIPES * STAT_STRIDE));
        KM_HIP(hipMemset(h->sb_stats, 0, (size_t)KM_STAT_STRIPES * STAT_STRIDE * 8));
        KM_TRY(dev_alloc(h, &h->sb_bad, 2));
        KM_HIP(hipMemset(h->sb_bad, 0, 16));
        KM_TRY(dev_alloc(h, &h->sb_d, N));
        KM_TRY(dev_alloc(h, &h->sb_ub0, N));
        KM_TRY(dev_alloc(h, &h->sb_crank, MAXB));
        KM_TRY(dev_alloc(h, &h->sb_hint_j, N));
        KM_TRY(dev_alloc(h, &h->sb_hint_mask, (size_t)N * 4));
        h->sb_audit = getenv("RP_LLOYD_AUDIT") != nullptr;
        {
            // production self-check: every SB_SAMPLE_STRIDE-th point goes through the unpruned search after every pruned pass;
            // a disagreement fails the next call that hands results out (never a silently different bucket)
            std::vector<uint32_t> smp;
            for (uint64_t i = 0; i < N; i += SB_SAMPLE_STRIDE) smp.push_back((uint32_t)i);
            h->sb_nsample = (uint32_t)smp.size();
            KM_TRY(dev_alloc(h, &h->sb_sample, smp.size()));
            if (!smp.empty()) KM_HIP(hipMemcpy(h->sb_sample, smp.data(), smp.size() * 4, hipMemcpyHostToDevice));
            KM_TRY(dev_alloc(h, &h->audit_j, N));
            KM_TRY(dev_alloc(h, &h->audit_d, N));
        }
        h->sb_on = true;
        if (h->kpp_lb && !getenv("RP_LLOYD_NO_KPP_BOUND2")) {  // the second k-means++ filter shares K = exp(-C/T) and the margins
            for (int c = 0; c < 3; ++c) KM_TRY(dev_alloc(h, &h->kpp2.list[c], (size_t)N + 4));
            KM_TRY(dev_alloc(h, &h->kpp2.count, 4));
            KM_TRY(dev_alloc(h, &h->kb_cursor, 4));
            KM_TRY(dev_alloc(h, &h->kb_stats, (size_t)KM_STAT_STRIPES * STAT_STRIDE));
            KM_HIP(hipMemset(h->kb_stats, 0, (size_t)KM_STAT_STRIPES * STAT_STRIDE * 8));
            h->kb_on = true;
            if (getenv("RP_KPP_DUAL")) h->kb_dual = std::max(1, std::min(2, atoi(getenv("RP_KPP_DUAL"))));
            if (!getenv("RP_LLOYD_NO_SAMPLE_CHECK")) {  // the rounds' tripwire: claims of the sampled points, checked by the solves (kpp_note)
                KM_TRY(dev_alloc(h, &h->M.kpp_claim, (size_t)N));
                KM_HIP(hipMemset(h->M.kpp_claim, 0, (size_t)N * 4));
                h->M.kpp_bad = h->sb_bad + 1;
                if (getenv("RP_KPP_CLAIM_TEST")) h->kb_claim_scale = (float)atof(getenv("RP_KPP_CLAIM_TEST"));
            }
        }
    }
    h->ns_host = ns;
    h->cent_m.assign(K, 0u);
    // RP_LLOYD_GROUPING (tests): "none" = one point per wavefront everywhere, "pairs" = no groups of four, "norefresh" = no regrouped
    // refresh pass; every grouping performs the same float operations per solve (tests/test_gpu_lloyd.py)
    const std::string grouping = getenv("RP_LLOYD_GROUPING") ? getenv("RP_LLOYD_GROUPING") : "";
    if (kind == RP_METRIC_SINKHORN && grouping != "none") {
        // grouping lists: <= QUAD_ROWS bins -> four per wavefront, <= PAIR_ROWS -> two, the others one
        std::vector<uint32_t> tiny, small, rest;
        const bool no_quads = grouping == "pairs";
        if (grouping != "norefresh") {
            h->refresh.nsup = d_ns;
            KM_TRY(dev_alloc(h, &h->refresh.count, (size_t)K));
            KM_TRY(dev_alloc(h, &h->refresh.offset, (size_t)K + 1));
            KM_TRY(dev_alloc(h, &h->refresh.list, (size_t)N + 2 * (size_t)K));
            // the interval-decided refresh needs the scaling-domain table of the MFMA bound, the remembered refreshes' versions and a
            // cost matrix whose rows are 16-byte aligned (RP_LLOYD_NO_REFRESH_BOUND=1: off, every refresh is the bit-faithful solve)
            if (h->sb_on && bins % 4 == 0 && !getenv("RP_LLOYD_NO_REFRESH_BOUND")) {
                KM_TRY(dev_alloc(h, &h->B.ulo, N));
                KM_TRY(dev_alloc(h, &h->B.uiv, N));
                KM_TRY(dev_alloc(h, &h->B.im_lo, N));
                KM_TRY(dev_alloc(h, &h->B.im_hi, N));
                KM_TRY(dev_alloc(h, &h->B.im_ver, N));
                KM_TRY(dev_alloc(h, &h->B.im_j, N));
                KM_HIP(hipMemset(h->B.im_ver, 0, (size_t)N * 4));
                KM_TRY(dev_alloc(h, &h->rb_code, N));
                KM_TRY(dev_alloc(h, &h->rb_cursor, 2));
                KM_TRY(dev_alloc(h, &h->rb_stats, (size_t)KM_STAT_STRIPES * STAT_STRIDE));
                KM_HIP(hipMemset(h->rb_stats, 0, (size_t)KM_STAT_STRIPES * STAT_STRIDE * 8));
                KM_HIP(hipMemset(h->B.uiv, 0, N));
                KM_HIP(hipMemset(h->B.ulo, 0, (size_t)N * 4));
                h->rb_on = true;
            }
        }
        for (uint64_t i = 0; i < N; ++i) (ns[i] <= QUAD_ROWS && !no_quads ? tiny : (ns[i] <= PAIR_ROWS ? small : rest)).push_back((uint32_t)i);
        while (tiny.size() & 3u) {
            small.push_back(tiny.back());
            tiny.pop_back();
        }
        if (small.size() & 1u) {
            rest.push_back(small.back());
            small.pop_back();
        }
        h->n_quads = tiny.size() / 4;
        h->n_pairs = small.size() / 2;
        h->n_singles = rest.size();
        KM_TRY(dev_alloc(h, &h->quads, tiny.size()));
        KM_TRY(dev_alloc(h, &h->pairs, small.size()));
        KM_TRY(dev_alloc(h, &h->singles, rest.size()));
        if (!tiny.empty()) KM_HIP(hipMemcpy(h->quads, tiny.data(), tiny.size() * 4, hipMemcpyHostToDevice));
        if (!small.empty()) KM_HIP(hipMemcpy(h->pairs, small.data(), small.size() * 4, hipMemcpyHostToDevice));
        if (!rest.empty()) KM_HIP(hipMemcpy(h->singles, rest.data(), rest.size() * 4, hipMemcpyHostToDevice));
    }
#undef KM_TRY
#undef KM_HIP
    // hipMemset on device memory returns before it has run, and a non-blocking stream does not wait for the null stream: the
    // first launch on this handle's stream could otherwise overtake the initialisation above and be overwritten by it
    (void)hipDeviceSynchronize();
    *out = h;
    return RP_OK;
}

size_t partial_sizes_offset(const rp_kmeans* h) { return ((((size_t)h->K * h->bins + h->K) * 4) + 7) & ~(size_t)7; }

int need_centroids(const rp_kmeans* h, const char* who) {
    if (!h->centroids_ready) return rp::fail(RP_ERR_INVALID, "%s: centroids not initialised (init_centroids / set_centroids)", who);
    return RP_OK;
}
int need_bounds(const rp_kmeans* h, const char* who) {
    if (!h->bounds_ready) return rp::fail(RP_ERR_INVALID, "%s: bounds not initialised (init_bounds)", who);
    return RP_OK;
}

// every (point, centroid) distance through the bit-faithful kernels: the reference's loop as it stands
// the production self-check of the pruned passes (launch_neighbor, the init_bounds shortcut): a sampled point whose pruned result
// differs from the unpruned search is a library bug that would silently move a bucket — the call fails instead
int prune_check(rp_kmeans* h, const char* who) {
    if (!h->sb_on || !h->sb_check_pending) return RP_OK;
    unsigned long long bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, h->sb_bad + 1, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->sb_check_pending = false;
    if (bad)
        return rp::fail(RP_ERR_INTERNAL, "%s: the pruned nearest-centroid search disagrees with the unpruned one on %llu of the sampled points "
                                         "(set RP_LLOYD_NO_MFMA_BOUND=1 and report)", who, bad);
    return RP_OK;
}
// Elkan::neighbor for the points of a (short) device list: K centroids per point split over NB_CHUNKS wavefronts, merged in order
#define NB_LIST_MAX 65536u
int launch_neighbor_list(rp_kmeans* h, const uint32_t* list, uint32_t n, uint8_t* out_j, float* out_d, Bounds init) {
    if (!n) return RP_OK;
    if (h->kind != RP_METRIC_SINKHORN || n > NB_LIST_MAX) {  // enough wavefronts as it is (or the variation metric's own kernel)
        hipLaunchKernelGGL(KSEL(h, k_neighbor), dim3(n), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M, h->kind, out_j, out_d, init, list);
        return RP_OK;
    }
    int rc;
    if (!h->nb_tmp_j && ((rc = dev_alloc(h, &h->nb_tmp_j, (size_t)NB_LIST_MAX * NB_CHUNKS)) || (rc = dev_alloc(h, &h->nb_tmp_d, (size_t)NB_LIST_MAX * NB_CHUNKS))))
        return rc;
    hipLaunchKernelGGL(KSEL(h, k_neighbor_chunk), dim3(n * NB_CHUNKS), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M, list, h->nb_tmp_j,
                       h->nb_tmp_d);
    hipLaunchKernelGGL(k_neighbor_merge, dim3((n + 255u) / 256u), dim3(256), 0, h->stream, list, n, h->K, (const uint8_t*)h->nb_tmp_j,
                       (const float*)h->nb_tmp_d, out_j, out_d, init);
    return RP_OK;
}

int launch_neighbor_full(rp_kmeans* h, uint8_t* out_j, float* out_d, Bounds init) {
    ck_begin(h, CK_NEIGHBOR);
    if (h->kind == RP_METRIC_VARIATION && h->bins == 101)  // turn layer: register-resident centroid CDFs
        hipLaunchKernelGGL(k_neighbor_var<101>, dim3((unsigned)((h->N + VB - 1) / VB)), dim3(256), 0, h->stream, h->P,
                           h->cs[h->cur], h->K, h->M, out_j, out_d, init);
    else if (h->kind == RP_METRIC_SINKHORN && (h->n_pairs || h->n_quads)) {
        if (h->n_quads)
            hipLaunchKernelGGL(KSEL(h, k_neighborG<4>), dim3((unsigned)h->n_quads), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M,
                               h->quads, out_j, out_d, init);
        if (h->n_pairs)
            hipLaunchKernelGGL(KSEL(h, k_neighborG<2>), dim3((unsigned)h->n_pairs), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M,
                               h->pairs, out_j, out_d, init);
        if (h->n_singles)
            hipLaunchKernelGGL(KSEL(h, k_neighbor), dim3((unsigned)h->n_singles), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M,
                               h->kind, out_j, out_d, init, h->singles);
    } else
        hipLaunchKernelGGL(KSEL(h, k_neighbor), dim3((unsigned)h->N), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M, h->kind, out_j,
                           out_d, init, (const uint32_t*)nullptr);
    ck_end(h, CK_NEIGHBOR);
    HIP_TRY(hipGetLastError());
    return RP_OK;
}

// An order of the centroids in which neighbours are similar (a nearest-neighbour chain over the last Elkan step's pairwise
// distances, started at the centroid farthest from centroid 0).  Only the GROUPING of the bound kernel's columns follows it — a block of
// sixteen columns runs until its slowest one is done, and against one point similar centroids need similar numbers of iterations —
// never a result.  Without pairwise distances (no Elkan step yet) the centroids keep their index order.
int centroid_order(rp_kmeans* h) {
    h->sb_crank_set = false;
    if (!h->pairw_seen || !h->sb_crank) return RP_OK;
    const uint32_t K = h->K;
    std::vector<float> pw((size_t)K * K);
    HIP_TRY(hipMemcpyAsync(pw.data(), h->pairw, pw.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    auto dist = [&](uint32_t a, uint32_t b) {
        const float d = pw[(size_t)a * K + b];
        return d == d ? d : INFINITY;
    };
    std::vector<uint8_t> rank(MAXB), used(K, 0);
    for (uint32_t k = 0; k < MAXB; ++k) rank[k] = (uint8_t)k;
    uint32_t at = 0;
    for (uint32_t k = 1; k < K; ++k)
        if (std::isfinite(dist(0, k)) && (at == 0 || dist(0, k) > dist(0, at))) at = k;
    for (uint32_t pos = 0; pos < K; ++pos) {
        used[at] = 1;
        rank[at] = (uint8_t)pos;
        uint32_t best = K;
        for (uint32_t k = 0; k < K; ++k)
            if (!used[k] && (best == K || dist(at, k) < dist(at, best))) best = k;
        if (best == K) break;
        at = best;
    }
    HIP_TRY(hipMemcpyAsync(h->sb_crank, rank.data(), MAXB, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));  // `rank` is a local
    h->sb_crank_set = true;
    return RP_OK;
}

// the MFMA bound over every point (one launch per support class), leaving the survivor masks in h->sb_mask
int launch_bound(rp_kmeans* h, float* dbg_lo, float* dbg_hi, const float* ub0 = nullptr) {
    HIP_TRY(hipMemsetAsync(h->sb_cursor, 0, 16, h->stream));
    const CentroidSet& cs = h->cs[h->cur];
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
    ck_begin(h, CK_BOUND);
    for (int t = 3; t >= 0; --t) {  // the longest-running class first
        const uint32_t n = h->sb_count[t];
        if (!n) continue;
        const uint32_t per_cu = t <= 1 ? 4u : 2u;  // workgroups (= points) a CU holds (LDS: SbLds<NT>; registers: 8 wavefronts of up to 256)
        const dim3 grid(std::min<uint32_t>(n, (uint32_t)cus * per_cu)), block(t <= 1 ? 128u : 256u);
#define SB_LAUNCH1(NT, LIP)                                                                                                        \
    hipLaunchKernelGGL((k_sinkhorn_bound<NT, LIP>), grid, block, 0, h->stream, h->P, cs, h->K, h->bins, h->sb, h->sb_list[t], n, \
                       h->sb_cursor + t, h->sb_mask, dbg_lo, dbg_hi, h->sb_stats, ub0,                                           \
                       (const uint8_t*)(h->sb_crank_set ? h->sb_crank : nullptr))
#define SB_LAUNCH(NT)                   \
    do {                                \
        if (h->sb.lip >= 2) SB_LAUNCH1(NT, 2); \
        else if (h->sb.lip) SB_LAUNCH1(NT, 1); \
        else SB_LAUNCH1(NT, 0);         \
    } while (0)
        if (t == 0) SB_LAUNCH(1);
        else if (t == 1) SB_LAUNCH(2);
        else if (t == 2) SB_LAUNCH(3);
        else SB_LAUNCH(4);
#undef SB_LAUNCH
#undef SB_LAUNCH1
    }
    if (h->sb_nbig) hipLaunchKernelGGL(k_mask_all, dim3((h->sb_nbig + 255) / 256), dim3(256), 0, h->stream, h->sb_big, h->sb_nbig, h->sb_mask);
    ck_end(h, CK_BOUND);
    HIP_TRY(hipGetLastError());
    return RP_OK;
}

// Elkan::neighbor for every point (init_bounds / Layer::lookup / step_naive).  Sinkhorn layers: the MFMA bound discards
// the centroids that cannot be the argmin, the bit-faithful kernel runs on the survivors (same bits).
enum { NB_INIT_BOUNDS, NB_LOOKUP, NB_NAIVE };
int launch_neighbor(rp_kmeans* h, uint8_t* out_j, float* out_d, Bounds init, int pass = NB_NAIVE) {
    if (!h->sb_on) return launch_neighbor_full(h, out_j, out_d, init);
    const unsigned nblk = (unsigned)((h->N + 255) / 256);
    const float* ub0 = nullptr;
    const uint8_t* hint_j = nullptr;
    if (h->sb.use_lb0) {
        if (pass == NB_INIT_BOUNDS && h->pot_is_min_d2) {
            // right after k-means++: potentials = min_k d(c_k, x)^2 for exactly these centroids
            hipLaunchKernelGGL(k_ub_from_pot, dim3(nblk), dim3(256), 0, h->stream, h->pot, h->N, h->sb_ub0);
            ub0 = h->sb_ub0;
        } else if (pass == NB_LOOKUP && h->bounds_ready) {
            // after the Elkan iterations: the exact distance to the point's assigned centroid (one solve per point, reused below)
            HIP_TRY(hipMemcpyAsync(h->sb_hint_j, h->B.j, h->N, hipMemcpyDeviceToDevice, h->stream));
            hipLaunchKernelGGL(k_hint_masks, dim3(nblk), dim3(256), 0, h->stream, h->sb_hint_j, h->N, h->K, h->sb_hint_mask);
            ck_begin(h, CK_NEIGHBOR);
            Bounds none{};
            hipLaunchKernelGGL(KSEL(h, k_neighbor_masked), dim3((unsigned)h->N), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M,
                               h->sb_hint_mask, (uint8_t*)nullptr, h->sb_ub0, none, (const uint8_t*)nullptr, (const float*)nullptr);
            ck_end(h, CK_NEIGHBOR);
            ub0 = h->sb_ub0;
            hint_j = h->sb_hint_j;
        }
    }
    int rc = centroid_order(h);
    if (rc || (rc = launch_bound(h, nullptr, nullptr, ub0))) return rc;
    float* dd = out_d ? out_d : h->sb_d;
    ck_begin(h, CK_NEIGHBOR);
    hipLaunchKernelGGL(KSEL(h, k_neighbor_masked), dim3((unsigned)h->N), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M, h->sb_mask,
                       out_j, dd, init, hint_j, hint_j ? ub0 : (const float*)nullptr);
    ck_end(h, CK_NEIGHBOR);
    HIP_TRY(hipGetLastError());
    if (!h->sb_audit && out_j && h->sb_nsample) {  // the sampled points once more, unpruned (k_neighbor takes a point list)
        Bounds none{};
        ck_begin(h, CK_NEIGHBOR);
        if ((rc = launch_neighbor_list(h, h->sb_sample, h->sb_nsample, h->audit_j, h->audit_d, none))) return rc;
        ck_end(h, CK_NEIGHBOR);
        hipLaunchKernelGGL(k_audit_compare_list, dim3((h->sb_nsample + 255) / 256), dim3(256), 0, h->stream, out_j, dd, h->audit_j,
                           h->audit_d, h->sb_sample, h->sb_nsample, h->sb_bad + 1);
        HIP_TRY(hipGetLastError());
        h->sb_sampled += h->sb_nsample;
        h->sb_check_pending = true;
    }
    if (h->sb_audit && out_j) {  // RP_LLOYD_AUDIT: the unpruned pass next to it; disagreements are counted, never corrected
        Bounds none{};
        if ((rc = launch_neighbor_full(h, h->audit_j, h->audit_d, none))) return rc;
        hipLaunchKernelGGL(k_audit_compare, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, out_j, dd, h->audit_j,
                           h->audit_d, h->N, h->sb_bad);
        HIP_TRY(hipGetLastError());
        h->sb_audited += h->N;
    }
    return RP_OK;
}

int launch_recompute(rp_kmeans* h, const uint8_t* assign, int set) {
    ck_begin(h, CK_RECOMPUTE);
    const unsigned slices = h->N >= (1ull << 17) ? 8u : 1u;  // workgroups per centroid (k_recompute adds into zeroed outputs when > 1)
    if (slices > 1) {
        HIP_TRY(hipMemsetAsync(h->cs[set].counts, 0, (size_t)h->K * h->bins * 4, h->stream));
        HIP_TRY(hipMemsetAsync(h->cs[set].weight, 0, (size_t)h->K * 4, h->stream));
        HIP_TRY(hipMemsetAsync(h->sizes, 0, (size_t)h->K * 8, h->stream));
    }
    hipLaunchKernelGGL(k_recompute, dim3(h->K, slices), dim3(256), 0, h->stream, h->P, assign, h->bins, h->cs[set].counts,
                       h->cs[set].weight, h->sizes);
    ck_end(h, CK_RECOMPUTE);
    HIP_TRY(hipGetLastError());
    return RP_OK;
}

// the part of step_elkan before the centroid exchange: pairwise, midpoints, bound refresh, partial sums
int step_front(rp_kmeans* h) {
    const int cur = h->cur;
    if (h->memo_on && h->memo_dirty) {  // centroids installed outside an Elkan step: nothing remembered is valid
        h->memo_epoch += 1;
        hipLaunchKernelGGL(k_fill_u32, dim3((h->K + 255) / 256), dim3(256), 0, h->stream, h->cver, h->K, h->memo_epoch << 16);
        HIP_TRY(hipMemsetAsync(h->pver, 0, (size_t)h->K * h->K * 8, h->stream));
        HIP_TRY(hipMemsetAsync(h->B.memo_ver, 0, (size_t)h->N * 4, h->stream));
        if (h->B.im_ver) HIP_TRY(hipMemsetAsync(h->B.im_ver, 0, (size_t)h->N * 4, h->stream));
        h->memo_dirty = false;
    }
    ck_begin(h, CK_PAIRWISE);
    if (h->kind == RP_METRIC_VARIATION)
        hipLaunchKernelGGL(k_pairwise_var, dim3((h->K * h->K + 255) / 256), dim3(256), 0, h->stream, h->cs[cur], h->K, h->M, h->pairw);
    else
        hipLaunchKernelGGL(KSEL(h, k_pairwise), dim3(h->K * h->K), dim3(64), 0, h->stream, h->cs[cur], h->K, h->M, h->kind, h->pairw, h->cver, h->pver);
    hipLaunchKernelGGL(k_midpoints, dim3((h->K + 63) / 64), dim3(64), 0, h->stream, h->pairw, h->K, h->mid);
    ck_end(h, CK_PAIRWISE);
    h->pairw_seen = true;
    ck_begin(h, CK_STEP);
    if (h->kind == RP_METRIC_VARIATION && h->bins == 101)
        hipLaunchKernelGGL(k_elkan_step_var<101>, dim3((unsigned)((h->N + VB - 1) / VB)), dim3(256), 0, h->stream, h->P, h->cs[cur],
                           h->K, h->M, h->B, h->pairw, h->mid);
    else {
        if (h->kind == RP_METRIC_SINKHORN && h->refresh.nsup) {
            const size_t entries = (size_t)h->N + 2 * (size_t)h->K;
            // the stale-bound refresh as its own pass: points bucketed by centroid, two per wavefront (k_refresh_pairs)
            auto bucket = [&](uint8_t want) -> int {
                h->refresh.want = want;
                HIP_TRY(hipMemsetAsync(h->refresh.count, 0, (size_t)h->K * 4, h->stream));
                HIP_TRY(hipMemsetAsync(h->refresh.list, 0xff, entries * 4, h->stream));
                hipLaunchKernelGGL(k_refresh_count, dim3(1024), dim3(256), 0, h->stream, h->B, h->refresh, h->mid, h->N, h->K);
                hipLaunchKernelGGL(k_refresh_offsets, dim3(1), dim3(1), 0, h->stream, h->refresh, h->K);
                hipLaunchKernelGGL(k_refresh_fill, dim3(1024), dim3(256), 0, h->stream, h->B, h->refresh, h->mid, h->N);
                return RP_OK;
            };
            int rc = RP_OK;
            if (h->rb_on) {
                // interval mode (refresh_bound.hpp): (1) what every stale point needs; (2) interval-valued bounds the filter cannot be
                // decided on become exact: one solve against the PREVIOUS centroids (cs[cur ^ 1] is still last step's set) and the last
                // drift; (3) the refreshes: an interval first, settled where it stays under every threshold of the candidate loop;
                // (4) the bit-faithful refresh for the rest
                h->refresh.code = h->rb_code;
                hipLaunchKernelGGL(k_rb_prepare, dim3(1024), dim3(256), 0, h->stream, h->B, h->refresh.nsup, h->mid, h->drift, h->N, h->K,
                                   h->rb_code, h->stats);
                if ((rc = bucket(3))) return rc;
                hipLaunchKernelGGL(KSEL(h, k_refresh_pairs), dim3((unsigned)(entries / 2 + 1)), dim3(64), 0, h->stream, h->P, h->cs[cur ^ 1], h->K,
                                   h->M, h->B, h->refresh, 1, h->drift, h->mid, h->rb_code);
                if ((rc = bucket(1))) return rc;
                HIP_TRY(hipMemsetAsync(h->rb_cursor, 0, 8, h->stream));
                ck_begin(h, CK_REFRESH_BOUND);
                hipLaunchKernelGGL((k_refresh_interval<32>), dim3(4096), dim3(64), 0, h->stream, h->P, h->cs[cur], h->K, h->bins, h->M.Cm,
                                   h->sb, h->B, h->pairw, h->refresh.list, h->refresh.offset + h->K, h->rb_cursor, h->rb_code, h->rb_stats, h->stats);
                ck_end(h, CK_REFRESH_BOUND);
                if ((rc = bucket(2))) return rc;
            } else {
                h->refresh.code = nullptr;
                if (h->B.memo_ver) hipLaunchKernelGGL(k_refresh_memo, dim3(1024), dim3(256), 0, h->stream, h->B, h->mid, h->N, h->K, h->M);
                if ((rc = bucket(0))) return rc;
            }
            hipLaunchKernelGGL(KSEL(h, k_refresh_pairs), dim3((unsigned)(entries / 2 + 1)), dim3(64), 0, h->stream, h->P, h->cs[cur], h->K, h->M,
                               h->B, h->refresh, 0, (const float*)nullptr, (const float*)nullptr, (uint8_t*)nullptr);
        }
        hipLaunchKernelGGL(KSEL(h, k_elkan_step), dim3((unsigned)h->N), dim3(64), 0, h->stream, h->P, h->cs[cur], h->K, h->M, h->kind, h->B,
                           h->pairw, h->mid);
    }
    ck_end(h, CK_STEP);
    HIP_TRY(hipGetLastError());
    return launch_recompute(h, h->B.j, cur ^ 1);
}

// after the (optional) all-reduce of the integer sums in cs[cur^1]: drift, bounds update, install, tally
int step_back(rp_kmeans* h, float* drift, uint64_t* sizes, double* reassigned) {
    const int cur = h->cur, nxt = cur ^ 1;
    int rc = prepare_centroids(h, nxt, true);
    if (rc) return rc;
    ck_begin(h, CK_DRIFT);
    if (h->kind == RP_METRIC_SINKHORN)
        hipLaunchKernelGGL(KSEL(h, k_drift_block), dim3(h->K), dim3(256), 0, h->stream, h->cs[nxt], h->cs[cur], h->K, h->M, h->drift);
    else
        hipLaunchKernelGGL(KSEL(h, k_drift), dim3(h->K), dim3(64), 0, h->stream, h->cs[nxt], h->cs[cur], h->K, h->M, h->kind, h->drift);
    ck_end(h, CK_DRIFT);
    ck_begin(h, CK_BOUNDS);
    hipLaunchKernelGGL(k_bounds_update, dim3(2048), dim3(256), 0, h->stream, h->B, h->N, h->K, h->drift);
    ck_end(h, CK_BOUNDS);
    HIP_TRY(hipMemsetAsync(h->scal + 1, 0, 8, h->stream));
    hipLaunchKernelGGL(k_tally, dim3(1024), dim3(256), 0, h->stream, h->B.j, h->prior, h->N, h->scal + 1);
    HIP_TRY(hipGetLastError());
    h->cur = nxt;  // Kmeans::next installs the new centroids (kmeans.rs:88)
    h->pot_is_min_d2 = false;
    std::vector<unsigned long long> sz(h->K);
    unsigned long long moved = 0;
    if (drift) HIP_TRY(hipMemcpyAsync(drift, h->drift, h->K * 4, hipMemcpyDeviceToHost, h->stream));
    if (sizes) HIP_TRY(hipMemcpyAsync(sz.data(), h->sizes, h->K * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(&moved, h->scal + 1, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (sizes) for (uint32_t k = 0; k < h->K; ++k) sizes[k] = sz[k];
    if (reassigned) *reassigned = (double)moved / (double)h->N;
    ck_drain(h);
    return RP_OK;
}

}  // namespace

extern "C" {

int rp_kmeans_create(uint32_t K, uint64_t N, uint32_t bins, const uint8_t* counts, rp_metric_kind kind, const float* tri_metric,
                     const rp_sinkhorn_hp* hp, uint64_t seed, int device, rp_kmeans** out) {
    return create_common(K, N, bins, counts, false, kind, tri_metric, hp, seed, device, out);
}
int rp_kmeans_create_device(uint32_t K, uint64_t N, uint32_t bins, const void* counts_dev, rp_metric_kind kind,
                            const float* tri_metric, const rp_sinkhorn_hp* hp, uint64_t seed, int device, rp_kmeans** out) {
    return create_common(K, N, bins, counts_dev, true, kind, tri_metric, hp, seed, device, out);
}

int rp_kmeans_destroy(rp_kmeans* h) {
    if (!h) return RP_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    ck_drain(h);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return RP_OK;
}

int rp_kmeans_set_centroids(rp_kmeans* h, const uint64_t* point_index) {
    if (!h || !point_index) return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_centroids: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    for (uint32_t k = 0; k < h->K; ++k) {
        if (point_index[k] >= h->N) return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_centroids: point index out of range");
        hipLaunchKernelGGL(k_centroid_from_point, dim3(1), dim3(256), 0, h->stream, h->cs[h->cur], k, h->P, point_index[k], h->bins);
    }
    HIP_TRY(hipGetLastError());
    int rc = prepare_centroids(h, h->cur);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->centroids_ready = true;
    h->bounds_ready = false;
    h->pot_is_min_d2 = false;
    return RP_OK;
}

int rp_kmeans_set_rng(rp_kmeans* h, rp_rng_kind kind, int street) {
    if (!h || (kind != RP_RNG_COUNTER && kind != RP_RNG_REFERENCE) || street < 0 || street > 3)
        return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_rng: bad argument");
    h->rng = kind;
    h->street = street;
    return RP_OK;
}

// Which exp / ln the layer's Sinkhorn distances compute with.  RP_LIBM_GLIBC: the lm_glibc pass of every kernel that evaluates them
// (KSEL), point self costs recomputed in it; the filters in front of the exact solves stay (audited in both arithmetics).  To be
// called before the first centroid exists.
int rp_kmeans_set_libm(rp_kmeans* h, rp_libm_kind kind) {
    if (!h || (kind != RP_LIBM_CONTRACT && kind != RP_LIBM_GLIBC)) return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_libm: bad argument");
    if (h->centroids_ready || h->bounds_ready) return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_libm: the layer already has centroids; set the mode first");
    if (kind == h->libm) return RP_OK;
    if (kind == RP_LIBM_CONTRACT) return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_set_libm: a layer switched to glibc's arithmetic stays there");
    HIP_TRY(hipSetDevice(h->device));
    h->libm = kind;
    // The k-means++ column bound, its interval filter and the MFMA bound compute in their own f32 arithmetic and carry margins (4e-5
    // relative, 4e-6 absolute: sinkhorn_bound.hpp).  glibc's distances sit <= 7 ulps from the contract's, three orders inside the
    // margins, and the full-size audit has been run in this arithmetic too (profiles/r05_glibc_audit.json): the filters stay; the
    // sample check of every pruned pass guards them as in the contract pass.  rp_kmeans_set_prune(h, 0) drops them.
    if (h->kind == RP_METRIC_SINKHORN) {  // OT(p, p) of every point (sinkhorn.rs:175-191) in the new arithmetic
        hipLaunchKernelGGL(KSEL(h, k_point_self), dim3((unsigned)h->N), dim3(64), 0, h->stream, h->P, h->M, const_cast<float*>(h->P.self));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return RP_OK;
}

int rp_kmeans_set_prune(rp_kmeans* h, int enable) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_prune: NULL handle");
    if (h->centroids_ready || h->bounds_ready) return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_prune: the layer already has centroids; set the mode first");
    const bool has = h->kpp_lb || h->kb_on || h->sb_on || h->rb_on;
    if (enable) return has ? RP_OK : rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_set_prune: this layer has no filters (variation metric, switched off at creation, or given up)");
    h->kpp_lb = false;
    h->kb_on = false;
    h->sb_on = false;
    if (h->rb_on) {  // every refresh is the bit-faithful solve again: the kernels see no interval arrays
        h->rb_on = false;
        h->B.ulo = nullptr;  // (the allocations stay with the handle)
        h->B.uiv = nullptr;
    }
    return RP_OK;
}

int rp_kmeans_kpp_begin(rp_kmeans* h) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_kpp_begin: NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    CentroidSet& cs = h->cs[h->cur];
    // all centroids start empty so every derived table is well defined for the not-yet-chosen ones
    HIP_TRY(hipMemsetAsync(cs.counts, 0, (size_t)h->K * h->bins * 4, h->stream));
    HIP_TRY(hipMemsetAsync(cs.weight, 0, h->K * 4, h->stream));
    hipLaunchKernelGGL(KSEL(h, k_prepare_centroids), dim3(h->K), dim3(64), 0, h->stream, cs, h->K, h->M, h->kind, 0u, (const float*)nullptr);
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, h->stream, h->pot, h->N, 1.0f);  // potentials = 1 (layer.rs:161)
    if (h->M.kpp_d) hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, h->stream, h->M.kpp_d, h->N, rp_u2f(0x7f800000u));
    HIP_TRY(hipGetLastError());
    h->centroids_ready = false;
    h->bounds_ready = false;
    h->pot_is_min_d2 = false;
    std::fill(h->cent_m.begin(), h->cent_m.end(), 0u);
    return RP_OK;
}

int rp_kmeans_kpp_total(rp_kmeans* h, uint64_t* total) {
    if (!h || !total) return rp::fail(RP_ERR_INVALID, "rp_kmeans_kpp_total: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    const uint32_t nblocks = (uint32_t)((h->N + KPP_BLOCK - 1) / KPP_BLOCK);
    ck_begin(h, CK_KPP);
    hipLaunchKernelGGL(k_kpp_blocksum, dim3(nblocks), dim3(256), 0, h->stream, h->pot, h->N, h->bsum);
    hipLaunchKernelGGL(k_kpp_total, dim3(1), dim3(1024), 0, h->stream, h->bsum, nblocks, h->scal);
    ck_end(h, CK_KPP);
    HIP_TRY(hipGetLastError());
    unsigned long long t = 0;
    HIP_TRY(hipMemcpyAsync(&t, h->scal, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *total = t;
    return RP_OK;
}

// requires a preceding rp_kmeans_kpp_total (block sums) with unchanged potentials, and r < that total
int rp_kmeans_kpp_pick(rp_kmeans* h, uint64_t r, uint64_t* index) {
    if (!h || !index) return rp::fail(RP_ERR_INVALID, "rp_kmeans_kpp_pick: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    const uint32_t nblocks = (uint32_t)((h->N + KPP_BLOCK - 1) / KPP_BLOCK);
    hipLaunchKernelGGL(k_kpp_pick, dim3(1), dim3(1024), 0, h->stream, h->pot, h->M.kpp_d, h->N, h->bsum, nblocks, (unsigned long long)r, h->scal);
    HIP_TRY(hipGetLastError());
    unsigned long long p = 0;
    HIP_TRY(hipMemcpyAsync(&p, h->scal, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *index = p;
    return RP_OK;
}

int rp_kmeans_kpp_update(rp_kmeans* h, uint32_t k) {
    if (!h || k >= h->K) return rp::fail(RP_ERR_INVALID, "rp_kmeans_kpp_update: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    ck_begin(h, CK_KPP);
    if (h->kind == RP_METRIC_VARIATION)
        hipLaunchKernelGGL(k_kpp_update_var, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, h->P, h->cs[h->cur], k, h->K,
                           h->M, h->pot);
    else if (h->kpp_lb) {
        // the column-marginal bound first: only points whose potential can still drop are solved, regrouped by support class
        const bool grouped = h->n_pairs || h->n_quads;
        const uint32_t qrows = grouped && h->n_quads ? QUAD_ROWS : 0u, prows = grouped ? PAIR_ROWS : 0u;
        HIP_TRY(hipMemsetAsync(h->kpp.count, 0, 16, h->stream));
        hipLaunchKernelGGL(k_minc, dim3(1), dim3(256), 0, h->stream, h->cs[h->cur], k, h->M, h->minc);
        hipLaunchKernelGGL(k_kpp_filter, dim3((unsigned)((h->N + 63) / 64)), dim3(1024), 0, h->stream, h->P, h->cs[h->cur], k, h->M,
                           h->minc, h->pot, h->d_nsup, h->kpp, qrows, prows);
        // grids sized for the worst case (every eligible point active); surplus wavefronts leave at once
        const uint64_t cap4 = qrows ? h->kpp_cap[0] : 0, cap2 = prows ? h->kpp_cap[1] + (qrows ? 0 : h->kpp_cap[0]) : 0,
                       cap1 = h->N - cap4 - cap2;
        const KppLists* todo = &h->kpp;
        if (h->kb_on) {
            // the second filter: a scaling-domain interval of each remaining pair (kpp_bound.hpp); the solve runs where the interval's
            // lower end, squared, is still below the potential.  Two pairs per wavefront while both supports fit 32 lanes.
            uint32_t m = h->cent_m[k];
            if (!m) {  // a centroid installed by other means: ask the device
                HIP_TRY(hipMemcpyAsync(&m, h->cs[h->cur].n + k, 4, hipMemcpyDeviceToHost, h->stream));
                HIP_TRY(hipStreamSynchronize(h->stream));
            }
            if (m && m <= 48u) {
                ck_end(h, CK_KPP);  // the interval filter has its own clock ("kpp_bound"): it is not a softmin kernel
                ck_begin(h, CK_KPP_BOUND);
                HIP_TRY(hipMemsetAsync(h->kpp2.count, 0, 16, h->stream));
                HIP_TRY(hipMemsetAsync(h->kb_cursor, 0, 16, h->stream));
                int cus = 256;
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
                const uint64_t caps[3] = {cap4, cap2, cap1};
                for (int c = 0; c < 3; ++c) {
                    if (!caps[c]) continue;
                    const bool two = m <= 32u && c < 2;  // classes 0 / 1 hold points with <= 32 bins
                    const unsigned grid = (unsigned)std::min<uint64_t>((caps[c] + (two ? 1 : 0)) / (two ? 2 : 1), (uint64_t)cus * 16u);
                    if (two)
                        hipLaunchKernelGGL((k_kpp_bound<32, 32>), dim3(grid), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->bins, h->M.Cm, h->sb,
                                           (const float*)h->pot, (const uint32_t*)h->kpp.list[c], (const unsigned int*)(h->kpp.count + c),
                                           h->kb_cursor + c, h->kpp2.list[c], h->kpp2.count + c, h->kb_stats, (float*)nullptr, h->sb.lip >= 2 ? h->kb_dual : 0, h->M.kpp_claim, h->kb_claim_scale);
                    else
                        hipLaunchKernelGGL((k_kpp_bound<64, 48>), dim3(grid), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->bins, h->M.Cm, h->sb,
                                           (const float*)h->pot, (const uint32_t*)h->kpp.list[c], (const unsigned int*)(h->kpp.count + c),
                                           h->kb_cursor + c, h->kpp2.list[c], h->kpp2.count + c, h->kb_stats, (float*)nullptr, h->sb.lip >= 2 ? h->kb_dual : 0, h->M.kpp_claim, h->kb_claim_scale);
                }
                todo = &h->kpp2;
                if (h->M.kpp_claim) h->sb_check_pending = true;  // the solves below check the sampled claims (prune_check reads the counter)
                ck_end(h, CK_KPP_BOUND);
                ck_begin(h, CK_KPP);
            }
        }
        if (cap4)
            hipLaunchKernelGGL(KSEL(h, k_kpp_updateG<4>), dim3((unsigned)((cap4 + 3) / 4)), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->M,
                               todo->list[0], todo->count + 0, h->pot);
        if (cap2)
            hipLaunchKernelGGL(KSEL(h, k_kpp_updateG<2>), dim3((unsigned)((cap2 + 1) / 2)), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->M,
                               todo->list[1], todo->count + 1, h->pot);
        if (cap1)
            hipLaunchKernelGGL(KSEL(h, k_kpp_update), dim3((unsigned)cap1), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->K, h->M, h->kind,
                               h->pot, todo->list[2], todo->count + 2);
    } else {
        if (h->n_pairs || h->n_quads) {
            if (h->n_quads)
                hipLaunchKernelGGL(KSEL(h, k_kpp_updateG<4>), dim3((unsigned)h->n_quads), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->M,
                                   h->quads, (const unsigned int*)nullptr, h->pot);
            if (h->n_pairs)
                hipLaunchKernelGGL(KSEL(h, k_kpp_updateG<2>), dim3((unsigned)h->n_pairs), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->M,
                                   h->pairs, (const unsigned int*)nullptr, h->pot);
            if (h->n_singles)
                hipLaunchKernelGGL(KSEL(h, k_kpp_update), dim3((unsigned)h->n_singles), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->K,
                                   h->M, h->kind, h->pot, h->singles, (const unsigned int*)nullptr);
        } else {
            hipLaunchKernelGGL(KSEL(h, k_kpp_update), dim3((unsigned)h->N), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->K, h->M, h->kind,
                               h->pot, (const uint32_t*)nullptr, (const unsigned int*)nullptr);
        }
    }
    ck_end(h, CK_KPP);
    HIP_TRY(hipGetLastError());
    if (k + 1 == h->K) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->centroids_ready = true;
        h->pot_is_min_d2 = true;  // potentials = min_k d(c_k, x)^2 over the K centroids now installed
        ck_drain(h);
    }
    return RP_OK;
}

int rp_kmeans_get_point(rp_kmeans* h, uint64_t index, uint32_t* counts) {
    if (!h || !counts || index >= h->N) return rp::fail(RP_ERR_INVALID, "rp_kmeans_get_point: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    std::vector<uint8_t> row(h->bins);
    HIP_TRY(hipMemcpyAsync(row.data(), h->P.counts + index * h->P.stride, h->bins, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (uint32_t b = 0; b < h->bins; ++b) counts[b] = row[b];
    return RP_OK;
}

int rp_kmeans_set_centroid(rp_kmeans* h, uint32_t k, const uint32_t* counts) {
    if (!h || !counts || k >= h->K) return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_centroid: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->hist_stage, counts, (size_t)h->bins * 4, hipMemcpyHostToDevice, h->stream));
    {
        uint32_t m = 0;
        for (uint32_t b = 0; b < h->bins; ++b) m += counts[b] > 0;
        h->cent_m[k] = m;
    }
    hipLaunchKernelGGL(k_centroid_from_hist, dim3(1), dim3(256), 0, h->stream, h->cs[h->cur], k, h->hist_stage, h->bins);
    hipLaunchKernelGGL(KSEL(h, k_prepare_centroids), dim3(1), dim3(64), 0, h->stream, h->cs[h->cur], h->K, h->M, h->kind, k, (const float*)nullptr);
    h->memo_dirty = true, h->pairw_seen = false;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));  // `counts` may be a temporary on the caller's side
    h->bounds_ready = false;
    h->pot_is_min_d2 = false;
    return RP_OK;
}

// Layer::init_centroids (layer.rs:140-181) composed from the primitives above
int rp_kmeans_init_centroids(rp_kmeans* h, uint64_t* chosen) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_init_centroids: NULL handle");
    int rc = rp_kmeans_kpp_begin(h);
    if (rc) return rc;
    std::vector<uint32_t> hist(h->bins);
    rp_smallrng rng;
    if (h->rng == RP_RNG_REFERENCE) {  // DefaultHasher::default(); self.street().hash(hasher); SmallRng::seed_from_u64(hasher.finish())
        rp_sip sh;
        rp_defaulthasher_new(&sh);
        rp_defaulthasher_write_u64(&sh, (uint64_t)(int64_t)h->street);  // a fieldless enum hashes its discriminant as isize
        rp_smallrng_seed(&rng, rp_defaulthasher_finish(&sh));
        if (!h->kpp_cum && (rc = dev_alloc(h, &h->kpp_cum, ref_pick_work_floats(h->N)))) return rc;
        if (!h->kr_out && (rc = dev_alloc(h, &h->kr_out, 4))) return rc;
        h->kr_walked = h->kr_chunks = 0;
    }
    for (uint32_t k = 0; k < h->K; ++k) {
        uint64_t total = 0, pick = 0;
        if (h->rng == RP_RNG_REFERENCE) {
            const float v01 = rp_u2f((rp_smallrng_next_u32(&rng) >> 9) | 0x3f800000u) - 1.0f;  // UniformFloat<f32>: [1, 2) - 1
            ck_begin(h, CK_KPP);
            ref_pick_launch(h->stream, h->pot, h->M.kpp_d, h->N, h->kpp_cum, v01, h->kr_out);
            ck_end(h, CK_KPP);
            HIP_TRY(hipGetLastError());
            unsigned long long out[3] = {0, 0, 0};
            HIP_TRY(hipMemcpyAsync(out, h->kr_out, 24, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            const unsigned long long pk = out[0];
            h->kr_walked += out[2], h->kr_chunks += ref_pick_chunks(h->N);
            if (pk >= h->N) return rp::fail(RP_ERR_INVALID, "rp_kmeans_init_centroids: every potential is zero after %u picks (fewer distinct points than K; the reference panics here)", k);
            pick = pk;
        } else {
        if ((rc = rp_kmeans_kpp_total(h, &total))) return rc;
        const uint64_t hsh = rp_stream(h->seed, k);
        if (total == 0) {
            pick = rp_mulhi64(hsh, h->N);
        } else if ((rc = rp_kmeans_kpp_pick(h, rp_mulhi64(hsh, total), &pick))) {
            return rc;
        }
        }
        if (chosen) chosen[k] = pick;
        if (!h->ns_host.empty()) h->cent_m[k] = h->ns_host[pick];
        hipLaunchKernelGGL(k_centroid_from_point, dim3(1), dim3(256), 0, h->stream, h->cs[h->cur], k, h->P, pick, h->bins);
        hipLaunchKernelGGL(KSEL(h, k_prepare_centroids), dim3(1), dim3(64), 0, h->stream, h->cs[h->cur], h->K, h->M, h->kind, k,
                           h->kind == RP_METRIC_SINKHORN ? h->P.self + pick : (const float*)nullptr);  // OT(c, c) = the point's memoised OT(p, p)
        h->memo_dirty = true, h->pairw_seen = false;
        HIP_TRY(hipGetLastError());
        if ((rc = rp_kmeans_kpp_update(h, k))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->centroids_ready = true;
    h->bounds_ready = false;
    ck_drain(h);
    return RP_OK;
}

int rp_kmeans_init_bounds(rp_kmeans* h) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_init_bounds: NULL handle");
    int rc = need_centroids(h, "rp_kmeans_init_bounds");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if (h->M.kpp_d && h->pot_is_min_d2) {
        // right after k-means++ on these very centroids: the nearest centroid of almost every point is already known
        // (k_init_from_kpp); the exact search runs on the rest
        unsigned int n_todo = 0;
        HIP_TRY(hipMemsetAsync(h->kpp_ntodo, 0, 4, h->stream));
        ck_begin(h, CK_NEIGHBOR);
        hipLaunchKernelGGL(k_zero_f32, dim3(2048), dim3(256), 0, h->stream, h->B.lower, (uint64_t)h->N * h->K);
        hipLaunchKernelGGL(k_init_from_kpp, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, h->M, h->N, h->K, h->prior, h->B,
                           h->kpp_todo, h->kpp_ntodo);
        HIP_TRY(hipMemcpyAsync(&n_todo, h->kpp_ntodo, 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (n_todo) {
            Bounds into = h->B;
            into.lower = nullptr;  // zeroed above
            if ((rc = launch_neighbor_list(h, h->kpp_todo, n_todo, h->prior, nullptr, into))) return rc;
        }
        ck_end(h, CK_NEIGHBOR);
        HIP_TRY(hipGetLastError());
        // the shortcut trusts the k-means++ column-marginal filter: checked like the MFMA prune — on the sample in production,
        // on every point under RP_LLOYD_AUDIT — against the unpruned search (bucket and upper bound, bit for bit)
        if (h->sb_on && (h->sb_audit || h->sb_nsample)) {
            Bounds none{};
            if (h->sb_audit) {
                if ((rc = launch_neighbor_full(h, h->audit_j, h->audit_d, none))) return rc;
                hipLaunchKernelGGL(k_audit_compare, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, h->B.j, h->B.u, h->audit_j,
                                   h->audit_d, h->N, h->sb_bad);
                h->sb_audited += h->N;
            } else {
                if ((rc = launch_neighbor_list(h, h->sb_sample, h->sb_nsample, h->audit_j, h->audit_d, none))) return rc;
                hipLaunchKernelGGL(k_audit_compare_list, dim3((h->sb_nsample + 255) / 256), dim3(256), 0, h->stream, h->B.j, h->B.u,
                                   h->audit_j, h->audit_d, h->sb_sample, h->sb_nsample, h->sb_bad + 1);
                h->sb_sampled += h->sb_nsample;
                h->sb_check_pending = true;
            }
            HIP_TRY(hipGetLastError());
        }
    } else if ((rc = launch_neighbor(h, h->prior, nullptr, h->B, NB_INIT_BOUNDS))) {  // Prior::from_bounds (prior.rs:23-32)
        return rc;
    }
    if (h->rb_on) HIP_TRY(hipMemsetAsync(h->B.uiv, 0, h->N, h->stream));  // every upper bound is an exact distance again
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->bounds_ready = true;
    ck_drain(h);
    return prune_check(h, "rp_kmeans_init_bounds");
}

int rp_kmeans_step(rp_kmeans* h, float* drift, uint64_t* sizes, double* reassigned) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_step: NULL handle");
    int rc = need_centroids(h, "rp_kmeans_step");
    if (rc || (rc = need_bounds(h, "rp_kmeans_step"))) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if ((rc = step_front(h))) return rc;
    return step_back(h, drift, sizes, reassigned);
}

int rp_kmeans_partial_bytes(rp_kmeans* h, size_t* bytes) {
    if (!h || !bytes) return rp::fail(RP_ERR_INVALID, "rp_kmeans_partial_bytes: NULL argument");
    *bytes = partial_sizes_offset(h) + (size_t)h->K * 8;
    return RP_OK;
}

// partial layout: [K*bins u32 counts][K u32 weights][pad to 8][K u64 sizes] — exact integers, all-reduce(sum)-able
int rp_kmeans_step_local(rp_kmeans* h, void* partial_dev) {
    if (!h || !partial_dev) return rp::fail(RP_ERR_INVALID, "rp_kmeans_step_local: NULL argument");
    int rc = need_centroids(h, "rp_kmeans_step_local");
    if (rc || (rc = need_bounds(h, "rp_kmeans_step_local"))) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if ((rc = step_front(h))) return rc;
    const int nxt = h->cur ^ 1;
    unsigned char* out = reinterpret_cast<unsigned char*>(partial_dev);
    const size_t cb = (size_t)h->K * h->bins * 4, wb = (size_t)h->K * 4;
    HIP_TRY(hipMemcpyAsync(out, h->cs[nxt].counts, cb, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(out + cb, h->cs[nxt].weight, wb, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(out + cb + wb, 0, partial_sizes_offset(h) - (cb + wb), h->stream));
    HIP_TRY(hipMemcpyAsync(out + partial_sizes_offset(h), h->sizes, (size_t)h->K * 8, hipMemcpyDeviceToDevice, h->stream));
    return RP_OK;
}

int rp_kmeans_step_finish(rp_kmeans* h, const void* reduced_dev, float* drift, uint64_t* sizes, double* reassigned) {
    if (!h || !reduced_dev) return rp::fail(RP_ERR_INVALID, "rp_kmeans_step_finish: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    const int nxt = h->cur ^ 1;
    const unsigned char* in = reinterpret_cast<const unsigned char*>(reduced_dev);
    const size_t cb = (size_t)h->K * h->bins * 4, wb = (size_t)h->K * 4;
    HIP_TRY(hipMemcpyAsync(h->cs[nxt].counts, in, cb, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->cs[nxt].weight, in + cb, wb, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->sizes, in + partial_sizes_offset(h), (size_t)h->K * 8, hipMemcpyDeviceToDevice, h->stream));
    return step_back(h, drift, sizes, reassigned);
}

int rp_kmeans_step_comm(rp_kmeans* h, rp_comm* c, float* drift, uint64_t* sizes, double* reassigned) {
    if (!h || !c) return rp::fail(RP_ERR_INVALID, "rp_kmeans_step_comm: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t nb = partial_sizes_offset(h) + (size_t)h->K * 8;
    if (!h->comm_partial) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, nb));
        h->allocs.push_back(p);
        h->comm_partial = reinterpret_cast<unsigned char*>(p);
    }
    int rc = rp_kmeans_step_local(h, h->comm_partial);
    if (rc) return rc;
    // exact integers, order free: the u32 block (counts, weights) and the u64 block (sizes) of the partial
    if ((rc = rp::comm_all_reduce_sum(c, h->comm_partial, (size_t)h->K * h->bins + h->K, 0, h->stream))) return rc;
    if ((rc = rp::comm_all_reduce_sum(c, h->comm_partial + partial_sizes_offset(h), h->K, 1, h->stream))) return rc;
    return rp_kmeans_step_finish(h, h->comm_partial, drift, sizes, reassigned);
}

int rp_kmeans_step_naive(rp_kmeans* h) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_step_naive: NULL handle");
    int rc = need_centroids(h, "rp_kmeans_step_naive");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    Bounds none{};
    if ((rc = launch_neighbor(h, h->tmp_j, nullptr, none))) return rc;
    const int nxt = h->cur ^ 1;
    if ((rc = launch_recompute(h, h->tmp_j, nxt))) return rc;
    if ((rc = prepare_centroids(h, nxt))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->cur = nxt;
    h->pot_is_min_d2 = false;
    ck_drain(h);
    return prune_check(h, "rp_kmeans_step_naive");
}

int rp_kmeans_assign(rp_kmeans* h, uint8_t* bucket, float* distance) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_assign: NULL handle");
    int rc = need_centroids(h, "rp_kmeans_assign");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    Bounds none{};
    if ((rc = launch_neighbor(h, h->tmp_j, h->pdist, none, NB_LOOKUP))) return rc;
    if (bucket) HIP_TRY(hipMemcpyAsync(bucket, h->tmp_j, h->N, hipMemcpyDeviceToHost, h->stream));
    if (distance) HIP_TRY(hipMemcpyAsync(distance, h->pdist, h->N * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    ck_drain(h);
    return prune_check(h, "rp_kmeans_assign");
}

int rp_kmeans_bounds(rp_kmeans* h, uint8_t* j, float* upper, float* lower) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_bounds: NULL handle");
    int rc = need_bounds(h, "rp_kmeans_bounds");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if (j) HIP_TRY(hipMemcpyAsync(j, h->B.j, h->N, hipMemcpyDeviceToHost, h->stream));
    if (upper) HIP_TRY(hipMemcpyAsync(upper, h->B.u, h->N * 4, hipMemcpyDeviceToHost, h->stream));
    if (lower) HIP_TRY(hipMemcpyAsync(lower, h->B.lower, h->N * h->K * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RP_OK;
}

int rp_kmeans_centroids(rp_kmeans* h, uint32_t* counts, uint64_t* weight) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_centroids: NULL handle");
    int rc = need_centroids(h, "rp_kmeans_centroids");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    std::vector<uint32_t> w(h->K);
    if (counts) HIP_TRY(hipMemcpyAsync(counts, h->cs[h->cur].counts, (size_t)h->K * h->bins * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(w.data(), h->cs[h->cur].weight, h->K * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (weight) for (uint32_t k = 0; k < h->K; ++k) weight[k] = w[k];
    return RP_OK;
}

int rp_kmeans_metric(rp_kmeans* h, float* tri) {
    if (!h || !tri) return rp::fail(RP_ERR_INVALID, "rp_kmeans_metric: NULL argument");
    int rc = need_centroids(h, "rp_kmeans_metric");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    ck_begin(h, CK_PAIRWISE);
    hipLaunchKernelGGL(KSEL(h, k_pairwise), dim3(h->K * h->K), dim3(64), 0, h->stream, h->cs[h->cur], h->K, h->M, h->kind, h->pairw, (const uint32_t*)nullptr, (uint32_t*)nullptr);
    ck_end(h, CK_PAIRWISE);
    HIP_TRY(hipGetLastError());
    std::vector<float> pw((size_t)h->K * h->K);
    HIP_TRY(hipMemcpyAsync(pw.data(), h->pairw, pw.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    // Layer::metric (layer.rs:85-101): (emd(x,y) + emd(y,x)) / 2, then Metric::from normalises by the max (metric.rs:127-141)
    float mx = RP_EPSILON;
    for (uint32_t i = 0; i < h->K; ++i)
        for (uint32_t j = 0; j < i; ++j) {
            float d = pw[(size_t)i * h->K + j] + pw[(size_t)j * h->K + i];
            d = d / 2.0f;
            tri[rp_tri_index(i, j)] = d;
            mx = rp_maxf(mx, d);
        }
    for (uint32_t t = 0; t < h->K * (h->K - 1) / 2; ++t) tri[t] = tri[t] / mx;
    ck_drain(h);
    return RP_OK;
}

int rp_kmeans_rms(rp_kmeans* h, float* out) {
    if (!h || !out) return rp::fail(RP_ERR_INVALID, "rp_kmeans_rms: NULL argument");
    int rc = need_bounds(h, "rp_kmeans_rms");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if (h->kind == RP_METRIC_VARIATION)
        hipLaunchKernelGGL(k_point_dist_var, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, h->P, h->cs[h->cur], h->K,
                           h->M, h->B.j, h->pdist);
    else
        hipLaunchKernelGGL(KSEL(h, k_point_dist), dim3((unsigned)h->N), dim3(64), 0, h->stream, h->P, h->cs[h->cur], h->K, h->M, h->kind, h->B.j,
                           h->pdist);
    HIP_TRY(hipGetLastError());
    std::vector<float> d(h->N);
    HIP_TRY(hipMemcpyAsync(d.data(), h->pdist, h->N * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    double acc = 0.0;  // f64 accumulation in point order (the reference's rayon f32 sum has no fixed order)
    for (uint64_t i = 0; i < h->N; ++i) acc += (double)(d[i] * d[i]);
    *out = (float)std::sqrt(acc / (double)h->N);
    return RP_OK;
}

static int read_stats(rp_kmeans* h, unsigned long long s[3]) {
    std::vector<unsigned long long> all((size_t)KM_STAT_STRIPES * STAT_STRIDE);
    HIP_TRY(hipMemcpyAsync(all.data(), h->stats, all.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    s[0] = s[1] = s[2] = 0;
    for (uint32_t q = 0; q < KM_STAT_STRIPES; ++q)
        for (uint32_t k = 0; k < 3; ++k) s[k] += all[(size_t)q * STAT_STRIDE + k];
    return RP_OK;
}

int rp_kmeans_stats(rp_kmeans* h, uint64_t* distances, uint64_t* sinkhorn_iterations) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_stats: NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    unsigned long long s[3];
    int rc = read_stats(h, s);
    if (rc) return rc;
    if (distances) *distances = s[0];
    if (sinkhorn_iterations) *sinkhorn_iterations = s[1];
    return RP_OK;
}

// diagnostics: the raw counters.  out[0] distances evaluated by a solve (or, variation metric, taken from the computed tile by Elkan's
// rule), [1] Sinkhorn iterations, [2] softmin / cost terms, [3] distances the reference evaluates at that point of its loop and this
// library REMEMBERS (Bounds::memo_*, the pairwise versions): [0] + [3] over the Elkan steps is the reference's own distance count,
// [4] variation distances computed beyond the ones the rule evaluates (whole 64-centroid tiles)
int rp_kmeans_stats_ex(rp_kmeans* h, uint64_t* out5) {
    if (!h || !out5) return rp::fail(RP_ERR_INVALID, "rp_kmeans_stats_ex: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    std::vector<unsigned long long> all((size_t)KM_STAT_STRIPES * STAT_STRIDE);
    HIP_TRY(hipMemcpyAsync(all.data(), h->stats, all.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (uint32_t k = 0; k < 5; ++k) {
        out5[k] = 0;
        for (uint32_t q = 0; q < KM_STAT_STRIPES; ++q) out5[k] += all[(size_t)q * STAT_STRIDE + k];
    }
    return RP_OK;
}

int rp_kmeans_exp_evals(rp_kmeans* h, uint64_t* evals) {
    if (!h || !evals) return rp::fail(RP_ERR_INVALID, "rp_kmeans_exp_evals: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    unsigned long long s[3];
    int rc = read_stats(h, s);
    if (rc) return rc;
    *evals = s[2];
    return RP_OK;
}

static int prune_stats_full(rp_kmeans* h, rp_prune_stats* out);
// the caller says how large ITS rp_prune_stats is: the struct grew in 0.3 (sampled_points, sample_mismatches) and may grow again;
// a host compiled against an older header gets the fields it knows, never a write past its struct
int rp_kmeans_prune_stats_sized(rp_kmeans* h, void* out, size_t out_bytes) {
    if (!h || !out || out_bytes < 8) return rp::fail(RP_ERR_INVALID, "rp_kmeans_prune_stats_sized: bad argument");
    rp_prune_stats full;
    const int rc = prune_stats_full(h, &full);
    if (rc) return rc;
    memcpy(out, &full, std::min(out_bytes, sizeof(full)));
    if (out_bytes > sizeof(full)) memset(reinterpret_cast<unsigned char*>(out) + sizeof(full), 0, out_bytes - sizeof(full));
    return RP_OK;
}
int rp_kmeans_prune_stats(rp_kmeans* h, rp_prune_stats* out) { return rp_kmeans_prune_stats_sized(h, out, sizeof(rp_prune_stats)); }
static int prune_stats_full(rp_kmeans* h, rp_prune_stats* out) {
    if (!h || !out) return rp::fail(RP_ERR_INVALID, "rp_kmeans_prune_stats: NULL argument");
    memset(out, 0, sizeof(*out));
    out->enabled = h->sb_on ? 1u : 0u;
    out->ref_pick_chunks = h->kr_chunks;
    out->ref_pick_walked = h->kr_walked;
    if (!h->sb_on) return RP_OK;
    HIP_TRY(hipSetDevice(h->device));
    std::vector<unsigned long long> all((size_t)KM_STAT_STRIPES * STAT_STRIDE);
    unsigned long long bad[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(all.data(), h->sb_stats, all.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(bad, h->sb_bad, 16, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    unsigned long long s[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t q = 0; q < KM_STAT_STRIPES; ++q)
        for (uint32_t k = 0; k < 6; ++k) s[k] += all[(size_t)q * STAT_STRIDE + k];
    out->column_iterations = s[5];
    out->survivors = s[0];
    out->points = s[1];
    out->candidates = s[1] * h->K;
    out->block_iterations = s[2];
    out->cost_passes = s[3];
    out->mfma_instructions = s[4];
    out->audited_points = h->sb_audited;
    out->audit_mismatches = bad[0];
    out->sampled_points = h->sb_sampled;
    out->sample_mismatches = bad[1];
    if (h->kb_stats) {
        HIP_TRY(hipMemcpyAsync(all.data(), h->kb_stats, all.size() * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        unsigned long long q4[4] = {0, 0, 0, 0};
        for (uint32_t q = 0; q < KM_STAT_STRIPES; ++q)
            for (uint32_t k = 0; k < 4; ++k) q4[k] += all[(size_t)q * STAT_STRIDE + k];
        out->kpp_bound_pairs = q4[0];
        out->kpp_bound_kept = q4[1];
        out->kpp_bound_iterations = q4[2];
        out->kpp_bound_cost_passes = q4[3];
    }
    return RP_OK;
}

// diagnostics (rp_mi355x_diag.h): the second k-means++ filter's lower bound of distance(centroid k, point i) for EVERY point,
// against an infinite potential (no early exit); 0 where the pair is outside the register tile or its window did not close
int rp_kmeans_kpp_bound_probe(rp_kmeans* h, uint32_t k, float* lo) { return rp_kmeans_kpp_bound_probe_at(h, k, -1.0f, lo); }
// potential < 0: every stopping window is followed to its end (lo^2 >= potential always holds for the window's bound and the dual exit is
// off); potential >= 0: the production rule against that potential for every point, dual exit included
int rp_kmeans_kpp_bound_probe_at(rp_kmeans* h, uint32_t k, float potential, float* lo) {
    if (!h || !lo || k >= h->K || !(potential == potential)) return rp::fail(RP_ERR_INVALID, "rp_kmeans_kpp_bound_probe: bad argument");
    if (!h->kb_on) return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_kpp_bound_probe: the layer has no k-means++ interval filter");
    HIP_TRY(hipSetDevice(h->device));
    // one scratch allocation (diagnostics: freed on every path): lo[N], pot[N], list in[N + 4], list out[N + 4], ctl[4], and the
    // probe's OWN striped counters (the layer's kb_stats feed rp_kmeans_prune_stats: a diagnostic must not move them)
    const size_t N = (size_t)h->N;
    const size_t stat_words = (size_t)KM_STAT_STRIPES * STAT_STRIDE * 2;  // u64 as two u32
    unsigned char* scratch = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&scratch), (4 * N + 8 + 4 + 2 + stat_words) * 4));
    float* d_lo = reinterpret_cast<float*>(scratch);
    float* d_pot = d_lo + N;
    uint32_t* d_list = reinterpret_cast<uint32_t*>(d_pot + N);
    unsigned int* d_ctl = reinterpret_cast<unsigned int*>(d_list + 2 * (N + 4));
    unsigned long long* d_stats = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(d_ctl + 4) + 7u) & ~(uintptr_t)7u);
    std::vector<uint32_t> ids(N);
    for (size_t i = 0; i < N; ++i) ids[i] = (uint32_t)i;
    const unsigned int ctl[4] = {(unsigned int)N, 0u, 0u, 0u};  // [0] count in, [1] cursor, [2] count out
    hipError_t e = hipMemcpyAsync(d_list, ids.data(), N * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ctl, ctl, 16, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_lo, 0, N * 4, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_stats, 0, stat_words * 4, h->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, h->stream, d_pot, h->N, potential < 0.0f ? -1.0f : potential);  // lo^2 >= -1 always: every window runs to its end
        hipLaunchKernelGGL((k_kpp_bound<64, 48>), dim3(2048), dim3(64), 0, h->stream, h->P, h->cs[h->cur], k, h->bins, h->M.Cm, h->sb,
                           (const float*)d_pot, (const uint32_t*)d_list, (const unsigned int*)d_ctl, d_ctl + 1, d_list + N + 4, d_ctl + 2,
                           d_stats, d_lo, (potential >= 0.0f && h->sb.lip >= 2) ? h->kb_dual : 0, (float*)nullptr, 1.0f);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(lo, d_lo, N * 4, hipMemcpyDeviceToHost, h->stream);
    const hipError_t e2 = hipStreamSynchronize(h->stream);  // `ids` and `ctl` are locals
    (void)hipFree(scratch);
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return rp::fail(RP_ERR_HIP, "rp_kmeans_kpp_bound_probe: %s", hipGetErrorString(e));
    return RP_OK;
}

// diagnostics (rp_mi355x_diag.h): the interval-decided refresh (refresh_bound.hpp)
int rp_kmeans_refresh_stats(rp_kmeans* h, uint64_t* out6) {  // (eight words: see rp_mi355x_diag.h)
    if (!h || !out6) return rp::fail(RP_ERR_INVALID, "rp_kmeans_refresh_stats: NULL argument");
    for (int k = 0; k < 8; ++k) out6[k] = 0;
    out6[5] = h->rb_on ? 1u : 0u;
    if (!h->rb_stats) return RP_OK;
    HIP_TRY(hipSetDevice(h->device));
    std::vector<unsigned long long> a((size_t)KM_STAT_STRIPES * STAT_STRIDE), b(a.size());
    HIP_TRY(hipMemcpyAsync(a.data(), h->rb_stats, a.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(b.data(), h->stats, b.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (uint32_t q = 0; q < KM_STAT_STRIPES; ++q) {
        for (uint32_t k = 0; k < 4; ++k) out6[k] += a[(size_t)q * STAT_STRIDE + k];
        out6[4] += b[(size_t)q * STAT_STRIDE + 6];
        out6[6] += a[(size_t)q * STAT_STRIDE + 4];
    }
    return RP_OK;
}
int rp_kmeans_upper_interval(rp_kmeans* h, float* ulo, uint8_t* uiv) {
    if (!h || !ulo || !uiv) return rp::fail(RP_ERR_INVALID, "rp_kmeans_upper_interval: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->rb_on) {  // every upper bound is exact
        HIP_TRY(hipMemcpyAsync(ulo, h->B.u, h->N * 4, hipMemcpyDeviceToHost, h->stream));
        memset(uiv, 0, h->N);
    } else {
        HIP_TRY(hipMemcpyAsync(ulo, h->B.ulo, h->N * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(uiv, h->B.uiv, h->N, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RP_OK;
}

// diagnostics (rp_mi355x_diag.h): the K x K table of centroid-to-centroid distances the LAST Elkan step worked with (elkan.rs:83-88,
// unnormalised, both orders), i.e. of the centroids that were current when that step began
int rp_kmeans_pairwise_last(rp_kmeans* h, float* pairw) {
    if (!h || !pairw) return rp::fail(RP_ERR_INVALID, "rp_kmeans_pairwise_last: NULL argument");
    if (!h->pairw_seen) return rp::fail(RP_ERR_INVALID, "rp_kmeans_pairwise_last: no Elkan step has run on these centroids yet");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(pairw, h->pairw, (size_t)h->K * h->K * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RP_OK;
}

// diagnostics (rp_mi355x_diag.h): WeightedIndex::new(weights).sample() for a drawn value0_1 on n host weights, by the term-by-term
// kernel of round 5 (mode 0: one wavefront, n dependent additions) or by the chunked walk the layer uses (mode 1: kpp_refpick.hpp);
// out = [picked index (n if the total is 0), bits of the total, chunks walked term by term (mode 1)]
int rp_weighted_index_probe(int device, uint64_t n, const float* weights, float v01, int mode, uint64_t* out) {
    if (!weights || !out || n == 0 || (mode != 0 && mode != 1)) return rp::fail(RP_ERR_INVALID, "rp_weighted_index_probe: bad argument");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_weighted_index_probe: no HIP device");
    HIP_TRY(hipSetDevice(device));
    float* d = nullptr;
    const size_t work = mode ? ref_pick_work_floats(n) : (size_t)((n + KR_CHUNK - 1) / KR_CHUNK);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), (n + work + 8) * 4));
    float* d_work = d + n;
    unsigned long long* d_out = reinterpret_cast<unsigned long long*>(d_work + work + ((n + work) & 1u));
    hipError_t e = hipMemcpy(d, weights, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_out, 0, 24);
    if (e == hipSuccess) {
        if (mode) ref_pick_launch(nullptr, d, nullptr, n, d_work, v01, d_out);
        else hipLaunchKernelGGL(k_kpp_ref_pick, dim3(1), dim3(64), 0, nullptr, d, (float*)nullptr, n, d_work, v01, d_out, d_out + 1);
        e = hipGetLastError();
    }
    unsigned long long got[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(got, d_out, 24, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return rp::fail(RP_ERR_HIP, "rp_weighted_index_probe: %s", hipGetErrorString(e));
    for (int i = 0; i < 3; ++i) out[i] = got[i];
    return RP_OK;
}

int rp_kmeans_bound_intervals(rp_kmeans* h, float* lo, float* hi) {
    if (!h || !lo || !hi) return rp::fail(RP_ERR_INVALID, "rp_kmeans_bound_intervals: NULL argument");
    if (!h->sb_on) return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_bound_intervals: the layer has no MFMA bound (variation metric, or switched off)");
    int rc = need_centroids(h, "rp_kmeans_bound_intervals");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    float *d_lo = nullptr, *d_hi = nullptr;
    const size_t cells = (size_t)h->N * h->K;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_lo), cells * 4));
    if (hipMalloc(reinterpret_cast<void**>(&d_hi), cells * 4) != hipSuccess) {
        (void)hipFree(d_lo);
        return rp::fail(RP_ERR_HIP, "rp_kmeans_bound_intervals: out of device memory");
    }
    // points outside the bound's support classes keep [0, inf)
    std::vector<float> zeros(cells, 0.0f), infs(cells, INFINITY);
    (void)hipMemcpyAsync(d_lo, zeros.data(), cells * 4, hipMemcpyHostToDevice, h->stream);
    (void)hipMemcpyAsync(d_hi, infs.data(), cells * 4, hipMemcpyHostToDevice, h->stream);
    rc = launch_bound(h, d_lo, d_hi);
    if (!rc) {
        (void)hipMemcpyAsync(lo, d_lo, cells * 4, hipMemcpyDeviceToHost, h->stream);
        (void)hipMemcpyAsync(hi, d_hi, cells * 4, hipMemcpyDeviceToHost, h->stream);
    }
    const hipError_t e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_lo);
    (void)hipFree(d_hi);
    ck_drain(h);
    if (rc) return rc;
    if (e != hipSuccess) return rp::fail(RP_ERR_HIP, "rp_kmeans_bound_intervals: %s", hipGetErrorString(e));
    return RP_OK;
}

int rp_kmeans_set_stream(rp_kmeans* h, void* hip_stream) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_set_stream: NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->own_stream) {
        HIP_TRY(hipStreamDestroy(h->stream));
        h->own_stream = false;
    }
    if (hip_stream) h->stream = reinterpret_cast<hipStream_t>(hip_stream);
    else {
        HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    return RP_OK;
}

int rp_kmeans_profile(rp_kmeans* h, int enable) {
    if (!h) return rp::fail(RP_ERR_INVALID, "rp_kmeans_profile: NULL handle");
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    ck_drain(h);
    h->profiling = enable != 0;
    for (auto& c : h->clk) { c.total_ms = 0.0; c.launches = 0; }
    return RP_OK;
}

int rp_kmeans_kernel_time(rp_kmeans* h, const char* name, double* total_ms, uint64_t* launches) {
    if (!h || !name) return rp::fail(RP_ERR_INVALID, "rp_kmeans_kernel_time: NULL argument");
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    ck_drain(h);
    for (int i = 0; i < CK_COUNT; ++i)
        if (std::string(name) == CLOCK_NAMES[i]) {
            if (total_ms) *total_ms = h->clk[i].total_ms;
            if (launches) *launches = h->clk[i].launches;
            return RP_OK;
        }
    return rp::fail(RP_ERR_INVALID, "rp_kmeans_kernel_time: unknown kernel '%s'", name);
}

// ---- stand-alone batched distances -------------------------------------------------------------------
namespace {
std::atomic<int> g_pair_libm{RP_LIBM_CONTRACT};
int pair_common(uint32_t bins, uint64_t pairs, const uint32_t* mu, const uint32_t* nu, const float* tri, const rp_sinkhorn_hp* hp,
                int device, float* out, uint32_t* iterations, int divergence) {
    if (!mu || !nu || !out || pairs == 0 || bins == 0 || bins > MAXB || (!tri && bins > 1))
        return rp::fail(RP_ERR_INVALID, "rp_sinkhorn_*: bad argument");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_sinkhorn_*: no HIP device visible; no CPU fallback");
    rp_sinkhorn_hp hh;
    if (hp) hh = *hp; else rp_sinkhorn_hp_default(&hh);
    HIP_TRY(hipSetDevice(device));
    std::vector<float> C((size_t)bins * bins, 0.0f), R((size_t)bins * bins, 0.0f);
    for (uint32_t x = 0; x < bins; ++x)
        for (uint32_t y = 0; y < bins; ++y) {
            const float c = x == y ? 0.0f : tri[rp_tri_index(x, y)];
            C[(size_t)x * bins + y] = c;
            R[(size_t)x * bins + y] = c / hh.temperature;
        }
    float *dC = nullptr, *dR = nullptr, *dout = nullptr;
    uint32_t *dmu = nullptr, *dnu = nullptr, *dit = nullptr;
    unsigned long long *dstats = nullptr, *dscr = nullptr;
    const size_t hb = (size_t)pairs * bins * 4;
    HIP_TRY(hipMalloc(&dC, C.size() * 4));
    HIP_TRY(hipMalloc(&dR, R.size() * 4));
    HIP_TRY(hipMalloc(&dout, pairs * 4));
    HIP_TRY(hipMalloc(&dmu, hb));
    HIP_TRY(hipMalloc(&dnu, hb));
    HIP_TRY(hipMalloc(&dstats, 32));
    HIP_TRY(hipMemset(dstats, 0, 32));
    HIP_TRY(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dmu, mu, hb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dnu, nu, hb, hipMemcpyHostToDevice));
    Metric M{dC, dR, bins, hh.iterations, hh.tolerance, dstats, 1u};
    const bool glibc = g_pair_libm.load() == RP_LIBM_GLIBC;
    hipLaunchKernelGGL(KSEL_IF(glibc, k_pair_sinkhorn), dim3((unsigned)pairs), dim3(64), 0, 0, dmu, dnu, M, divergence, dout, (uint32_t*)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, pairs * 4, hipMemcpyDeviceToHost));
    if (iterations) {
        HIP_TRY(hipMalloc(&dit, pairs * 4));
        HIP_TRY(hipMalloc(&dscr, pairs * 32));
        HIP_TRY(hipMemset(dscr, 0, pairs * 32));
        hipLaunchKernelGGL(KSEL_IF(glibc, k_pair_iters), dim3((unsigned)pairs), dim3(64), 0, 0, dmu, dnu, M, dscr, dit);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(iterations, dit, pairs * 4, hipMemcpyDeviceToHost));
        (void)hipFree(dit);
        (void)hipFree(dscr);
    }
    (void)hipFree(dC); (void)hipFree(dR); (void)hipFree(dout); (void)hipFree(dmu); (void)hipFree(dnu); (void)hipFree(dstats);
    return RP_OK;
}
}  // namespace

int rp_sinkhorn_set_libm(rp_libm_kind kind) {
    if (kind != RP_LIBM_CONTRACT && kind != RP_LIBM_GLIBC) return rp::fail(RP_ERR_INVALID, "rp_sinkhorn_set_libm: unknown kind %d", (int)kind);
    g_pair_libm.store(kind);
    return RP_OK;
}
int rp_sinkhorn_divergence(uint32_t bins, uint64_t pairs, const uint32_t* mu, const uint32_t* nu, const float* tri_metric,
                           const rp_sinkhorn_hp* hp, int device, float* out) {
    return pair_common(bins, pairs, mu, nu, tri_metric, hp, device, out, nullptr, 1);
}
int rp_sinkhorn_cost(uint32_t bins, uint64_t pairs, const uint32_t* mu, const uint32_t* nu, const float* tri_metric,
                     const rp_sinkhorn_hp* hp, int device, float* out, uint32_t* iterations) {
    return pair_common(bins, pairs, mu, nu, tri_metric, hp, device, out, iterations, 0);
}
int rp_sinkhorn_flow(uint32_t bins, const uint32_t* mu, const uint32_t* nu, const float* tri, const rp_sinkhorn_hp* hp, int device,
                     float* flow, float* coupling) {
    if (!mu || !nu || !flow || bins == 0 || bins > MAXB || (!tri && bins > 1)) return rp::fail(RP_ERR_INVALID, "rp_sinkhorn_flow: bad argument");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_sinkhorn_flow: no HIP device visible; no CPU fallback");
    rp_sinkhorn_hp hh;
    if (hp) hh = *hp; else rp_sinkhorn_hp_default(&hh);
    HIP_TRY(hipSetDevice(device));
    const size_t cells = (size_t)bins * bins;
    std::vector<float> C(cells, 0.0f), R(cells, 0.0f);
    for (uint32_t x = 0; x < bins; ++x)
        for (uint32_t y = 0; y < bins; ++y) {
            const float c = x == y ? 0.0f : tri[rp_tri_index(x, y)];
            C[(size_t)x * bins + y] = c;
            R[(size_t)x * bins + y] = c / hh.temperature;
        }
    float* buf = nullptr;  // C | R | flow | coupling
    uint32_t* hist = nullptr;
    unsigned long long* dstats = nullptr;
    HIP_TRY(hipMalloc(&buf, cells * 4 * 4));
    HIP_TRY(hipMalloc(&hist, (size_t)bins * 8));
    HIP_TRY(hipMalloc(&dstats, 32));
    HIP_TRY(hipMemset(dstats, 0, 32));
    HIP_TRY(hipMemset(buf + 2 * cells, 0, cells * 8));
    HIP_TRY(hipDeviceSynchronize());  // null-stream memsets before launches on other streams
    HIP_TRY(hipMemcpy(buf, C.data(), cells * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(buf + cells, R.data(), cells * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(hist, mu, (size_t)bins * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(hist + bins, nu, (size_t)bins * 4, hipMemcpyHostToDevice));
    Metric M{buf, buf + cells, bins, hh.iterations, hh.tolerance, dstats, 1u};
    hipLaunchKernelGGL(KSEL_IF(g_pair_libm.load() == RP_LIBM_GLIBC, k_pair_flow), dim3(1), dim3(64), 0, 0, hist, hist + bins, M, buf + 2 * cells, buf + 3 * cells);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(flow, buf + 2 * cells, cells * 4, hipMemcpyDeviceToHost));
    if (coupling) HIP_TRY(hipMemcpy(coupling, buf + 3 * cells, cells * 4, hipMemcpyDeviceToHost));
    (void)hipFree(buf); (void)hipFree(hist); (void)hipFree(dstats);
    return RP_OK;
}
int rp_equity_variation(uint32_t bins, uint64_t pairs, const uint32_t* x, const uint32_t* y, int device, float* out) {
    if (!x || !y || !out || pairs == 0 || bins == 0) return rp::fail(RP_ERR_INVALID, "rp_equity_variation: bad argument");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_equity_variation: no HIP device visible; no CPU fallback");
    HIP_TRY(hipSetDevice(device));
    uint32_t *dx = nullptr, *dy = nullptr;
    float* dout = nullptr;
    const size_t hb = (size_t)pairs * bins * 4;
    HIP_TRY(hipMalloc(&dx, hb));
    HIP_TRY(hipMalloc(&dy, hb));
    HIP_TRY(hipMalloc(&dout, pairs * 4));
    HIP_TRY(hipMemcpy(dx, x, hb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dy, y, hb, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pair_variation, dim3((unsigned)((pairs + 63) / 64)), dim3(64), 0, 0, dx, dy, bins, pairs, dout);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, pairs * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout);
    return RP_OK;
}

}  // extern "C"

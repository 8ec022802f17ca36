/* rp_oracle_lloyd.c — CPU restatement of robopoker's `lloyd`/`elkan` hot path.      TEST INFRASTRUCTURE.
 *
 * ORACLE for the MI355X hand-abstraction path (Elkan k-means over Sinkhorn EMD / equity variation).
 * Plain C, single threaded, deterministic.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product never does.
 *
 * What it follows (paths under /root/reference/crates, cited per function):
 *   lloyd/src/sinkhorn.rs   Sinkhorn::{sinkhorn, lhs, rhs, softmin, delta, coupling, cost, divergence, self_cost}
 *   lloyd/src/{phi.rs, potential.rs, bins.rs, metric.rs, pair.rs, equity.rs, layer.rs, kmeans.rs}
 *   elkan/src/{elkan.rs, bounds.rs, drift.rs, prior.rs}
 *
 * PARITY STATUS.  The reference cannot be built here (Rust).  Its k-means is deterministic but every
 * distance goes through platform libm exp/ln (sinkhorn.rs:115-128), so it holds no golden vectors —
 * only property tests.  The oracle is pinned against all of them (tests/test_oracle_lloyd.py):
 * self-divergence < 1e-4 and symmetry < 1e-3 on the closed-form fixture (sinkhorn.rs:240-293),
 * OT(mu,mu) <= 0.01 / positivity / triangle (emd.rs:105-131), variation symmetric/zero/positive
 * (emd.rs:72-97), Elkan == naive (tests.rs:148-161), Pair bijection (pair.rs:171-189).
 * exp/ln are by default include/rp_math.h's rp_expf/rp_logf (the clustering kernels' f32 contract, <= 1 ulp from glibc's) and, under
 * ora_lloyd_set_libm, the platform's own functions (1) or glibc's restated (2: include/rp_libm_glibc.h, equal to glibc 2.35's on all
 * 2^32 inputs — the boundary is pinned for glibc hosts, tests/test_libm_glibc.py); the k-means++ draw is rp_math.h's order-independent fixed-point scheme by default and, under
 * ora_kmeans_set_rng(RP_RNG_REFERENCE), layer.rs:155-178's own SmallRng + WeightedIndex<f32> (include/rp_refrng.h).
 *
 * f32 operation order follows the reference: supports ascend by bin index (phi.rs:51-57), every softmin
 * term is clamped at MIN_POSITIVE before the left-fold sum (sinkhorn.rs:119-128), the cost is summed
 * x-major (sinkhorn.rs:206-217).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/rp_math.h"
#include "../include/rp_mi355x.h"
#include "../include/rp_refrng.h"
#include "../include/rp_libm_glibc.h"

/* THE LIBM BOUNDARY, made measurable.  The reference calls the platform's libm (f32::exp / f32::ln; sinkhorn.rs:115,120-127,136,
 * phi.rs:36); the clustering kernels' contract is rp_expf / rp_logf (include/rp_math.h: <= 1 ulp from glibc's, host == device).  ora_lloyd_set_libm(1)
 * makes THIS oracle call the platform's expf / logf instead — the reference's arithmetic on this machine — so that tests can state what
 * the boundary is worth: how many Sinkhorn costs change and by how much, whether any k-means++ pick or bucket moves
 * (tests/test_oracle_lloyd.py::test_platform_libm_*).  Mode 2 = glibc's functions restated: what the device's lm_glibc pass computes
 * (rp_kmeans_set_libm / rp_sinkhorn_set_libm with RP_LIBM_GLIBC), checked against this oracle in that mode. */
#define ORA_API __attribute__((visibility("default")))
static int g_libm = 0;
ORA_API void ora_lloyd_set_libm(int on) { g_libm = on; }
/* 0: the contract (rp_math.h); 1: this machine's libm; 2: glibc's algorithms restated (include/rp_libm_glibc.h: = 1 on glibc / x86-64) */
static inline float ora_expf(float x) { return g_libm == 1 ? expf(x) : (g_libm == 2 ? rp_glibc_expf(x) : rp_expf(x)); }
static inline float ora_logf(float x) { return g_libm == 1 ? logf(x) : (g_libm == 2 ? rp_glibc_logf(x) : rp_logf(x)); }

#define ORA_MAXBINS 256

typedef struct ora_hist {
    uint32_t counts[ORA_MAXBINS];
    uint64_t weight;
} ora_hist;

/* Bins::density (bins.rs:58-60): integer count over integer weight, both cast to f32 */
static float h_density(const ora_hist* h, uint32_t i) { return (float)h->counts[i] / (float)h->weight; }

/* Metric::raw_distance (metric.rs:41-55) over Pair::merge (pair.rs:58-65) */
static float raw_distance(const float* tri, uint32_t x, uint32_t y) {
    if (x == y) return 0.0f;
    return tri[rp_tri_index(x, y)];
}

/* SinkhornHyperParams::DEFAULT (lloyd/src/hyperparams/sinkhorn.rs:17-23), oracle-local copy */
static void ora_hp_default(rp_sinkhorn_hp* out) {
    out->temperature = 0.025f;
    out->iterations = 128;
    out->tolerance = 0.0005f;
}

static uint64_t g_sinkhorn_iters = 0;
static uint64_t g_distances = 0;
/* CPU-baseline threading (bench.py's all-core figure): the reference's point-parallel structure — rayon par_iter over
 * points in init_bounds / step_elkan / lookup and over centroid pairs in pairwises (elkan.rs:39-47,80-93,153-168) —
 * as OpenMP loops.  Every distance is a pure function of its two histograms, so results do not depend on the count. */
static int g_threads = 1;
ORA_API void ora_lloyd_set_threads(int n) { g_threads = n > 0 ? n : 1; }
ORA_API int ora_lloyd_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Sinkhorn::from(..).minimize().cost() (sinkhorn.rs:77-92,194-230) */
static float sinkhorn_cost_traced(uint32_t bins, const ora_hist* mu, const ora_hist* nu, const float* tri,
                                  const rp_sinkhorn_hp* hp, uint32_t* iters_out, float* trace_err, float* trace_cost);
static float sinkhorn_cost(uint32_t bins, const ora_hist* mu, const ora_hist* nu, const float* tri,
                           const rp_sinkhorn_hp* hp, uint32_t* iters_out) {
    return sinkhorn_cost_traced(bins, mu, nu, tri, hp, iters_out, NULL, NULL);
}
/* x-major sum of coupling * distance (sinkhorn.rs:206-217) for the potentials (lhs, rhs) */
static float coupling_cost(const float* tri, const uint32_t* sx, uint32_t m, const uint32_t* sy, uint32_t n,
                           const float* lhs, const float* rhs, float T) {
    float cost = -0.0f; /* Iterator::sum::<f32>() folds from -0.0 (libcore iter/traits/accum.rs since Rust 1.83; the workspace asks for 1.90) */
    for (uint32_t i = 0; i < m; ++i)
        for (uint32_t j = 0; j < n; ++j) {
            float c = raw_distance(tri, sx[i], sy[j]);
            cost += ora_expf(lhs[i] + rhs[j] - c / T) * c;
        }
    return cost;
}
/* trace_err / trace_cost (tests only, each hp->iterations long): the stopping statistic lhs_err + rhs_err of every
 * iteration and the cost the solve would return had it stopped there; with a trace the loop does not stop early, the
 * returned cost and *iters_out are still those of the reference's stopping rule. */
static float sinkhorn_cost_traced(uint32_t bins, const ora_hist* mu, const ora_hist* nu, const float* tri,
                                  const rp_sinkhorn_hp* hp, uint32_t* iters_out, float* trace_err, float* trace_cost) {
    uint32_t sx[ORA_MAXBINS], sy[ORA_MAXBINS];
    uint32_t m = 0, n = 0;
    for (uint32_t i = 0; i < bins; ++i) { /* Bins::support ascending (bins.rs:84-88) */
        if (mu->counts[i] > 0) sx[m++] = i;
        if (nu->counts[i] > 0) sy[n++] = i;
    }
    if (iters_out) *iters_out = 0;
    if (m == 0 || n == 0) return -0.0f; /* empty support: the cost sum is empty, and an empty f32 sum is -0.0 */
    float lhs[ORA_MAXBINS], rhs[ORA_MAXBINS], nxt[ORA_MAXBINS];
    /* Potential::uniform (phi.rs:34-39): ln(1 / n()) on the support */
    float lu = ora_logf(1.0f / (float)m), ru = ora_logf(1.0f / (float)n);
    for (uint32_t i = 0; i < m; ++i) lhs[i] = lu;
    for (uint32_t j = 0; j < n; ++j) rhs[j] = ru;
    float T = hp->temperature;
    uint32_t t, stop_t = 0, done_iters = 0;
    int stopped = 0;
    float stop_cost = 0.0f;
    for (t = 0; t < hp->iterations; ++t) {
        /* lhs (sinkhorn.rs:94-102) via softmin (sinkhorn.rs:119-128) */
        float lhs_err = 0.0f;
        for (uint32_t i = 0; i < m; ++i) {
            float s = 0.0f;
            for (uint32_t j = 0; j < n; ++j) {
                float e = ora_expf(rhs[j] - raw_distance(tri, sx[i], sy[j]) / T);
                s += rp_maxf(e, RP_EPSILON);
            }
            nxt[i] = ora_logf(h_density(mu, sx[i])) - ora_logf(s);
        }
        for (uint32_t i = 0; i < m; ++i) lhs_err += rp_absf(ora_expf(nxt[i]) - ora_expf(lhs[i])); /* delta :134-139 */
        for (uint32_t i = 0; i < m; ++i) lhs[i] = nxt[i];
        /* rhs sees the fresh lhs (Gauss-Seidel, sinkhorn.rs:80-87) */
        float rhs_err = 0.0f;
        for (uint32_t j = 0; j < n; ++j) {
            float s = 0.0f;
            for (uint32_t i = 0; i < m; ++i) {
                float e = ora_expf(lhs[i] - raw_distance(tri, sy[j], sx[i]) / T);
                s += rp_maxf(e, RP_EPSILON);
            }
            nxt[j] = ora_logf(h_density(nu, sy[j])) - ora_logf(s);
        }
        for (uint32_t j = 0; j < n; ++j) rhs_err += rp_absf(ora_expf(nxt[j]) - ora_expf(rhs[j]));
        for (uint32_t j = 0; j < n; ++j) rhs[j] = nxt[j];
        done_iters += 1; /* counted per solve and added once at the end: no shared counter inside the threaded hot loop */
        if (trace_err) {
            trace_err[t] = lhs_err + rhs_err;
            trace_cost[t] = coupling_cost(tri, sx, m, sy, n, lhs, rhs, T);
            if (!stopped && lhs_err + rhs_err < hp->tolerance) {
                stopped = 1;
                stop_t = t + 1;
                stop_cost = trace_cost[t];
            }
            continue;
        }
        if (lhs_err + rhs_err < hp->tolerance) { t += 1; break; }
    }
#pragma omp atomic
    g_sinkhorn_iters += done_iters;
    if (trace_err && stopped) {
        if (iters_out) *iters_out = stop_t;
        return stop_cost;
    }
    if (iters_out) *iters_out = t;
    /* cost (sinkhorn.rs:206-217): x-major sum of coupling * distance */
    return coupling_cost(tri, sx, m, sy, n, lhs, rhs, T);
}

/* Sinkhorn::divergence (sinkhorn.rs:166-171) with the self terms passed in (self_cost :175-191 memoises
 * them by histogram content; a pure function of the histogram, so precomputing is equivalent) */
static float divergence_with(uint32_t bins, const ora_hist* mu, float xx, const ora_hist* nu, float yy,
                             const float* tri, const rp_sinkhorn_hp* hp) {
    float xy = sinkhorn_cost(bins, mu, nu, tri, hp, NULL);
#pragma omp atomic
    g_distances += 1;
    return rp_maxf(xy - 0.5f * xx - 0.5f * yy, 0.0f);
}

/* Equity::variation (equity.rs:41-53): running CDFs over the `bins` equity buckets */
static float variation(uint32_t bins, const ora_hist* x, const ora_hist* y) {
    float cdf_x = 0.0f, cdf_y = 0.0f, sum = 0.0f;
    for (uint32_t i = 0; i < bins; ++i) {
        cdf_x += h_density(x, i);
        cdf_y += h_density(y, i);
        sum += rp_absf(cdf_x - cdf_y);
    }
#pragma omp atomic
    g_distances += 1;
    return sum / (float)bins;
}

static void hist_from_u32(ora_hist* h, uint32_t bins, const uint32_t* c) {
    memset(h, 0, sizeof(*h));
    for (uint32_t i = 0; i < bins; ++i) {
        h->counts[i] = c[i];
        h->weight += c[i];
    }
}

ORA_API float ora_sinkhorn_cost(uint32_t bins, const uint32_t* mu, const uint32_t* nu, const float* tri,
                                const rp_sinkhorn_hp* hp, uint32_t* iters) {
    ora_hist a, b;
    hist_from_u32(&a, bins, mu);
    hist_from_u32(&b, bins, nu);
    return sinkhorn_cost(bins, &a, &b, tri, hp, iters);
}
/* tests only: per-iteration stopping statistic and would-be cost (see sinkhorn_cost_traced) */
ORA_API float ora_sinkhorn_trace(uint32_t bins, const uint32_t* mu, const uint32_t* nu, const float* tri,
                                 const rp_sinkhorn_hp* hp, uint32_t* iters, float* trace_err, float* trace_cost) {
    ora_hist a, b;
    hist_from_u32(&a, bins, mu);
    hist_from_u32(&b, bins, nu);
    return sinkhorn_cost_traced(bins, &a, &b, tri, hp, iters, trace_err, trace_cost);
}
/* Coupling::flow for every (x, y) of one minimised pair (sinkhorn.rs:114-116,202-204); dense bins*bins outputs, zero
 * outside the supports; coupling may be NULL */
ORA_API void ora_sinkhorn_flow(uint32_t bins, const uint32_t* mu_c, const uint32_t* nu_c, const float* tri, const rp_sinkhorn_hp* hp,
                               float* flow, float* coupling) {
    ora_hist mu, nu;
    hist_from_u32(&mu, bins, mu_c);
    hist_from_u32(&nu, bins, nu_c);
    memset(flow, 0, (size_t)bins * bins * 4);
    if (coupling) memset(coupling, 0, (size_t)bins * bins * 4);
    uint32_t sx[ORA_MAXBINS], sy[ORA_MAXBINS], m = 0, n = 0;
    for (uint32_t i = 0; i < bins; ++i) {
        if (mu.counts[i] > 0) sx[m++] = i;
        if (nu.counts[i] > 0) sy[n++] = i;
    }
    if (m == 0 || n == 0) return;
    /* the same minimisation as sinkhorn_cost_traced, potentials kept */
    float lhs[ORA_MAXBINS], rhs[ORA_MAXBINS], nxt[ORA_MAXBINS];
    float lu = ora_logf(1.0f / (float)m), ru = ora_logf(1.0f / (float)n), T = hp->temperature;
    for (uint32_t i = 0; i < m; ++i) lhs[i] = lu;
    for (uint32_t j = 0; j < n; ++j) rhs[j] = ru;
    for (uint32_t t = 0; t < hp->iterations; ++t) {
        float le = 0.0f, re = 0.0f;
        for (uint32_t i = 0; i < m; ++i) {
            float s = 0.0f;
            for (uint32_t j = 0; j < n; ++j) s += rp_maxf(ora_expf(rhs[j] - raw_distance(tri, sx[i], sy[j]) / T), RP_EPSILON);
            nxt[i] = ora_logf(h_density(&mu, sx[i])) - ora_logf(s);
        }
        for (uint32_t i = 0; i < m; ++i) le += rp_absf(ora_expf(nxt[i]) - ora_expf(lhs[i]));
        for (uint32_t i = 0; i < m; ++i) lhs[i] = nxt[i];
        for (uint32_t j = 0; j < n; ++j) {
            float s = 0.0f;
            for (uint32_t i = 0; i < m; ++i) s += rp_maxf(ora_expf(lhs[i] - raw_distance(tri, sy[j], sx[i]) / T), RP_EPSILON);
            nxt[j] = ora_logf(h_density(&nu, sy[j])) - ora_logf(s);
        }
        for (uint32_t j = 0; j < n; ++j) re += rp_absf(ora_expf(nxt[j]) - ora_expf(rhs[j]));
        for (uint32_t j = 0; j < n; ++j) rhs[j] = nxt[j];
        if (le + re < hp->tolerance) break;
    }
    for (uint32_t i = 0; i < m; ++i)
        for (uint32_t j = 0; j < n; ++j) {
            float c = raw_distance(tri, sx[i], sy[j]);
            float pi = ora_expf(lhs[i] + rhs[j] - c / T);
            if (coupling) coupling[(size_t)sx[i] * bins + sy[j]] = pi;
            flow[(size_t)sx[i] * bins + sy[j]] = pi * c;
        }
}
ORA_API float ora_sinkhorn_divergence(uint32_t bins, const uint32_t* mu, const uint32_t* nu, const float* tri,
                                      const rp_sinkhorn_hp* hp) {
    ora_hist a, b;
    hist_from_u32(&a, bins, mu);
    hist_from_u32(&b, bins, nu);
    float xx = sinkhorn_cost(bins, &a, &a, tri, hp, NULL);
    float yy = sinkhorn_cost(bins, &b, &b, tri, hp, NULL);
    return divergence_with(bins, &a, xx, &b, yy, tri, hp);
}
ORA_API float ora_equity_variation(uint32_t bins, const uint32_t* x, const uint32_t* y) {
    ora_hist a, b;
    hist_from_u32(&a, bins, x);
    hist_from_u32(&b, bins, y);
    return variation(bins, &a, &b);
}

/* ------------------------------------------------------------------ Elkan k-means (Layer<K,N>) */
typedef struct ora_bound { /* Bounds<K> (bounds.rs:19-25) */
    uint32_t j;
    float* lower;
    float error;
    int stale;
} ora_bound;

typedef struct ora_kmeans {
    uint32_t K, bins;
    uint64_t N;
    int kind;
    float* tri;
    rp_sinkhorn_hp hp;
    uint64_t seed;
    int rng;    /* rp_rng_kind: RP_RNG_REFERENCE = layer.rs:155-166's own generator and WeightedIndex<f32> */
    int street; /* Street discriminant hashed into the generator's seed (layer.rs:156-158; deuce/src/street.rs:21-27) */
    ora_hist* points;
    float* self_p; /* OT(p,p) per point */
    ora_hist* cent;
    float* self_c;
    ora_bound* bounds;
    float* lower_store;
    uint8_t* prior; /* Prior<N> (prior.rs:20) */
    float* pot;     /* k-means++ potentials */
    int has_prior;
} ora_kmeans;

static float point_self(const ora_kmeans* h, const ora_hist* p) {
    if (h->kind != RP_METRIC_SINKHORN) return 0.0f;
    return sinkhorn_cost(h->bins, p, p, h->tri, &h->hp, NULL);
}
/* Layer::distance -> Metric::emd (layer.rs:136-138, metric.rs:109-115) */
static float dist(const ora_kmeans* h, const ora_hist* a, float sa, const ora_hist* b, float sb) {
    if (h->kind == RP_METRIC_SINKHORN) return divergence_with(h->bins, a, sa, b, sb, h->tri, &h->hp);
    return variation(h->bins, a, b);
}
static void refresh_self_c(ora_kmeans* h) {
    for (uint32_t j = 0; j < h->K; ++j) h->self_c[j] = point_self(h, &h->cent[j]);
}

ORA_API ora_kmeans* ora_kmeans_create(uint32_t K, uint64_t N, uint32_t bins, const uint8_t* counts, int kind,
                                      const float* tri, const rp_sinkhorn_hp* hp, uint64_t seed) {
    if (bins > ORA_MAXBINS || K == 0 || K > 256 || N == 0) return NULL;
    ora_kmeans* h = (ora_kmeans*)calloc(1, sizeof(*h));
    h->K = K; h->N = N; h->bins = bins; h->kind = kind; h->seed = seed;
    if (hp) h->hp = *hp; else ora_hp_default(&h->hp);
    if (kind == RP_METRIC_SINKHORN) {
        size_t nt = (size_t)bins * (bins - 1) / 2;
        h->tri = (float*)malloc(4 * (nt ? nt : 1));
        memcpy(h->tri, tri, 4 * nt);
    }
    h->points = (ora_hist*)calloc(N, sizeof(ora_hist));
    for (uint64_t i = 0; i < N; ++i)
        for (uint32_t b = 0; b < bins; ++b) {
            h->points[i].counts[b] = counts[i * bins + b];
            h->points[i].weight += counts[i * bins + b];
        }
    h->self_p = (float*)malloc(4 * N);
    for (uint64_t i = 0; i < N; ++i) h->self_p[i] = point_self(h, &h->points[i]);
    h->cent = (ora_hist*)calloc(K, sizeof(ora_hist));
    h->self_c = (float*)calloc(K, 4);
    h->bounds = (ora_bound*)calloc(N, sizeof(ora_bound));
    h->lower_store = (float*)calloc((size_t)N * K, 4);
    for (uint64_t i = 0; i < N; ++i) h->bounds[i].lower = h->lower_store + i * K;
    h->prior = (uint8_t*)calloc(N, 1);
    return h;
}
ORA_API void ora_kmeans_destroy(ora_kmeans* h) {
    if (!h) return;
    free(h->tri); free(h->points); free(h->self_p); free(h->cent); free(h->self_c);
    free(h->bounds); free(h->lower_store); free(h->prior); free(h->pot); free(h);
}

ORA_API void ora_kmeans_set_centroids(ora_kmeans* h, const uint64_t* idx) {
    for (uint32_t j = 0; j < h->K; ++j) h->cent[j] = h->points[idx[j]];
    refresh_self_c(h);
}

/* k-means++ primitives (include/rp_mi355x.h rp_kmeans_kpp_*), Layer::init_centroids (layer.rs:140-181) is
 * their composition.  Draw = rp_math.h fixed-point scheme. */
ORA_API void ora_kmeans_kpp_begin(ora_kmeans* h) {
    free(h->pot);
    h->pot = (float*)malloc(4 * h->N);
    for (uint64_t i = 0; i < h->N; ++i) h->pot[i] = 1.0f;
    memset(h->cent, 0, sizeof(ora_hist) * h->K);
    memset(h->self_c, 0, 4 * h->K);
}
ORA_API uint64_t ora_kmeans_kpp_total(const ora_kmeans* h) {
    uint64_t total = 0;
    for (uint64_t i = 0; i < h->N; ++i) total += rp_kpp_quant(h->pot[i]);
    return total;
}
ORA_API uint64_t ora_kmeans_kpp_pick(ora_kmeans* h, uint64_t r) {
    uint64_t acc = 0, pick = h->N - 1;
    for (uint64_t i = 0; i < h->N; ++i) {
        acc += rp_kpp_quant(h->pot[i]);
        if (acc > r) { pick = i; break; }
    }
    h->pot[pick] = 0.0f;
    return pick;
}
ORA_API void ora_kmeans_get_point(const ora_kmeans* h, uint64_t idx, uint32_t* counts) {
    memcpy(counts, h->points[idx].counts, 4 * h->bins);
}
ORA_API void ora_kmeans_set_centroid(ora_kmeans* h, uint32_t k, const uint32_t* counts) {
    hist_from_u32(&h->cent[k], h->bins, counts);
    h->self_c[k] = point_self(h, &h->cent[k]);
}
ORA_API void ora_kmeans_kpp_update(ora_kmeans* h, uint32_t k) {
    for (uint64_t i = 0; i < h->N; ++i) { /* distance(&x, h) with the new centroid first (layer.rs:172) */
        float d = dist(h, &h->cent[k], h->self_c[k], &h->points[i], h->self_p[i]);
        h->pot[i] = rp_minf(d * d, h->pot[i]);
    }
}
ORA_API void ora_kmeans_set_rng(ora_kmeans* h, int kind, int street) {
    h->rng = kind;
    h->street = street;
}
/* WeightedIndex::new(potentials.iter()).sample(rng) (layer.rs:164-166; rand 0.9.2 weighted_index.rs): f32 running sums in index
 * order, x = Uniform::new(0, total).sample(rng), partition_point(cum <= x) over the sums of all but the last weight.
 * Returns N when the weights are invalid (total == 0: the reference panics on "valid weights array"). */
static uint64_t weighted_index_f32(const float* w, uint64_t n, rp_smallrng* rng, float* cum) {
    float total = w[0];
    for (uint64_t i = 0; i + 1 < n; ++i) {
        cum[i] = total;
        total += w[i + 1];
    }
    if (!(total > 0.0f)) return n;
    const float x = rp_rand_uniform_f32(rng, rp_uniform_f32_scale(total));
    uint64_t lo = 0, hi = n - 1; /* partition_point over cum[0 .. n-1) */
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (cum[mid] <= x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
ORA_API void ora_kmeans_init_centroids(ora_kmeans* h, uint64_t* chosen) {
    uint32_t* hist = (uint32_t*)malloc(4 * ORA_MAXBINS);
    ora_kmeans_kpp_begin(h);
    rp_smallrng rng;
    float* cum = NULL;
    if (h->rng == RP_RNG_REFERENCE) { /* DefaultHasher::default(); self.street().hash(hasher); SmallRng::seed_from_u64(finish()) */
        rp_sip s;
        rp_defaulthasher_new(&s);
        rp_defaulthasher_write_u64(&s, (uint64_t)(int64_t)h->street); /* derive(Hash) on a fieldless enum: the discriminant as isize */
        rp_smallrng_seed(&rng, rp_defaulthasher_finish(&s));
        cum = (float*)malloc(4 * h->N);
    }
    for (uint32_t k = 0; k < h->K; ++k) {
        uint64_t pick;
        if (h->rng == RP_RNG_REFERENCE) {
            pick = weighted_index_f32(h->pot, h->N, &rng, cum);
            if (pick >= h->N) pick = h->N - 1; /* all potentials zero: the reference panics; a defined answer here */
            h->pot[pick] = 0.0f;
        } else {
        uint64_t total = ora_kmeans_kpp_total(h);
        uint64_t hsh = rp_stream(h->seed, k);
        pick = total ? ora_kmeans_kpp_pick(h, rp_mulhi64(hsh, total)) : rp_mulhi64(hsh, h->N);
        }
        if (chosen) chosen[k] = pick;
        ora_kmeans_get_point(h, pick, hist);
        ora_kmeans_set_centroid(h, k, hist);
        ora_kmeans_kpp_update(h, k);
    }
    free(hist);
    free(cum);
}

/* Elkan::neighbor (elkan.rs:68-77): distance(centroid, point), first minimum wins */
static void neighbor(const ora_kmeans* h, uint64_t i, uint32_t* jo, float* dout) {
    uint32_t bj = 0;
    float bd = 0.0f;
    for (uint32_t j = 0; j < h->K; ++j) {
        float d = dist(h, &h->cent[j], h->self_c[j], &h->points[i], h->self_p[i]);
        if (j == 0 || d < bd) { bj = j; bd = d; }
    }
    *jo = bj;
    *dout = bd;
}

/* Elkan::init_bounds (elkan.rs:39-47) + Bounds::from (bounds.rs:111-120) */
ORA_API void ora_kmeans_init_bounds(ora_kmeans* h) {
#pragma omp parallel for schedule(dynamic, 8) num_threads(g_threads) if (g_threads > 1)
    for (uint64_t i = 0; i < h->N; ++i) {
        ora_bound* b = &h->bounds[i];
        neighbor(h, i, &b->j, &b->error);
        for (uint32_t k = 0; k < h->K; ++k) b->lower[k] = 0.0f;
        b->stale = 0;
    }
    for (uint64_t i = 0; i < h->N; ++i) h->prior[i] = (uint8_t)h->bounds[i].j; /* Prior::from_bounds (prior.rs:23-32) */
    h->has_prior = 1;
}

static void absorb(ora_hist* acc, const ora_hist* p, uint32_t bins) { /* Bins::merge (bins.rs:75-82) */
    acc->weight += p->weight;
    for (uint32_t b = 0; b < bins; ++b) acc->counts[b] += p->counts[b];
}

/* Kmeans::next (kmeans.rs:82-110) = Elkan::step_elkan (elkan.rs:153-168) + install + Prior::tally, split at
 * the centroid sums so a point-sharded job can all-reduce them (rp_kmeans_step_local / rp_kmeans_step_finish).
 * partial layout: [K*bins u32 counts][K u32 weights][pad to 8][K u64 sizes] */
ORA_API size_t ora_kmeans_partial_bytes(const ora_kmeans* h) {
    size_t a = ((size_t)h->K * h->bins + h->K) * 4;
    a = (a + 7) & ~(size_t)7;
    return a + (size_t)h->K * 8;
}
ORA_API void ora_kmeans_step_local(ora_kmeans* h, void* partial) {
    uint32_t K = h->K;
    float* pw = (float*)malloc(4 * (size_t)K * K);
    float* mid = (float*)malloc(4 * K);
    /* pairwises (elkan.rs:80-93): both orders, not symmetrised */
#pragma omp parallel for schedule(dynamic, 16) num_threads(g_threads) if (g_threads > 1)
    for (uint32_t e = 0; e < K * K; ++e) {
        uint32_t i = e / K, j = e % K;
        pw[i * K + j] = (i == j) ? 0.0f : dist(h, &h->cent[i], h->self_c[i], &h->cent[j], h->self_c[j]);
    }
    /* midpoints (elkan.rs:96-105) */
    for (uint32_t i = 0; i < K; ++i) {
        float r = RP_F32_MAX;
        for (uint32_t j = 0; j < K; ++j)
            if (j != i) r = rp_minf(r, pw[i * K + j] * 0.5f);
        mid[i] = r;
    }
#pragma omp parallel for schedule(dynamic, 8) num_threads(g_threads) if (g_threads > 1)
    for (uint64_t i = 0; i < h->N; ++i) {
        ora_bound* b = &h->bounds[i];
        if (!(b->error > mid[b->j])) continue; /* filter u > s[j] (elkan.rs:159) */
        const ora_hist* x = &h->points[i];
        if (b->stale) { /* refresh (elkan.rs:113-117, bounds.rs:79-83): distance(point, centroid) */
            float d = dist(h, x, h->self_p[i], &h->cent[b->j], h->self_c[b->j]);
            b->lower[b->j] = d;
            b->error = d;
            b->stale = 0;
        }
        for (uint32_t j = 0; j < K; ++j) { /* rebound (elkan.rs:119-123), has_shifted (bounds.rs:57-61) */
            if (b->j != j && b->error > b->lower[j] && b->error > 0.5f * pw[b->j * K + j]) {
                float d = dist(h, x, h->self_p[i], &h->cent[j], h->self_c[j]);
                b->lower[j] = d; /* witness (bounds.rs:85-91) */
                if (d < b->error) {
                    b->j = j;
                    b->error = d;
                }
            }
        }
    }
    /* recompute (elkan.rs:128-142): integer sums of members (this shard's share) */
    ora_hist* nc = (ora_hist*)calloc(K, sizeof(ora_hist));
    uint64_t* sizes = (uint64_t*)calloc(K, 8);
    for (uint64_t i = 0; i < h->N; ++i) {
        absorb(&nc[h->bounds[i].j], &h->points[i], h->bins);
        sizes[h->bounds[i].j] += 1;
    }
    unsigned char* out = (unsigned char*)partial;
    uint32_t* pc = (uint32_t*)out;
    uint32_t* pwt = pc + (size_t)K * h->bins;
    size_t off = (((size_t)K * h->bins + K) * 4 + 7) & ~(size_t)7;
    uint64_t* ps = (uint64_t*)(out + off);
    for (uint32_t k = 0; k < K; ++k) {
        memcpy(pc + (size_t)k * h->bins, nc[k].counts, 4 * h->bins);
        pwt[k] = (uint32_t)nc[k].weight;
        ps[k] = sizes[k];
    }
    free(pw); free(mid); free(nc); free(sizes);
}
ORA_API void ora_kmeans_step_finish(ora_kmeans* h, const void* reduced, float* drift_out, uint64_t* sizes_out,
                                    double* reassigned) {
    uint32_t K = h->K;
    const unsigned char* in = (const unsigned char*)reduced;
    const uint32_t* pc = (const uint32_t*)in;
    const uint32_t* pwt = pc + (size_t)K * h->bins;
    size_t off = (((size_t)K * h->bins + K) * 4 + 7) & ~(size_t)7;
    const uint64_t* ps = (const uint64_t*)(in + off);
    ora_hist* nc = (ora_hist*)calloc(K, sizeof(ora_hist));
    for (uint32_t k = 0; k < K; ++k) {
        memcpy(nc[k].counts, pc + (size_t)k * h->bins, 4 * h->bins);
        nc[k].weight = pwt[k];
    }
    /* drift (elkan.rs:108-110): distance(new, old) */
    float* drift = (float*)malloc(4 * K);
    float* self_n = (float*)malloc(4 * K);
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads) if (g_threads > 1)
    for (uint32_t j = 0; j < K; ++j) self_n[j] = point_self(h, &nc[j]);
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads) if (g_threads > 1)
    for (uint32_t j = 0; j < K; ++j) drift[j] = dist(h, &nc[j], self_n[j], &h->cent[j], h->self_c[j]);
    /* Bounds::update (bounds.rs:69-77) */
    for (uint64_t i = 0; i < h->N; ++i) {
        ora_bound* b = &h->bounds[i];
        for (uint32_t j = 0; j < K; ++j) b->lower[j] = rp_maxf(b->lower[j] - drift[j], 0.0f);
        b->error += drift[b->j];
        b->stale = 1;
    }
    memcpy(h->cent, nc, sizeof(ora_hist) * K);
    memcpy(h->self_c, self_n, 4 * K);
    /* Prior::tally (prior.rs:35-47): reassignments of this shard; sizes are the reduced (global) ones */
    uint64_t moved = 0;
    for (uint64_t i = 0; i < h->N; ++i) {
        uint32_t j = h->bounds[i].j;
        if ((uint8_t)j != h->prior[i]) {
            moved += 1;
            h->prior[i] = (uint8_t)j;
        }
    }
    if (sizes_out) memcpy(sizes_out, ps, 8 * K);
    if (reassigned) *reassigned = (double)moved / (double)h->N;
    if (drift_out) memcpy(drift_out, drift, 4 * K);
    free(nc); free(drift); free(self_n);
}
ORA_API void ora_kmeans_step(ora_kmeans* h, float* drift_out, uint64_t* sizes_out, double* reassigned) {
    void* partial = malloc(ora_kmeans_partial_bytes(h));
    ora_kmeans_step_local(h, partial);
    ora_kmeans_step_finish(h, partial, drift_out, sizes_out, reassigned);
    free(partial);
}

/* Elkan::step_naive (elkan.rs:171-188) + install */
ORA_API void ora_kmeans_step_naive(ora_kmeans* h) {
    ora_hist* nc = (ora_hist*)calloc(h->K, sizeof(ora_hist));
    for (uint64_t i = 0; i < h->N; ++i) {
        uint32_t j;
        float d;
        neighbor(h, i, &j, &d);
        absorb(&nc[j], &h->points[i], h->bins);
    }
    memcpy(h->cent, nc, sizeof(ora_hist) * h->K);
    refresh_self_c(h);
    free(nc);
}

/* Layer::lookup (layer.rs:62-82) */
ORA_API void ora_kmeans_assign(const ora_kmeans* h, uint8_t* bucket, float* distance) {
#pragma omp parallel for schedule(dynamic, 8) num_threads(g_threads) if (g_threads > 1)
    for (uint64_t i = 0; i < h->N; ++i) {
        uint32_t j;
        float d;
        neighbor(h, i, &j, &d);
        if (bucket) bucket[i] = (uint8_t)j;
        if (distance) distance[i] = d;
    }
}
ORA_API void ora_kmeans_bounds(const ora_kmeans* h, uint8_t* j, float* upper, float* lower) {
    for (uint64_t i = 0; i < h->N; ++i) {
        if (j) j[i] = (uint8_t)h->bounds[i].j;
        if (upper) upper[i] = h->bounds[i].error;
        if (lower) memcpy(lower + i * h->K, h->bounds[i].lower, 4 * h->K);
    }
}
ORA_API void ora_kmeans_centroids(const ora_kmeans* h, uint32_t* counts, uint64_t* weight) {
    for (uint32_t j = 0; j < h->K; ++j) {
        if (counts) memcpy(counts + (size_t)j * h->bins, h->cent[j].counts, 4 * h->bins);
        if (weight) weight[j] = h->cent[j].weight;
    }
}
/* Layer::metric (layer.rs:85-101) then Metric::from(BTreeMap) (metric.rs:127-141) */
ORA_API void ora_kmeans_metric(const ora_kmeans* h, float* tri_out) {
    uint32_t K = h->K;
    float mx = RP_EPSILON;
    for (uint32_t i = 0; i < K; ++i)
        for (uint32_t j = 0; j < i; ++j) {
            float d = dist(h, &h->cent[i], h->self_c[i], &h->cent[j], h->self_c[j]) +
                      dist(h, &h->cent[j], h->self_c[j], &h->cent[i], h->self_c[i]);
            d = d / 2.0f;
            tri_out[rp_tri_index(i, j)] = d;
            mx = rp_maxf(mx, d);
        }
    for (uint32_t t = 0; t < K * (K - 1) / 2; ++t) tri_out[t] = tri_out[t] / mx;
}
/* Elkan::rms_with (elkan.rs:191-200): per-point d^2 accumulated in f64 in point order */
ORA_API float ora_kmeans_rms(const ora_kmeans* h) {
    double acc = 0.0;
    for (uint64_t i = 0; i < h->N; ++i) {
        uint32_t j = h->bounds[i].j;
        float d = dist(h, &h->points[i], h->self_p[i], &h->cent[j], h->self_c[j]);
        acc += (double)(d * d);
    }
    return (float)sqrt(acc / (double)h->N);
}
ORA_API void ora_lloyd_stats(uint64_t* distances, uint64_t* iters, int reset) {
    if (distances) *distances = g_distances;
    if (iters) *iters = g_sinkhorn_iters;
    if (reset) { g_distances = 0; g_sinkhorn_iters = 0; }
}


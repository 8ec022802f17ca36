// lloyd_shared.hpp — what the host and BOTH arithmetic passes of the lloyd kernels share: the kernel-argument structures and the sizes.
// (lloyd.hip includes lloyd_kernels.hpp twice: namespace lm_contract = include/rp_math.h's f32 exp / ln, lm_glibc = glibc's.)
#pragma once
#include <cstdint>

namespace rp {

#define MAXB 256  // bins and K are both <= 256 (Abstraction index is 8 bits, kicker/src/abstraction.rs:22-23)

struct Metric {
    const float* Cm;  // [bins][bins] raw_distance (0 on the diagonal)          metric.rs:41-55
    const float* Rt;  // [bins][bins] raw_distance / temperature                sinkhorn.rs:129-131
    uint32_t bins;
    uint32_t iters;
    float tol;
    unsigned long long* stats;  // [0] distances, [1] sinkhorn iterations, [2] exp evaluations of the softmin/cost loops
    uint32_t stat_stripes;      // the counters are striped over this many 128-byte lines (STAT): every wavefront adds to
                                // them once per solve, and adds to ONE address serialise at its L2 channel (~9 ns each)
    // k-means++ computes distance(centroid_k, point) — the very value Elkan::neighbor needs (elkan.rs:68-77: the same
    // centroid-first call, first minimum wins).  Every solved distance is noted: nearest centroid so far and its distance.
    // -1 = not known (the picked point, whose potential is set to 0 without a solve).  See k_init_from_kpp.
    float* kpp_d;               // [N] or NULL
    uint8_t* kpp_j;             // [N]
    // the k-means++ interval filter's tripwire (kpp_bound.hpp): every 521st point the filter would drop is solved anyway and the lower
    // bound it was dropped with (kpp_claim[i] > 0) compared with the exact distance; a bound above it counts into kpp_bad
    float* kpp_claim;               // [N] or NULL
    unsigned long long* kpp_bad;    // one counter (the production sample check's)
};

#define STAT_STRIDE 16u  // u64 per stripe
#define KM_STAT_STRIPES 256u

// one prepared centroid set: integer sums + the derived support / log-density tables
struct CentroidSet {
    uint32_t* counts;  // [K][bins]
    uint32_t* weight;  // [K]
    uint32_t* n;       // [K]  support size
    uint16_t* sup;     // [K][MAXB] support bins ascending
    float* lnd;        // [K][MAXB] ln(density) on the support
    float* dens;       // [bins][K] density, transposed (variation path)
    float* densR;      // [K][256] density by centroid, 0 off the support and past `bins` (the MFMA bound's mu operand)
    float* mincT;      // [256][256] mincT[y][k] = min over x in supp(centroid k) of C(x, y): the column-marginal bound
    float* self;       // [K] OT(c,c)
};

struct Points {
    const uint8_t* counts;  // [N][stride]
    const uint32_t* weight; // [N]
    const float* self;      // [N] OT(p,p)
    uint32_t stride;
    uint64_t N;
};

#define PAIR_ROWS 32u
#define QUAD_ROWS 16u

struct Bounds {
    uint8_t* j;      // [N]
    float* u;        // [N]   Bounds::error
    uint8_t* stale;  // [N]
    float* lower;    // [N][K]
    // the last EXACT distance(point, its centroid) and what it was measured against.  Bounds::refresh (bounds.rs:79-83)
    // recomputes distance(point, centroid j) whenever the bound is stale; the distance is a pure function of the two
    // histograms, so while centroid j has not changed (cver[j], bumped when its integer sums change) and the point still
    // belongs to it the refresh would return memo_d bit for bit — the solve is skipped.  Late iterations move a few dozen
    // points: most centroids, hence most refreshes, repeat.
    float* memo_d;          // [N]
    uint32_t* memo_ver;     // [N]  cver[memo_j] at the time; 0 = nothing remembered
    uint8_t* memo_j;        // [N]
    const uint32_t* cver;   // [K]  content version of the centroids in use (starts at 1)
    // interval-decided refresh (refresh_bound.hpp): while uiv[i] is set, [ulo[i], u[i]] contains the reference's Bounds::error
    float* ulo;             // [N]  (NULL: the layer has no refresh bound)
    uint8_t* uiv;           // [N]
    // the last interval of distance(point, its centroid) and what it was computed against: like memo_d, for refreshes the interval settled
    float* im_lo;           // [N]
    float* im_hi;           // [N]
    uint32_t* im_ver;       // [N]  cver[im_j] at the time; 0 = nothing remembered
    uint8_t* im_j;          // [N]
};

struct KppLists {
    uint32_t* list[3];     // active points that are solved four / two / one per wavefront
    unsigned int* count;   // [3]
};

struct Refresh {
    const uint8_t* code;  // [N] interval mode: what each point needs this step (refresh_bound.hpp); NULL: the stale-bound predicate
    uint8_t want;         //     the code this pass collects
    const uint8_t* nsup;  // [N] support sizes
    uint32_t* count;      // [K]   points needing a refresh per cluster, then the fill cursor
    uint32_t* offset;     // [K+1] start of each cluster's (even-padded) bucket; offset[K] = entries in the list
    uint32_t* list;       // [N + 2K] point indices, 0xffffffff = padding
};

#define KPP_BLOCK 1024

#define KR_CHUNK 1024u

#define RC_CHUNK 16384  // points per round (256 threads x 64 assignment bytes)

#define VB 32  // points per workgroup

}  // namespace rp

// deuce.hip — the inputs of the abstraction pipeline (include/rp_mi355x.h "abstraction inputs"):
// hand strength, river equity, suit isomorphism, the isomorphism iterator and the histogram projection that turns
// one street's lookup table into the previous street's k-means points.
// Reference: crates/deuce/src/{evaluator,strength,ranking,kicks,hand,hand_iter,observation,observation_iter,
// permutation,isomorphism,isomorphism_iter,street}.rs, crates/lloyd/src/lookup.rs, crates/kicker/src/abstraction.rs.
// Oracle: oracle/rp_oracle_deuce.c.  All of it is integer work except one IEEE division per river observation.
//
// MI355X mapping.
//  * Strength.  The reference walks nibbles of the u64 card set; here a hand is four 13-bit suit words and the
//    rank multiplicities come out bit-sliced (two half adders over the suit words): quads / trips / pairs are
//    13-bit masks, every "highest rank with n of a kind" is one count-leading-zeros.  The result is ONE u32 whose
//    integer order is the reference's derived Ord on Strength, so a showdown is an integer compare.
//  * River equity: one wavefront per observation; the 990 opposing holes are spread over the 64 lanes (15.5 rounds,
//    97 % lane use), the board's suit words are wave-uniform, wins and losses meet in a wave reduction.
//  * Isomorphism iterator: index arithmetic instead of iteration — observation (p, r) is pocket number p and
//    board number r in the combinatorial number system (colex order IS ascending bit-set order, the reference
//    iterator's order), so 2.8 G river candidates are tested for canonicity in parallel and compacted in order by a
//    count / scan / write pair of launches.  Pocket ranges shard across GPUs with no exchange.
//  * Lookup: the table stays in iterator order; its search key is (pocket number << 52 | board bit set), monotone
//    in that order, so Lookup::lookup is a binary search over u64 and Lookup::projections is one wavefront per
//    observation, one lane per revealed card.
#include <hip/hip_runtime.h>
#include "sortscan.hpp"

#include <vector>

#include "cards.hpp"
#include "obs.hpp"
#include "rp_internal.h"

namespace rp {

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

static thread_local double g_last_ms = 0.0;

// orders this wavefront's LDS traffic (one wavefront's lanes exchange data through LDS without a workgroup barrier)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// From<i64> for Observation (observation.rs:144-165): one byte (card + 1) per card, the two lowest are the pocket
__device__ __forceinline__ void obs_decode(int64_t bits, uint64_t* pocket, uint64_t* public_) {
    uint64_t po = 0, pu = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t b = (uint64_t)bits >> (8 * i);
        if (bits <= 0 || b == 0) break;
        const uint32_t c = (uint32_t)(b & 0xffu) - 1u;
        const uint64_t card = c < 52u ? 1ull << c : 0ull;  // a byte that is no card leaves the hand short: callers check sizes
        if (i < 2) po |= card;
        else pu |= card;
    }
    *pocket = po, *public_ = pu;
}
// ---------------------------------------------------------------------------------------------------------------
// combinatorial number system.  Colex rank r <-> k-subset of {0..n-1}; ascending rank = ascending bit set, the order
// of HandIterator (hand_iter.rs:18-28, Gosper's successor).
// ---------------------------------------------------------------------------------------------------------------
struct Binom {
    uint32_t c[53][6];  // c[n][k] = C(n, k), k <= 5
};
static Binom make_binom() {
    Binom b;
    for (int n = 0; n <= 52; ++n)
        for (int k = 0; k <= 5; ++k) b.c[n][k] = k == 0 ? 1u : (n == 0 ? 0u : b.c[n - 1][k - 1] + b.c[n - 1][k]);
    return b;
}
__device__ __forceinline__ uint64_t unrank(const uint32_t (*C)[6], uint32_t r, uint32_t k, uint32_t n) {
    uint64_t m = 0;
    uint32_t c = n;
    for (uint32_t i = k; i >= 1; --i) {
        do --c;
        while (C[c][i] > r);  // largest c with C(c, i) <= r
        r -= C[c][i];
        m |= 1ull << c;
    }
    return m;
}
__device__ __forceinline__ uint64_t gosper(uint64_t x) {
    const uint64_t a = x | (x - 1), b = a + 1;
    return b | (((~a & b) - 1) >> (1 + __builtin_ctzll(x)));
}
// spread a bit set over the cards that are not c1 < c2: open a zero at bit c1, then at bit c2 (order preserving)
__device__ __forceinline__ uint64_t spread2(uint64_t m, uint32_t c1, uint32_t c2) {
    uint64_t lo = m & ((1ull << c1) - 1);
    m = lo | ((m ^ lo) << 1);
    lo = m & ((1ull << c2) - 1);
    return lo | ((m ^ lo) << 1);
}
__device__ __forceinline__ void pocket_cards(uint32_t p, uint32_t* c1, uint32_t* c2) {  // p-th two-card hand
    uint32_t hi = 1;
    while ((hi + 1) * hi / 2 <= p) ++hi;  // largest hi with C(hi, 2) <= p
    *c2 = hi, *c1 = p - hi * (hi - 1) / 2;
}
// ---------------------------------------------------------------------------------------------------------------
// IsomorphismIterator (isomorphism_iter.rs:7-20 over observation_iter.rs:13-104)
// ---------------------------------------------------------------------------------------------------------------
#define EN_THREADS 256u
#define EN_RUN 16u  // consecutive boards per thread
struct EnumArgs {
    uint32_t pocket_lo, n_pockets, n_board, blocks_per_pocket;
    uint32_t boards;  // C(50, n_board)
};
template <bool WRITE>
__global__ __launch_bounds__(EN_THREADS) void k_enumerate(EnumArgs a, const Binom* bn, uint32_t* counts, const uint64_t* offsets,
                                                          int64_t* out, uint64_t cap) {
    __shared__ uint32_t C[53][6];
    __shared__ uint32_t wave_tot[EN_THREADS / 64];
    for (uint32_t i = threadIdx.x; i < 53 * 6; i += EN_THREADS) (&C[0][0])[i] = (&bn->c[0][0])[i];
    __syncthreads();
    const uint32_t pk = blockIdx.x / a.blocks_per_pocket, blk = blockIdx.x % a.blocks_per_pocket;
    uint32_t c1, c2;
    pocket_cards(a.pocket_lo + pk, &c1, &c2);
    const uint64_t pocket = (1ull << c1) | (1ull << c2);
    const uint32_t r0 = (blk * EN_THREADS + threadIdx.x) * EN_RUN;
    uint32_t found = 0;
    uint64_t keep[EN_RUN];
    if (r0 < a.boards) {
        uint64_t m = a.n_board ? unrank(C, r0, a.n_board, 50) : 0ull;
        const uint32_t cnt = min(EN_RUN, a.boards - r0);
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint64_t board = spread2(m, c1, c2);
            if (is_canonical(pocket, board)) {
                if (WRITE) keep[found] = board;
                ++found;
            }
            if (i + 1 < cnt) m = gosper(m);
        }
    }
    // block total / in-block exclusive prefix, in thread order
    uint32_t incl = found;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63) >= d) incl += o;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t w = 0; w < EN_THREADS / 64; ++w) {
        if (w < (threadIdx.x >> 6)) base += wave_tot[w];
        tot += wave_tot[w];
    }
    if (!WRITE) {
        if (threadIdx.x == 0) counts[blockIdx.x] = tot;
        return;
    }
    uint64_t at = offsets[blockIdx.x] + base + incl - found;
    for (uint32_t i = 0; i < found; ++i, ++at)
        if (at < cap) out[at] = obs_encode(pocket, keep[i]);
}

// ---------------------------------------------------------------------------------------------------------------
// Observation::equity (observation.rs:45-63) and Abstraction::from(Probability) (kicker/src/abstraction.rs:61-63,93-99)
// ---------------------------------------------------------------------------------------------------------------
#define EQ_THREADS 256u
struct PairLut {
    uint16_t ij[990];  // i | j << 8, i < j < 45
};
static PairLut make_pairs() {
    PairLut l;
    uint32_t v = 0;
    for (uint32_t j = 1; j < 45; ++j)
        for (uint32_t i = 0; i < j; ++i) l.ij[v++] = (uint16_t)(i | (j << 8));
    return l;
}
__global__ __launch_bounds__(EQ_THREADS) void k_river_equity(const int64_t* obs, uint64_t n, const PairLut* lut, float* equity,
                                                             uint8_t* bucket, uint32_t* bad) {
    __shared__ uint16_t pairs[990];
    __shared__ uint8_t rem[EQ_THREADS / 64][64];
    for (uint32_t i = threadIdx.x; i < 990; i += EQ_THREADS) pairs[i] = lut->ij[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t waves = (uint64_t)gridDim.x * (EQ_THREADS / 64);
    for (uint64_t o = (uint64_t)blockIdx.x * (EQ_THREADS / 64) + wave; o < n; o += waves) {
        uint64_t pocket, public_;
        obs_decode(obs[o], &pocket, &public_);
        const uint64_t used = pocket | public_;
        if (__popcll(pocket) != 2 || __popcll(public_) != 5 || (pocket & public_)) {  // not a river observation
            if (lane == 0) atomicAdd(bad, 1u);
            continue;
        }
        // the 45 cards nobody shows, ascending
        wave_lds_sync();  // the previous observation's reads are done
        if (lane < 52 && !((used >> lane) & 1ull)) rem[wave][__popcll(~used & ((1ull << lane) - 1))] = (uint8_t)lane;
        wave_lds_sync();
        const uint64_t board = sw_of_hand(public_);
        const uint32_t hero = strength_key(board | sw_of_hand(pocket));
        uint32_t won = 0, lost = 0;
        for (uint32_t v = lane; v < 990; v += 64) {
            const uint32_t ij = pairs[v];
            const uint32_t k = strength_key(board | sw_of_card(rem[wave][ij & 0xffu]) | sw_of_card(rem[wave][ij >> 8]));
            won += hero > k;
            lost += hero < k;
        }
        for (int d = 32; d > 0; d >>= 1) {
            won += __shfl_xor(won, d, 64);
            lost += __shfl_xor(lost, d, 64);
        }
        if (lane == 0) {
            const uint32_t sum = won + lost;
            const float e = sum == 0 ? 0.5f : (float)won / (float)sum;
            if (equity) equity[o] = e;
            if (bucket) bucket[o] = (uint8_t)(uint32_t)roundf(e * 100.0f);
        }
    }
}

__global__ void k_strength(const uint64_t* hands, uint64_t n, uint32_t* keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = strength_key(sw_of_hand(hands[i] & HAND_MASK));
}
__global__ void k_canonical(const int64_t* obs, uint64_t n, int64_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t po, pu, cp, cb;
    obs_decode(obs[i], &po, &pu);
    canonical(po, pu, &cp, &cb);
    out[i] = obs_encode(cp, cb);
}

// ---------------------------------------------------------------------------------------------------------------
// Lookup (lloyd/src/lookup.rs)
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_search_keys(const int64_t* obs, uint64_t n, uint64_t* keys, uint32_t n_cards, uint32_t* bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t po, pu;
    obs_decode(obs[i], &po, &pu);
    if (__popcll(po) != 2 || (uint32_t)__popcll(pu) + 2u != n_cards || (po & pu)) {
        atomicAdd(bad, 1u);
        keys[i] = 0;
        return;
    }
    keys[i] = search_key(po, pu);
}
__global__ void k_check_sorted(const uint64_t* keys, uint64_t n, uint32_t* bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < n && keys[i] >= keys[i + 1]) atomicAdd(bad, 1u);
}
// Lookup::lookup(&Isomorphism::from(obs)) (lookup.rs:23-25)
__global__ void k_lookup_get(const uint64_t* keys, const uint8_t* abs_, uint64_t n_keys, const int64_t* obs, uint64_t n, uint8_t* out,
                             uint32_t* missing) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t po, pu, cp, cb;
    obs_decode(obs[i], &po, &pu);
    if (__popcll(po) != 2 || (po & pu)) {
        atomicAdd(missing, 1u);
        return;
    }
    canonical(po, pu, &cp, &cb);
    const int64_t at = table_find(keys, n_keys, search_key(cp, cb));
    if (at < 0) atomicAdd(missing, 1u);
    else out[i] = abs_[at];
}
// Lookup::future (lookup.rs:35-45): children (observation.rs:35-40) -> Isomorphism::from -> lookup -> Histogram
// (histogram.rs:207-212).  One wavefront per observation, one lane per revealed card.
#define PJ_THREADS 256u
#define PJ_MAX_BINS 256u
__global__ __launch_bounds__(PJ_THREADS) void k_project(const uint64_t* keys, const uint8_t* abs_, uint64_t n_keys, const int64_t* obs,
                                                        uint64_t n, uint32_t n_cards, uint32_t bins, uint8_t* hist, uint32_t* missing) {
    __shared__ uint32_t h[PJ_THREADS / 64][PJ_MAX_BINS];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t waves = (uint64_t)gridDim.x * (PJ_THREADS / 64);
    for (uint64_t o = (uint64_t)blockIdx.x * (PJ_THREADS / 64) + wave; o < n; o += waves) {
        wave_lds_sync();
        for (uint32_t b = lane; b < bins; b += 64) h[wave][b] = 0;
        wave_lds_sync();
        uint64_t po, pu;
        obs_decode(obs[o], &po, &pu);
        const uint64_t used = po | pu;
        if (__popcll(po) != 2 || (uint32_t)__popcll(used) + 1u != n_cards || (po & pu)) {
            if (lane == 0) atomicAdd(missing, 1u);
            continue;
        }
        if (lane < 52 && !((used >> lane) & 1ull)) {  // the revealed card is `lane`
            uint64_t cp, cb;
            canonical(po, pu | (1ull << lane), &cp, &cb);
            const int64_t at = table_find(keys, n_keys, search_key(cp, cb));
            const uint32_t a = at < 0 ? bins : abs_[at];
            if (a >= bins) atomicAdd(missing, 1u);
            else atomicAdd(&h[wave][a], 1u);
        }
        wave_lds_sync();
        for (uint32_t b = lane; b < bins; b += 64) hist[o * bins + b] = (uint8_t)h[wave][b];  // counts <= 47
    }
}

static int pick_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return rp::fail(RP_ERR_NO_DEVICE, "no HIP device: the library has no CPU path");
    if (device < 0 || device >= n) return rp::fail(RP_ERR_INVALID, "device %d out of range (%d devices)", device, n);
    HIP_TRY(hipSetDevice(device));
    return RP_OK;
}
// small device scratch that frees itself
struct Scratch {
    void* p = nullptr;
    ~Scratch() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
};
struct Timer {
    hipEvent_t a = nullptr, b = nullptr;
    Timer() {
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, nullptr);
    }
    void stop() {
        (void)hipEventRecord(b, nullptr);
        (void)hipEventSynchronize(b);
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, a, b);
        g_last_ms = ms;
    }
    ~Timer() {
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
    }
};
static const int N_BOARD[4] = {0, 3, 4, 5};     // street.rs:67-74
static const int N_OBSERVED[4] = {2, 5, 6, 7};  // street.rs:59-66

}  // namespace rp

using namespace rp;

struct rp_lookup {
    int device;
    int street;
    uint64_t n;
    uint64_t* keys;
    uint8_t* abs_;
};

namespace rp {
// nlmc.hip: the encoder's table of one street (device pointers; valid while the handle lives)
int lookup_view(const rp_lookup* t, const uint64_t** keys, const uint8_t** abs_, uint64_t* n, int* street) {
    if (!t) return rp::fail(RP_ERR_INVALID, "lookup_view: NULL handle");
    *keys = t->keys;
    *abs_ = t->abs_;
    *n = t->n;
    *street = t->street;
    return RP_OK;
}
}  // namespace rp

extern "C" {

int rp_deuce_kernel_ms(double* ms) {
    if (!ms) return rp::fail(RP_ERR_INVALID, "null argument");
    *ms = g_last_ms;
    return RP_OK;
}

int rp_hand_strength(int device, uint64_t n, const uint64_t* hands, uint32_t* keys) {
    if ((!hands || !keys) && n) return rp::fail(RP_ERR_INVALID, "null argument");
    if (int rc = pick_device(device)) return rc;
    if (!n) return RP_OK;
    Scratch in, out;
    HIP_TRY(in.alloc(n * 8));
    HIP_TRY(out.alloc(n * 4));
    HIP_TRY(hipMemcpy(in.p, hands, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_strength, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, in.as<uint64_t>(), n, out.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(keys, out.p, n * 4, hipMemcpyDeviceToHost));
    return RP_OK;
}

int rp_obs_canonical(int device, uint64_t n, const int64_t* obs, int64_t* canon) {
    if ((!obs || !canon) && n) return rp::fail(RP_ERR_INVALID, "null argument");
    if (int rc = pick_device(device)) return rc;
    if (!n) return RP_OK;
    Scratch in, out;
    HIP_TRY(in.alloc(n * 8));
    HIP_TRY(out.alloc(n * 8));
    HIP_TRY(hipMemcpy(in.p, obs, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_canonical, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, in.as<int64_t>(), n, out.as<int64_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(canon, out.p, n * 8, hipMemcpyDeviceToHost));
    return RP_OK;
}

int rp_isomorphisms(int device, int street, uint32_t pocket_lo, uint32_t pocket_hi, int64_t* obs_dev, uint64_t cap, uint64_t* n) {
    if (!n) return rp::fail(RP_ERR_INVALID, "null argument");
    if (street < 0 || street > 3) return rp::fail(RP_ERR_INVALID, "street %d: 0 preflop, 1 flop, 2 turn, 3 river", street);
    if (pocket_hi > 1326) pocket_hi = 1326;
    *n = 0;
    if (pocket_lo >= pocket_hi) return RP_OK;
    if (int rc = pick_device(device)) return rc;
    static const Binom bn = make_binom();
    EnumArgs a;
    a.pocket_lo = pocket_lo;
    a.n_pockets = pocket_hi - pocket_lo;
    a.n_board = (uint32_t)N_BOARD[street];
    a.boards = bn.c[50][a.n_board];
    a.blocks_per_pocket = (a.boards + EN_THREADS * EN_RUN - 1) / (EN_THREADS * EN_RUN);
    const uint64_t blocks = (uint64_t)a.n_pockets * a.blocks_per_pocket;
    Scratch dbn, counts, offs, tmp;
    HIP_TRY(dbn.alloc(sizeof(Binom)));
    HIP_TRY(hipMemcpy(dbn.p, &bn, sizeof(Binom), hipMemcpyHostToDevice));
    HIP_TRY(counts.alloc(blocks * 4));
    HIP_TRY(offs.alloc((blocks + 1) * 8));
    Timer t;
    hipLaunchKernelGGL((k_enumerate<false>), dim3((unsigned)blocks), dim3(EN_THREADS), 0, nullptr, a, dbn.as<Binom>(), counts.as<uint32_t>(),
                       (const uint64_t*)nullptr, (int64_t*)nullptr, (uint64_t)0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(tmp.alloc(ss::scan_scratch_bytes(blocks)));
    HIP_TRY(ss::exclusive_scan<uint64_t>(counts.as<uint32_t>(), offs.as<uint64_t>(), (uint32_t)blocks, tmp.p, nullptr));
    uint64_t last_off = 0;
    uint32_t last_cnt = 0;
    HIP_TRY(hipMemcpy(&last_off, offs.as<uint64_t>() + (blocks - 1), 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&last_cnt, counts.as<uint32_t>() + (blocks - 1), 4, hipMemcpyDeviceToHost));
    *n = last_off + last_cnt;
    if (obs_dev && cap) {
        hipLaunchKernelGGL((k_enumerate<true>), dim3((unsigned)blocks), dim3(EN_THREADS), 0, nullptr, a, dbn.as<Binom>(), (uint32_t*)nullptr,
                           offs.as<uint64_t>(), obs_dev, cap);
        HIP_TRY(hipGetLastError());
    }
    t.stop();
    HIP_TRY(hipDeviceSynchronize());
    return RP_OK;
}

int rp_river_equity(int device, uint64_t n, const int64_t* obs_dev, float* equity_dev, uint8_t* bucket_dev) {
    if (!obs_dev && n) return rp::fail(RP_ERR_INVALID, "null argument");
    if (int rc = pick_device(device)) return rc;
    if (!n) return RP_OK;
    static const PairLut lut = make_pairs();
    Scratch dl, bad;
    HIP_TRY(dl.alloc(sizeof(PairLut)));
    HIP_TRY(hipMemcpy(dl.p, &lut, sizeof(PairLut), hipMemcpyHostToDevice));
    HIP_TRY(bad.alloc(4));
    HIP_TRY(hipMemset(bad.p, 0, 4));
    const uint64_t want = (n + EQ_THREADS / 64 - 1) / (EQ_THREADS / 64);
    const unsigned grid = (unsigned)(want < 256ull * 64 ? want : 256ull * 64);
    Timer t;
    hipLaunchKernelGGL(k_river_equity, dim3(grid), dim3(EQ_THREADS), 0, nullptr, obs_dev, n, dl.as<PairLut>(), equity_dev, bucket_dev,
                       bad.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    t.stop();
    uint32_t nbad = 0;
    HIP_TRY(hipMemcpy(&nbad, bad.p, 4, hipMemcpyDeviceToHost));
    if (nbad) return rp::fail(RP_ERR_INVALID, "%u of the observations are not river observations (2 + 5 distinct cards)", nbad);
    return RP_OK;
}

int rp_lookup_create(int device, int street, uint64_t n, const int64_t* obs_dev, const uint8_t* abs_dev, rp_lookup** out) {
    if (!out || !obs_dev || !abs_dev || !n) return rp::fail(RP_ERR_INVALID, "null or empty argument");
    if (street < 0 || street > 3) return rp::fail(RP_ERR_INVALID, "street %d: a lookup table is for the preflop (0), flop (1), turn (2) or river (3)", street);
    if (int rc = pick_device(device)) return rc;
    rp_lookup* h = new rp_lookup{device, street, n, nullptr, nullptr};
    Scratch bad;
    hipError_t e = hipMalloc((void**)&h->keys, n * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&h->abs_, n);
    if (e == hipSuccess) e = bad.alloc(4);
    if (e == hipSuccess) e = hipMemset(bad.p, 0, 4);
    if (e == hipSuccess) e = hipMemcpy(h->abs_, abs_dev, n, hipMemcpyDeviceToDevice);
    uint32_t nbad = 0;
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_search_keys, dim3(grid), dim3(256), 0, nullptr, obs_dev, n, h->keys, (uint32_t)N_OBSERVED[street], bad.as<uint32_t>());
        hipLaunchKernelGGL(k_check_sorted, dim3(grid), dim3(256), 0, nullptr, h->keys, n, bad.as<uint32_t>());
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&nbad, bad.p, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess || nbad) {
        (void)hipFree(h->keys);
        (void)hipFree(h->abs_);
        delete h;
        if (e != hipSuccess) return rp::fail(RP_ERR_HIP, "rp_lookup_create: %s", hipGetErrorString(e));
        return rp::fail(RP_ERR_INVALID, "the table is not %d-card observations in IsomorphismIterator order (%u violations)", N_OBSERVED[street], nbad);
    }
    *out = h;
    return RP_OK;
}

int rp_lookup_destroy(rp_lookup* h) {
    if (!h) return RP_OK;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->keys);
    (void)hipFree(h->abs_);
    delete h;
    return RP_OK;
}

int rp_lookup_get(rp_lookup* h, uint64_t n, const int64_t* obs_dev, uint8_t* abs_dev) {
    if (!h || ((!obs_dev || !abs_dev) && n)) return rp::fail(RP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (!n) return RP_OK;
    Scratch miss;
    HIP_TRY(miss.alloc(4));
    HIP_TRY(hipMemset(miss.p, 0, 4));
    Timer t;
    hipLaunchKernelGGL(k_lookup_get, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, h->keys, h->abs_, h->n, obs_dev, n, abs_dev,
                       miss.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    t.stop();
    uint32_t nm = 0;
    HIP_TRY(hipMemcpy(&nm, miss.p, 4, hipMemcpyDeviceToHost));
    if (nm) return rp::fail(RP_ERR_INVALID, "%u observations have no entry in the table", nm);  // the reference panics (lookup.rs:24)
    return RP_OK;
}

int rp_lookup_project(rp_lookup* h, uint64_t n, const int64_t* obs_dev, uint32_t bins, uint8_t* hist_dev) {
    if (!h || ((!obs_dev || !hist_dev) && n)) return rp::fail(RP_ERR_INVALID, "null argument");
    if (bins == 0 || bins > PJ_MAX_BINS) return rp::fail(RP_ERR_INVALID, "bins must be in 1..%u", PJ_MAX_BINS);
    if (h->street < 2) return rp::fail(RP_ERR_UNSUPPORTED, "projection onto the preflop (19 600 flops per pocket) is not built");
    HIP_TRY(hipSetDevice(h->device));
    if (!n) return RP_OK;
    Scratch miss;
    HIP_TRY(miss.alloc(4));
    HIP_TRY(hipMemset(miss.p, 0, 4));
    const uint64_t want = (n + PJ_THREADS / 64 - 1) / (PJ_THREADS / 64);
    const unsigned grid = (unsigned)(want < 256ull * 32 ? want : 256ull * 32);
    Timer t;
    hipLaunchKernelGGL(k_project, dim3(grid), dim3(PJ_THREADS), 0, nullptr, h->keys, h->abs_, h->n, obs_dev, n, (uint32_t)N_OBSERVED[h->street], bins,
                       hist_dev, miss.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    t.stop();
    uint32_t nm = 0;
    HIP_TRY(hipMemcpy(&nm, miss.p, 4, hipMemcpyDeviceToHost));
    if (nm)
        return rp::fail(RP_ERR_INVALID, "%u children are missing from the table, out of range of `bins`, or not observations of the previous street", nm);
    return RP_OK;
}

}  // extern "C"

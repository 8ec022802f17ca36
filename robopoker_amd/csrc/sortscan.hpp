// sortscan.hpp — the device-wide primitives the row-addressed profile (sparse.hip) and the isomorphism enumeration (deuce.hip)
// need: exclusive scan, stable LSD radix sort of (u32 key, u32 value) pairs, run-length encoding of sorted keys.  Written for
// wave64 / gfx950 (ballot-based digit matching, LDS-resident digit cursors); no library underneath.
//
//   scan     three levels of 1024-element tiles (tile sums -> scan of the sums, recursively -> tile scans with their base);
//            inputs of at most SCAN_ONE elements go through one workgroup in one launch.
//   sort     8 or 9 bits per pass (27-bit row indices: three passes of 9).  k_rs_hist: per 2048-key tile an LDS histogram ->
//            hist[digit][tile]; every (digit, tile) gets its first output slot from the exclusive scan of that digit-major
//            array — computed by the scatter workgroups themselves while a pass has few tiles (two launches per pass), by the
//            tiled scan otherwise; k_rs_scatter re-reads the tile in eight
//            rounds of 256 keys (ascending index), ranks the keys of a round that share a digit by lane order inside a
//            wavefront (eight ballots) and by wavefront order across the workgroup (per-wave digit counts in LDS), so equal
//            keys keep their input order: the sort is STABLE, which the ordered update relies on (touches of a row are applied
//            in batch order).
//   rle      heads of runs flagged, scanned, written: distinct keys, the start of every run, the run lengths, their number.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

namespace rp {
namespace ss {

constexpr uint32_t SCAN_TILE = 1024;        // elements per workgroup (256 threads x 4)
constexpr uint32_t SCAN_ONE = 64 * 1024;    // at most this many elements: one workgroup, one launch
constexpr uint32_t ONE_THREADS = 1024;      // ... of this many threads (a thread walks n / 1024 consecutive elements: 15 -> 5 us at 10^4)
constexpr uint32_t RS_TILE = 2048;          // keys per workgroup and pass
constexpr uint32_t RS_ROUNDS = RS_TILE / 256;

// exclusive scan of v over the threads of the workgroup (whole wavefronts, at most 16: wave_tot holds blockDim.x / 64 words), thread
// order; *total = the sum
__device__ __forceinline__ uint64_t block_exscan64(uint64_t v, uint64_t* wave_tot, uint64_t* total) {
    const uint32_t tid = threadIdx.x, ln = tid & 63u;
    uint64_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = __shfl_up((uint32_t)incl, d, 64), hi = __shfl_up((uint32_t)(incl >> 32), d, 64);
        if ((int)ln >= d) incl += (uint64_t)lo | ((uint64_t)hi << 32);
    }
    if (ln == 63u) wave_tot[tid >> 6] = incl;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    const uint32_t nw = blockDim.x >> 6;
    for (uint32_t w = 0; w < nw; ++w) {
        const uint64_t c = wave_tot[w];
        if (w < (tid >> 6)) base += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// ---- scan ----------------------------------------------------------------------------------------------------------
// one workgroup scans everything: thread t owns the contiguous slice [t * per, (t + 1) * per)
template <class OUT>
static __global__ __launch_bounds__(1024) void k_scan_one(const uint32_t* in, OUT* out, uint32_t n, OUT* total_out) {
    __shared__ uint64_t wt[16];
    const uint32_t per = (n + blockDim.x - 1u) / blockDim.x, lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += in[i];
    uint64_t tot;
    uint64_t run = block_exscan64(s, wt, &tot);
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t v = in[i];  // read before the write: in and out may alias
        out[i] = (OUT)run;
        run += v;
    }
    if (total_out && threadIdx.x == 0) *total_out = (OUT)tot;
}
// level kernels of the tiled scan: sums of 1024-element tiles; tile scans with a base
static __global__ __launch_bounds__(256) void k_scan_sums(const uint32_t* in, uint32_t n, uint64_t* sums) {
    __shared__ uint64_t wt[4];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 4u;
    uint64_t s = 0;
    for (uint32_t k = 0; k < 4u; ++k) s += base + k < n ? in[base + k] : 0u;
    uint64_t tot;
    (void)block_exscan64(s, wt, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
static __global__ __launch_bounds__(256) void k_scan_sums64(uint64_t* data, uint32_t n, uint64_t* sums) {  // the levels above the first
    __shared__ uint64_t wt[4];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 4u;
    uint64_t s = 0;
    for (uint32_t k = 0; k < 4u; ++k) s += base + k < n ? data[base + k] : 0ull;
    uint64_t tot;
    (void)block_exscan64(s, wt, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
static __global__ __launch_bounds__(256) void k_scan_one64(uint64_t* data, uint32_t n) {  // in place, one workgroup
    __shared__ uint64_t wt[4];
    const uint32_t per = (n + 255u) / 256u, lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += data[i];
    uint64_t tot;
    uint64_t run = block_exscan64(s, wt, &tot);
    for (uint32_t i = lo; i < hi; ++i) {
        const uint64_t v = data[i];
        data[i] = run;
        run += v;
    }
}
static __global__ __launch_bounds__(256) void k_scan_tiles64(uint64_t* data, uint32_t n, const uint64_t* bases) {  // in place
    __shared__ uint64_t wt[4];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 4u;
    uint64_t v[4], s = 0;
    for (uint32_t k = 0; k < 4u; ++k) {
        v[k] = base + k < n ? data[base + k] : 0ull;
        s += v[k];
    }
    uint64_t tot;
    uint64_t run = bases[blockIdx.x] + block_exscan64(s, wt, &tot);
    for (uint32_t k = 0; k < 4u; ++k) {
        if (base + k < n) data[base + k] = run;
        run += v[k];
    }
}
template <class OUT>
static __global__ __launch_bounds__(256) void k_scan_tiles(const uint32_t* in, OUT* out, uint32_t n, const uint64_t* bases, OUT* total_out) {
    __shared__ uint64_t wt[4];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 4u;
    uint32_t v[4];
    uint64_t s = 0;
    for (uint32_t k = 0; k < 4u; ++k) {
        v[k] = base + k < n ? in[base + k] : 0u;
        s += v[k];
    }
    uint64_t tot;
    uint64_t run = bases[blockIdx.x] + block_exscan64(s, wt, &tot);
    for (uint32_t k = 0; k < 4u; ++k) {
        if (base + k < n) out[base + k] = (OUT)run;
        run += v[k];
    }
    if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *total_out = (OUT)run;
}

// bytes of scratch exclusive_scan needs for n elements
inline size_t scan_scratch_bytes(uint64_t n) {
    size_t words = 0;
    for (uint64_t m = (n + SCAN_TILE - 1) / SCAN_TILE; ; m = (m + SCAN_TILE - 1) / SCAN_TILE) {
        words += m;
        if (m <= SCAN_ONE) break;
    }
    return words * 8 + 64;
}
// out[i] = sum of in[0..i) (OUT = uint32_t or uint64_t); in may alias out when OUT is uint32_t; total (device, optional) = the sum
template <class OUT>
inline hipError_t exclusive_scan(const uint32_t* in, OUT* out, uint32_t n, void* scratch, hipStream_t st, OUT* total = nullptr) {
    if (n == 0) return hipSuccess;
    if (n <= SCAN_ONE) {
        hipLaunchKernelGGL((k_scan_one<OUT>), dim3(1), dim3(ONE_THREADS), 0, st, in, out, n, total);
        return hipGetLastError();
    }
    // level 0 tile sums, then the sums are scanned in place (recursively), then the tiles
    uint64_t* lvl[4];
    uint32_t cnt[4];
    int levels = 0;
    uint64_t* p = reinterpret_cast<uint64_t*>(scratch);
    for (uint32_t m = (n + SCAN_TILE - 1) / SCAN_TILE;; m = (m + SCAN_TILE - 1) / SCAN_TILE) {
        lvl[levels] = p;
        cnt[levels] = m;
        p += m;
        levels += 1;
        if (m <= SCAN_ONE || levels == 4) break;
    }
    hipLaunchKernelGGL(k_scan_sums, dim3(cnt[0]), dim3(256), 0, st, in, n, lvl[0]);
    for (int l = 1; l < levels; ++l) hipLaunchKernelGGL(k_scan_sums64, dim3(cnt[l]), dim3(256), 0, st, lvl[l - 1], cnt[l - 1], lvl[l]);
    hipLaunchKernelGGL(k_scan_one64, dim3(1), dim3(256), 0, st, lvl[levels - 1], cnt[levels - 1]);
    for (int l = levels - 2; l >= 0; --l) hipLaunchKernelGGL(k_scan_tiles64, dim3(cnt[l + 1]), dim3(256), 0, st, lvl[l], cnt[l], lvl[l + 1]);
    hipLaunchKernelGGL((k_scan_tiles<OUT>), dim3(cnt[0]), dim3(256), 0, st, in, out, n, lvl[0], total);
    return hipGetLastError();
}

// ---- radix sort ----------------------------------------------------------------------------------------------------
// RB bits per pass (8 or 9: 27-bit row indices sort in three passes of 9)
// TILE_MAJOR: hist[tile][digit] (the layout the self-offset scatter reads coalesced); otherwise hist[digit][tile] (what a scan
// of the whole array turns into first slots)
template <uint32_t RB, bool TILE_MAJOR>
static __global__ __launch_bounds__(256) void k_rs_hist(const uint32_t* keys, uint32_t n, uint32_t shift, uint32_t ntiles, uint32_t* hist) {
    constexpr uint32_t NB = 1u << RB;
    __shared__ uint32_t h[NB];
    for (uint32_t d = threadIdx.x; d < NB; d += 256u) h[d] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
    for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
        const uint32_t i = base + r * 256u + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & (NB - 1u)], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < NB; d += 256u) hist[TILE_MAJOR ? (size_t)blockIdx.x * NB + d : (size_t)d * ntiles + blockIdx.x] = h[d];
}
// SCANNED: `offs` is the exclusive scan of the digit-major histogram (large inputs).  Otherwise `offs` is the raw histogram
// and every workgroup derives its own first slots from it (digit totals, their scan, the counts of the earlier tiles): no
// scan launch between the two kernels of a pass — the cheaper way while a pass has at most a few hundred tiles.
template <uint32_t RB, bool SCANNED>
static __global__ __launch_bounds__(256) void k_rs_scatter(const uint32_t* keys, const uint32_t* vals, uint32_t n, uint32_t shift,
                                                           uint32_t ntiles, const uint32_t* offs, uint32_t* keys_out, uint32_t* vals_out) {
    constexpr uint32_t NB = 1u << RB, PER = NB / 256u;
    __shared__ uint32_t cursor[NB];      // next output slot of each digit for this tile
    __shared__ uint32_t wcnt[4][NB];     // keys of the round with each digit, per wavefront
    __shared__ uint64_t wt[4];
    const uint32_t tid = threadIdx.x, ln = tid & 63u, wv = tid >> 6;
    if (SCANNED) {
        for (uint32_t d = tid; d < NB; d += 256u) cursor[d] = offs[(size_t)d * ntiles + blockIdx.x];
    } else {
        // thread t owns the PER consecutive digits t * PER ..: their totals over all tiles and over the tiles before this one
        // (offs is tile-major here: consecutive threads read consecutive words)
        uint32_t tot[PER], before[PER];
        uint64_t mine = 0;
        for (uint32_t q = 0; q < PER; ++q) tot[q] = before[q] = 0;
        for (uint32_t k = 0; k < ntiles; ++k) {
            for (uint32_t q = 0; q < PER; ++q) {
                const uint32_t c = offs[(size_t)k * NB + tid * PER + q];
                before[q] += k < blockIdx.x ? c : 0u;
                tot[q] += c;
            }
        }
        for (uint32_t q = 0; q < PER; ++q) mine += tot[q];
        uint64_t all;
        uint64_t run = block_exscan64(mine, wt, &all);
        for (uint32_t q = 0; q < PER; ++q) {
            cursor[tid * PER + q] = (uint32_t)run + before[q];
            run += tot[q];
        }
    }
    // the tile's keys and values are requested up front (eight independent loads per array), the rounds then run out of
    // registers and LDS
    const uint32_t base = blockIdx.x * RS_TILE;
    uint32_t kreg[RS_ROUNDS], vreg[RS_ROUNDS];
#pragma unroll
    for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
        const uint32_t i = base + r * 256u + tid;
        kreg[r] = i < n ? keys[i] : 0u;
        vreg[r] = i < n ? vals[i] : 0u;
    }
    for (uint32_t w = 0; w < 4u; ++w)
        for (uint32_t d = tid; d < NB; d += 256u) wcnt[w][d] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
        const uint32_t i = base + r * 256u + tid;
        const bool live = i < n;
        const uint32_t key = kreg[r], val = vreg[r];
        const uint32_t d = (key >> shift) & (NB - 1u);
        // the lanes of this wavefront with the same digit (dead lanes match nobody)
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (uint32_t b = 0; b < RB; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = (uint32_t)__popcll(peers & ((1ull << ln) - 1ull));
        const bool first = live && before == 0u;  // the first of the peers speaks for them
        const uint32_t mine = (uint32_t)__popcll(peers);
        if (first) wcnt[wv][d] = mine;
        __syncthreads();
        if (live) {
            uint32_t pos = cursor[d] + before;
            for (uint32_t w = 0; w < wv; ++w) pos += wcnt[w][d];
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        if (first) {
            atomicAdd(&cursor[d], mine);
            wcnt[wv][d] = 0;  // clean for the next round
        }
        __syncthreads();
    }
}

// ---- the whole sort by ONE workgroup of 1024 (at most SORT_ONE keys): what a batch of the reference's size needs (128 trees, ~10^4
// Decisions: six launches of five workgroups each through the tiled passes above).  Wavefront w owns a contiguous chunk of the input;
// it ranks its keys digit by digit in rounds of 64 against ITS OWN running digit counts (LDS, no workgroup barrier inside the rounds);
// one scan over (digit, wavefront) turns the counts into first slots; keys, values and ranks wait in registers meanwhile.  Between the
// passes keys and values stay in LDS (64 + 32 KB: a value is a position below 16 384); only the first pass reads global memory and
// only the last writes it.  A pass is four workgroup barriers.
constexpr uint32_t SORT_ONE = 16384;  // 16 rounds of 64 per wavefront
struct SortOneLds {
    uint32_t key[SORT_ONE];
    uint16_t val[SORT_ONE];
    uint32_t wcount[16 * 512];
    uint64_t wt[16];
};
template <uint32_t RB, bool FIRST, bool LAST>
__device__ __forceinline__ void sort_one_pass(const uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n, uint32_t shift, SortOneLds& L) {
    constexpr uint32_t NB = 1u << RB, ROUNDS = SORT_ONE / 1024u;
    static_assert(NB * 16u % 1024u == 0 && NB <= 512u, "whole (digit, wavefront) slices per thread");
    const uint32_t tid = threadIdx.x, ln = tid & 63u, wv = tid >> 6;
    const uint32_t chunk = ((n + 1023u) / 1024u) * 64u;  // keys per wavefront: whole rounds
    for (uint32_t e = tid; e < 16u * NB; e += 1024u) L.wcount[e] = 0;
    uint32_t key[ROUNDS], val[ROUNDS], rank[ROUNDS];
    uint32_t* mine_cnt = L.wcount + wv * NB;
#pragma unroll
    for (uint32_t r = 0; r < ROUNDS; ++r) {
        const uint32_t i = wv * chunk + r * 64u + ln;
        const bool live = r * 64u < chunk && i < n;
        key[r] = live ? (FIRST ? keys_in[i] : L.key[i]) : 0u;
        val[r] = live ? (FIRST ? i : (uint32_t)L.val[i]) : 0u;
    }
    __syncthreads();  // the counters are zero; every key and value of the previous pass is in registers
#pragma unroll
    for (uint32_t r = 0; r < ROUNDS; ++r) {
        if (r * 64u >= chunk) break;  // workgroup uniform
        const uint32_t i = wv * chunk + r * 64u + ln;
        const bool live = i < n;
        const uint32_t d = (key[r] >> shift) & (NB - 1u);
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (uint32_t b = 0; b < RB; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = (uint32_t)__popcll(peers & ((1ull << ln) - 1ull));
        rank[r] = live ? mine_cnt[d] + before : 0u;  // the wavefront's keys of this digit in earlier rounds + the peers in front
        __builtin_amdgcn_wave_barrier();             // every peer has read the count
        if (live && before == 0u) mine_cnt[d] += (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // first slots: exclusive scan over (digit major, wavefront minor); thread t owns PER consecutive cells of that order
        constexpr uint32_t PER = NB * 16u / 1024u;
        uint32_t c[PER];
        uint64_t sum = 0;
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) {
            const uint32_t cell = tid * PER + q, d = cell >> 4, w = cell & 15u;
            c[q] = L.wcount[w * NB + d];
            sum += c[q];
        }
        uint64_t tot;
        uint32_t run = (uint32_t)block_exscan64(sum, L.wt, &tot);
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) {
            const uint32_t cell = tid * PER + q, d = cell >> 4, w = cell & 15u;
            L.wcount[w * NB + d] = run;
            run += c[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < ROUNDS; ++r) {
        if (r * 64u >= chunk) break;
        const uint32_t i = wv * chunk + r * 64u + ln;
        if (i < n) {
            const uint32_t pos = mine_cnt[(key[r] >> shift) & (NB - 1u)] + rank[r];
            L.key[pos] = key[r];
            if (LAST) {
                keys_out[pos] = key[r];
                vals_out[pos] = val[r];
            } else {
                L.val[pos] = (uint16_t)val[r];
            }
        }
    }
    __syncthreads();  // the pass' keys (and values) are in LDS for the next pass / the caller
}
// stable sort of (keys_in[i], i) by the low `bits` bits (at most 27) into (keys_out, vals_out) by the calling workgroup of 1024; the
// sorted keys are in L.key as well when it returns
__device__ __forceinline__ void sort_one_body(const uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n, uint32_t bits, SortOneLds& L) {
    if (bits <= 9u) {
        sort_one_pass<9, true, true>(keys_in, keys_out, vals_out, n, 0u, L);
    } else if (bits <= 18u) {
        sort_one_pass<9, true, false>(keys_in, keys_out, vals_out, n, 0u, L);
        sort_one_pass<9, false, true>(keys_in, keys_out, vals_out, n, 9u, L);
    } else {
        sort_one_pass<9, true, false>(keys_in, keys_out, vals_out, n, 0u, L);
        sort_one_pass<9, false, false>(keys_in, keys_out, vals_out, n, 9u, L);
        sort_one_pass<9, false, true>(keys_in, keys_out, vals_out, n, 18u, L);
    }
}

inline uint32_t rs_tiles(uint32_t n) { return (n + RS_TILE - 1) / RS_TILE; }
constexpr uint32_t RS_SELF_OFFSETS = 512;  // tiles up to which a scatter workgroup derives its offsets itself
// scratch: the digit-major histogram [512][tiles] + what its scan needs + one ping-pong pair of n keys and n values
inline size_t sort_scratch_bytes(uint32_t n) {
    const size_t hist = (size_t)512 * rs_tiles(n) * 4;
    return ((hist + 255) & ~(size_t)255) + ((scan_scratch_bytes((uint64_t)512 * rs_tiles(n)) + 255) & ~(size_t)255) + (size_t)n * 8 + 256;
}
template <uint32_t RB>
inline hipError_t sort_pass(const uint32_t* sk, const uint32_t* sv, uint32_t* dk, uint32_t* dv, uint32_t n, uint32_t shift, uint32_t* hist,
                            void* scan_tmp, hipStream_t st) {
    const uint32_t tiles = rs_tiles(n);
    if (tiles <= RS_SELF_OFFSETS) {
        hipLaunchKernelGGL((k_rs_hist<RB, true>), dim3(tiles), dim3(256), 0, st, sk, n, shift, tiles, hist);
        hipLaunchKernelGGL((k_rs_scatter<RB, false>), dim3(tiles), dim3(256), 0, st, sk, sv, n, shift, tiles, hist, dk, dv);
    } else {
        hipLaunchKernelGGL((k_rs_hist<RB, false>), dim3(tiles), dim3(256), 0, st, sk, n, shift, tiles, hist);
        hipError_t e = exclusive_scan<uint32_t>(hist, hist, (1u << RB) * tiles, scan_tmp, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_rs_scatter<RB, true>), dim3(tiles), dim3(256), 0, st, sk, sv, n, shift, tiles, hist, dk, dv);
    }
    return hipGetLastError();
}
// stable sort of (keys_in[i], vals_in[i]) by the low `bits` bits of the key into (keys_out, vals_out); inputs are not modified
inline hipError_t sort_pairs(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n,
                             uint32_t bits, void* scratch, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const uint32_t tiles = rs_tiles(n);
    const bool nine = bits > 24u && bits <= 27u;  // three passes of 9 instead of four of 8
    const uint32_t rb = nine ? 9u : 8u, passes = bits == 0 ? 1u : (bits + rb - 1u) / rb;
    unsigned char* sp = reinterpret_cast<unsigned char*>(scratch);
    uint32_t* hist = reinterpret_cast<uint32_t*>(sp);
    sp += ((size_t)512 * tiles * 4 + 255) & ~(size_t)255;
    void* scan_tmp = sp;
    sp += (scan_scratch_bytes((uint64_t)512 * tiles) + 255) & ~(size_t)255;
    uint32_t* tk = reinterpret_cast<uint32_t*>(sp);
    uint32_t* tv = tk + n;
    const uint32_t* sk = keys_in;
    const uint32_t* sv = vals_in;
    for (uint32_t p = 0; p < passes; ++p) {
        // the last pass lands in (keys_out, vals_out): destinations alternate backwards from there
        const bool to_out = ((passes - 1u - p) & 1u) == 0u;
        uint32_t* dk = to_out ? keys_out : tk;
        uint32_t* dv = to_out ? vals_out : tv;
        const hipError_t e = nine ? sort_pass<9>(sk, sv, dk, dv, n, p * rb, hist, scan_tmp, st) : sort_pass<8>(sk, sv, dk, dv, n, p * rb, hist, scan_tmp, st);
        if (e != hipSuccess) return e;
        sk = dk;
        sv = dv;
    }
    return hipGetLastError();
}

// ---- run-length encoding of sorted keys ----------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_rle_heads(const uint32_t* keys, uint32_t n, uint32_t* flag) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) flag[i] = (i == 0u || keys[i] != keys[i - 1u]) ? 1u : 0u;
}
// seg = exclusive scan of the head flags; a head writes its key and its start
static __global__ __launch_bounds__(256) void k_rle_write(const uint32_t* keys, const uint32_t* seg, uint32_t n, uint32_t* uniq, uint32_t* starts) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (i == 0u || keys[i] != keys[i - 1u]) {
        uniq[seg[i]] = keys[i];
        starts[seg[i]] = i;
    }
}
static __global__ __launch_bounds__(256) void k_rle_counts(const uint32_t* starts, const uint32_t* n_runs, uint32_t n, uint32_t* counts) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x, runs = *n_runs;
    if (g < runs) counts[g] = (g + 1u < runs ? starts[g + 1u] : n) - starts[g];
}
// ---- the same, with the head flags inside the scan (no flag array, no separate heads / write launches) ----------------
// a thread's four consecutive keys and which of them start a run
__device__ __forceinline__ uint32_t rle_heads4(const uint32_t* keys, uint32_t n, uint32_t base, uint32_t (&key)[4], bool (&head)[4]) {
    uint32_t prev = base > 0u && base - 1u < n ? keys[base - 1u] : 0u, heads = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t i = base + k;
        key[k] = i < n ? keys[i] : 0u;
        head[k] = i < n && (i == 0u || key[k] != prev);
        heads += head[k] ? 1u : 0u;
        prev = key[k];
    }
    return heads;
}
static __global__ __launch_bounds__(256) void k_rle_sums(const uint32_t* keys, uint32_t n, uint64_t* sums) {
    __shared__ uint64_t wt[4];
    uint32_t key[4];
    bool head[4];
    const uint64_t s = rle_heads4(keys, n, blockIdx.x * SCAN_TILE + threadIdx.x * 4u, key, head);
    uint64_t tot;
    (void)block_exscan64(s, wt, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
// bases = the exclusive scan of the tile sums; a head writes its key and its start at its rank
static __global__ __launch_bounds__(256) void k_rle_tiles(const uint32_t* keys, uint32_t n, const uint64_t* bases, uint32_t* uniq, uint32_t* starts,
                                                          uint32_t* n_runs) {
    __shared__ uint64_t wt[4];
    uint32_t key[4];
    bool head[4];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 4u;
    const uint64_t s = rle_heads4(keys, n, base, key, head);
    uint64_t tot;
    uint32_t run = (uint32_t)(bases[blockIdx.x] + block_exscan64(s, wt, &tot));
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k)
        if (head[k]) {
            uniq[run] = key[k];
            starts[run] = base + k;
            run += 1u;
        }
    if (blockIdx.x == gridDim.x - 1u && threadIdx.x == 255u) *n_runs = run;  // the last thread ends at the number of runs
}
// at most SCAN_ONE keys: one workgroup does all of it, counts included, in one launch
__device__ __forceinline__ void rle_one_body(const uint32_t* keys, uint32_t n, uint32_t* uniq, uint32_t* starts, uint32_t* counts, uint32_t* n_runs,
                                             uint64_t* wt /* [blockDim.x / 64] words of LDS */) {
    const uint32_t per = (n + blockDim.x - 1u) / blockDim.x, lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += (i == 0u || keys[i] != keys[i - 1u]) ? 1u : 0u;
    uint64_t tot;
    uint32_t run = (uint32_t)block_exscan64(s, wt, &tot);
    for (uint32_t i = lo; i < hi; ++i)
        if (i == 0u || keys[i] != keys[i - 1u]) {
            uniq[run] = keys[i];
            starts[run] = i;
            run += 1u;
        }
    if (threadIdx.x == 0) *n_runs = (uint32_t)tot;
    if (!counts) return;
    __syncthreads();  // every start of the workgroup is written
    const uint32_t runs = (uint32_t)tot;
    for (uint32_t g = threadIdx.x; g < runs; g += blockDim.x) counts[g] = (g + 1u < runs ? starts[g + 1u] : n) - starts[g];
}
static __global__ __launch_bounds__(1024) void k_rle_one(const uint32_t* keys, uint32_t n, uint32_t* uniq, uint32_t* starts, uint32_t* counts,
                                                         uint32_t* n_runs) {
    __shared__ uint64_t wt[16];
    rle_one_body(keys, n, uniq, starts, counts, n_runs, wt);
}

// sorted keys -> uniq[r], starts[r] (= the exclusive scan of counts), counts[r] for r < *n_runs; work = n words of scratch.
// counts may be NULL (a caller that derives them itself).  Three launches (one for at most SCAN_ONE keys) + one for the counts;
// more than SCAN_ONE tiles: the general form (flag array: heads, scan, write, counts).
inline hipError_t run_length_encode(const uint32_t* keys, uint32_t n, uint32_t* uniq, uint32_t* starts, uint32_t* counts, uint32_t* n_runs,
                                    uint32_t* work, void* scan_tmp, hipStream_t st) {
    if (n == 0) return hipMemsetAsync(n_runs, 0, 4, st);
    const dim3 grid((n + 255u) / 256u), block(256);
    const uint32_t tiles = (n + SCAN_TILE - 1u) / SCAN_TILE;
    if (n <= SCAN_ONE) {
        hipLaunchKernelGGL(k_rle_one, dim3(1), dim3(ONE_THREADS), 0, st, keys, n, uniq, starts, counts, n_runs);
        return hipGetLastError();
    }
    if (tiles <= SCAN_ONE) {
        uint64_t* sums = reinterpret_cast<uint64_t*>(scan_tmp);  // scan_scratch_bytes(n) holds the tile sums
        hipLaunchKernelGGL(k_rle_sums, dim3(tiles), block, 0, st, keys, n, sums);
        hipLaunchKernelGGL(k_scan_one64, dim3(1), block, 0, st, sums, tiles);
        hipLaunchKernelGGL(k_rle_tiles, dim3(tiles), block, 0, st, keys, n, sums, uniq, starts, n_runs);
        if (counts) hipLaunchKernelGGL(k_rle_counts, grid, block, 0, st, starts, n_runs, n, counts);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_rle_heads, grid, block, 0, st, keys, n, work);
    hipError_t e = exclusive_scan<uint32_t>(work, work, n, scan_tmp, st, n_runs);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_rle_write, grid, block, 0, st, keys, work, n, uniq, starts);
    if (counts) hipLaunchKernelGGL(k_rle_counts, grid, block, 0, st, starts, n_runs, n, counts);
    return hipGetLastError();
}

}  // namespace ss
}  // namespace rp

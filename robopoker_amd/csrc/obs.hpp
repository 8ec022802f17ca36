// obs.hpp — Observation / Isomorphism helpers shared by the abstraction inputs (deuce.hip) and the NLHE encoder (nlmc.hip):
// the i64 form of an observation, the canonical suit permutation, the lookup tables' monotone search key.
// Reference: crates/deuce/src/{observation.rs:132-165, permutation.rs:9-71, isomorphism.rs:8-45}, crates/lloyd/src/lookup.rs:23-25.
#ifndef RP_OBS_HPP
#define RP_OBS_HPP

#include <hip/hip_runtime.h>

#include "cards.hpp"

namespace rp {

// From<Observation> for i64 (observation.rs:132-141): public then pocket, ascending, first card most significant
__device__ __forceinline__ int64_t obs_encode(uint64_t pocket, uint64_t public_) {
    uint64_t acc = 0;
    for (uint64_t h = public_; h; h &= h - 1) acc = acc << 8 | (uint64_t)(1 + __builtin_ctzll(h));
    for (uint64_t h = pocket; h; h &= h - 1) acc = acc << 8 | (uint64_t)(1 + __builtin_ctzll(h));
    return (int64_t)acc;
}

// ---------------------------------------------------------------------------------------------------------------
// Permutation::from(&Observation) (permutation.rs:9-21,45-60).  A suit's sort key packs, most significant first:
// pocket size, public size, pocket min rank, public min rank, pocket max rank, public max rank (None < Some: +1),
// suit.  Keys are distinct (the suit breaks ties), so sorted position = number of smaller keys.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t suit_key(uint64_t pocket, uint64_t public_, uint32_t s) {
    const uint64_t po = pocket & (SUIT0 << s), pu = public_ & (SUIT0 << s);
    const uint32_t lo_po = po ? (uint32_t)__builtin_ctzll(po) / 4u + 1u : 0u, hi_po = po ? (63u - (uint32_t)__builtin_clzll(po)) / 4u + 1u : 0u;
    const uint32_t lo_pu = pu ? (uint32_t)__builtin_ctzll(pu) / 4u + 1u : 0u, hi_pu = pu ? (63u - (uint32_t)__builtin_clzll(pu)) / 4u + 1u : 0u;
    return ((uint32_t)__popcll(po) << 21) | ((uint32_t)__popcll(pu) << 18) | (lo_po << 14) | (lo_pu << 10) | (hi_po << 6) | (hi_pu << 2) | s;
}
// Isomorphism::is_canonical (isomorphism.rs:41-45): the permutation is the identity iff the suits are already sorted
__device__ __forceinline__ bool is_canonical(uint64_t pocket, uint64_t public_) {
    const uint32_t k0 = suit_key(pocket, public_, 0), k1 = suit_key(pocket, public_, 1), k2 = suit_key(pocket, public_, 2),
                   k3 = suit_key(pocket, public_, 3);
    return k0 < k1 && k1 < k2 && k2 < k3;
}
// Isomorphism::from(Observation) (isomorphism.rs:8-14) with Permutation::image / shift (permutation.rs:27-32,61-71)
__device__ __forceinline__ void canonical(uint64_t pocket, uint64_t public_, uint64_t* opocket, uint64_t* opublic) {
    uint32_t k[4];
#pragma unroll
    for (uint32_t s = 0; s < 4; ++s) k[s] = suit_key(pocket, public_, s);
    uint64_t po = 0, pu = 0;
#pragma unroll
    for (uint32_t s = 0; s < 4; ++s) {
        const uint32_t to = (k[0] < k[s]) + (k[1] < k[s]) + (k[2] < k[s]) + (k[3] < k[s]);  // suit s is renamed to `to`
        po |= ((pocket >> s) & SUIT0) << to;
        pu |= ((public_ >> s) & SUIT0) << to;
    }
    *opocket = po, *opublic = pu;
}

__device__ __forceinline__ uint32_t pocket_number(uint64_t pocket) {
    const uint32_t c1 = (uint32_t)__builtin_ctzll(pocket), c2 = 63u - (uint32_t)__builtin_clzll(pocket);
    return c2 * (c2 - 1) / 2 + c1;
}
// the table's search key: monotone in IsomorphismIterator order
__device__ __forceinline__ uint64_t search_key(uint64_t pocket, uint64_t public_) { return ((uint64_t)pocket_number(pocket) << 52) | public_; }

__device__ __forceinline__ int64_t table_find(const uint64_t* keys, uint64_t n, uint64_t key) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        const uint64_t k = keys[mid];
        if (k == key) return (int64_t)mid;
        if (k < key) lo = mid + 1;
        else hi = mid;
    }
    return -1;
}

}  // namespace rp

#endif

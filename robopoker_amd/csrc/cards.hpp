// cards.hpp — card sets and hand strength on the device, shared by deuce.hip (river equity) and nlhe.hip (showdowns).
// Reference: crates/deuce/src/{card,hand,evaluator,strength,ranking,kicks}.rs.
#ifndef RP_CARDS_HPP
#define RP_CARDS_HPP

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rp {

// ---------------------------------------------------------------------------------------------------------------
// cards.  card = rank * 4 + suit (card.rs:16-20); Hand = u64 bit set of cards (hand.rs:7).
// Suit words: SW = four 16-bit fields of one u64, field s = the 13-bit rank set of suit s.
// ---------------------------------------------------------------------------------------------------------------
#define HAND_MASK 0x000FFFFFFFFFFFFFull
#define SUIT0 0x0001111111111111ull

__device__ __forceinline__ uint64_t sw_of_card(uint32_t c) { return 1ull << (16u * (c & 3u) + (c >> 2)); }
__device__ __forceinline__ uint64_t sw_of_hand(uint64_t h) {
    uint64_t t = 0;
    while (h) {
        t |= sw_of_card((uint32_t)__builtin_ctzll(h));
        h &= h - 1;
    }
    return t;
}
__device__ __forceinline__ uint32_t msb32(uint32_t x) { return 31u - (uint32_t)__clz((int)x); }
// keep the n highest bits of m (Evaluator::find_kickers drops the lowest until n remain, evaluator.rs:55-72)
__device__ __forceinline__ uint32_t top_n(uint32_t m, int n) {
#pragma unroll
    for (int i = 0; i < 6; ++i) m = __popc(m) > n ? (m & (m - 1)) : m;  // a hand shows at most 7 ranks
    return m;
}
// Evaluator::find_rank_of_straight (evaluator.rs:122-137); 0x100F = A2345, ranked Five (3)
__device__ __forceinline__ int straight_of(uint32_t r) {
    const uint32_t b = r & (r << 1) & (r << 2) & (r << 3) & (r << 4);
    if (b) return (int)msb32(b);
    return (r & 0x100Fu) == 0x100Fu ? 3 : -1;
}
#define KEY(variant, r1, r2, kicks) (((uint32_t)(variant) << 21) | ((uint32_t)(r1) << 17) | ((uint32_t)(r2) << 13) | (uint32_t)(kicks))
// Strength::from(Hand) (strength.rs:18-24) = find_ranking (evaluator.rs:38-50) + find_kickers, as an order key:
// variant (ranking.rs:17-29, default build: HighCard < OnePair < TwoPair < ThreeOAK < Straight < FullHouse < Flush <
// FourOAK < StraightFlush), then the ranking's rank(s), then the kicker set — the fields of the derived Ord, in order.
__device__ __forceinline__ uint32_t strength_key(uint64_t sw) {
    const uint32_t s0 = (uint32_t)sw & 0xffffu, s1 = (uint32_t)sw >> 16, s2 = (uint32_t)(sw >> 32) & 0xffffu, s3 = (uint32_t)(sw >> 48);
    const uint32_t any = s0 | s1 | s2 | s3;
    // per-rank multiplicity, bit-sliced
    const uint32_t x = s0 ^ s1, c01 = s0 & s1, y = s2 ^ s3, c23 = s2 & s3;
    const uint32_t quads = c01 & c23;
    const uint32_t ge2 = c01 | c23 | (x & y);
    const uint32_t ge3 = ((x ^ y) & (c01 | c23)) | quads;
    // find_suit_of_flush (evaluator.rs:144-152): first suit with >= 5 cards (at most one suit of <= 9 cards can)
    const uint32_t fl = __popc(s0) >= 5 ? s0 : (__popc(s1) >= 5 ? s1 : (__popc(s2) >= 5 ? s2 : (__popc(s3) >= 5 ? s3 : 0u)));
    if (fl) {
        const int sf = straight_of(fl);
        if (sf >= 0) return KEY(8, sf, 0, 0);
    }
    if (quads) {
        const uint32_t r = msb32(quads);
        return KEY(7, r, 0, top_n(any & ~(1u << r), 1));
    }
    if (ge3) {
        const uint32_t t = msb32(ge3), rest = ge2 & ~(1u << t);
        if (rest) return KEY(5, t, msb32(rest), 0);
    }
    if (fl) return KEY(6, msb32(fl), 0, 0);  // evaluator.rs:109-115: the flush's top card only, no kickers
    {
        const int st = straight_of(any);
        if (st >= 0) return KEY(4, st, 0, 0);
    }
    if (ge3) {
        const uint32_t t = msb32(ge3);
        return KEY(3, t, 0, top_n(any & ~(1u << t), 2));
    }
    if (ge2) {
        const uint32_t hi = msb32(ge2), rest = ge2 & ~(1u << hi);
        if (rest) {
            const uint32_t lo = msb32(rest);
            return KEY(2, hi, lo, top_n(any & ~((1u << hi) | (1u << lo)), 1));
        }
        return KEY(1, hi, 0, top_n(any & ~(1u << hi), 3));
    }
    const uint32_t hi = msb32(any);
    return KEY(0, hi, 0, top_n(any & ~(1u << hi), 4));
}


}  // namespace rp

#endif

/* CPU oracle for the abstraction inputs: cards, hand strength, river equity, suit isomorphism,
 * the observation / isomorphism iterators and the histogram projection.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (robopoker_amd/, include/) includes, links or calls this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * A plain-C restatement of the reference's algorithms, written from their behaviour (the reference is Rust and
 * cannot be built here).  Every function cites the reference file:line it follows.  It is pinned by the reference's
 * own known-answer tests (tests/test_oracle_deuce.py): the evaluator's 20 hands (deuce/src/evaluator.rs:176-372),
 * the permutation / isomorphism identities (permutation.rs:160-259, isomorphism.rs:55-222), the hand iterator's
 * sequences (hand_iter.rs:85-170) and the isomorphism counts 169 / 1 286 792 / 13 960 050 / 123 156 254
 * (street.rs:120-127).
 *
 * Encoding (card.rs:16-20,41-45; hand.rs:7): card = rank * 4 + suit, rank 0 = Two .. 12 = Ace, suit 0 = c, 1 = d,
 * 2 = h, 3 = s; a Hand is the u64 bit set of its cards.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORA_API __attribute__((visibility("default")))

#define HAND_MASK 0x000FFFFFFFFFFFFFull /* hand.rs:86-88 */
static const uint64_t SUIT_BITS[4] = {0x0001111111111111ull, 0x0002222222222222ull, 0x0004444444444444ull,
                                      0x0008888888888888ull}; /* suit.rs:43-52 */

static int popc64(uint64_t x) { return __builtin_popcountll(x); }

/* Hand::ranks (hand.rs:62-84): which ranks are present, as a 13-bit set */
static uint16_t hand_ranks(uint64_t h) {
    uint16_t r = 0;
    for (int k = 0; k < 13; ++k)
        if (h & (0xFull << (4 * k))) r |= (uint16_t)(1u << k);
    return r;
}
static int msb16(uint16_t x) { return 31 - __builtin_clz((unsigned)x); } /* Rank::from(u16) rank.rs:62-67 */

/* Ranking variants in the default build's declaration order = their derived Ord (ranking.rs:17-29) */
enum { HIGH_CARD = 0, ONE_PAIR, TWO_PAIR, THREE_OAK, STRAIGHT, FULL_HOUSE, FLUSH, FOUR_OAK, STRAIGHT_FLUSH };

typedef struct {
    int variant, r1, r2;
    uint16_t kicks;
} strength_t;

/* Evaluator::find_rank_of_n_oak_skip (evaluator.rs:153-173): highest rank held at least n times, skipping one */
static int rank_of_n_oak(uint64_t hand, int n, int skip) {
    for (int r = 12; r >= 0; --r) {
        if (r == skip) continue;
        if (popc64(hand & (0xFull << (4 * r))) >= n) return r;
    }
    return -1;
}
/* Evaluator::find_rank_of_straight (evaluator.rs:122-137); WHEEL = A2345, ranked Five */
static int rank_of_straight(uint16_t ranks) {
    uint16_t bits = ranks;
    for (int i = 0; i < 4; ++i) bits &= (uint16_t)(bits << 1);
    if (bits) return msb16(bits);
    if ((ranks & 0x100F) == 0x100F) return 3;
    return -1;
}
/* Evaluator::find_suit_of_flush (evaluator.rs:144-152): the first suit with five or more cards */
static int suit_of_flush(uint64_t hand) {
    for (int s = 0; s < 4; ++s)
        if (popc64(hand & SUIT_BITS[s]) >= 5) return s;
    return -1;
}
/* Ranking::n_kickers / Ranking::mask (ranking.rs:32-51) + Evaluator::find_kickers (evaluator.rs:55-72) */
static uint16_t kickers_of(uint64_t hand, int variant, int r1, int r2) {
    int n;
    uint16_t mask;
    switch (variant) {
        case HIGH_CARD: n = 4; mask = (uint16_t)~(1u << r1); break;
        case ONE_PAIR: n = 3; mask = (uint16_t)~(1u << r1); break;
        case THREE_OAK: n = 2; mask = (uint16_t)~(1u << r1); break;
        case FOUR_OAK: n = 1; mask = (uint16_t)~(1u << r1); break;
        case TWO_PAIR: n = 1; mask = (uint16_t)~((1u << r1) | (1u << r2)); break;
        default: return 0;
    }
    uint16_t rank = hand_ranks(hand) & mask;
    while (n < __builtin_popcount(rank)) rank &= (uint16_t)(rank - 1); /* drop the lowest */
    return rank;
}
/* Evaluator::find_ranking (evaluator.rs:38-50): strongest first, first match wins */
static strength_t strength_of(uint64_t hand) {
    strength_t s = {HIGH_CARD, 0, 0, 0};
    int r, q;
    const int fs = suit_of_flush(hand);
    if (fs >= 0 && (r = rank_of_straight(hand_ranks(hand & SUIT_BITS[fs]))) >= 0) {
        s.variant = STRAIGHT_FLUSH, s.r1 = r;
    } else if ((r = rank_of_n_oak(hand, 4, -1)) >= 0) {
        s.variant = FOUR_OAK, s.r1 = r;
    } else if ((r = rank_of_n_oak(hand, 3, -1)) >= 0 && (q = rank_of_n_oak(hand, 2, r)) >= 0) {
        s.variant = FULL_HOUSE, s.r1 = r, s.r2 = q;
    } else if (fs >= 0) {
        s.variant = FLUSH, s.r1 = msb16(hand_ranks(hand & SUIT_BITS[fs])); /* evaluator.rs:109-115: top card only */
    } else if ((r = rank_of_straight(hand_ranks(hand))) >= 0) {
        s.variant = STRAIGHT, s.r1 = r;
    } else if ((r = rank_of_n_oak(hand, 3, -1)) >= 0) {
        s.variant = THREE_OAK, s.r1 = r;
    } else if ((r = rank_of_n_oak(hand, 2, -1)) >= 0) {
        if ((q = rank_of_n_oak(hand, 2, r)) >= 0) s.variant = TWO_PAIR, s.r1 = r, s.r2 = q;
        else s.variant = ONE_PAIR, s.r1 = r;
    } else {
        s.variant = HIGH_CARD, s.r1 = rank_of_n_oak(hand, 1, -1);
    }
    s.kicks = kickers_of(hand, s.variant, s.r1, s.r2);
    return s;
}
/* Strength's derived Ord (strength.rs:6-10): ranking (variant, then its ranks), then kickers as a u16 */
static int strength_cmp(strength_t a, strength_t b) {
    if (a.variant != b.variant) return a.variant < b.variant ? -1 : 1;
    if (a.r1 != b.r1) return a.r1 < b.r1 ? -1 : 1;
    if (a.r2 != b.r2) return a.r2 < b.r2 ? -1 : 1;
    if (a.kicks != b.kicks) return a.kicks < b.kicks ? -1 : 1;
    return 0;
}
/* the same order as one integer (the boundary's key, include/rp_mi355x.h rp_hand_strength) */
ORA_API uint32_t ora_strength_key(uint64_t hand) {
    const strength_t s = strength_of(hand & HAND_MASK);
    return ((uint32_t)s.variant << 21) | ((uint32_t)s.r1 << 17) | ((uint32_t)s.r2 << 13) | s.kicks;
}
ORA_API void ora_strength(uint64_t hand, int32_t* variant, int32_t* r1, int32_t* r2, uint32_t* kicks) {
    const strength_t s = strength_of(hand & HAND_MASK);
    *variant = s.variant, *r1 = s.r1, *r2 = s.r2, *kicks = s.kicks;
}

/* ---- HandIterator (hand_iter.rs:3-81) ---------------------------------------------------------------------- */
static int hi_exhausted(uint64_t next) { return next == 0 || (next >> 52) != 0; } /* hand_iter.rs:15-17 */
static uint64_t hi_permute(uint64_t x) {                                          /* hand_iter.rs:18-28 */
    const uint64_t a = x | (x - 1), b = a + 1, c = ~a, d = c & b, e = d - 1;
    return b | (e >> (1 + __builtin_ctzll(x)));
}
ORA_API uint64_t ora_hand_iter_first(uint32_t n, uint64_t mask) { /* hand_iter.rs:65-78 */
    uint64_t next = n ? (1ull << n) - 1 : 0;
    while ((next & mask) && !hi_exhausted(next)) next = hi_permute(next);
    return hi_exhausted(next) ? 0 : next;
}
/* hand_iter.rs:34-41,47-58: the hand after `cur`; 0 once exhausted */
ORA_API uint64_t ora_hand_iter_next(uint64_t cur, uint64_t mask) {
    uint64_t next = cur;
    do next = hi_permute(next);
    while (next & mask);
    return hi_exhausted(next) ? 0 : next;
}

/* ---- Observation (observation.rs) ---------------------------------------------------------------------------- */
/* From<Observation> for i64 (observation.rs:132-141): public cards then pocket cards, each ascending, one byte
 * (card + 1) per card, the first in the most significant position */
ORA_API int64_t ora_obs_to_i64(uint64_t pocket, uint64_t public_) {
    uint64_t acc = 0;
    for (uint64_t h = public_; h; h &= h - 1) acc = acc << 8 | (uint64_t)(1 + __builtin_ctzll(h));
    for (uint64_t h = pocket; h; h &= h - 1) acc = acc << 8 | (uint64_t)(1 + __builtin_ctzll(h));
    return (int64_t)acc;
}
/* From<i64> for Observation (observation.rs:144-165): the two lowest bytes are the pocket */
ORA_API void ora_obs_from_i64(int64_t bits, uint64_t* pocket, uint64_t* public_) {
    *pocket = 0, *public_ = 0;
    for (int i = 0; i < 8; ++i) {
        const int64_t b = bits >> (8 * i);
        if (b <= 0) break;
        const uint64_t card = 1ull << (((uint64_t)b & 0xff) - 1);
        if (i < 2) *pocket |= card;
        else *public_ |= card;
    }
}
/* Observation::equity (observation.rs:45-63): wins / (wins + losses) over the 990 opposing holes, ties dropped */
ORA_API float ora_river_equity(uint64_t pocket, uint64_t public_, uint32_t* won_out, uint32_t* sum_out) {
    const strength_t hero = strength_of(pocket | public_);
    const uint64_t mask = pocket | public_;
    uint32_t won = 0, sum = 0;
    for (uint64_t hole = ora_hand_iter_first(2, mask); hole; hole = ora_hand_iter_next(hole, mask)) {
        const int c = strength_cmp(hero, strength_of(hole | public_));
        if (c > 0) won += 1, sum += 1;
        else if (c < 0) sum += 1;
    }
    if (won_out) *won_out = won;
    if (sum_out) *sum_out = sum;
    return sum == 0 ? 0.5f : (float)won / (float)sum;
}
/* Abstraction::quantize (kicker/src/abstraction.rs:61-63): round(p * N), N = KMEANS_EQTY_CLUSTER_COUNT - 1 = 100 */
ORA_API uint32_t ora_quantize(float p) { return (uint32_t)roundf(p * 100.0f); }

/* ---- Permutation / Isomorphism (permutation.rs, isomorphism.rs) ------------------------------------------------ */
typedef struct {
    int suit, k[6];
} colex_t;
static int lo_rank(uint64_t h) { return h ? __builtin_ctzll(h) / 4 : -1; }       /* Hand::min_rank, None < Some */
static int hi_rank(uint64_t h) { return h ? (63 - __builtin_clzll(h)) / 4 : -1; } /* Hand::max_rank */
/* Permutation::order (permutation.rs:45-54) */
static int colex_cmp(const colex_t* a, const colex_t* b) {
    for (int i = 0; i < 6; ++i)
        if (a->k[i] != b->k[i]) return a->k[i] < b->k[i] ? -1 : 1;
    return a->suit < b->suit ? -1 : (a->suit > b->suit ? 1 : 0);
}
/* Permutation::from(&Observation) (permutation.rs:9-21): sort the suits by their colex key; the suit found at
 * sorted position i is renamed to suit i */
ORA_API void ora_permutation(uint64_t pocket, uint64_t public_, uint8_t perm[4]) {
    colex_t c[4];
    for (int s = 0; s < 4; ++s) {
        const uint64_t po = pocket & SUIT_BITS[s], pu = public_ & SUIT_BITS[s];
        c[s].suit = s;
        c[s].k[0] = popc64(po), c[s].k[1] = popc64(pu);
        c[s].k[2] = lo_rank(po), c[s].k[3] = lo_rank(pu);
        c[s].k[4] = hi_rank(po), c[s].k[5] = hi_rank(pu);
    }
    for (int i = 1; i < 4; ++i) /* insertion sort: the key is total (suit tiebreak), any sort gives the same */
        for (int j = i; j > 0 && colex_cmp(&c[j], &c[j - 1]) < 0; --j) {
            const colex_t t = c[j];
            c[j] = c[j - 1], c[j - 1] = t;
        }
    for (int i = 0; i < 4; ++i) perm[c[i].suit] = (uint8_t)i;
}
/* Permutation::image / shift (permutation.rs:27-32,61-71) */
static uint64_t perm_image(const uint8_t perm[4], uint64_t hand) {
    uint64_t out = 0;
    for (int s = 0; s < 4; ++s) {
        const uint64_t cards = hand & SUIT_BITS[s];
        const int shift = (int)perm[s] - s;
        out |= shift >= 0 ? cards << shift : cards >> -shift;
    }
    return out & HAND_MASK;
}
/* Isomorphism::from(Observation) (isomorphism.rs:8-14) */
ORA_API void ora_isomorphism(uint64_t pocket, uint64_t public_, uint64_t* opocket, uint64_t* opublic) {
    uint8_t perm[4];
    ora_permutation(pocket, public_, perm);
    *opocket = perm_image(perm, pocket);
    *opublic = perm_image(perm, public_);
}
ORA_API void ora_permute(const uint8_t perm[4], uint64_t hand, uint64_t* out) { *out = perm_image(perm, hand); }
/* Isomorphism::is_canonical (isomorphism.rs:41-45) */
ORA_API int ora_is_canonical(uint64_t pocket, uint64_t public_) {
    uint8_t perm[4];
    ora_permutation(pocket, public_, perm);
    return perm[0] == 0 && perm[1] == 1 && perm[2] == 2 && perm[3] == 3;
}

/* ---- ObservationIterator / IsomorphismIterator (observation_iter.rs:13-104, isomorphism_iter.rs:7-20) ----------- */
static const int N_BOARD[4] = {0, 3, 4, 5}; /* street.rs:67-74 */
/* Canonical observations of `street` whose pocket is the p-th two-card hand, p in [pocket_lo, pocket_hi), in the
 * iterator's order: pockets ascending as bit sets, then boards ascending as bit sets.  Writes up to `cap` i64 forms
 * into `out` (may be NULL) and returns how many there are. */
ORA_API uint64_t ora_isomorphisms(int street, uint32_t pocket_lo, uint32_t pocket_hi, int64_t* out, uint64_t cap) {
    uint64_t n = 0;
    uint32_t p = 0;
    for (uint64_t pocket = ora_hand_iter_first(2, 0); pocket; pocket = ora_hand_iter_next(pocket, 0), ++p) {
        if (p < pocket_lo) continue;
        if (p >= pocket_hi) break;
        if (street == 0) { /* observation_iter.rs:97-99: preflop has no board */
            if (ora_is_canonical(pocket, 0)) {
                if (out && n < cap) out[n] = ora_obs_to_i64(pocket, 0);
                ++n;
            }
            continue;
        }
        for (uint64_t board = ora_hand_iter_first((uint32_t)N_BOARD[street], pocket); board; board = ora_hand_iter_next(board, pocket)) {
            if (!ora_is_canonical(pocket, board)) continue;
            if (out && n < cap) out[n] = ora_obs_to_i64(pocket, board);
            ++n;
        }
    }
    return n;
}

/* ---- Lookup (lloyd/src/lookup.rs) ------------------------------------------------------------------------------ */
/* Lookup::lookup (lookup.rs:23-25) over a table in IsomorphismIterator order: binary search on (pocket, public) */
static int obs_cmp(uint64_t p0, uint64_t b0, uint64_t p1, uint64_t b1) {
    if (p0 != p1) return p0 < p1 ? -1 : 1;
    return b0 < b1 ? -1 : (b0 > b1 ? 1 : 0);
}
ORA_API int64_t ora_lookup_index(const int64_t* keys, uint64_t n, uint64_t pocket, uint64_t public_) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        uint64_t kp, kb;
        ora_obs_from_i64(keys[mid], &kp, &kb);
        const int c = obs_cmp(kp, kb, pocket, public_);
        if (c == 0) return (int64_t)mid;
        if (c < 0) lo = mid + 1;
        else hi = mid;
    }
    return -1;
}
/* Lookup::future (lookup.rs:35-45) + Observation::children (observation.rs:35-40) + Histogram::from(Vec<Abstraction>)
 * (histogram.rs:207-212): the histogram of the next street's abstractions over all ways to reveal the next card(s).
 * keys/abs: the next street's table.  Returns 0, or -1 if a child is missing from the table. */
ORA_API int ora_project(int64_t obs, const int64_t* keys, const uint8_t* abs_, uint64_t n, uint32_t bins, uint32_t* hist) {
    uint64_t pocket, public_;
    ora_obs_from_i64(obs, &pocket, &public_);
    const int ncards = popc64(pocket | public_);
    const uint32_t reveal = ncards == 2 ? 3u : 1u; /* street.rs:75-82 n_revealed of the next street */
    memset(hist, 0, bins * sizeof(uint32_t));
    const uint64_t mask = pocket | public_;
    for (uint64_t r = ora_hand_iter_first(reveal, mask); r; r = ora_hand_iter_next(r, mask)) {
        uint64_t cp, cb;
        ora_isomorphism(pocket, public_ | r, &cp, &cb);
        const int64_t at = ora_lookup_index(keys, n, cp, cb);
        if (at < 0 || abs_[at] >= bins) return -1;
        hist[abs_[at]] += 1;
    }
    return 0;
}
/* The turn layer's points straight from the definition: the river abstraction of a child is its quantised equity
 * (Lookup::grow, lookup.rs:172-178), so no table is needed. */
ORA_API void ora_project_river(int64_t turn_obs, uint32_t* hist101) {
    uint64_t pocket, public_;
    ora_obs_from_i64(turn_obs, &pocket, &public_);
    memset(hist101, 0, 101 * sizeof(uint32_t));
    const uint64_t mask = pocket | public_;
    for (uint64_t r = ora_hand_iter_first(1, mask); r; r = ora_hand_iter_next(r, mask))
        hist101[ora_quantize(ora_river_equity(pocket, public_ | r, NULL, NULL))] += 1;
}

/* rp_oracle_nlhe.h — types shared by the NLHE rules oracle (rp_oracle_nlhe.c) and the NLHE MCCFR oracle (rp_oracle_nlmc.c).
 * TEST INFRASTRUCTURE ONLY. */
#ifndef RP_ORACLE_NLHE_H
#define RP_ORACLE_NLHE_H
#include <stdint.h>

#define ORA_API __attribute__((visibility("default")))
#define MAXP 10
#define S_BLIND 1
#define B_BLIND 2

uint32_t ora_strength_key(uint64_t hand); /* oracle/rp_oracle_deuce.c */

enum { BETTING = 0, SHOVING = 1, FOLDING = 2 };                                      /* seat.rs:79-84 */
enum { A_DRAW = 0, A_FOLD, A_CALL, A_CHECK, A_RAISE, A_SHOVE, A_BLIND };             /* action.rs:8-16 */
enum { T_TERMINAL = -2, T_CHANCE = -1 };                                             /* turn.rs:2-6; >= 0: Choice(i) */

typedef struct {
    int32_t state;
    int16_t stack, stake, spent;
    uint64_t cards;
} ora_seat;
typedef struct {
    int32_t n;      /* P */
    int32_t dealer; /* game.rs:30-36 */
    int32_t ticker;
    int16_t pot;
    uint64_t board;
    ora_seat seats[MAXP];
} ora_game;
typedef struct {
    int32_t kind;
    int16_t chips;
    uint64_t cards;
} ora_action;

enum { E_DRAW = 1, E_FOLD = 2, E_CHECK = 3, E_CALL = 4, E_SHOVE = 5, E_OPEN0 = 6, E_RAISE0 = 10 }; /* edge codes (edge.rs:101-120) */
#define MAX_PATH_EDGES 12 /* lib.rs:73 */

/* rp_oracle_nlhe.c */
void ora_nlhe_from_start(ora_game* g, int n, int dealer, const int16_t* stacks, const uint64_t* holes);
int ora_nlhe_turn(const ora_game* g);
int ora_nlhe_street(const ora_game* g);
uint64_t ora_nlhe_choices(const ora_game* g, int depth);
int ora_nlhe_apply_edge(ora_game* g, uint8_t edge, const uint64_t* draws);
int ora_nlhe_payoff(const ora_game* g, int seat, float* out);
uint64_t ora_path_pack(const uint8_t* edges, int n);
int ora_path_unpack(uint64_t p, uint8_t* edges);
int ora_path_aggression(uint64_t p);
/* rp_oracle_deuce.c */
void ora_isomorphism(uint64_t pocket, uint64_t public_, uint64_t* opocket, uint64_t* opublic);
int64_t ora_obs_to_i64(uint64_t pocket, uint64_t public_);
int64_t ora_lookup_index(const int64_t* keys, uint64_t n, uint64_t pocket, uint64_t public_);

#endif

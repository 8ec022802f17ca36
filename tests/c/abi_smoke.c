/* A plain-C host of librp_mi355x.so: what a cgo / Rust-FFI / JNI binding sees.  Compiled with gcc by
 * tests/test_abi.py (the header must be valid C11) and run on the GPU box by tests/test_gpu_c_host.py.
 * Exit code 0 = ok, 2 = no device (the library refuses to compute: there is no CPU fallback), 1 = failure. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/rp_mi355x.h"

#define CHECK(call)                                                       \
    do {                                                                  \
        int _rc = (call);                                                 \
        if (_rc != RP_OK) {                                               \
            fprintf(stderr, "%s -> %d: %s\n", #call, _rc, rp_last_error()); \
            return _rc == RP_ERR_NO_DEVICE ? 2 : 1;                       \
        }                                                                 \
    } while (0)

int main(void) {
    printf("%s, %d device(s)\n", rp_version(), rp_device_count());
    rp_game* game = NULL;
    CHECK(rp_game_create(RP_GAME_KUHN, &game));
    rp_game_table view;
    CHECK(rp_game_view(game, &view));
    rp_hyper hp;
    rp_hyper_default(&hp);
    rp_mccfr* solver = NULL;
    /* Kuhn::default().solve(1 << 18) with the reference's schedules (kuhn/src/solver.rs:85-87) but a GPU-sized batch */
    CHECK(rp_mccfr_create(&view, RP_REGRET_FLOORED, RP_WEIGHT_LINEAR, RP_SAMPLING_EXTERNAL, 4096, &hp, 7, 0, &solver));
    CHECK(rp_mccfr_solve(solver, 1u << 20));
    float expl = 1.0f;
    CHECK(rp_mccfr_exploitability(solver, &expl));
    uint64_t epoch = 0, nodes = 0, infos = 0;
    CHECK(rp_mccfr_epoch(solver, &epoch));
    CHECK(rp_mccfr_counters(solver, &nodes, &infos));
    uint32_t id = 0, n = 0;
    CHECK(rp_game_info_id(game, "J|", &id));
    float avg[16];
    CHECK(rp_mccfr_policy(solver, id, RP_DIST_AVERAGED, avg, &n));
    printf("kuhn: epoch=%llu infos=%llu exploitability=%.5f  P(bet | J) = %.4f\n", (unsigned long long)epoch,
           (unsigned long long)infos, expl, n > 1 ? avg[1] : -1.0f);
    CHECK(rp_mccfr_destroy(solver));
    CHECK(rp_game_destroy(game));
    /* Kuhn's Nash has exploitability 0; the reference asserts < 0.005 after 2^18 epochs (kuhn/src/solver.rs tests) */
    return expl < 0.02f && epoch == 256 ? 0 : 1;
}

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
    if (!(expl < 0.02f && epoch == 256)) return 1;

    /* the blueprint trainer: Flagship = Nlhe<LinearRegret, LinearWeight, PluribusSampling> (nlhe/src/lib.rs:86-90) driven by
     * Trainer::train (forge/src/trainer.rs:18-66) for six steps of the reference's batch of 128 (nlhe/src/solver.rs:11) */
    rp_nlhe* nl = NULL;
    CHECK(rp_nlhe_create(0, 18, RP_REGRET_LINEAR, RP_WEIGHT_LINEAR, &hp, 11, 128, NULL, &nl));
    CHECK(rp_nlhe_set_sampling(nl, RP_SAMPLING_PLURIBUS));
    char summary[256];
    CHECK(rp_nlhe_train(nl, RP_UPDATE_COMPOSED, 6, 0.0, 1e9, 1e9, NULL, NULL, NULL, summary, sizeof summary));
    uint64_t nl_epoch = 0, nl_nodes = 0, nl_infos = 0, nl_keys = 0, n_rows = 0;
    CHECK(rp_nlhe_epoch(nl, &nl_epoch));
    CHECK(rp_nlhe_counters(nl, &nl_nodes, &nl_infos, &nl_keys));
    CHECK(rp_nlhe_export(nl, 0, &n_rows, NULL, NULL, NULL, NULL)); /* how many infosets the blueprint holds */
    printf("nlhe: epoch=%llu nodes=%llu infos=%llu infosets=%llu\n%s\n", (unsigned long long)nl_epoch, (unsigned long long)nl_nodes,
           (unsigned long long)nl_infos, (unsigned long long)n_rows, summary);
    CHECK(rp_nlhe_destroy(nl));
    /* 6 x 128 trees of a few hundred nodes and a few dozen walker infosets each; every infoset of the table was inserted once */
    return nl_epoch == 6 && nl_nodes > 6u * 128u * 100u && nl_infos > 6u * 128u * 10u && n_rows == nl_keys && n_rows > 1000 ? 0 : 1;
}

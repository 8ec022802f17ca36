/* rp_mi355x.h — C ABI of librp_mi355x.so: robopoker's two numeric hot paths on MI355X (gfx950).
 *
 * The reference (krukah/robopoker, Rust) has NO FFI seam for these paths: they sit behind
 * Rust traits with default methods.  This header IS the seam: every entry
 * point names the reference trait item it replaces so a thin Rust shim can implement
 * `mccfr::Solver` / `elkan::Elkan` / `monge::Coupling` by delegation (INTEGRATION.md shows
 * the binding).  Conventions kept from the reference: values are plain-old-data, the batch
 * (tree sampling + regret vectors) is pure w.r.t. the profile and the table mutation happens
 * afterwards in tree-id order; no Result on the hot path — here every call returns an
 * `rp_status` (0 = ok) and never unwinds across the boundary.
 *
 * All handles are opaque.  All buffers are caller-owned flat arrays.  `device` is a HIP
 * device ordinal; there is NO CPU fallback in this library: a call that needs the GPU on a
 * machine without one returns RP_ERR_NO_DEVICE.
 *
 * Arithmetic (exp/ln/pow, RNG, summation orders) is specified in rp_math.h and DESIGN.md.
 *
 * COMPILED LIMITS (a call outside them fails with RP_ERR_INVALID / RP_ERR_CAPACITY / RP_ERR_UNSUPPORTED, never silently):
 *   mccfr    max_actions <= 16 (RP_MAX_ACTIONS); a sampled tree has < 255 nodes (rp_game_table.max_tree_nodes) and each tree
 *            emits at most 255 Decisions.  Games with <= 62 nodes per sampled tree, depth <= 10, <= 32 internal nodes and
 *            <= 8191 infosets (Kuhn, Leduc, RPS, the wide Leduc) take the LDS-resident traversal; larger ones its HBM-scratch
 *            variant.  NLHE-sized trees (10^3-10^4 nodes) are NOT reachable through rp_game_table: see rp_nlhe_* below.
 *   profile  (rp_profile_*) max_actions <= 16, rows < 2^32, Decisions per batch < 2^31.
 *   nlhe     (rp_nlhe_*) 2 players, stacks of 200 chips; at most 9 choices per infoset; a batch's trees together may hold
 *            1 536 nodes per tree on average (92 B each) per PASS (a batch that needs more is split into passes automatically)
 *            and 160 Decisions per tree on average, one tree at most 48 levels,
 *            65 535 nodes and 2 048 walker nodes; the infoset table holds 2^cap_log2 rows (a full table fails the step).
 *   lloyd    K <= 256 and bins <= 256 (an Abstraction index is 8 bits, kicker/src/abstraction.rs:22-23), counts are u8
 *            (a point's mass per bin <= 255; the flop / turn layers have mass 47 / 46), N < 2^32.  The MFMA bound prunes
 *            points with 1..64 support bins; others go through the unpruned kernels.
 *   sampling Discounted / Asymmetric regret have a sign-dependent discount: ordered update only (composed, sharded and
 *            windowed steps return RP_ERR_UNSUPPORTED).
 */
#ifndef RP_MI355X_H
#define RP_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RP_API __attribute__((visibility("default")))

typedef enum rp_status {
    RP_OK = 0,
    RP_ERR_INVALID = 1,     /* bad argument / inconsistent table                */
    RP_ERR_NO_DEVICE = 2,   /* no HIP device (the product path has no CPU mode) */
    RP_ERR_HIP = 3,         /* a HIP runtime call failed (see rp_last_error)    */
    RP_ERR_UNSUPPORTED = 4, /* combination not implemented on the device path   */
    RP_ERR_CAPACITY = 5,    /* game exceeds the compiled per-tree limits        */
    RP_ERR_INTERNAL = 6     /* a self-check of the library failed (rp_last_error) */
} rp_status;

/* human-readable text of the last failure on the calling thread */
RP_API const char* rp_last_error(void);
/* Diagnostics — device self tests, HIP-event kernel clocks, traversal shape and census, the MFMA bound's intervals — are declared in
 * rp_mi355x_diag.h (same library, same conventions): what tests and bench.py call, nothing a drop-in caller needs. */
/* number of visible HIP devices (0 when there is none); never fails */
RP_API int rp_device_count(void);
/* library build info: "rp_mi355x <version> gfx950 hip <ver>" */
RP_API const char* rp_version(void);

/* ======================================================================= mccfr ==
 * crates/mccfr: Solver (solver/solver.rs:38-351), RefProf/MutProf/CfrSampling
 * (strategy/{profile.rs:12-29,storage.rs:9-18,training.rs:10-30}), schedules
 * (regret/mod.rs:18-29, policy/mod.rs:18-25), samplers (sample/mod.rs:53-65).
 */

/* RegretSchedule impls: regret/{summed,linear,discounted,floored,asymmetric}.rs */
typedef enum rp_regret_kind {
    RP_REGRET_SUMMED = 0,
    RP_REGRET_LINEAR = 1,
    RP_REGRET_DISCOUNTED = 2,
    RP_REGRET_FLOORED = 3,
    RP_REGRET_ASYMMETRIC = 4
} rp_regret_kind;

/* WeightSchedule impls: policy/{constant,linear,quadratic,exponential}.rs */
typedef enum rp_weight_kind {
    RP_WEIGHT_CONSTANT = 0,
    RP_WEIGHT_LINEAR = 1,
    RP_WEIGHT_QUADRATIC = 2,
    RP_WEIGHT_EXPONENTIAL = 3
} rp_weight_kind;

/* SamplingScheme impls: sample/{external,pruning,pluribus}.rs */
typedef enum rp_sampling_kind {
    RP_SAMPLING_EXTERNAL = 0,
    RP_SAMPLING_PRUNABLE = 1,
    RP_SAMPLING_PLURIBUS = 2
} rp_sampling_kind;

/* hyperparams/{sampling.rs:39-50, pruning.rs:36-53, training.rs:49-60}; rp_hyper_default() fills the
 * reference defaults {tau 1.0, beta 2.0, eps 0.05, threshold -3e5, explore 0.05, warmup 16384, regret_min -4e6}. */
typedef struct rp_hyper {
    float temperature;
    float smoothing;
    float curiosity;
    float prune_threshold;
    float prune_explore;
    uint64_t prune_warmup;
    float regret_min;
    uint32_t _pad;
} rp_hyper;
RP_API void rp_hyper_default(rp_hyper* out);

/* Encounter: solver/encounter.rs:22-27 (16-byte AoS at the boundary; SoA by field in HBM) */
typedef struct rp_encounter {
    float weight;
    float regret;
    float payoff;
    uint32_t visits;
} rp_encounter;

/* Flat description of an extensive-form game: the boundary's answer to "games are Rust generics"
 * (CfrGame/CfrTurn/CfrEdge/CfrInfo/CfrEncoder; kuhn/src/game.rs:115-167, leduc/src/game.rs:177-245).
 * A state is a node of the full game tree; children of a player state are listed in `choices()` order
 * (kuhn/src/info.rs:36-42, leduc/src/info.rs:29-35), children of a chance state in deal order. */
#define RP_TURN_CHANCE 254u
#define RP_TURN_TERMINAL 255u
#define RP_NO_INFO 0xffffffffu
typedef struct rp_state {
    uint8_t turn;       /* acting player 0..n_players-1, RP_TURN_CHANCE or RP_TURN_TERMINAL */
    uint8_t n_children; /* branching factor (0 for terminals)                                */
    uint16_t chance_info; /* chance states, reference-seed mode only: 1 + index of the node's info among rp_hash_streams.chance;
                             0 = this chance state has no counterpart in the reference's trees (the root deal, which the
                             reference takes from the thread RNG, kuhn/src/game.rs:115-123) and keeps the counter hash.
                             Was `reserved` before 0.4: a caller that builds its own rp_game_table MUST zero it unless it hands
                             hash streams to rp_mccfr_set_rng (a stray value in range selects the wrong chance stream)       */
    uint32_t info;      /* infoset id at player states, RP_NO_INFO otherwise                  */
    uint32_t offset;    /* player/chance: first child in children[]; terminal: row in payoffs */
} rp_state;

typedef struct rp_game_table {
    uint32_t n_states;
    uint32_t n_infos;
    uint32_t n_players;
    uint32_t max_actions;  /* A: row stride of the regret/strategy tables          */
    uint32_t n_children;   /* length of children[]                                  */
    uint32_t n_terminals;  /* rows of payoffs[]                                     */
    uint32_t train_root;   /* CfrGame::root(): chance over deals -> first decision  */
    uint32_t exploit_root; /* CfrGame::exploitability_root()                        */
    uint32_t max_depth;    /* longest root->leaf path (states)                      */
    uint32_t max_tree_nodes; /* bound on nodes of one externally-sampled tree       */
    const rp_state* states;
    const uint32_t* children;
    const float* payoffs;          /* [n_terminals][n_players]                       */
    const uint8_t* info_actions;   /* [n_infos] number of choices()                  */
    const uint8_t* info_player;    /* [n_infos] acting player                        */
    const float* default_regret;   /* [n_infos][max_actions] or NULL (= 0): CfrEdge::default_regret */
} rp_game_table;

/* Built-in game models (crates/kuhn, crates/leduc, crates/roshambo), owned by the library. */
/* RP_GAME_LEDUC_WIDE is synthetic (not a reference game): Leduc's rules over seven ranks, 616 infosets — it exists to
 * exercise the large-game kernels (more than 256 infosets) with a game the oracle can also play */
typedef enum rp_game_kind { RP_GAME_KUHN = 0, RP_GAME_LEDUC = 1, RP_GAME_RPS = 2, RP_GAME_LEDUC_WIDE = 3 } rp_game_kind;
typedef struct rp_game rp_game;
RP_API int rp_game_create(rp_game_kind kind, rp_game** out);
RP_API int rp_game_view(const rp_game* g, rp_game_table* out); /* pointers valid until destroy */
RP_API int rp_game_destroy(rp_game* g);
/* Named infoset lookup for the built-in games, e.g. "K|XB" (Kuhn: rank|history, kuhn/src/info.rs:58-66)
 * or "Q|K|XRC|R" (Leduc: rank|board|round1|round2). Returns RP_ERR_INVALID when unknown. */
RP_API int rp_game_info_id(const rp_game* g, const char* name, uint32_t* out);
RP_API int rp_game_info_name(const rp_game* g, uint32_t info, char* buf, size_t cap);
/* structural validation of a caller-built table (tree shape, offsets, perfect recall of ids) */
RP_API int rp_game_table_check(const rp_game_table* t);

typedef struct rp_mccfr rp_mccfr;

/* ---- which generator draws the sampled branches ------------------------------------------------------------------------------
 * RP_RNG_COUNTER (default): one 64-bit counter hash per (seed, epoch, tree, infoset), include/rp_math.h rp_node_hash — the
 *   build's own definition, cheap on the device.
 * RP_RNG_REFERENCE ("reference-seed" mode): the reference's own chain, restated from the published algorithms in
 *   include/rp_refrng.h: DefaultHasher (SipHash-1-3, zero key) over t.hash(), info.hash(), node.seed().hash()
 *   (crates/mccfr/src/strategy/flow.rs:285-295) -> SmallRng::seed_from_u64 (xoshiro256++) -> one draw: WeightedIndex<f32> at an
 *   opponent node (sample/external.rs:41-64), random_range(0..n) at a chance node (sample/mod.rs:68-82), random::<f32>() for
 *   Pluribus' exploration coin (sample/pluribus.rs:91).  `t` = the epoch, node.seed() = the tree's index in the batch
 *   (solver.rs: the par_iter index), info.hash() = the byte stream the caller's `impl Hash for I` writes, handed over once as
 *   rp_hash_streams (INTEGRATION.md shows the ten-line recording Hasher that produces it from any CfrInfo).  The library's
 *   `seed` then only enters the root deal, which the reference leaves to the unseeded thread RNG. */
typedef enum rp_rng_kind { RP_RNG_COUNTER = 0, RP_RNG_REFERENCE = 1 } rp_rng_kind;
#define RP_HASH_STREAM_MAX 55u
typedef struct rp_hash_stream {
    uint8_t len;                       /* bytes written by I::hash                          */
    uint8_t bytes[RP_HASH_STREAM_MAX]; /* in write order                                    */
} rp_hash_stream;
typedef struct rp_hash_streams {
    uint32_t n_infos;              /* = rp_game_table.n_infos                                     */
    uint32_t n_chance;             /* distinct infos of in-tree chance nodes                      */
    const rp_hash_stream* infos;   /* [n_infos]                                                   */
    const rp_hash_stream* chance;  /* [n_chance], indexed by rp_state.chance_info - 1             */
} rp_hash_streams;
/* the streams of a built-in game (kuhn/src/info.rs:20-24,71; leduc/src/info.rs:11-17,85; roshambo/src/turn.rs:10-18) */
RP_API int rp_game_hash_streams(const rp_game* g, rp_hash_streams* out); /* pointers valid until destroy */
RP_API int rp_mccfr_set_rng(rp_mccfr* h, rp_rng_kind kind, const rp_hash_streams* streams /* NULL for RP_RNG_COUNTER */);

typedef enum rp_update_mode {
    RP_UPDATE_ORDERED = 0, /* per-key sequential application in tree-id order: solver.rs:96-105 exactly */
    RP_UPDATE_COMPOSED = 1 /* per-key composed (a,b,floor) maps; used for the multi-GPU exchange         */
} rp_update_mode;
/* WHICH MODE TO USE.  ORDERED is the reference's arithmetic bit for bit and the default of rp_mccfr_create (a drop-in caller
 * gets the reference's numbers); it is a serial chain per table cell: 0.35 G infoset-updates/s on Leduc.  COMPOSED re-associates
 * the same touches (tables within rtol 1e-4 per step of ORDERED, the "stated fp32 tolerance" of the build's north star), runs
 * 100x faster (36 G/s) and is the only mode that shards across GPUs — it is the mode bench.py's headline `value` is measured in
 * (`other_update_mode` in the same line quotes ORDERED).  Select it at creation with rp_mccfr_create_mode, or later with
 * rp_mccfr_set_update_mode.  Discounted / Asymmetric regret (sign-dependent discount) exist in ORDERED only. */

/* Composed update: a BLOCK is the set of Decisions of one infoset produced by one chunk of RP_COMPOSE_CHUNK
 * consecutive trees (of this rank); it is composed sequentially, in tree-id order, from the identity into a map
 * F(x) = max(a x + b, m) per table cell; the blocks of an infoset are then folded in chunk order and the result is
 * applied to the table (rank by rank in the multi-GPU exchange).  Exact in real arithmetic; in f32 it is a fixed
 * re-association of the reference's sequential update (tests state the tolerance). */
#define RP_COMPOSE_CHUNK 256u
/* The block maps of a cell are folded sequentially inside groups of RP_FOLD_GROUP consecutive blocks, the group maps
 * sequentially into the cell's map: a fixed two-level shape, so that the groups of a hot infoset (or hot row of the
 * sparse profile) fold in parallel on the device and the oracle can restate the association exactly. */
#define RP_FOLD_GROUP 64u

/* mccfr!(Prefix, Encoder, T, E, G, I, batch) + <R, W, S> (strategy/macros.rs:7-151): one solver instance.
 * `batch_size` = Solver::batch_size() (trees per step). */
RP_API int rp_mccfr_create(const rp_game_table* game, rp_regret_kind r, rp_weight_kind w, rp_sampling_kind s,
                           uint32_t batch_size, const rp_hyper* hp, uint64_t seed, int device, rp_mccfr** out);
RP_API int rp_mccfr_create_mode(const rp_game_table* game, rp_regret_kind r, rp_weight_kind w, rp_sampling_kind s,
                                uint32_t batch_size, const rp_hyper* hp, uint64_t seed, int device, rp_update_mode mode,
                                rp_mccfr** out);
RP_API int rp_mccfr_destroy(rp_mccfr* h);
/* Solver::step (solver.rs:96-105): batch() then update_{regret,weight,payoff,visits} then epoch += 1 */
RP_API int rp_mccfr_step(rp_mccfr* h);
/* Solver::solve (solver.rs:111-122): trees / batch_size steps */
RP_API int rp_mccfr_solve(rp_mccfr* h, uint64_t trees);
/* Solver::spend (solver.rs:130-137) */
RP_API int rp_mccfr_spend(rp_mccfr* h, double seconds, uint64_t* iterations, double* elapsed);
/* Trainer::train (crates/forge/src/trainer.rs:18-66) over this solver: loop { step; checkpoint; flush; interrupt? }.
 * A checkpoint fires when `log_interval` seconds passed since the last one (Metrics::checkpoint, mccfr/src/metrics/
 * mod.rs:67-80: rate = new infos / max(1, whole seconds)); its line is Checkpoint's Display (metrics/checkpoint.rs:
 * 39-50): "batch E", "nodes N", "infos I", "I/sec R.R", each left-aligned in 20 columns.  A flush event fires every
 * `flush_interval` seconds (TrainingHyperParams defaults: 60 s and 30 min, hyperparams/training.rs:50-54); the
 * callback may export the table there (rp_mccfr_export).  The loop ends when *interrupt becomes non-zero
 * (pokerkit::interrupted), after max_steps (0 = unbounded) or max_seconds (<= 0 = unbounded); `summary` receives
 * Progress::summary (progress.rs:24-26).  Counters are read from the device only when a checkpoint is due. */
typedef struct rp_checkpoint { uint64_t epoch, nodes, infos; double rate; } rp_checkpoint;
typedef enum rp_train_event { RP_TRAIN_CHECKPOINT = 0, RP_TRAIN_FLUSH = 1 } rp_train_event;
typedef void (*rp_train_event_fn)(int event, const rp_checkpoint* cp, const char* line, void* user);
RP_API int rp_mccfr_train(rp_mccfr* h, uint64_t max_steps, double max_seconds, double log_interval,
                          double flush_interval, rp_train_event_fn on_event, void* user,
                          const volatile int* interrupt, char* summary, size_t summary_cap);
/* enqueue `steps` steps on the stream without host synchronisation (FastSession::step loop shape) */
RP_API int rp_mccfr_step_async(rp_mccfr* h, uint32_t steps);
RP_API int rp_mccfr_sync(rp_mccfr* h);
/* RefProf::t (profile.rs:14) */
RP_API int rp_mccfr_epoch(rp_mccfr* h, uint64_t* epoch);
/* Metrics nodes / infos counters (metrics/mod.rs:21-80; infos = the "infoset-updates" unit, solver.rs:273) */
RP_API int rp_mccfr_counters(rp_mccfr* h, uint64_t* nodes, uint64_t* infos);
/* RefProf::cum_{weight,regret,payoff,visits} / MutProf::mut_* (profile.rs:16-23, storage.rs:9-18) */
RP_API int rp_mccfr_get(rp_mccfr* h, uint32_t info, uint32_t edge, rp_encounter* out);
RP_API int rp_mccfr_set(rp_mccfr* h, uint32_t info, uint32_t edge, const rp_encounter* in);
/* CfrData::encounters_ref / hydrate (book.rs:14-24, nlhe/src/profile.rs:97-163): rows[n_infos*max_actions] */
RP_API int rp_mccfr_export(rp_mccfr* h, rp_encounter* rows, uint64_t cap);
RP_API int rp_mccfr_import(rp_mccfr* h, const rp_encounter* rows, uint64_t n, uint64_t epoch);
typedef enum rp_dist_kind {
    RP_DIST_ITERATED = 0, /* RefProf::iterated_distribution (profile.rs:47-51) */
    RP_DIST_AVERAGED = 1, /* RefProf::averaged_distribution (profile.rs:40-44) */
    RP_DIST_SAMPLING = 2  /* CfrFlow::sampling_distribution (flow.rs:33-42)    */
} rp_dist_kind;
RP_API int rp_mccfr_policy(rp_mccfr* h, uint32_t info, rp_dist_kind kind, float* out, uint32_t* n);
/* Solver::exploitability (solver.rs:327-337) -> CfrNash::exploitability (nash.rs:31-38); host-side validation */
RP_API int rp_mccfr_exploitability(rp_mccfr* h, float* out);
/* RefProf::sum_regret (book.rs:124-131), summed in (info, edge) order */
RP_API int rp_mccfr_sum_regret(rp_mccfr* h, float* out);
/* batch size may be changed between steps (no reference equivalent: batch_size is a const fn there) */
RP_API int rp_mccfr_set_batch(rp_mccfr* h, uint32_t batch_size);
RP_API int rp_mccfr_set_update_mode(rp_mccfr* h, rp_update_mode mode);
/* run on a caller-provided hipStream_t (NULL = the library's own stream) */
RP_API int rp_mccfr_set_stream(rp_mccfr* h, void* hip_stream);

/* ---- multi-GPU: trees sharded by rank, per-key composed maps exchanged -------------
 * rank r samples tree ids [r*B, (r+1)*B) of a world*B-tree batch.  step_local() runs the traversal and
 * reduces this rank's Decisions to one composed map per table cell; the caller all-gathers the
 * `summary_bytes` blobs (RCCL) and every rank folds them in rank order with step_apply(). */
RP_API int rp_mccfr_set_shard(rp_mccfr* h, uint32_t rank, uint32_t world);
RP_API int rp_mccfr_summary_bytes(rp_mccfr* h, size_t* bytes);
RP_API int rp_mccfr_step_local(rp_mccfr* h, void* summary_dev);
RP_API int rp_mccfr_step_apply(rp_mccfr* h, const void* gathered_dev, uint32_t world);
/* The PERIODIC exchange: window_local() is one local step against the table as it stood when the window began (the
 * table is not touched; the epoch — hence the sampled trees, the walker and the discounts — advances) whose composed
 * maps are folded into `window_dev` (summary_bytes; first != 0 starts a new window); after `S` such steps the caller
 * all-gathers the window summaries once and every rank applies them in rank order with window_apply() (no epoch
 * change).  S = 1 is step_local + step_apply.  Oracle: ora_mccfr_window_world. */
RP_API int rp_mccfr_window_local(rp_mccfr* h, void* window_dev, int first);
RP_API int rp_mccfr_window_apply(rp_mccfr* h, const void* gathered_dev, uint32_t world);

/* ---- rp_comm: the RCCL communicator of a one-process-per-GPU job, for hosts without torch.distributed (a Rust or C
 * trainer).  Rank 0 calls rp_comm_unique_id and ships the 128 bytes to the other ranks by its own means (MPI, a socket,
 * a file); every rank then calls rp_comm_create (ncclCommInitRank: collective, blocks until all ranks arrive).  A host that
 * already owns an ncclComm_t wraps it with rp_comm_adopt (not destroyed by rp_comm_destroy).  librccl is loaded on the
 * first rp_comm_* call.  SURVEY §8b: rp_mccfr_allreduce(h, rp_comm*). */
#define RP_COMM_ID_BYTES 128
typedef struct rp_comm rp_comm;
RP_API int rp_comm_unique_id(uint8_t* id /* [RP_COMM_ID_BYTES] */);
RP_API int rp_comm_create(const uint8_t* id, int rank, int world, int device, rp_comm** out);
RP_API int rp_comm_adopt(void* nccl_comm, int rank, int world, int device, rp_comm** out);
RP_API int rp_comm_destroy(rp_comm* c);
/* `steps` Solver::step's of a tree-sharded job (rank = the communicator's): exchange windows of `window` local steps
 * (rp_mccfr_window_local), ONE ncclAllGather of the window summaries per window on the solver's stream, then
 * rp_mccfr_window_apply — no host synchronisation inside; a trailing partial window is exchanged too.  Calls
 * rp_mccfr_set_shard(rank, world) itself.  Every rank must pass the same steps / window. */
RP_API int rp_mccfr_step_comm(rp_mccfr* h, rp_comm* c, uint32_t steps, uint32_t window);

/* which Solver::batch kernel this handle launches under its current sampling scheme: 0 = per-tree scratch in HBM (any
 * game), 1 = per-lane DFS with the tree in LDS (small games), 2 = instantiated over the game's compile-time action
 * skeleton (Kuhn / Leduc shapes, external sampling; csrc/traverse_static.hpp).  All three produce identical Decisions. */
RP_API int rp_mccfr_traversal_variant(rp_mccfr* h, int* out);
/* Which compile-time action skeleton a game table matches node for node, for every chance outcome (host only, no device
 * needed): 0 none (the generic traversal kernels), 1 Kuhn's, 2 Leduc's (any number of ranks). */
RP_API int rp_game_skeleton(const rp_game_table* game, int* out);

/* ============================================================= sparse profile ==
 * The update half of Solver::step (solver/solver.rs:96-105,143-192: update_regret, update_weight, update_payoff,
 * update_visits in batch order, then epoch += 1) for tables addressed by ROW INDEX: the NLHE-scale shape
 * (~2^27 infosets x <= 9 actions, crates/forge/README.md:175), where a per-infoset launch grid is impossible and the
 * Decisions come from a producer other than the built-in traversal (synthetic batches;
 * §8f f1: the on-device NLHE engine).  Rows are 16*max_actions contiguous bytes in HBM
 * {regret[A], weight[A], payoff[A], visits[A]} so that one touch reads and writes ONE contiguous row.
 *
 *   ORDERED   stable radix sort by row, then every row's touches applied sequentially in batch order:
 *             the reference loop, bit for bit.
 *   COMPOSED  per-row maps F(x) = max(a x + b, m) composed over blocks of RP_SPARSE_BLOCK consecutive touches,
 *             folded in block order (hot rows parallelise); the multi-GPU exchange uses the same entries:
 *             summarize -> all-gather -> fold in rank order.
 */
#define RP_SPARSE_BLOCK 64u
typedef struct rp_profile rp_profile;
typedef struct rp_decisions { /* one batch of Decisions in DEVICE memory, in application (tree-id) order */
    uint32_t n;
    const uint32_t* row;        /* [n] table row of the infoset                                   */
    const uint8_t* n_actions;   /* [n] |choices()| of the infoset (the same for every touch of a row) */
    const uint16_t* expanded;   /* [n] edges present in the regret vector (solver.rs:143-152)       */
    const float* regret;        /* [n][max_actions] regret_vector                                  */
    const float* policy;        /* [n][max_actions] policy_vector                                  */
    const float* payoff;        /* [n] infoset value                                               */
} rp_decisions;
/* default_regret: [max_actions] CfrEdge::default_regret per action slot (NULL = 0), book.rs:101-106 */
RP_API int rp_profile_create(int device, uint64_t n_rows, uint32_t max_actions, rp_regret_kind regret,
                             rp_weight_kind weight, const rp_hyper* hp, const float* default_regret,
                             uint32_t max_batch, rp_profile** out);
RP_API int rp_profile_destroy(rp_profile* h);
/* one Solver::step worth of updates: applies the batch (asynchronously on the profile's stream), epoch += 1 */
RP_API int rp_profile_apply(rp_profile* h, const rp_decisions* batch, rp_update_mode mode);
RP_API int rp_profile_sync(rp_profile* h);
RP_API int rp_profile_epoch(const rp_profile* h, uint64_t* epoch);
RP_API int rp_profile_set_epoch(rp_profile* h, uint64_t epoch);
/* out[i*max_actions + a] = Encounter of (rows[i], action a); rows is a HOST array */
RP_API int rp_profile_get_rows(rp_profile* h, uint64_t n, const uint32_t* rows, rp_encounter* out);
/* overwrite rows from the host (hydrate / resynchronisation): in[i*max_actions + a] -> (rows[i], action a) */
RP_API int rp_profile_set_rows(rp_profile* h, uint64_t n, const uint32_t* rows, const rp_encounter* in);
RP_API int rp_profile_set_stream(rp_profile* h, void* hip_stream);
/* multi-GPU: bytes of one summary entry (16 + 2*max_actions*16), and the two halves of a sharded step.
 * summarize: this rank's batch -> entries sorted by row in `entries_dev` (capacity >= batch->n entries);
 *            *n_entries is written on the host after a stream sync.
 * fold:      `n_entries` entries (all ranks' lists back to back, rank-major) -> stable sort by row -> per-row fold in
 *            rank order -> table; epoch += 1.  Every replica that folds the same list stays bit-identical. */
RP_API int rp_profile_entry_bytes(const rp_profile* h, size_t* bytes);
RP_API int rp_profile_summarize(rp_profile* h, const rp_decisions* batch, void* entries_dev, uint32_t* n_entries);
RP_API int rp_profile_fold(rp_profile* h, const void* entries_dev, uint32_t n_entries);

/* ===================================================================== nlhe ==
 * The blueprint trainer's solver: mccfr!(Nlhe, NlheEncoder, NlheTurn, NlheEdge, NlheGame, NlheInfo, 128)
 * (crates/nlhe/src/solver.rs:11) — external-sampling MCCFR over heads-up no-limit hold'em, the game generated on the
 * device (kicker::Game rules, the Pluribus action abstraction), infosets keyed by NlheInfo = (subgame Path, abstraction
 * bucket, choices Path) (nlhe/src/info.rs:145-160; columns past BIGINT, present SMALLINT, choices BIGINT of the blueprint
 * table, nlhe/src/profile.rs:20-31) and hashed to rows of an rp_profile table.  The encoder's isomorphism -> abstraction
 * map (NlheEncoder, nlhe/src/encoder.rs:30-36) is the four rp_lookup tables the clustering pipeline produces
 * (tables[street], street = 0 pref .. 3 river); tables = NULL selects a hash of the canonical observation (tests).
 * 2^cap_log2 table rows of 9 actions (144 B each) + one 32-byte key slot per row; batch = trees per step (0 = the
 * reference's 128).  2 players, stacks of 100 big blinds.  The batch is grown LEVEL-SYNCHRONOUSLY (all trees one level per
 * pair of launches, kernels sorted by node kind: robopoker_amd/csrc/nlmc_level.hpp) in 1 536 nodes of budget per tree
 * (92 B each); a batch that needs more is traversed in several passes over contiguous ranges of its trees (same Decisions).  Sampling scheme: ExternalSampling (the mccfr! macro's default) until rp_nlhe_set_sampling selects
 * PrunableSampling / PluribusSampling (Flagship, nlhe/src/lib.rs:86-90; thresholds from `hp`).
 * Device memory beside the table: 224 Decisions of buffer per tree (~120 B each) and the node arrays.  A batch of at most 2 048
 * trees (the reference runs 128) takes the one-tree-per-workgroup kernel and reserves a region of 8 192 nodes (92 + 192 B each: the
 * reference-order evaluation arrays are always allocated there) and 9 floats per walker slot PER TREE: 0.3 GB at 128 trees, 4.8 GB at
 * 2 048; the region is not released when a tree outgrows it and the handle falls back to the level-synchronous kernels.
 * Oracle: oracle/rp_oracle_nlmc.c. */
typedef struct rp_nlhe rp_nlhe;
typedef struct rp_lookup rp_lookup;
RP_API int rp_nlhe_create(int device, uint32_t cap_log2, rp_regret_kind regret, rp_weight_kind weight, const rp_hyper* hp,
                          uint64_t seed, uint32_t batch, const rp_lookup* const* tables, rp_nlhe** out);
RP_API int rp_nlhe_destroy(rp_nlhe* h);
/* SamplingScheme::sample at walker nodes (mccfr/src/sample/{external.rs:17-64, pruning.rs:44-66, pluribus.rs:72-101}) */
RP_API int rp_nlhe_set_sampling(rp_nlhe* h, rp_sampling_kind sampling);
/* rp_rng_kind for the NLHE trainer.  RP_RNG_REFERENCE: the opponent's WeightedIndex draw (sample/external.rs:41-64) and Pluribus'
 * exploration coin (sample/pluribus.rs:91) come from DefaultHasher(t, NlheInfo, tree id) -> SmallRng (flow.rs:285-295), NlheInfo's
 * Hash stream being subgame: Path(u64), choices: Path(u64), Abstraction(u16) (nlhe/src/{info.rs:41-42, public.rs:19-23,
 * secret.rs:10-11}) — the three fields of the infoset key this library already carries.  Hole cards and board cards stay on the
 * library's counter hash in both modes: the reference deals them from the unseeded thread RNG (kicker game.rs). */
RP_API int rp_nlhe_set_rng(rp_nlhe* h, rp_rng_kind kind);
/* The regret vectors' float order.  The reference values a leaf at rel / smp * payoff with the two reach products multiplied from the
 * walker node's CHILD down, and sums children in choices() order (CfrFlow::recursed_value / ancestor_reach, flow.rs:166-216).  A batch of
 * at most 2 048 trees (the reference's is 128) is ALWAYS evaluated that way (one tree per workgroup): Decisions and tables equal the
 * reference's arithmetic bit for bit.  Larger batches default to the factorised form D(node) = sum f(edge) D(child) (the same real
 * number; regret vectors within rtol 2e-4 / atol 2e-3); rp_nlhe_set_exact(h, 1) makes them carry the per-ancestor reach rows too
 * (192 B per node on top of 92: 77 GB at 262 144 trees) and equal the reference's order bit for bit as well. */
RP_API int rp_nlhe_set_exact(rp_nlhe* h, int on);
/* Solver::step (solver.rs:96-105): the batch's trees, their Decisions, the table update (ordered or composed), epoch += 1 */
RP_API int rp_nlhe_step(rp_nlhe* h, rp_update_mode mode);
/* Trainer::train (crates/forge/src/trainer.rs:18-66) over this solver — the loop forge runs on the Flagship type; the contract of
 * rp_mccfr_train (checkpoint / flush events, Checkpoint's display line, interrupt flag, Progress::summary) */
RP_API int rp_nlhe_train(rp_nlhe* h, rp_update_mode mode, uint64_t max_steps, double max_seconds, double log_interval,
                         double flush_interval, rp_train_event_fn on_event, void* user, const volatile int* interrupt,
                         char* summary, size_t summary_cap);
/* Solver::batch (solver.rs:225-250) alone, for inspection: the Decisions of the current epoch in tree order, each with
 * the infoset behind its row; *n = their number, at most `cap` are copied out; any output may be NULL.
 * regret / policy: [n][9]. */
RP_API int rp_nlhe_batch(rp_nlhe* h, uint32_t cap, uint32_t* n, uint32_t* tree, uint64_t* past, uint32_t* present, uint64_t* choices,
                         uint8_t* n_actions, uint16_t* expanded, float* regret, float* policy, float* payoff);
RP_API int rp_nlhe_epoch(rp_nlhe* h, uint64_t* epoch);
/* nodes / infos: Solver::inc_nodes / inc_infos (solver.rs:252-275); keys: infosets in the table */
RP_API int rp_nlhe_counters(rp_nlhe* h, uint64_t* nodes, uint64_t* infos, uint64_t* keys);
/* NlheProfile::rows (nlhe/src/profile.rs:144-163) without the edge expansion: every infoset with its 9 Encounters
 * (slot a = the a-th edge of `choices`); *n = infosets in the table, at most `cap` are copied */
RP_API int rp_nlhe_export(rp_nlhe* h, uint64_t cap, uint64_t* n, uint64_t* past, uint32_t* present, uint64_t* choices, rp_encounter* enc);
/* Hydrate (profile.rs:90-141): load Encounters by infoset; sets the epoch */
RP_API int rp_nlhe_import(rp_nlhe* h, uint64_t n, const uint64_t* past, const uint32_t* present, const uint64_t* choices,
                          const rp_encounter* enc, uint64_t epoch);

/* Multi-GPU (BASELINE configs[3]): trees sharded by rank (rank r samples tree ids [r*B, (r+1)*B) of a world*B-tree epoch
 * against a replicated table).  step_local: this rank's traversal reduced to one composed entry per infoset touched
 * (rp_profile_summarize's records, entry_bytes each, at most max_entries) plus the infoset KEY of every entry — each
 * rank's table assigns rows in its own insertion order, so the exchange is by key; the caller all-gathers entries and keys
 * (rank-major, packed); step_apply maps the keys to this table's rows (inserting unseen infosets with their default
 * regrets) and folds in rank order, epoch += 1.  Replicas stay identical as key -> Encounter maps.  Oracle:
 * ora_nlmc_step_world. */
RP_API int rp_nlhe_set_shard(rp_nlhe* h, uint32_t rank, uint32_t world);
RP_API int rp_nlhe_entry_bytes(rp_nlhe* h, size_t* bytes, uint32_t* max_entries);
RP_API int rp_nlhe_step_local(rp_nlhe* h, void* entries_dev, uint64_t* past_dev, uint32_t* present_dev, uint64_t* choices_dev,
                              uint32_t* n_entries);
RP_API int rp_nlhe_step_apply(rp_nlhe* h, void* entries_dev, const uint64_t* past_dev, const uint32_t* present_dev,
                              const uint64_t* choices_dev, uint32_t n_entries);
/* the whole sharded step over the library's own RCCL communicator (rp_comm, below): step_local, the ranks' entry counts
 * (the one host read of a step: ncclAllGather takes host-known sizes), entries + keys gathered padded to the longest list,
 * packed rank-major on the device, step_apply; `steps` times.  Sets the shard from the communicator's rank / world. */
RP_API int rp_nlhe_step_comm(rp_nlhe* h, rp_comm* c, uint32_t steps);
/* every rp_nlhe kernel runs on ONE stream (the profile's): hand it the stream the collectives run on and the exchange is
 * ordered without host synchronisation (NULL = back to a stream of the library's own); rp_nlhe_sync waits for it.
 * step_local returns with *n_entries valid and the key arrays QUEUED on that stream. */
RP_API int rp_nlhe_set_stream(rp_nlhe* h, void* hip_stream);
RP_API int rp_nlhe_sync(rp_nlhe* h);

/* ===================================================================== lloyd ==
 * crates/elkan: Elkan<K,N> (elkan.rs:27-207), Bounds (bounds.rs:19-120), Prior::tally (prior.rs:35-47)
 * crates/lloyd: Layer (layer.rs:23-273), Kmeans (kmeans.rs:29-111), Sinkhorn (sinkhorn.rs:62-230),
 *               Metric::emd (metric.rs:109-115), Equity::variation (equity.rs:41-53)
 * crates/monge: Coupling::{minimize, flow, cost} (coupling.rs:23-51)
 */
typedef enum rp_metric_kind {
    RP_METRIC_SINKHORN = 0, /* Histogram over Flop/Turn buckets -> Sinkhorn::divergence */
    RP_METRIC_VARIATION = 1 /* Histogram over river equity bins -> Equity::variation    */
} rp_metric_kind;

/* SinkhornHyperParams::DEFAULT {0.025, 128, 5e-4} (lloyd/src/hyperparams/sinkhorn.rs:17-23) */
typedef struct rp_sinkhorn_hp {
    float temperature;
    uint32_t iterations;
    float tolerance;
} rp_sinkhorn_hp;
RP_API void rp_sinkhorn_hp_default(rp_sinkhorn_hp* out);

/* Pair::merge (lloyd/src/pair.rs:58-65): triangular index of an unordered bin pair, i != j */
static inline uint32_t rp_tri_index(uint32_t i, uint32_t j) {
    uint32_t lo = i < j ? i : j, hi = i < j ? j : i;
    return hi == 0 ? 0 : hi * (hi - 1) / 2 + lo;
}

typedef struct rp_kmeans rp_kmeans;

/* Layer::build (layer.rs:250-272): K clusters over N points, each a dense histogram of `bins` u8 counts
 * (Bins<N>, bins.rs:30-37; reference stores usize counts, point mass <= 47 so u8 is lossless).
 * `tri_metric` is Metric's triangular table bins*(bins-1)/2 (metric.rs:26-55) for RP_METRIC_SINKHORN,
 * NULL for RP_METRIC_VARIATION.  `counts` is a host pointer; use the _device variant when the
 * histograms already live in HBM. */
RP_API int rp_kmeans_create(uint32_t K, uint64_t N, uint32_t bins, const uint8_t* counts, rp_metric_kind kind,
                            const float* tri_metric, const rp_sinkhorn_hp* hp, uint64_t seed, int device,
                            rp_kmeans** out);
RP_API int rp_kmeans_create_device(uint32_t K, uint64_t N, uint32_t bins, const void* counts_dev,
                                   rp_metric_kind kind, const float* tri_metric, const rp_sinkhorn_hp* hp,
                                   uint64_t seed, int device, rp_kmeans** out);
RP_API int rp_kmeans_destroy(rp_kmeans* h);
/* Elkan::init_centroids = k-means++ (layer.rs:140-181); chosen[] receives the K point indices (may be NULL) */
RP_API int rp_kmeans_init_centroids(rp_kmeans* h, uint64_t* chosen);
/* rp_rng_kind of rp_kmeans_init_centroids.  RP_RNG_REFERENCE = Layer::init_centroids' own chain (crates/lloyd/src/layer.rs:155-178):
 * DefaultHasher over the Street (`street` = its discriminant: 0 Pref, 1 Flop, 2 Turn, 3 Rive; deuce/src/street.rs:21-27) ->
 * SmallRng::seed_from_u64, ONE generator for the K picks, each pick WeightedIndex::<f32>::new(potentials).sample(rng): f32
 * running sums in index order (sequential by definition: one wavefront, ~4 ns per point), x = Uniform::new(0, total).sample,
 * partition_point (include/rp_refrng.h).  Single GPU: the running sums span all N points in order, so the sharded k-means++
 * (rp_kmeans_kpp_* composed across ranks) keeps the fixed-point draw.  `seed` is not used in this mode. */
RP_API int rp_kmeans_set_rng(rp_kmeans* h, rp_rng_kind kind, int street);
/* Which exp / ln the Sinkhorn distances compute with.
 *   RP_LIBM_CONTRACT  (default) include/rp_math.h's rp_expf / rp_logf: f32 only, packed and pipelined in the softmin loops,
 *                     <= 1 ulp from glibc's on the Sinkhorn's domain.
 *   RP_LIBM_GLIBC     glibc's expf / logf as a Rust build on Linux calls them (f32::exp / f32::ln, sinkhorn.rs:115,120-127,136;
 *                     phi.rs:36), restated in include/rp_libm_glibc.h (equal to glibc 2.35's on all 2^32 inputs) and evaluated in
 *                     double on the device: every distance, bound, drift and bucket is the reference's, bit for bit, at a few
 *                     times the cost (every kernel that evaluates exp / ln exists in both arithmetics: csrc/lloyd_kernels.hpp).
 * What the default's <= 1 ulp is worth is measured (costs within 7 ulps, no k-means++ pick and no bucket moves). */
typedef enum rp_libm_kind { RP_LIBM_CONTRACT = 0, RP_LIBM_GLIBC = 1 } rp_libm_kind;
/* a layer: call before the first centroid exists (recomputes the points' self costs).  One way only.  The three filters in front of
 * the bit-faithful solves — the k-means++ column bound, the k-means++ interval filter, the MFMA bound of the neighbor passes — stay
 * in place: their margins (4e-5 relative, 4e-6 absolute) sit three orders above the <= 7 ulps between the two arithmetics, and the
 * full flop layer has been audited in both (profiles/r05_glibc_audit.json, profiles/r03_mfma_audit.json: 0 of 1 286 792 points differ
 * from the unpruned search).  rp_kmeans_set_prune(h, 0) runs every distance through the bit-faithful kernel instead.
 * Host assumption of RP_LIBM_GLIBC ("bit for bit with a Rust build"): x86-64 glibc whose expf is the FMA variant (the ifunc every
 * AVX2 machine selects); rp_libm_glibc.h restates that one, and tests/test_libm_glibc.py compares it with the platform's libm. */
RP_API int rp_kmeans_set_libm(rp_kmeans* h, rp_libm_kind kind);
/* enable = 0: no filter in front of the exact solves (Elkan::neighbor / Layer::init_centroids as the reference loops them, every
 * (point, centroid) pair solved, every stale-bound refresh solved): the yardstick the filtered passes are audited against.  Before
 * the first centroid; a layer that gave its filters up does not get them back (enable = 1 on such a layer: RP_ERR_UNSUPPORTED).
 *
 * WHAT THE FILTERED PASSES GUARANTEE — read before relying on "bit-exact buckets" through them.  The four filters (k-means++ column
 * bound and interval filter, the MFMA bound of init_bounds / lookup, the interval-decided refresh of the Elkan iterations) replace a
 * bit-faithful solve by a scaling-domain INTERVAL that must contain the value the reference would compute.  The column bound is
 * rigorous, and so is the rule by which the MFMA bound skips cost evaluations inside a stopping window (round 6: a Lipschitz bound of
 * <P, C> in the coupling's L1 travel, exact arithmetic, applied only to columns that cannot be the argmin either way), and so is the
 * rule by which a far column (MFMA bound) or a far pair (k-means++ interval filter) LEAVES before its stopping window closes (round 6:
 * weak duality for a Kantorovich pair read off the iterate, f = T ln u with g = -T ln K^T u or f's c-transform, plus the L1 error of the
 * row marginals, which never grows; float slack only — csrc/sinkhorn_bound.hpp "THE DUAL EXIT").  The intervals of the columns that are
 * followed to the end are not proven: their margins (SbParams: kappa, rho, dc_abs 4e-6, dc_rel 4e-5) are a multiple of the worst
 * float noise MEASURED between the scaling-domain and the log-domain iteration, and the smallest slack observed on a sampled pair was
 * 0.88 of the margin (profiles/r05_glibc_audit.json), i.e. a safety factor of about 8 over the worst observed case, on synthetic
 * points.  The evidence that they hold: full-size audits in both arithmetics with 0 of 1 286 792 points differing from the unpruned
 * search on the synthetic layer (profiles/r05_glibc_audit.json, r05_kpp_audit*.json) and on the real flop layer (r03_mfma_audit.json),
 * 0 differences over 32 Elkan iterations with and without the refresh bound (profiles/r06_refresh_audit_*.json; all of them again
 * after the dual exits: profiles/r06p_*audit*.json, 0 of 2 573 584 audited points in each of the four), and a runtime
 * tripwire: every 521st point is searched again without the MFMA prune after every pruned pass (and every 521st point a k-means++ round's
 * interval filter drops is solved anyway and its bound compared with the exact distance), a mismatch makes the next call that
 * hands results out fail with RP_ERR_INTERNAL (never a silently different bucket).  That is a statistical claim with a tripwire, not a
 * bound: a caller that needs exactness BY CONSTRUCTION uses rp_kmeans_set_prune(h, 0), the only such mode (and RP_LLOYD_AUDIT=1 runs
 * the unpruned search behind every pruned pass and counts disagreements). */
RP_API int rp_kmeans_set_prune(rp_kmeans* h, int enable);
/* the three stand-alone operators (rp_sinkhorn_divergence / _cost / _flow): process-wide */
RP_API int rp_sinkhorn_set_libm(rp_libm_kind kind);
/* install centroids = copies of the given points (TestLayer-style explicit seeding, tests.rs:100-102) */
RP_API int rp_kmeans_set_centroids(rp_kmeans* h, const uint64_t* point_index);
/* install / read one centroid as an integer histogram (counts[bins] u32): resume, and the multi-GPU
 * k-means++ where the chosen point lives on another rank */
RP_API int rp_kmeans_set_centroid(rp_kmeans* h, uint32_t k, const uint32_t* counts);
RP_API int rp_kmeans_get_point(rp_kmeans* h, uint64_t index, uint32_t* counts);
/* k-means++ (layer.rs:140-181) one primitive at a time, so a point-sharded job can interleave collectives:
 *   kpp_begin   potentials <- 1, centroids cleared
 *   kpp_total   this shard's sum of quantised potentials (exact u64, rp_math.h rp_kpp_quant)
 *   kpp_pick    first local i whose inclusive quantised prefix exceeds r; its potential <- 0
 *   kpp_update  potentials <- min(potentials, distance(centroid k, point)^2)
 * rp_kmeans_init_centroids is exactly: for k in 0..K { total; r = mulhi64(rp_stream(seed,k), total); pick;
 * set_centroid(k, point); update(k) }. */
RP_API int rp_kmeans_kpp_begin(rp_kmeans* h);
RP_API int rp_kmeans_kpp_total(rp_kmeans* h, uint64_t* total);
RP_API int rp_kmeans_kpp_pick(rp_kmeans* h, uint64_t r, uint64_t* index);
RP_API int rp_kmeans_kpp_update(rp_kmeans* h, uint32_t k);
/* Elkan::init_bounds (elkan.rs:39-47) */
RP_API int rp_kmeans_init_bounds(rp_kmeans* h);
/* Kmeans::next (kmeans.rs:82-110): step_elkan (elkan.rs:153-168), install centroids, Prior::tally.
 * drift[K], sizes[K], reassigned may be NULL. */
RP_API int rp_kmeans_step(rp_kmeans* h, float* drift, uint64_t* sizes, double* reassigned);
/* Elkan::step_naive (elkan.rs:171-188) + install */
RP_API int rp_kmeans_step_naive(rp_kmeans* h);
/* Layer::lookup (layer.rs:62-82): fresh neighbor(i) for every point; bucket[N], distance[N] (may be NULL) */
RP_API int rp_kmeans_assign(rp_kmeans* h, uint8_t* bucket, float* distance);
/* current Elkan bounds: j[N] (u8), upper[N]; lower[N*K] may be NULL.  With the interval-decided refresh (the default for Sinkhorn
 * layers whose filters are on; csrc/refresh_bound.hpp) upper[i] may be the UPPER END of an interval around the reference's
 * Bounds::error and lower[i][j[i]] its lower end: rp_kmeans_upper_interval (rp_mi355x_diag.h) says where and returns the lower ends.
 * j[], and lower[i][k] for k != j[i], are the reference's bit for bit in every mode; after rp_kmeans_set_prune(h, 0) so is everything. */
RP_API int rp_kmeans_bounds(rp_kmeans* h, uint8_t* j, float* upper, float* lower);
/* centroids as integer sums: counts[K*bins] (u32), weight[K] (u64) (histogram.rs:286-294, bins.rs:75-82) */
RP_API int rp_kmeans_centroids(rp_kmeans* h, uint32_t* counts, uint64_t* weight);
/* Layer::metric (layer.rs:85-101) + Metric::from(BTreeMap) normalisation (metric.rs:127-141): tri[K*(K-1)/2] */
RP_API int rp_kmeans_metric(rp_kmeans* h, float* tri);
/* Elkan::rms (elkan.rs:191-200), summed in point order */
RP_API int rp_kmeans_rms(rp_kmeans* h, float* out);
/* number of distance evaluations so far (Sinkhorn/variation calls incl. self terms), for reporting */
RP_API int rp_kmeans_stats(rp_kmeans* h, uint64_t* distances, uint64_t* sinkhorn_iterations);
/* exp evaluations spent in Sinkhorn softmin / cost loops so far: sum over solves of (2*iterations + 1) * m * n */
RP_API int rp_kmeans_exp_evals(rp_kmeans* h, uint64_t* evals);
/* The MFMA Sinkhorn bound in front of the N x K neighbor passes of a Sinkhorn layer (init_bounds, assign, step_naive):
 * a scaling-domain iteration u = mu ./ (K v), v = nu ./ (K^T u), K = exp(-C/T), on v_mfma_f32_16x16x4_f32 gives every
 * (point, centroid) pair an interval that contains the value Sinkhorn::divergence (sinkhorn.rs:166-171) returns; the
 * centroids whose lower bound exceeds the smallest upper bound are discarded and the bit-faithful kernel runs on the
 * survivors only, in ascending centroid order — buckets, distances and tie-breaks are those of the unpruned loop
 * (elkan.rs:68-77).  Environment: RP_LLOYD_NO_MFMA_BOUND=1 switches it off; RP_LLOYD_AUDIT=1 also runs the unpruned
 * pass and counts the points on which the two disagree (audit_mismatches; must stay 0). */
typedef struct rp_prune_stats {
    uint32_t enabled;          /* 0: the layer runs every distance through the bit-faithful kernel */
    uint32_t reserved;
    uint64_t points;           /* points that went through the bound (all neighbor passes so far) */
    uint64_t candidates;       /* points * K */
    uint64_t survivors;        /* (point, centroid) pairs handed to the bit-faithful kernel */
    uint64_t block_iterations; /* MFMA work: iterations of a 16-centroid column block (each 2 * 16 * ceil(n/16) * 4 MFMAs of 2048 flop) */
    uint64_t cost_passes;      /* extra K.*C contractions for the cost of an iterate inside the stopping window */
    uint64_t mfma_instructions;/* v_mfma_f32_16x16x4_f32 issued per wavefront, summed (2048 flop each) */
    uint64_t audited_points;   /* RP_LLOYD_AUDIT: points compared with the unpruned pass */
    uint64_t audit_mismatches; /* ... and how many differed in bucket or distance bits */
    uint64_t sampled_points;   /* the production self-check: every 521st point is searched again WITHOUT the prune after every */
    uint64_t sample_mismatches;/* pruned pass; a mismatch fails the next call that hands results out (RP_ERR_INTERNAL) */
    /* the second k-means++ filter (csrc/kpp_bound.hpp; layer.rs:170-178 needs d(c_k, x) only where d^2 < potential): */
    uint64_t kpp_bound_pairs;      /* (new centroid, point) pairs the column-marginal bound let through and the interval examined */
    uint64_t kpp_bound_kept;       /* ... of which the bit-faithful solve was still run */
    uint64_t kpp_bound_iterations; /* scaling-domain iterations over all examined pairs */
    uint64_t kpp_bound_cost_passes;/* cost evaluations inside the stopping windows */
    uint64_t column_iterations;    /* MFMA bound: iterations summed over single centroid columns (a block of 16 runs until its slowest) */
    /* the reference-seed k-means++ draw (csrc/kpp_refpick.hpp; WeightedIndex<f32>'s sequential running sums, layer.rs:160-166): */
    uint64_t ref_pick_chunks;      /* 256-term chunks over the layer's K draws */
    uint64_t ref_pick_walked;      /* ... of which were walked term by term (first chunk, binade crossings, ties); the others are one exact add */
} rp_prune_stats;
RP_API int rp_kmeans_prune_stats(rp_kmeans* h, rp_prune_stats* out);
/* the same with the size of the CALLER's struct: a host compiled against an older (shorter) rp_prune_stats passes its own sizeof
 * and gets the fields it knows; bytes beyond this library's struct are zeroed.  Prefer this one from plain-C hosts. */
RP_API int rp_kmeans_prune_stats_sized(rp_kmeans* h, void* out, size_t out_bytes);
RP_API int rp_kmeans_set_stream(rp_kmeans* h, void* hip_stream);

/* ---- multi-GPU: points sharded by rank, integer centroid sums all-reduced ----------
 * step_local(): pairwise + bound refresh on this rank's points, then partial centroid sums into
 * partial_dev (K*bins u32 counts, K u64 weights, K u64 sizes: partial_bytes()).  After an
 * all-reduce(sum) over ranks, step_finish() installs the centroids, computes drift and updates bounds. */
RP_API int rp_kmeans_partial_bytes(rp_kmeans* h, size_t* bytes);
/* Kmeans::next of a point-sharded job over an rp_comm: step_local, ncclAllReduce(sum) of the integer centroid sums
 * (u32 block and u64 block of the partial, in place, on the layer's stream), step_finish. */
RP_API int rp_kmeans_step_comm(rp_kmeans* h, rp_comm* c, float* drift, uint64_t* sizes, double* reassigned);
RP_API int rp_kmeans_step_local(rp_kmeans* h, void* partial_dev);
RP_API int rp_kmeans_step_finish(rp_kmeans* h, const void* reduced_dev, float* drift, uint64_t* sizes,
                                 double* reassigned);

/* Sinkhorn::divergence (sinkhorn.rs:166-171) / Metric::emd for P independent pairs:
 * mu[P*bins], nu[P*bins] u32 counts (host), out[P].  One wavefront per pair. */
RP_API int rp_sinkhorn_divergence(uint32_t bins, uint64_t pairs, const uint32_t* mu, const uint32_t* nu,
                                  const float* tri_metric, const rp_sinkhorn_hp* hp, int device, float* out);
/* Coupling::minimize().cost() (sinkhorn.rs:194-218): raw entropic OT cost, and iterations used */
RP_API int rp_sinkhorn_cost(uint32_t bins, uint64_t pairs, const uint32_t* mu, const uint32_t* nu,
                            const float* tri_metric, const rp_sinkhorn_hp* hp, int device, float* out,
                            uint32_t* iterations);
/* impl Coupling for Sinkhorn (monge/src/coupling.rs:23-51; lloyd/src/sinkhorn.rs:194-218): minimize() one pair, then
 * flow(x, y) = coupling(x, y) * raw_distance(x, y) with coupling = exp(lhs(x) + rhs(y) - C/T) (sinkhorn.rs:114-116).
 * flow[bins*bins] row-major over (x, y), 0 outside supp(mu) x supp(nu); coupling (same shape) may be NULL.  The x-major
 * left fold of `flow` is cost() (sinkhorn.rs:206-217) bit for bit. */
RP_API int rp_sinkhorn_flow(uint32_t bins, const uint32_t* mu, const uint32_t* nu, const float* tri_metric,
                            const rp_sinkhorn_hp* hp, int device, float* flow, float* coupling);
/* Equity::variation (equity.rs:41-53) for P pairs of `bins`-bin histograms (bins = 101 on the turn layer) */
RP_API int rp_equity_variation(uint32_t bins, uint64_t pairs, const uint32_t* x, const uint32_t* y, int device,
                               float* out);

/* =================================================================================================
 * Abstraction inputs: what feeds the k-means layers above.
 *   crates/deuce/src — cards, hand strength, river equity, suit isomorphism and its iterator;
 *   crates/lloyd/src/lookup.rs — the (isomorphism -> abstraction) table and its projection onto the
 *   previous street's histograms (the k-means points).
 * Encodings are the reference's: card = rank * 4 + suit (card.rs:16-20; rank 0 = Two .. 12 = Ace, suit
 * c, d, h, s = 0..3); Hand = u64 bit set of cards (hand.rs:7); Observation as i64 = one byte
 * (card + 1) per card, public cards then pocket cards, each ascending, first card most significant
 * (observation.rs:132-165) — the `obs BIGINT` column of the reference's tables.
 * `*_dev` arguments are DEVICE pointers the caller has finished writing; calls return after the
 * device finished.  Bulk work shards across GPUs by pocket range / observation slice, no exchange.
 * ================================================================================================= */
typedef enum rp_street { RP_STREET_PREF = 0, RP_STREET_FLOP = 1, RP_STREET_TURN = 2, RP_STREET_RIVE = 3 } rp_street;

/* Strength::from(Hand) (strength.rs:18-32 = Evaluator::find_ranking + find_kickers, evaluator.rs:38-72) for n
 * hands of 5..7 cards (host arrays).  key = variant << 21 | rank1 << 17 | rank2 << 13 | kickers, variant in the
 * default build's Ranking order (ranking.rs:17-29: HighCard 0 .. StraightFlush 8; Flush above FullHouse as
 * declared there): integer order of keys == the reference's derived Ord on Strength. */
RP_API int rp_hand_strength(int device, uint64_t n, const uint64_t* hands, uint32_t* keys);
/* i64::from(Isomorphism::from(Observation::from(obs))) (isomorphism.rs:8-14, permutation.rs:9-71), host arrays */
RP_API int rp_obs_canonical(int device, uint64_t n, const int64_t* obs, int64_t* canon);
/* IsomorphismIterator::from(street) (isomorphism_iter.rs:7-26 over observation_iter.rs:13-104), restricted to the
 * pockets numbered [pocket_lo, pocket_hi) of the 1326 two-card hands in ascending bit-set order (the iterator's outer
 * loop).  Writes the canonical observations, in the iterator's order, to obs_dev[0..min(*n, cap)) and the number
 * there are to *n; obs_dev may be NULL to count.  All pockets: 169 / 1 286 792 / 13 960 050 / 123 156 254
 * (street.rs:120-127). */
RP_API int rp_isomorphisms(int device, int street, uint32_t pocket_lo, uint32_t pocket_hi, int64_t* obs_dev,
                           uint64_t cap, uint64_t* n);
/* Observation::equity (observation.rs:45-63) of n river observations: wins / (wins + losses) over the 990 opposing
 * holes, 0.5 when every showdown ties; bucket = Abstraction::from(Probability)'s index (kicker/src/abstraction.rs:
 * 61-63,93-99: round(p * 100)) = Lookup::grow(Street::Rive) (lookup.rs:172-178).  Either output may be NULL. */
RP_API int rp_river_equity(int device, uint64_t n, const int64_t* obs_dev, float* equity_dev, uint8_t* bucket_dev);

/* Lookup (lookup.rs:9-25): `n` isomorphisms of one street with their abstraction indices, in
 * IsomorphismIterator order (checked).  The table is copied. */
typedef struct rp_lookup rp_lookup;
RP_API int rp_lookup_create(int device, int street, uint64_t n, const int64_t* obs_dev, const uint8_t* abs_dev,
                            rp_lookup** out);
RP_API int rp_lookup_destroy(rp_lookup* h);
/* Lookup::lookup(&Isomorphism::from(obs)) for n observations of the table's street (any suit labelling).  A miss
 * is RP_ERR_INVALID (the reference panics, lookup.rs:24). */
RP_API int rp_lookup_get(rp_lookup* h, uint64_t n, const int64_t* obs_dev, uint8_t* abs_dev);
/* Lookup::projections / future (lookup.rs:27-45): for each of n observations of the PREVIOUS street, the histogram
 * of the table's abstractions over its children (Observation::children, observation.rs:35-40: 47 turns of a flop,
 * 46 rivers of a turn), Histogram::from(Vec<Abstraction>) (histogram.rs:207-212).  hist_dev[n][bins] u8 — the
 * `counts` layout of rp_kmeans_create_device. */
RP_API int rp_lookup_project(rp_lookup* h, uint64_t n, const int64_t* obs_dev, uint32_t bins, uint8_t* hist_dev);
/* device time of the calling thread's last call in this section, from HIP events around its launches */
RP_API int rp_deuce_kernel_ms(double* ms);

/* =================================================================================================
 * Artifact files: what the reference streams to PostgreSQL, written as files in the same
 * byte format — daybook::Streamable::stream (daybook/src/traits/streamable.rs:36-46) over
 * tokio_postgres::binary_copy::BinaryCopyInWriter, i.e. PostgreSQL's binary COPY format: load with
 *     COPY isomorphism (obs, abs) FROM '<file>' (FORMAT binary)       etc.
 * Host code, host arrays.
 * ================================================================================================= */
/* Generic writer / reader, struct-of-arrays.  `types`: one character per column — 'h' int2, 'i' int4, 'q' int8,
 * 'f' float4 (the shapes of daybook/src/traits/row.rs:21-57 are "qh", "if", "hhf", "qhqqfffi").  columns[c] points
 * at n_rows values of column c.  read(): fills up to `cap` rows (columns may be NULL to count) and reports *n_rows. */
RP_API int rp_pgcopy_write(const char* path, const char* types, uint64_t n_rows, const void* const* columns);
RP_API int rp_pgcopy_read(const char* path, const char* types, uint64_t cap, void* const* columns, uint64_t* n_rows);
/* Lookup rows (lloyd/src/lookup.rs:141-147): (obs i64, abs i16 = street << 8 | index) in table order */
RP_API int rp_artifact_write_lookup(const char* path, int street, uint64_t n, const int64_t* obs,
                                    const uint8_t* abs_index);
/* Metric rows (lloyd/src/metric.rs:219-226, distances.rs:69-84): (tri i32 = street << 30 | t, dx f32), t ascending;
 * tri[] as rp_kmeans_metric returns it */
RP_API int rp_artifact_write_metric(const char* path, int street, uint32_t K, const float* tri);
/* Future rows (lloyd/src/future.rs:99-111): per abstraction, its centroid's distribution() (bins.rs:113-117: density
 * descending, stable) as (prev i16, next i16, dx f32); counts/weight as rp_kmeans_centroids returns them */
RP_API int rp_artifact_write_transitions(const char* path, int street, uint32_t K, uint32_t bins,
                                         const uint32_t* counts, const uint64_t* weight);
/* NlheProfile::rows (nlhe/src/profile.rs:144-163) as the blueprint table's COPY file: columns (past BIGINT, present SMALLINT,
 * choices BIGINT, edge BIGINT, weight REAL, regret REAL, payoff REAL, visits INTEGER) (profile.rs:20-31), one row per
 * (infoset, edge of its choices); edge = From<Edge> for u64 (kicker/src/edge.rs:122-160).  Input: rp_nlhe_export's arrays.
 * Read back with rp_pgcopy_read(path, "qhqqfffi", ...) and rp_nlhe_import (Hydrate, profile.rs:90-141). */
RP_API int rp_artifact_write_blueprint(const char* path, uint64_t n_infosets, const uint64_t* past, const uint32_t* present,
                                       const uint64_t* choices, const rp_encounter* enc, int only_visited, uint64_t* rows_written);

/* =================================================================================================
 * The NLHE rules engine on the device (first device step).  The betting state machine of
 * kicker::GameN<P> (crates/kicker/src/game.rs), the action abstraction (edge.rs, size.rs: Pluribus grids) with
 * NlheGame::apply's actionize + snap (crates/nlhe/src/game.rs:33-53) and Showdown::settle (showdown.rs) run as
 * device functions; this entry point plays `n_games` random abstract hands of `n_players` (2..10, 200-chip stacks,
 * blinds 1 / 2, dealer = game % n_players), one lane per game: deals and choices from rp_node_hash(seed, 0, game,
 * counter) (rp_math.h).  payoffs_dev[game][seat] = NlheGame::payoff (settlement.won()), digests_dev[game] folds every
 * intermediate state, steps_dev[game] = actions taken (0xffffffff if the hand did not finish in max_steps).  The
 * MCCFR traversal over this engine is not built yet.
 * ================================================================================================= */
RP_API int rp_nlhe_playouts(int device, uint32_t n_players, uint64_t n_games, uint64_t seed, uint32_t max_steps,
                            float* payoffs_dev, uint64_t* digests_dev, uint32_t* steps_dev);

#ifdef __cplusplus
}
#endif
#endif /* RP_MI355X_H */

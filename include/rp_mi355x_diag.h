/* rp_mi355x_diag.h — diagnostics of librp_mi355x.so: device self tests of the arithmetic contract and of the sort / scan primitives,
 * HIP-event kernel clocks, the NLHE traversal's shape and node census, the MFMA Sinkhorn bound's intervals.  Tests and bench.py call
 * these; a drop-in caller of the hot paths (include/rp_mi355x.h) needs none of them.  Same conventions: int status, rp_last_error(). */
#ifndef RP_MI355X_DIAG_H
#define RP_MI355X_DIAG_H

#include "rp_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Arithmetic-contract self test: evaluates rp_math.h's primitives ON THE DEVICE for n input pairs so a
 * caller can compare them bit for bit with a host evaluation of the same header.
 * out[0*n..] = rp_expf(x), [1*n..] = rp_logf(|x|), [2*n..] = x / y, [3*n..] = sqrtf(|x|),
 * [4*n..] = fmaf(x, y, x), [5*n..] = (float)(uint32)|x| as u32->f32 conversion of y's bits. */
RP_API int rp_math_selftest(int device, uint64_t n, const float* x, const float* y, float* out);
/* Sweeps ALL 2^32 f32 bit patterns on the device and counts where the gfx950 spellings of exp differ from the
 * contract's spec sequence (rp_expf_spec): mismatches[0] rp_expf, [1] rp_exp_floor vs max(spec, MIN_POSITIVE),
 * [2] rp_exp_floor2 (packed), [3] smallest mismatching bit pattern (~0 if none).  All counts must be 0. */
RP_API int rp_math_exp_sweep(int device, uint64_t* mismatches);
/* include/rp_libm_glibc.h's expf / logf evaluated ON THE DEVICE over the f32 bit patterns [lo, hi) (a multiple of 65536 of them; the
 * whole range is [0, 2^32)): sums[0] = sum of expf's result bits, [1] = sum of result bits x (2 u + 1) for input pattern u, [2], [3]
 * the same for logf; wrapping u64 arithmetic, NaN results counted as 0x7fc00000.  A host evaluation of the same header gives the
 * same four numbers iff the two agree on every input (tests/test_gpu_z_glibc_mode.py). */
RP_API int rp_libm_glibc_sweep(int device, uint64_t lo, uint64_t hi, uint64_t* sums);
/* The forms of the same two functions that the lloyd kernels' glibc pass evaluates (branch-free, tables in LDS:
 * csrc/lm_glibc_dev.hpp, include/rp_libm_glibc.h's rp_glibc_expf_tab / rp_glibc_exp_floor_tab / rp_glibc_logf_tab) against the ladder
 * forms the sweep above pins to the host, on the device, over the same kind of range: mismatches[0] expf, [1] max(expf, MIN_POSITIVE),
 * [2] logf, [3] the smallest mismatching bit pattern (~0 if none).  All counts must be 0. */
RP_API int rp_libm_glibc_tab_sweep(int device, uint64_t lo, uint64_t hi, uint64_t* mismatches);
/* The device-wide primitives under the row-addressed profile and the isomorphism enumeration (csrc/sortscan.hpp: stable LSD
 * radix sort of (key, index) pairs by the low `bits` bits of the key, run-length encoding of the sorted keys, exclusive
 * scan), run on n host keys so a test can compare them with a host sort: sorted_keys / perm [n]; uniq / starts / counts
 * [n] of which the first *n_runs are set; scan[i] = sum of keys[0..i) as u64. */
RP_API int rp_sortscan_selftest(int device, uint32_t n, uint32_t bits, const uint32_t* keys, uint32_t* sorted_keys, uint32_t* perm,
                                uint32_t* uniq, uint32_t* starts, uint32_t* counts, uint32_t* n_runs, uint64_t* scan);

/* ---- profiling hooks used by bench.py (HIP events on the launch stream) ------------------------- */
RP_API int rp_mccfr_profile(rp_mccfr* h, int enable);
/* name in {"traverse","compact","update"}; total milliseconds and launch count since profiling was enabled */
RP_API int rp_mccfr_kernel_time(rp_mccfr* h, const char* name, double* total_ms, uint64_t* launches);

/* name in {"sort","apply"}: HIP-event milliseconds since rp_profile_profile(h, 1) */
RP_API int rp_profile_profile(rp_profile* h, int enable);
RP_API int rp_profile_kernel_time(rp_profile* h, const char* name, double* total_ms, uint64_t* launches);

/* levels grown and nodes created by the last traversed batch (diagnostics of the level-synchronous traversal) */
RP_API int rp_nlhe_last_shape(rp_nlhe* h, uint32_t* levels, uint32_t* nodes);
/* profiling hooks used by bench.py (HIP events on the launch stream); name in {"expand","children","sweeps","decide",
 * "apply"}: total milliseconds and launches since profiling was enabled; census: nodes of those steps by kind {terminal,
 * chance, walker, opponent} and the children of their walker nodes (what k_nl_expand's algorithmic bytes are counted from) */
RP_API int rp_nlhe_profile(rp_nlhe* h, int enable);
RP_API int rp_nlhe_kernel_time(rp_nlhe* h, const char* name, double* total_ms, uint64_t* launches);
RP_API int rp_nlhe_census(rp_nlhe* h, uint64_t* kinds4, uint64_t* walker_children);

/* the divergence intervals of the bound against the current centroids: lo[N*K], hi[N*K] (tests / diagnostics) */
RP_API int rp_kmeans_bound_intervals(rp_kmeans* h, float* lo, float* hi);
/* the second k-means++ filter (csrc/kpp_bound.hpp): lo[N] = its lower bound of distance(centroid k, point i) (Elkan::neighbor's
 * centroid-first call, elkan.rs:68-77) for every point, each stopping window followed to its end; 0 where the pair does not fit the
 * register tile (either support above 48 bins) or no bound was obtained.  Tests compare it with the bit-faithful distances. */
RP_API int rp_kmeans_kpp_bound_probe(rp_kmeans* h, uint32_t k, float* lo);
/* the same filter as production runs it against ONE potential for every point (potential >= 0: a pair leaves as soon as its bound squared
 * reaches it — the window's bound or, from the second iteration on, the Kantorovich dual bound of csrc/kpp_bound.hpp); lo[i] = the
 * lower bound the pair left with, 0 where it was kept for the bit-faithful solve.  potential < 0: rp_kmeans_kpp_bound_probe. */
RP_API int rp_kmeans_kpp_bound_probe_at(rp_kmeans* h, uint32_t k, float potential, float* lo);
/* The interval-decided refresh of the Elkan iterations (csrc/refresh_bound.hpp; rp_kmeans_set_prune(h, 0) switches it off with the other
 * filters).  out8: [0] refreshes examined by the interval kernel, [1] settled by it (no bit-faithful solve), [2] scaling-domain
 * iterations, [3] cost evaluations inside the stopping windows, [4] exactify solves (an interval of the previous step replaced by its
 * exact value: a solve the reference does not repeat, not part of rp_kmeans_stats' distance count), [5] 1 if the layer uses it,
 * [6] examined refreshes whose interval was remembered (the centroid's content had not changed: no iteration), [7] reserved.
 * The reference's own distance count of the Elkan steps is evaluated + remembered (rp_kmeans_stats_ex) + out8[1]. */
RP_API int rp_kmeans_refresh_stats(rp_kmeans* h, uint64_t* out8);
/* While uiv[i] != 0 the upper bound rp_kmeans_bounds returns for point i is the upper end of an interval [ulo[i], upper[i]] that
 * contains the reference's Bounds::error, and lower[i][bucket i] holds its lower end; where uiv[i] == 0 all three are the
 * reference's values bit for bit (ulo[i] is then not meaningful).  Without the refresh bound: uiv = 0 everywhere. */
RP_API int rp_kmeans_upper_interval(rp_kmeans* h, float* ulo, uint8_t* uiv);
/* pairw[K*K]: the centroid-to-centroid distances (Elkan::pairwises, elkan.rs:83-88; unnormalised, row i = distance(c_i, c_k)) that the
 * last rp_kmeans_step worked with — those of the centroids that were current when it began.  RP_ERR_INVALID before the first step. */
RP_API int rp_kmeans_pairwise_last(rp_kmeans* h, float* pairw);
/* rand 0.9.2's WeightedIndex::new(weights).sample() on n host weights for a given value0_1 (the UniformFloat draw in [0, 1)): the
 * reference-seed k-means++ draw (crates/lloyd/src/layer.rs:160-166) in isolation.  mode 0: one wavefront walks the n dependent f32
 * additions (round 5's kernel, kept as the checker); mode 1: the chunked walk the layer uses (csrc/kpp_refpick.hpp), exact by
 * construction.  out[0] = the index (n if the total is 0), out[1] = the bits of the total weight, out[2] = chunks of 256 terms that
 * mode 1 walked term by term (ties, binade crossings, mispredictions).  Both modes must agree with a host loop on every input. */
RP_API int rp_weighted_index_probe(int device, uint64_t n, const float* weights, float v01, int mode, uint64_t* out);
/* the layer's raw counters, out[5]: [0] distances evaluated (rp_kmeans_stats), [1] Sinkhorn iterations, [2] softmin + cost terms,
 * [3] distances the reference evaluates at that point of Elkan::step_elkan (elkan.rs:153-168) and this library remembers instead of
 * solving again (same centroid content, same point: the value is a pure function of the two), [4] variation distances computed
 * beyond the ones Elkan's rule evaluates (the turn kernels compute whole 64-centroid tiles).  Over Elkan steps [0] + [3] is the
 * reference's own distance count (tests/test_gpu_lloyd.py::test_elkan_iterations_bit_exact). */
RP_API int rp_kmeans_stats_ex(rp_kmeans* h, uint64_t* out5);

RP_API int rp_kmeans_profile(rp_kmeans* h, int enable);
/* name in {"pairwise","step","recompute","bounds","neighbor","selfcost","kpp","drift","mfma_bound","kpp_bound"} */
RP_API int rp_kmeans_kernel_time(rp_kmeans* h, const char* name, double* total_ms, uint64_t* launches);

#ifdef __cplusplus
}
#endif

#endif /* RP_MI355X_DIAG_H */

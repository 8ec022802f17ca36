// TEMPORARY stubs (replaced as the device paths land)
#include "rp_internal.h"
extern "C" {
int rp_kmeans_create(uint32_t K, uint64_t N, uint32_t bins, const uint8_t* counts, rp_metric_kind kind, const float* tri_metric, const rp_sinkhorn_hp* hp, uint64_t seed, int device, rp_kmeans** out) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_create: not implemented yet"); }
int rp_kmeans_create_device(uint32_t K, uint64_t N, uint32_t bins, const void* counts_dev, rp_metric_kind kind, const float* tri_metric, const rp_sinkhorn_hp* hp, uint64_t seed, int device, rp_kmeans** out) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_create_device: not implemented yet"); }
int rp_kmeans_destroy(rp_kmeans* h) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_destroy: not implemented yet"); }
int rp_kmeans_init_centroids(rp_kmeans* h, uint64_t* chosen) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_init_centroids: not implemented yet"); }
int rp_kmeans_set_centroids(rp_kmeans* h, const uint64_t* point_index) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_set_centroids: not implemented yet"); }
int rp_kmeans_init_bounds(rp_kmeans* h) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_init_bounds: not implemented yet"); }
int rp_kmeans_step(rp_kmeans* h, float* drift, uint64_t* sizes, double* reassigned) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_step: not implemented yet"); }
int rp_kmeans_step_naive(rp_kmeans* h) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_step_naive: not implemented yet"); }
int rp_kmeans_assign(rp_kmeans* h, uint8_t* bucket, float* distance) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_assign: not implemented yet"); }
int rp_kmeans_bounds(rp_kmeans* h, uint8_t* j, float* upper, float* lower) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_bounds: not implemented yet"); }
int rp_kmeans_centroids(rp_kmeans* h, uint32_t* counts, uint64_t* weight) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_centroids: not implemented yet"); }
int rp_kmeans_metric(rp_kmeans* h, float* tri) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_metric: not implemented yet"); }
int rp_kmeans_rms(rp_kmeans* h, float* out) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_rms: not implemented yet"); }
int rp_kmeans_stats(rp_kmeans* h, uint64_t* distances, uint64_t* sinkhorn_iterations) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_stats: not implemented yet"); }
int rp_kmeans_set_stream(rp_kmeans* h, void* hip_stream) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_set_stream: not implemented yet"); }
int rp_kmeans_profile(rp_kmeans* h, int enable) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_profile: not implemented yet"); }
int rp_kmeans_kernel_time(rp_kmeans* h, const char* name, double* total_ms, uint64_t* launches) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_kernel_time: not implemented yet"); }
int rp_kmeans_partial_bytes(rp_kmeans* h, size_t* bytes) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_partial_bytes: not implemented yet"); }
int rp_kmeans_step_local(rp_kmeans* h, void* partial_dev) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_step_local: not implemented yet"); }
int rp_kmeans_step_finish(rp_kmeans* h, const void* reduced_dev, float* drift, uint64_t* sizes, double* reassigned) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_kmeans_step_finish: not implemented yet"); }
int rp_sinkhorn_divergence(uint32_t bins, uint64_t pairs, const uint32_t* mu, const uint32_t* nu, const float* tri_metric, const rp_sinkhorn_hp* hp, int device, float* out) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_sinkhorn_divergence: not implemented yet"); }
int rp_sinkhorn_cost(uint32_t bins, uint64_t pairs, const uint32_t* mu, const uint32_t* nu, const float* tri_metric, const rp_sinkhorn_hp* hp, int device, float* out, uint32_t* iterations) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_sinkhorn_cost: not implemented yet"); }
int rp_equity_variation(uint32_t bins, uint64_t pairs, const uint32_t* x, const uint32_t* y, int device, float* out) { return rp::fail(RP_ERR_UNSUPPORTED, "rp_equity_variation: not implemented yet"); }
}

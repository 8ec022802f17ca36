// common.cpp — error reporting, version and device probing for librp_mi355x.so
#include "rp_internal.h"

#include <cstdarg>
#include <hip/hip_runtime_api.h>

namespace rp {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// Checkpoint's Display / Progress::format (metrics/checkpoint.rs:39-50, progress.rs:8-18): four 20-column fields
void format_progress(char* buf, size_t cap, uint64_t epoch, uint64_t nodes, uint64_t infos, double rate) {
    char f[4][48];
    snprintf(f[0], sizeof f[0], "batch %llu", (unsigned long long)epoch);
    snprintf(f[1], sizeof f[1], "nodes %llu", (unsigned long long)nodes);
    snprintf(f[2], sizeof f[2], "infos %llu", (unsigned long long)infos);
    snprintf(f[3], sizeof f[3], "I/sec %.1f", rate);
    snprintf(buf, cap, "%-20s%-20s%-20s%-20s", f[0], f[1], f[2], f[3]);
}

}  // namespace rp

extern "C" {

const char* rp_last_error(void) { return rp::g_err; }

int rp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char* rp_version(void) { return "rp_mi355x 0.3.0 gfx950 hip " RP_STR(HIP_VERSION_MAJOR) "." RP_STR(HIP_VERSION_MINOR); }

void rp_hyper_default(rp_hyper* out) {
    if (!out) return;
    memset(out, 0, sizeof(*out));
    out->temperature = 1.0f;
    out->smoothing = 2.0f;
    out->curiosity = 0.05f;
    out->prune_threshold = -3e5f;
    out->prune_explore = 0.05f;
    out->prune_warmup = 16384;
    out->regret_min = -4e6f;
}

void rp_sinkhorn_hp_default(rp_sinkhorn_hp* out) {
    if (!out) return;
    out->temperature = 0.025f;
    out->iterations = 128;
    out->tolerance = 0.0005f;
}

}  // extern "C"

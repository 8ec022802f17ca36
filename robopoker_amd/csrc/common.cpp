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

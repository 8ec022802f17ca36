// selftest.hip — device-side evaluation of the arithmetic contract (include/rp_math.h) for parity tests.
#include <hip/hip_runtime.h>

#include "../../include/rp_math.h"
#include "rp_internal.h"

namespace rp {
__global__ void k_math_selftest(uint64_t n, const float* x, const float* y, float* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[i], b = y[i];
    out[0 * n + i] = rp_expf(a);
    out[1 * n + i] = rp_logf(rp_absf(a));
    out[2 * n + i] = a / b;
    out[3 * n + i] = sqrtf(rp_absf(a));
    out[4 * n + i] = fmaf(a, b, a);
    out[5 * n + i] = (float)rp_f2u(b);
}
}  // namespace rp

extern "C" int rp_math_selftest(int device, uint64_t n, const float* x, const float* y, float* out) {
    if (!x || !y || !out || n == 0) return rp::fail(RP_ERR_INVALID, "rp_math_selftest: bad argument");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_math_selftest: no HIP device");
#define ST_TRY(e)                                                                                   \
    do {                                                                                            \
        hipError_t _e = (e);                                                                        \
        if (_e != hipSuccess) return rp::fail(RP_ERR_HIP, "%s failed: %s", #e, hipGetErrorString(_e)); \
    } while (0)
    ST_TRY(hipSetDevice(device));
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    ST_TRY(hipMalloc(&dx, n * 4));
    ST_TRY(hipMalloc(&dy, n * 4));
    ST_TRY(hipMalloc(&dout, n * 4 * 6));
    ST_TRY(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
    ST_TRY(hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rp::k_math_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, dx, dy, dout);
    ST_TRY(hipGetLastError());
    ST_TRY(hipMemcpy(out, dout, n * 4 * 6, hipMemcpyDeviceToHost));
    (void)hipFree(dx);
    (void)hipFree(dy);
    (void)hipFree(dout);
    return RP_OK;
}

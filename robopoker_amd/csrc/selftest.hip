// selftest.hip — device-side evaluation of the arithmetic contract (include/rp_math.h) for parity tests.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "../../include/rp_math.h"
#include "../../include/rp_libm_glibc.h"
#include "lm_glibc_dev.hpp"
#include "rp_internal.h"
#include "sortscan.hpp"

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

// every f32 bit pattern: the gfx950 spellings of exp against the contract's spec sequence, bit for bit
__global__ void k_exp_sweep(unsigned long long* bad) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;  // 2^24 threads x 256 patterns
    unsigned long long b0 = 0, b1 = 0, b2 = 0;
    uint32_t first = 0xffffffffu;
    for (uint32_t k = 0; k < 256; k += 2) {
        const uint32_t u0 = (k << 24) | tid, u1 = ((k + 1) << 24) | tid;
        const float x0 = rp_u2f(u0), x1 = rp_u2f(u1);
        const float s0 = rp_expf_spec(x0), s1 = rp_expf_spec(x1);
        const bool e0 = rp_f2u(rp_expf(x0)) != rp_f2u(s0), e1 = rp_f2u(rp_expf(x1)) != rp_f2u(s1);
        b0 += e0 + e1;
        const float f0 = rp_maxf(s0, RP_EPSILON), f1 = rp_maxf(s1, RP_EPSILON);
        const bool g0 = rp_f2u(rp_exp_floor(x0)) != rp_f2u(f0), g1 = rp_f2u(rp_exp_floor(x1)) != rp_f2u(f1);
        b1 += g0 + g1;
        rp_f2 v;
        v.x = x0;
        v.y = x1;
        const rp_f2 r = rp_exp_floor2(v);
        const bool h0 = rp_f2u(r.x) != rp_f2u(f0), h1 = rp_f2u(r.y) != rp_f2u(f1);
        b2 += h0 + h1;
        if (e0 || g0 || h0) first = min(first, u0);
        if (e1 || g1 || h1) first = min(first, u1);
    }
    if (b0) atomicAdd(&bad[0], b0);
    if (b1) atomicAdd(&bad[1], b1);
    if (b2) atomicAdd(&bad[2], b2);
    if (first != 0xffffffffu) atomicMin(&bad[3], (unsigned long long)first);
}

// include/rp_libm_glibc.h on the device: order-independent checksums of expf and logf over the bit patterns [lo, lo + 256 * threads)
// (thread t takes lo + 256 t .. + 255): sums of the result bits and of the result bits times the odd number 2 u + 1 of the input,
// NaN results canonicalised (which payload an x + x keeps is the hardware's business).  The host computes the same four numbers.
__global__ void k_glibc_sweep(uint64_t lo, unsigned long long* acc) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long se = 0, we = 0, sl = 0, wl = 0;
    for (uint32_t k = 0; k < 256; ++k) {
        const uint32_t u = (uint32_t)(lo + t * 256u + k);
        const float x = rp_u2f(u);
        const float e = rp_glibc_expf(x), l = rp_glibc_logf(x);
        const uint64_t be = e != e ? 0x7fc00000u : rp_f2u(e), bl = l != l ? 0x7fc00000u : rp_f2u(l);
        const uint64_t odd = 2ull * u + 1ull;
        se += be;
        we += be * odd;
        sl += bl;
        wl += bl * odd;
    }
    atomicAdd(&acc[0], se);
    atomicAdd(&acc[1], we);
    atomicAdd(&acc[2], sl);
    atomicAdd(&acc[3], wl);
}
// the forms the lloyd kernels evaluate (branch-free, tables in LDS: lm_glibc_dev.hpp) against the header's ladder forms above, on the
// device, over [lo, lo + 256 * threads): bad[0] expf, [1] max(expf, MIN_POSITIVE), [2] logf (both NaN counts as equal), [3] the
// smallest mismatching pattern
__global__ void k_glibc_tab_sweep(uint64_t lo, unsigned long long* bad) {
    lmg::tables_init();
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long b0 = 0, b1 = 0, b2 = 0;
    uint32_t first = 0xffffffffu;
    for (uint32_t k = 0; k < 256; ++k) {
        const uint32_t u = (uint32_t)(lo + t * 256u + k);
        const float x = rp_u2f(u);
        const float e = rp_glibc_expf(x), l = rp_glibc_logf(x);
        const float a = rp_glibc_expf_tab(x, lmg::lds_exp);
        const float c = rp_glibc_exp_floor_tab(x, lmg::lds_exp);
        const float g = rp_glibc_logf_tab(x, lmg::lds_log);
        const bool m0 = rp_f2u(a) != rp_f2u(e) && !(a != a && e != e);
        const bool m1 = rp_f2u(c) != rp_f2u(rp_maxf(e, RP_EPSILON));
        const bool m2 = rp_f2u(g) != rp_f2u(l) && !(g != g && l != l);
        b0 += m0;
        b1 += m1;
        b2 += m2;
        if (m0 || m1 || m2) first = min(first, u);
    }
    if (b0) atomicAdd(&bad[0], b0);
    if (b1) atomicAdd(&bad[1], b1);
    if (b2) atomicAdd(&bad[2], b2);
    if (first != 0xffffffffu) atomicMin(&bad[3], (unsigned long long)first);
}
}  // namespace rp

extern "C" int rp_libm_glibc_sweep(int device, uint64_t lo, uint64_t hi, uint64_t* sums) {
    if (!sums || hi <= lo || hi > (1ull << 32) || ((hi - lo) & 0xffffull)) return rp::fail(RP_ERR_INVALID, "rp_libm_glibc_sweep: the range must be a multiple of 65536 bit patterns inside [0, 2^32]");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_libm_glibc_sweep: no HIP device");
#define GL_TRY(e)                                                                                   \
    do {                                                                                            \
        hipError_t _e = (e);                                                                        \
        if (_e != hipSuccess) return rp::fail(RP_ERR_HIP, "%s failed: %s", #e, hipGetErrorString(_e)); \
    } while (0)
    GL_TRY(hipSetDevice(device));
    unsigned long long* d = nullptr;
    GL_TRY(hipMalloc(&d, 32));
    GL_TRY(hipMemset(d, 0, 32));
    hipLaunchKernelGGL(rp::k_glibc_sweep, dim3((unsigned)((hi - lo) >> 16)), dim3(256), 0, 0, lo, d);
    GL_TRY(hipGetLastError());
    unsigned long long out[4];
    GL_TRY(hipMemcpy(out, d, 32, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    for (int i = 0; i < 4; ++i) sums[i] = out[i];
    return RP_OK;
}

extern "C" int rp_libm_glibc_tab_sweep(int device, uint64_t lo, uint64_t hi, uint64_t* mismatches) {
    if (!mismatches || hi <= lo || hi > (1ull << 32) || ((hi - lo) & 0xffffull)) return rp::fail(RP_ERR_INVALID, "rp_libm_glibc_tab_sweep: the range must be a multiple of 65536 bit patterns inside [0, 2^32]");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_libm_glibc_tab_sweep: no HIP device");
    GL_TRY(hipSetDevice(device));
    unsigned long long* d = nullptr;
    const unsigned long long init[4] = {0, 0, 0, ~0ull};
    GL_TRY(hipMalloc(&d, sizeof(init)));
    GL_TRY(hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rp::k_glibc_tab_sweep, dim3((unsigned)((hi - lo) >> 16)), dim3(256), 0, 0, lo, d);
    GL_TRY(hipGetLastError());
    unsigned long long out[4];
    GL_TRY(hipMemcpy(out, d, sizeof(out), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    for (int i = 0; i < 4; ++i) mismatches[i] = out[i];
    return RP_OK;
}

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

extern "C" int rp_math_exp_sweep(int device, uint64_t* mismatches) {
    if (!mismatches) return rp::fail(RP_ERR_INVALID, "rp_math_exp_sweep: bad argument");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_math_exp_sweep: no HIP device");
    ST_TRY(hipSetDevice(device));
    unsigned long long* d = nullptr;
    const unsigned long long init[4] = {0, 0, 0, ~0ull};
    ST_TRY(hipMalloc(&d, sizeof(init)));
    ST_TRY(hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rp::k_exp_sweep, dim3(1u << 16), dim3(256), 0, 0, d);
    ST_TRY(hipGetLastError());
    unsigned long long out[4];
    ST_TRY(hipMemcpy(out, d, sizeof(out), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    for (int i = 0; i < 4; ++i) mismatches[i] = out[i];
    return RP_OK;
}

// the device-wide primitives of csrc/sortscan.hpp on host arrays: stable sort of (key, index) pairs by the low `bits` bits,
// run-length encoding of the sorted keys, 64-bit exclusive scan of the keys
extern "C" int rp_sortscan_selftest(int device, uint32_t n, uint32_t bits, const uint32_t* keys, uint32_t* sorted_keys, uint32_t* perm,
                                    uint32_t* uniq, uint32_t* starts, uint32_t* counts, uint32_t* n_runs, uint64_t* scan) {
    if (!keys || !sorted_keys || !perm || !uniq || !starts || !counts || !n_runs || !scan)
        return rp::fail(RP_ERR_INVALID, "rp_sortscan_selftest: NULL argument");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_sortscan_selftest: no HIP device visible");
#define SS_TRY(expr)                                                                                    \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            for (void* q : bufs) (void)hipFree(q);                                                      \
            return rp::fail(RP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));                 \
        }                                                                                               \
    } while (0)
    std::vector<void*> bufs;
    SS_TRY(hipSetDevice(device));
    const size_t m = std::max<uint32_t>(n, 1);
    uint32_t *dk, *di, *dko, *dpo, *du, *ds, *dc, *dn, *dw;
    uint64_t* dscan;
    void *tmp, *stmp;
    auto dalloc = [&](void** q, size_t bytes) {
        hipError_t e = hipMalloc(q, bytes);
        if (e == hipSuccess) bufs.push_back(*q);
        return e;
    };
    SS_TRY(dalloc((void**)&dk, m * 4));
    SS_TRY(dalloc((void**)&di, m * 4));
    SS_TRY(dalloc((void**)&dko, m * 4));
    SS_TRY(dalloc((void**)&dpo, m * 4));
    SS_TRY(dalloc((void**)&du, m * 4));
    SS_TRY(dalloc((void**)&ds, m * 4));
    SS_TRY(dalloc((void**)&dc, m * 4));
    SS_TRY(dalloc((void**)&dn, 4));
    SS_TRY(dalloc((void**)&dw, m * 4));
    SS_TRY(dalloc((void**)&dscan, m * 8));
    SS_TRY(dalloc(&tmp, rp::ss::sort_scratch_bytes((uint32_t)m)));
    SS_TRY(dalloc(&stmp, rp::ss::scan_scratch_bytes(m)));
    std::vector<uint32_t> iota(m);
    for (size_t i = 0; i < m; ++i) iota[i] = (uint32_t)i;
    SS_TRY(hipMemcpy(dk, keys, (size_t)n * 4, hipMemcpyHostToDevice));
    SS_TRY(hipMemcpy(di, iota.data(), m * 4, hipMemcpyHostToDevice));
    SS_TRY(rp::ss::sort_pairs(dk, di, dko, dpo, n, bits, tmp, nullptr));
    SS_TRY(rp::ss::run_length_encode(dko, n, du, ds, dc, dn, dw, stmp, nullptr));
    SS_TRY(rp::ss::exclusive_scan<uint64_t>(dk, dscan, n, stmp, nullptr));
    SS_TRY(hipDeviceSynchronize());
    SS_TRY(hipMemcpy(sorted_keys, dko, (size_t)n * 4, hipMemcpyDeviceToHost));
    SS_TRY(hipMemcpy(perm, dpo, (size_t)n * 4, hipMemcpyDeviceToHost));
    SS_TRY(hipMemcpy(n_runs, dn, 4, hipMemcpyDeviceToHost));
    SS_TRY(hipMemcpy(uniq, du, (size_t)n * 4, hipMemcpyDeviceToHost));
    SS_TRY(hipMemcpy(starts, ds, (size_t)n * 4, hipMemcpyDeviceToHost));
    SS_TRY(hipMemcpy(counts, dc, (size_t)n * 4, hipMemcpyDeviceToHost));
    SS_TRY(hipMemcpy(scan, dscan, (size_t)n * 8, hipMemcpyDeviceToHost));
    for (void* q : bufs) (void)hipFree(q);
#undef SS_TRY
    return RP_OK;
}

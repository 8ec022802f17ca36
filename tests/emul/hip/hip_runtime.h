// tests/emul/hip/hip_runtime.h — TEST INFRASTRUCTURE, never part of the product.
//
// A wave64 execution model for the library's kernel SOURCES on the host: tests/emul/build.py compiles
// robopoker_amd/csrc/*.hip as plain C++ against this header instead of <hip/hip_runtime.h>, into
// tests/emul/_build/librp_emul.so (same C-ABI as librp_mi355x.so).  Only tests/ may load that library: it exists to
// check the kernels' LOGIC (indexing, LDS protocols, ballots / shuffles under divergence, atomics, launch
// geometry) against the oracle here, where there is no GPU.  It says nothing about speed and nothing about the
// memory model (x86 is stronger than the device), and robopoker_amd/ never falls back to it.
//
// Model: a workgroup = fibers (one per work-item) on one host thread; a fiber runs until it reaches a wavefront
// collective (ballot, shuffle, readfirstlane, mfma, wave_barrier) or __syncthreads, or returns.  When every lane of
// a wavefront is parked, the lanes parked at the SAME source line and operation form the collective's EXEC mask; when
// they are parked at different collectives only the earliest in the source is settled (a divergent region runs to its
// end before the code after it: lanes that skipped a branch or left a loop wait for the others) — which is what
// structured divergence / reconvergence gives on the hardware.  The workgroups of a launch are spread over host threads
// (RP_EMUL_THREADS), so races between workgroups are real.
#ifndef RP_EMUL_HIP_RUNTIME_H
#define RP_EMUL_HIP_RUNTIME_H

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <type_traits>
#include <utility>

#define RP_EMUL 1
#define __HIP__ 1
#define __HIPCC__ 1
#define __HIP_DEVICE_COMPILE__ 1  // the device formulations of include/rp_math.h are the ones under test
#define __gfx950__ 1

#define __host__
#define __device__
#define __global__
#define __constant__ static const
#define __shared__ static thread_local
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __align__(n) alignas(n)

#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5

// ---- vector types (clang extended vectors: .x/.y/.z/.w, [] and arithmetic, 16-byte alignment like HIP's) ----
typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint2 __attribute__((ext_vector_type(2)));
typedef unsigned int uint4 __attribute__((ext_vector_type(4)));
typedef int int2 __attribute__((ext_vector_type(2)));
typedef int int4 __attribute__((ext_vector_type(4)));
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }

struct dim3 {
    uint32_t x, y, z;
    constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_idx3 {
    uint32_t x, y, z;
};
extern thread_local emu_idx3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;
static const int warpSize = 64;

// ---- runtime API (host memory stands in for HBM; everything is synchronous) ----
typedef int hipError_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipErrorOutOfMemory 2
struct ihipStream_t;
typedef ihipStream_t* hipStream_t;
struct ihipEvent_t;
typedef ihipEvent_t* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
#define hipStreamNonBlocking 1u
#define hipStreamDefault 0u
#define hipEventDisableTiming 2u

extern "C" {
hipError_t emu_hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemset(void* dst, int v, size_t n);
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipDeviceSynchronize(void);
hipError_t hipGetLastError(void);
hipError_t hipPeekAtLastError(void);
const char* hipGetErrorString(hipError_t e);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int dev);
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b);
}
// pinned, mapped host memory: plain host memory here (the "device" reads host addresses); a launch has run when it returns
#define hipHostMallocMapped 2u
#define hipHostMallocCoherent 0x40000000u
#define hipErrorNotReady 600
static inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
    *p = calloc(1, bytes);
    return *p ? hipSuccess : (hipError_t)2;
}
static inline hipError_t hipHostFree(void* p) {
    free(p);
    return hipSuccess;
}
static inline hipError_t hipHostGetDevicePointer(void** dp, void* hp, unsigned) {
    *dp = hp;
    return hipSuccess;
}
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
template <class T>
static inline hipError_t hipMalloc(T** p, size_t bytes) {
    return emu_hipMalloc(reinterpret_cast<void**>(p), bytes);
}

// ---- the execution model (tests/emul/wavesim.cpp) ----
namespace emu {
enum Op : int { OP_BALLOT = 1, OP_SHFL, OP_SHFL_UP, OP_SHFL_DOWN, OP_SHFL_XOR, OP_FIRST, OP_WAVE_BARRIER, OP_MFMA16 };
// parks the calling lane until its wavefront's lanes at (site, op) are known; returns this lane's result
uint64_t collective(int op, int site, uint64_t value, int arg, int width);
void mfma16x16x4(int site, float a, float b, const float* c, float* d);
void syncthreads();
unsigned lane();
void* dyn_smem();
typedef void (*body_fn)(void*);
void launch(dim3 grid, dim3 block, size_t shmem, const char* name, body_fn fn, void* ctx);
template <class F>
static inline void launch_lambda(dim3 grid, dim3 block, size_t shmem, const char* name, F&& f) {
    typedef typename std::remove_reference<F>::type FT;
    launch(grid, block, shmem, name, [](void* c) { (*static_cast<FT*>(c))(); }, const_cast<void*>(static_cast<const void*>(&f)));
}
template <class T>
static inline uint64_t to_bits(T v) {
    static_assert(sizeof(T) <= 8 && std::is_trivially_copyable<T>::value, "shuffle operand");
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    return u;
}
template <class T>
static inline T from_bits(uint64_t u) {
    T v;
    memcpy(&v, &u, sizeof(T));
    return v;
}
template <class T>
static inline T shfl(int site, T v, int src, int width = 64) {
    return from_bits<T>(collective(OP_SHFL, site, to_bits(v), src, width));
}
template <class T>
static inline T shfl_up(int site, T v, unsigned d, int width = 64) {
    return from_bits<T>(collective(OP_SHFL_UP, site, to_bits(v), (int)d, width));
}
template <class T>
static inline T shfl_down(int site, T v, unsigned d, int width = 64) {
    return from_bits<T>(collective(OP_SHFL_DOWN, site, to_bits(v), (int)d, width));
}
template <class T>
static inline T shfl_xor(int site, T v, int m, int width = 64) {
    return from_bits<T>(collective(OP_SHFL_XOR, site, to_bits(v), m, width));
}
static inline uint64_t ballot(int site, int pred) { return collective(OP_BALLOT, site, pred ? 1u : 0u, 0, 64); }
template <class T>
static inline T first(int site, T v) {
    return from_bits<T>(collective(OP_FIRST, site, to_bits(v), 0, 64));
}
typedef float f32x4_t __attribute__((ext_vector_type(4)));
static inline f32x4_t mfma(int site, float a, float b, f32x4_t c) {
    float ci[4] = {c[0], c[1], c[2], c[3]}, d[4];
    mfma16x16x4(site, a, b, ci, d);
    return f32x4_t{d[0], d[1], d[2], d[3]};
}
struct rsrc {
    const char* base;
    uint32_t bytes;
};
static inline uint32_t buf_load32(rsrc r, int voff, int soff) {
    const uint64_t o = (uint64_t)(uint32_t)voff + (uint32_t)soff;
    uint32_t v = 0;
    if (o + 4 <= r.bytes) memcpy(&v, r.base + o, 4);  // out-of-range reads return 0, as the hardware's raw buffers do
    return v;
}
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
static inline u32x2_t buf_load64(rsrc r, int voff, int soff) {
    const uint64_t o = (uint64_t)(uint32_t)voff + (uint32_t)soff;
    uint32_t v[2] = {0, 0};
    if (o + 8 <= r.bytes) memcpy(v, r.base + o, 8);
    return u32x2_t{v[0], v[1]};
}
}  // namespace emu

#define EMU_SITE (__LINE__)
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...)                                                   \
    ::emu::launch_lambda(dim3(grid), dim3(block), (size_t)(shmem), #kern, [=, emu_args = std::make_tuple(__VA_ARGS__)]() { \
        std::apply([&](auto... emu_a) { kern(emu_a...); }, emu_args);                                               \
    })

#define __syncthreads() ::emu::syncthreads()
#define __ballot(p) ::emu::ballot(EMU_SITE, (p))
#define __any(p) (::emu::ballot(EMU_SITE, (p)) != 0)
#define __all(p) (::emu::ballot(EMU_SITE, !(p)) == 0)
#define __shfl(...) ::emu::shfl(EMU_SITE, __VA_ARGS__)
#define __shfl_up(...) ::emu::shfl_up(EMU_SITE, __VA_ARGS__)
#define __shfl_down(...) ::emu::shfl_down(EMU_SITE, __VA_ARGS__)
#define __shfl_xor(...) ::emu::shfl_xor(EMU_SITE, __VA_ARGS__)
#define __lane_id() ::emu::lane()
#define __activemask() ::emu::ballot(EMU_SITE, 1)
#define __builtin_amdgcn_readfirstlane(v) ::emu::first(EMU_SITE, (v))
#define __builtin_amdgcn_wave_barrier() ((void)::emu::collective(::emu::OP_WAVE_BARRIER, EMU_SITE, 0, 0, 64))
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __threadfence() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __threadfence_block() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __threadfence_system() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) ::emu::mfma(EMU_SITE, (a), (b), (c))
#define __builtin_amdgcn_mbcnt_lo(mask, v) ((uint32_t)(v) + (uint32_t)__builtin_popcount((uint32_t)(mask) & (::emu::lane() >= 32u ? 0xffffffffu : ((1u << ::emu::lane()) - 1u))))
#define __builtin_amdgcn_mbcnt_hi(mask, v) ((uint32_t)(v) + (uint32_t)__builtin_popcount((uint32_t)(mask) & (::emu::lane() <= 32u ? 0u : ((1u << (::emu::lane() - 32u)) - 1u))))
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_logf(x) (::log2f(x))
#define __builtin_amdgcn_fmed3f(a, b, c) (::fmaxf(::fminf((a), (b)), ::fminf(::fmaxf((a), (b)), (c))))
typedef ::emu::rsrc __amdgpu_buffer_rsrc_t;
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) (::emu::rsrc{reinterpret_cast<const char*>(p), (uint32_t)(num)})
#define __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, aux) ::emu::buf_load32((r), (voff), (soff))
#define __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, aux) ::emu::buf_load64((r), (voff), (soff))
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)

static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline unsigned __float_as_uint(float f) { return ::emu::from_bits<unsigned>(::emu::to_bits(f)); }
static inline int __float_as_int(float f) { return ::emu::from_bits<int>(::emu::to_bits(f)); }
static inline float __uint_as_float(unsigned u) { return ::emu::from_bits<float>(::emu::to_bits(u)); }
static inline float __int_as_float(int u) { return ::emu::from_bits<float>(::emu::to_bits(u)); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }

// HIP's device-side min / max overloads
template <class A, class B>
static inline typename std::common_type<A, B>::type min(A a, B b) {
    typedef typename std::common_type<A, B>::type T;
    return (T)b < (T)a ? (T)b : (T)a;
}
template <class A, class B>
static inline typename std::common_type<A, B>::type max(A a, B b) {
    typedef typename std::common_type<A, B>::type T;
    return (T)a < (T)b ? (T)b : (T)a;
}

// ---- atomics (integer only in this library; sequentially consistent on the host) ----
template <class T, class U>
static inline T atomicAdd(T* p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U>
static inline T atomicSub(T* p, U v) { return __atomic_fetch_sub(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U>
static inline T atomicOr(T* p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U>
static inline T atomicAnd(T* p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U>
static inline T atomicExch(T* p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U, class V>
static inline T atomicCAS(T* p, U cmp, V val) {
    T expected = (T)cmp;
    __atomic_compare_exchange_n(p, &expected, (T)val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return expected;
}
template <class T, class U>
static inline T atomicMin(T* p, U v) {
    T cur = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while ((T)v < cur && !__atomic_compare_exchange_n(p, &cur, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return cur;
}
template <class T, class U>
static inline T atomicMax(T* p, U v) {
    T cur = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while ((T)v > cur && !__atomic_compare_exchange_n(p, &cur, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return cur;
}

#endif

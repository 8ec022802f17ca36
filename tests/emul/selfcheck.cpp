// tests/emul/selfcheck.cpp — the execution model checking ITSELF: small kernels whose results follow from the wave64 rules the
// model claims (tests/emul/hip/hip_runtime.h), compared on the host.  emu_selfcheck() returns the number of wrong values.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" unsigned emu_host_threads(void);

namespace {

int wrong(const char* what, unsigned i, double got, double want) {
    fprintf(stderr, "emu_selfcheck: %s[%u] = %.9g, expected %.9g\n", what, i, got, want);
    return 1;
}
#define CHECK(what, i, got, want) \
    if ((got) != (want)) bad += wrong(what, (unsigned)(i), (double)(got), (double)(want))

// a ballot inside a loop whose trip count differs per lane: the mask of iteration k holds the lanes still iterating
__global__ void k_loop_ballots(unsigned long long* masks /* [4] */, uint32_t* seen /* [64] */) {
    const uint32_t lane = threadIdx.x, trips = (lane & 3u) + 1u;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < trips; ++k) {
        const unsigned long long m = __ballot(1);
        if (lane == 3u) masks[k] = m;  // lane 3 runs all four iterations
        mine += (uint32_t)__popcll(m);
    }
    seen[lane] = mine;
}
// a shuffle whose source lane is outside EXEC returns 0; inside a width-16 segment sources wrap inside the segment
__global__ void k_shuffles(uint32_t* out /* [4][64] */) {
    const uint32_t lane = threadIdx.x;
    uint32_t a = 0xdeadbeefu;
    if (lane < 32u) a = __shfl(lane + 100u, (int)(lane + 32u), 64);  // lanes 32..63 are not here
    out[lane] = a;
    out[64 + lane] = __shfl(lane, 5, 16);          // lane 5 of the own 16-segment
    out[128 + lane] = __shfl_up(lane, 3u, 64);     // lanes 0..2 keep their own value
    out[192 + lane] = __shfl_xor(lane, 17, 64);
}
// __syncthreads with a wavefront that has left; readfirstlane = the lowest lane that is there
__global__ void k_barrier_and_first(uint32_t* out /* [256] */) {
    __shared__ uint32_t box[256];
    const uint32_t tid = threadIdx.x;
    box[tid] = tid * 3u;
    if (tid >= 192u) return;  // the last wavefront leaves before the barrier
    __syncthreads();
    uint32_t v = box[(tid + 64u) % 192u];
    if ((tid & 63u) >= 10u) v += __builtin_amdgcn_readfirstlane(tid);  // first lane there: 10 of each wavefront
    out[tid] = v;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(const float* A /* [16][4] */, const float* B /* [4][16] */, const float* C /* [16][16] */, float* D /* [16][16] */) {
    const uint32_t l = threadIdx.x;
    f32x4 c;
    for (uint32_t r = 0; r < 4u; ++r) c[r] = C[(4u * (l >> 4) + r) * 16u + (l & 15u)];
    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15u) * 4u + (l >> 4)], B[(l >> 4) * 16u + (l & 15u)], c, 0, 0, 0);
    for (uint32_t r = 0; r < 4u; ++r) D[(4u * (l >> 4) + r) * 16u + (l & 15u)] = d[r];
}
// workgroups really run side by side: every work-item adds to one counter, every workgroup has its own LDS
__global__ void k_blocks(uint32_t* counter, uint32_t* per_block /* [grid] */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    __shared__ uint32_t local;
    if (threadIdx.x == 0) local = 0;
    __syncthreads();
    atomicAdd(&local, threadIdx.x);
    dyn[threadIdx.x] = blockIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) per_block[blockIdx.x] = local + dyn[blockDim.x - 1u];
    atomicAdd(counter, 1u);
}

// wave-synchronous exchange through LDS: legal only with a wave_barrier between the store and the load
__global__ void k_wave_barrier(uint32_t* out /* [128] */) {
    __shared__ uint32_t box[128];
    const uint32_t tid = threadIdx.x;
    box[tid] = tid + 1000u;
    __builtin_amdgcn_wave_barrier();
    out[tid] = box[tid ^ 63u];  // the mirror lane of the own wavefront
}
// __any / __all under a partial mask; a workgroup waiting for another one's flag (workgroups run side by side)
__global__ void k_votes_and_flags(uint32_t* out /* [64] */, uint32_t* flag, uint32_t* order /* [2] */) {
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x == 0) {
        if (lane < 40u) out[lane] = (__any(lane == 39u) ? 1u : 0u) | (__all(lane < 40u) ? 2u : 0u) | (__all(lane < 39u) ? 4u : 0u);
        else out[lane] = 8u;
        if (lane == 0) {
            while (atomicAdd(flag, 0u) == 0u) {}  // set by workgroup 1
            order[0] = atomicAdd(flag, 1u);
        }
    } else if (lane == 0) {
        order[1] = atomicAdd(flag, 1u);
    }
}

template <class T>
T* dev(size_t n) {
    T* p = nullptr;
    (void)hipMalloc(&p, n * sizeof(T));
    (void)hipMemset(p, 0, n * sizeof(T));
    return p;
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int emu_selfcheck(void) {
    int bad = 0;
    {
        unsigned long long* masks = dev<unsigned long long>(4);
        uint32_t* seen = dev<uint32_t>(64);
        hipLaunchKernelGGL(k_loop_ballots, dim3(1), dim3(64), 0, nullptr, masks, seen);
        for (uint32_t k = 0; k < 4; ++k) {
            unsigned long long want = 0;
            for (uint32_t l = 0; l < 64; ++l)
                if ((l & 3u) + 1u > k) want |= 1ull << l;
            CHECK("loop ballot mask", k, masks[k], want);
        }
        for (uint32_t l = 0; l < 64; ++l) {
            uint32_t want = 0;
            for (uint32_t k = 0; k < (l & 3u) + 1u; ++k) want += 64u - 16u * k;
            CHECK("loop ballot popcounts", l, seen[l], want);
        }
        (void)hipFree(masks);
        (void)hipFree(seen);
    }
    {
        uint32_t* out = dev<uint32_t>(256);
        hipLaunchKernelGGL(k_shuffles, dim3(1), dim3(64), 0, nullptr, out);
        for (uint32_t l = 0; l < 64; ++l) {
            CHECK("shfl from a lane outside EXEC", l, out[l], (l < 32u ? 0u : 0xdeadbeefu));
            CHECK("shfl width 16", l, out[64 + l], (l & ~15u) + 5u);
            CHECK("shfl_up", l, out[128 + l], (l < 3u ? l : l - 3u));
            CHECK("shfl_xor", l, out[192 + l], (l ^ 17u));
        }
        (void)hipFree(out);
    }
    {
        uint32_t* out = dev<uint32_t>(256);
        hipLaunchKernelGGL(k_barrier_and_first, dim3(1), dim3(256), 0, nullptr, out);
        for (uint32_t t = 0; t < 192; ++t) {
            uint32_t want = ((t + 64u) % 192u) * 3u;
            if ((t & 63u) >= 10u) want += (t & ~63u) + 10u;
            CHECK("barrier / readfirstlane", t, out[t], want);
        }
        (void)hipFree(out);
    }
    {
        float *A = dev<float>(64), *B = dev<float>(64), *C = dev<float>(256), *D = dev<float>(256);
        for (int i = 0; i < 64; ++i) {
            A[i] = (float)((i * 7) % 11) - 5.0f;
            B[i] = (float)((i * 5) % 13) - 6.0f;
        }
        for (int i = 0; i < 256; ++i) C[i] = (float)(i % 9) * 0.5f;
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, nullptr, A, B, C, D);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float acc = C[i * 16 + j];
                for (int k = 0; k < 4; ++k) acc = fmaf(A[i * 4 + k], B[k * 16 + j], acc);
                CHECK("mfma", i * 16 + j, D[i * 16 + j], acc);
            }
        (void)hipFree(A);
        (void)hipFree(B);
        (void)hipFree(C);
        (void)hipFree(D);
    }
    {
        const uint32_t grid = 500, block = 128;
        uint32_t *counter = dev<uint32_t>(1), *per = dev<uint32_t>(grid);
        hipLaunchKernelGGL(k_blocks, dim3(grid), dim3(block), block * 4, nullptr, counter, per);
        CHECK("global counter", 0, *counter, grid * block);
        for (uint32_t b = 0; b < grid; ++b) CHECK("per-workgroup LDS", b, per[b], block * (block - 1u) / 2u + b);
        (void)hipFree(counter);
        (void)hipFree(per);
    }
    {
        uint32_t* out = dev<uint32_t>(128);
        hipLaunchKernelGGL(k_wave_barrier, dim3(1), dim3(128), 0, nullptr, out);
        for (uint32_t t = 0; t < 128; ++t) CHECK("wave_barrier exchange", t, out[t], (t ^ 63u) + 1000u);
        (void)hipFree(out);
    }
    if (emu_host_threads() > 1) {  // the flag test needs two workgroups in flight
        uint32_t *out = dev<uint32_t>(64), *flag = dev<uint32_t>(1), *order = dev<uint32_t>(2);
        hipLaunchKernelGGL(k_votes_and_flags, dim3(2), dim3(64), 0, nullptr, out, flag, order);
        for (uint32_t l = 0; l < 64; ++l) CHECK("any / all under a partial mask", l, out[l], (l < 40u ? 3u : 8u));
        CHECK("flag order", 0, order[1], 0u);
        CHECK("flag order", 1, order[0], 1u);
        (void)hipFree(out);
        (void)hipFree(flag);
        (void)hipFree(order);
    }
    return bad;
}

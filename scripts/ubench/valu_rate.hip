// Micro-benchmark (measurement tool, not part of the library): sustained issue cost of wave64 VALU instruction kinds on
// gfx950, in SIMD-cycles per wave-instruction at the nominal 2.4 GHz.  Used to state the VALU rooflines in DESIGN.md.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define ITER 4096
#define UNROLL 16

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    float x[UNROLL];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 y[UNROLL];
    unsigned u[UNROLL];
    double d[UNROLL];
    for (int i = 0; i < UNROLL; ++i) {
        d[i] = threadIdx.x * 1e-3 + i;
        x[i] = threadIdx.x * 1e-3f + i;
        y[i] = f2{x[i], x[i] + 1.0f};
        u[i] = threadIdx.x * 2654435761u + i;
    }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(f2{a, a}), "v"(f2{b, b}));
            if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]));
            if (KIND == 3) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]));
            if (KIND == 4) asm volatile("v_ffbh_u32 %0, %0" : "+v"(u[i]));
            if (KIND == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            if (KIND == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]));
            if (KIND == 7) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(*reinterpret_cast<unsigned long long*>(&y[i])));
            if (KIND == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(f2{a, a}));
            // round 6: the f64 instructions of glibc's expf evaluated in double (include/rp_libm_glibc.h)
            if (KIND == 10) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"((double)a), "v"((double)b));
            if (KIND == 11) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)a));
            if (KIND == 12) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)b));
            if (KIND == 13) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(x[i]));
            if (KIND == 14) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x[i]) : "v"(d[i]));
            if (KIND == 15) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (KIND == 16) asm volatile("v_lshl_add_u32 %0, %0, 15, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]));
            if (KIND == 17) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"((u[i] & 31u) << 3));
            if (KIND == 18) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));
            if (KIND == 19) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(d[i]));
            if (KIND == 20) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(u[i]));
            if (KIND == 21) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        }
    }
    if (KIND == 17) asm volatile("s_waitcnt lgkmcnt(0)");
    float s = 0;
    for (int i = 0; i < UNROLL; ++i) s += x[i] + y[i].x + y[i].y + (float)u[i] + (float)d[i];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
double run(const char* name) {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * ITER * UNROLL;
    const double simd_cycles = ms * 1e-3 * 2.4e9 * 256 * 4;
    std::printf("%-16s %8.3f ms  %6.2f SIMD-cycles per wave64 instruction (at 2.4 GHz)\n", name, ms, simd_cycles / wave_instr);
    hipFree(out);
    return simd_cycles / wave_instr;
}

int main() {
    run<0>("v_fma_f32");
    run<1>("v_pk_fma_f32");
    run<8>("v_mul_f32");
    run<9>("v_pk_mul_f32");
    run<2>("v_add_u32");
    run<6>("v_and_b32");
    run<3>("v_bcnt_u32_b32");
    run<4>("v_ffbh_u32");
    run<7>("v_lshlrev_b64");
    run<5>("v_exp_f32");
    run<10>("v_fma_f64");
    run<11>("v_mul_f64");
    run<12>("v_add_f64");
    run<13>("v_cvt_f64_f32");
    run<14>("v_cvt_f32_f64");
    run<15>("v_med3_f32");
    run<16>("v_lshl_add_u32");
    run<17>("ds_read_b64");
    run<18>("v_rndne_f64");
    run<19>("v_cvt_i32_f64");
    run<20>("v_ldexp_f64");
    run<21>("v_max_f32");
    return 0;
}

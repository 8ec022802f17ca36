// lm_glibc_dev.hpp — where the lloyd kernels' glibc-arithmetic pass (namespace rp::lm_glibc) keeps glibc's two tables: in LDS.
//
// include/rp_libm_glibc.h's rp_glibc_expf / rp_glibc_logf are glibc's control flow; compiled for gfx950 an exponential was a
// global_load_dwordx2 of its table entry with an s_waitcnt vmcnt(0) inside the dependent chain, ~25 exec-mask instructions of
// special-case ladder, 55 instructions per softmin term against the contract pass' 19 (round 5's ISA).  The kernels
// now evaluate the header's branch-free forms (rp_glibc_exp_floor_tab, rp_glibc_expf_tab, rp_glibc_logf_tab: the same bits for every
// input, swept over all 2^32 patterns on the host and on the device) on a copy of the tables in LDS:
//   expf  32 x u64 = 256 B: entry i occupies banks 2i, 2i+1, so a ds_read_b64 gather of 64 lanes is conflict-free whatever the indices
//   logf  16 x (1/c, log c) = 256 B: one ds_read_b128 per call, likewise
// A term of the softmin is then sub, clamp, 11 f64 instructions (cvt, mul, rndne, cvt_i32, 5 x fma / mul, mul, cvt), 4 integer ones and
// one LDS read — no branch, no vector-memory access.  Measured issue cost (scripts/ubench/valu_rate.hip -> profiles/r06_valu_issue_rates.txt):
// an f64 fma / mul is 5.1 SIMD-cycles per wave64 instruction against 2.9 - 3.1 for a plain f32 one, so the pass costs ~1.35 x the
// contract's per term; that is the price of evaluating glibc's own double-precision polynomial, not an overhead.
//
// Every kernel that evaluates an exponential or a logarithm starts with LM_TABLES() (lloyd_kernels.hpp): the workgroup's first 64
// work-items copy the 512 bytes from constant memory, one barrier.
#pragma once

#include "../../include/rp_libm_glibc.h"

namespace rp {
namespace lmg {
__device__ const uint64_t EXP_TAB[32] = RP_GLIBC_EXP2F_TAB_INIT;
__device__ const double LOG_TAB[16][2] = RP_GLIBC_LOGF_TAB_INIT;
__shared__ uint64_t lds_exp[32];
__shared__ double lds_log[16][2];

__device__ __forceinline__ void tables_init() {
    for (uint32_t t = threadIdx.x; t < 64u; t += blockDim.x) {
        if (t < 32u) lds_exp[t] = EXP_TAB[t];
        else lds_log[(t - 32u) >> 1][(t - 32u) & 1u] = LOG_TAB[(t - 32u) >> 1][(t - 32u) & 1u];
    }
    __syncthreads();
}
}  // namespace lmg
}  // namespace rp

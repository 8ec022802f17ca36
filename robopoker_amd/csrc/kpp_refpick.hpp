// kpp_refpick.hpp — the reference-seed k-means++ draw (Layer::init_centroids, layer.rs:160-166): WeightedIndex::new(potentials) +
// sample, i.e. the SEQUENTIAL f32 running sum of N potentials and a partition point on it, without walking the N dependent additions.
// Included by lloyd_kernels.hpp in the contract namespace only (no exponential in here: one copy serves both arithmetics).
//
// rand 0.9.2's WeightedIndex<f32> keeps cumulative_weights[i] = (...((w0 + w1) + w2)... + wi) rounded after every addition; the pick
// is the first i whose running sum exceeds the drawn x.  f32 addition does not associate, so the sums cannot be re-bracketed — but
// between two powers of two they are INTEGER arithmetic.  With run in [2^e, 2^(e+1)), u = 2^(e-23) its ulp and m = run / u:
//     RN(run + w) = u * (m + R(w / u)),  R = round to nearest, a tie resolved towards the even m + R,
// as long as the result stays below 2^(e+1) (w >= 0, so the sums never decrease).  A term that is not a tie adds the same integer
// a = floor(w/u) + (frac(w/u) > 1/2) whatever the sum in front of it, so a chunk of terms without ties that stays inside one binade
// adds the integer D = sum of its a — in any order, by any number of lanes.
//
//   k_kr_sums    approximate chunk sums (one wavefront per chunk of 256 potentials)
//   k_kr_scan    exclusive scan of those: the binade e_c each chunk is EXPECTED to start in (a prediction, checked later)
//   k_kr_chunks  per chunk, for u = 2^(e_c - 23): D_c, and a flag if any term is a tie, is not finite or is >= 2^23 u
//   k_kr_pick    one wavefront walks the CHUNKS in order with the exact running sum: where the prediction holds (the sum's exponent
//                is e_c, no flag, m + D_c <= 2^24 - 1) the chunk is one exact addition run += D_c u; everywhere else (the first
//                chunk, the ~20 binade crossings, the ~ln N ties, a misprediction) it walks the chunk's 256 additions as the
//                reference does.  Chunk-end sums are kept; the partition point is a binary search over them and one re-walked chunk.
// The result is the reference's running sum bit for bit by construction, not by tolerance: every shortcut is taken only under the
// conditions of the identity above, checked against the exact sum at the chunk's start.  tests/test_reference_seed.py compares the
// picks and the total with a host walk of the N additions on adversarial weight sets (ties, crossings, zeros, huge terms).
// Before (round 5): one wavefront, N dependent v_add_f32: 8.9 ms per pick at N = 1 286 792, 2.27 s per layer.
#pragma once

#define KR_ELEMS 256u  // potentials per chunk: four per lane

// meta[c]: .x = D_c, .y = biased exponent the chunk was summarised for (0 = none) | flag << 31
__global__ __launch_bounds__(256) void k_kr_sums(const float* pot, uint64_t N, uint32_t nchunks, float* csum) {
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (c >= nchunks) return;
    const uint64_t base = (uint64_t)c * KR_ELEMS;
    float s = 0.0f;
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        const uint64_t i = base + q * 64u + lane;
        s += i < N ? pot[i] : 0.0f;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) csum[c] = s;
}
// one workgroup: start[c] = csum[0] + ... + csum[c-1] (any bracketing: a prediction), kept as the biased exponent of that float
__global__ __launch_bounds__(1024) void k_kr_scan(const float* csum, uint32_t nchunks, uint32_t* expo) {
    __shared__ float part[1024];
    const uint32_t t = threadIdx.x, per = (nchunks + 1023u) / 1024u, lo = t * per, hi = min(lo + per, nchunks);
    float s = 0.0f;
    for (uint32_t c = lo; c < hi; ++c) s += csum[c];
    part[t] = s;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {  // Hillis-Steele inclusive scan of the per-thread totals
        const float v = t >= o ? part[t - o] : 0.0f;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    float run = t ? part[t - 1] : 0.0f;
    for (uint32_t c = lo; c < hi; ++c) {
        expo[c] = (rp_f2u(run) >> 23) & 0xffu;
        run += csum[c];
    }
}
__global__ __launch_bounds__(256) void k_kr_chunks(const float* pot, uint64_t N, uint32_t nchunks, const uint32_t* expo, uint2* meta) {
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (c >= nchunks) return;
    const uint32_t be = expo[c];  // biased exponent of the predicted start
    // u = 2^(be - 127 - 23); the scale 1/u = 2^(150 - be) must be a normal float and so must u: be in [24, 254]
    const bool usable = be >= 24u && be <= 254u;
    const float inv_u = rp_u2f((usable ? 277u - be : 127u) << 23);
    const uint64_t base = (uint64_t)c * KR_ELEMS;
    uint32_t d = 0;
    bool flag = !usable;
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        const uint64_t i = base + q * 64u + lane;
        const float w = i < N ? pot[i] : 0.0f;
        const float x = w * inv_u;  // exact (a power of two) unless it underflows, and then x < 1/2 either way
        const float fl = floorf(x), fr = x - fl;  // exact: x < 2^23 below
        flag |= !(x < 8388608.0f) || !(w >= 0.0f) || fr == 0.5f;  // too large for the binade, NaN / negative, or a tie
        d += (x < 8388608.0f && w >= 0.0f) ? (uint32_t)fl + (fr > 0.5f ? 1u : 0u) : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);  // < 2^8 x 2^23
    const bool any = __ballot(flag) != 0ull;
    if (lane == 0) meta[c] = make_uint2(d, be | (any ? 0x80000000u : 0u));
}

// cum[c] = the running sum after chunk c (exact).  picked[0] = the index, picked[1] = the total's bits, picked[2] = chunks walked
__global__ __launch_bounds__(64) void k_kr_pick(float* pot, float* kpp_d, uint64_t N, uint32_t nchunks, const uint2* meta, float* cum, float v01,
                                                unsigned long long* picked) {
    __shared__ __attribute__((aligned(16))) float buf[KR_ELEMS];
    const uint32_t ln = threadIdx.x;
    auto load_chunk = [&](uint32_t c) {
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint64_t i = (uint64_t)c * KR_ELEMS + q * 64u + ln;
            buf[q * 64u + ln] = i < N ? pot[i] : 0.0f;  // + 0 past the end leaves the sum as it is
        }
        __syncthreads();
    };
    auto walk = [&](float run) {  // the reference's additions over the chunk in buf
#pragma unroll 8
        for (uint32_t j = 0; j < KR_ELEMS; j += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(buf + j);
            run += w4.x; run += w4.y; run += w4.z; run += w4.w;
        }
        return run;
    };
    float run = 0.0f;  // total_weight (0 + w0 = w0 exactly)
    uint32_t walked = 0;
    for (uint32_t c0 = 0; c0 < nchunks; c0 += 64u) {
        const uint32_t mine = c0 + ln;
        const uint2 mm = mine < nchunks ? meta[mine] : make_uint2(0u, 0x80000000u);
        float my_end = 0.0f;
        const uint32_t lim = min(64u, nchunks - c0);
        for (uint32_t j = 0; j < lim; ++j) {
            const uint32_t d = (uint32_t)__shfl((int)mm.x, (int)j, 64), info = (uint32_t)__shfl((int)mm.y, (int)j, 64);
            const uint32_t rb = rp_f2u(run), be = rb >> 23;  // run >= 0: no sign bit
            const uint32_t m = (rb & 0x7fffffu) | 0x800000u;
            const bool fast = info == be && be >= 24u && be <= 254u && d <= 0xffffffu - m;  // prediction holds, no flag, stays in the binade
            if (fast) {
                run += (float)d * rp_u2f((be - 23u) << 23);  // both exact: d < 2^24, and m + d <= 2^24 - 1 is representable at this ulp
            } else {  // wave uniform
                load_chunk(c0 + j);
                run = walk(run);
                walked += 1;
            }
            if (ln == j) my_end = run;
        }
        if (mine < nchunks) cum[mine] = my_end;
    }
    const float total = run;
    uint64_t win = N;  // invalid weights (total == 0): the reference panics ("valid weights array"); the host falls back
    if (total > 0.0f) {  // wave uniform
        const float x = v01 * rp_uniform_f32_scale(total) + 0.0f;  // UniformFloat::sample: value0_1 * scale + low
        // partition_point(|w| w <= x) over cum[0 .. N-1): the first index whose running sum exceeds x, N - 1 if none does
        __threadfence();
        __syncthreads();  // every lane's chunk-end sums are stored before any lane searches them
        uint32_t lo = 0, hi = nchunks;  // first chunk whose END sum exceeds x
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            // (an agent-scope load: the sums were written by this wavefront a moment ago)
            if (rp_u2f(__hip_atomic_load(reinterpret_cast<const uint32_t*>(cum) + mid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) <= x) lo = mid + 1;
            else hi = mid;
        }
        win = N - 1;
        if (lo < nchunks) {
            float r2 = lo ? rp_u2f(__hip_atomic_load(reinterpret_cast<const uint32_t*>(cum) + lo - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0f;
            load_chunk(lo);
            uint32_t at = KR_ELEMS;
            for (uint32_t j = 0; j < KR_ELEMS; ++j) {  // the reference's adds from the exact sum in front of the chunk
                r2 += buf[j];
                if (at == KR_ELEMS && r2 > x) at = j;
            }
            if (at < KR_ELEMS) win = min((uint64_t)lo * KR_ELEMS + at, N - 1);
        }
        if (ln == 0) {
            pot[win] = 0.0f;                // potentials[i] = 0 (layer.rs:168)
            if (kpp_d) kpp_d[win] = -1.0f;  // no solve stands behind that 0
        }
    }
    if (ln == 0) {
        picked[0] = win;
        picked[1] = rp_f2u(total);
        picked[2] = walked;
    }
}

// gg_arith.h -- the exactly-specified arithmetic of the walk sampler,
// device side.  Every rounding step is explicit (fmaf where fused, plain ops elsewhere;
// the library is built with -ffp-contract=off) so that results are bit-identical to
// any other implementation of the same text (the CPU oracle has its own).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gg {

// S4: Philox4x32-10 (Salmon et al., SC'11), counter = (hop, walk, root, stream), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t &o0, uint32_t &o1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o0 = c0;
    o1 = c1;
}

// 53-bit numerator m of the hop's uniform u = m / 2^53 (numpy legacy random_sample layout).
__device__ __forceinline__ uint64_t uniform53(uint64_t seed, uint32_t stream, uint32_t root,
                                              uint32_t walk, uint32_t hop) {
    uint32_t a, b;
    philox4x32_10(hop, walk, root, stream, (uint32_t)seed, (uint32_t)(seed >> 32), a, b);
    return ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6);
}

// S2: exp(x) for x <= 0 in pure fp32; x < -28 -> 0 (its weight would truncate to 0 anyway).
__device__ __forceinline__ float exp_spec(float x) {
    if (x < -28.0f) return 0.0f;
    const float kf = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(kf, -0.693359375f, x);
    r = __builtin_fmaf(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    p = __builtin_fmaf(p, r2, r);
    p = p + 1.0f;
    const int k = (int)kf;  // [-41, 0]
    return p * __uint_as_float((uint32_t)(k + 127) << 23);
}

// S3: fixed-point softmax weight, exact integer arithmetic from here on.
__device__ __forceinline__ uint64_t weight_fix40(float e) { return (uint64_t)(e * 1099511627776.0f); }

// S5: t = floor(m * W / 2^53) with m < 2^53, W < 2^63.
__device__ __forceinline__ uint64_t threshold(uint64_t m, uint64_t W) {
    const uint64_t hi = __umul64hi(m, W), lo = m * W;
    return (hi << 11) | (lo >> 53);
}

}  // namespace gg

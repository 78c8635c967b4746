// Device helpers shared by the kernels of the diffusion process: the reverse-step expression with the
// reference's rounding, the classifier-free-guidance blend, and a counter-based Gaussian noise stream.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace ddpm {

// torch.lerp(start = e_u, end = e_c, weight = w): two-branch formula, every operation rounded separately.
__device__ __forceinline__ float guided_eps(float e_c, float e_u, float w, bool low) {
    const float d = __fsub_rn(e_c, e_u);
    return low ? __fadd_rn(e_u, __fmul_rn(w, d)) : __fsub_rn(e_c, __fmul_rn(d, __fsub_rn(1.0f, w)));
}

// c1*(x - c2*eps) + c3*z as the reference evaluates it: five separately rounded fp32 operations (the
// explicit _rn intrinsics keep the compiler from contracting them into FMAs).
__device__ __forceinline__ float step1(float x, float e, float z, float c1, float c2, float c3, bool has_z) {
    const float inner = __fsub_rn(x, __fmul_rn(c2, e));
    const float a = __fmul_rn(c1, inner);
    // t == 0: the reference still adds sqrt(beta)*0 = +0, which leaves `a` unchanged except that
    // -0 + +0 = +0; adding 0.0f reproduces exactly that.
    return has_z ? __fadd_rn(a, __fmul_rn(c3, z)) : __fadd_rn(a, 0.0f);
}

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): four 32-bit words per
// (128-bit counter, 64-bit key). Stateless: any thread can produce any element of the stream.
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * ctr.x;
        const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * ctr.z;
        ctr = make_uint4(static_cast<uint32_t>(p1 >> 32) ^ ctr.y ^ key.x, static_cast<uint32_t>(p1),
                         static_cast<uint32_t>(p0 >> 32) ^ ctr.w ^ key.y, static_cast<uint32_t>(p0));
        key.x += 0x9E3779B9u;
        key.y += 0xBB67AE85u;
    }
    return ctr;
}

// Standard-normal element `index` of the stream (seed, call): elements 4q..4q+3 come from one Philox
// block (counter = (q, call)), as two Box-Muller pairs. u in (0, 1], so the logarithm is finite.
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t call, uint64_t index) {
    const uint64_t q = index >> 2;
    const uint4 r = philox4x32_10(make_uint4(static_cast<uint32_t>(q), static_cast<uint32_t>(q >> 32), static_cast<uint32_t>(call),
                                             static_cast<uint32_t>(call >> 32)),
                                  make_uint2(static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32)));
    const uint32_t a = (index & 2) ? r.z : r.x, b = (index & 2) ? r.w : r.y;
    const float u1 = (static_cast<float>(a >> 8) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1]
    const float u2 = static_cast<float>(b >> 8) * (1.0f / 16777216.0f);              // [0, 1)
    const float rad = sqrtf(-2.0f * logf(u1));
    const float ang = 6.28318530717958647692f * u2;
    return (index & 1) ? rad * sinf(ang) : rad * cosf(ang);
}

}  // namespace ddpm

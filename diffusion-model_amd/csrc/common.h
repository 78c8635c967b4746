// Shared host-side helpers for the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "../../include/ddpm_mi355x.h"

namespace ddpm {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in order: a loop whose index is a compile-time
// constant in the body (for `if constexpr` and array indices) without relying on the optimiser's unrolling.
template <typename F, int... I>
__host__ __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

void set_error(const char* fmt, ...);

inline int fail_invalid(const char* what) {
    set_error("invalid argument: %s", what);
    return DDPM_E_INVALID;
}

// Checks the launch itself (not completion: nothing here synchronises).
inline int check_launch(const char* kernel) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", kernel, hipGetErrorString(e));
        return DDPM_E_LAUNCH;
    }
    return DDPM_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kMaxGridX = 1 << 20;

// Arithmetic of the MFMA-fed convolutions. The process starts in the mode DDPM_CONV_MATH names in the environment and
// ddpm_set_conv_math() changes it at run time (the host mirror does so to repeat a call whose operands left fp16's range):
//   "f16x2"  every fp32 operand as a pair of fp16 numbers (hi, and the remainder scaled by 2^11), three cross products
//            per fp32 product on v_mfma_f32_32x32x16_f16 (the default);
//   "bf16x3" every fp32 operand as the exact sum of three bf16 numbers, six cross products on
//            v_mfma_f32_32x32x16_bf16;
//   "fp32"   the plain fp32 MFMA kernels and the fp32 packed-weight format: a diagnostic switch for A/B measurements.
// The packed-weight format follows the mode, so packed weights must be used in the mode they were packed in.
enum ConvMath { kMathFp32 = 0, kMathBf16x3 = 1, kMathF16x2 = 2 };
int conv_math();                       // runtime.hip
inline bool split_math() { return conv_math() != kMathFp32; }
inline bool f16_math() { return conv_math() == kMathF16x2; }

// Device address, on the current GPU, of the word that the fp16-pair kernels set to 1 when an operand lies beyond
// +-65504 (the pair representation clips there); null (with the error message set) when it cannot be looked up.
// ddpm_range_flag_read() hands it to the host.
unsigned* range_flag();                // runtime.hip
constexpr float kF16Max = 65504.0f;

// DDPM_ATTN_SMALL=0 (read once) routes feature maps of at most 64 pixels through the tiled attention
// kernels instead of the one-workgroup-per-sample ones: a diagnostic switch for A/B measurements.
inline bool small_attention() {
    static const bool on = [] {
        const char* e = getenv("DDPM_ATTN_SMALL");
        return !(e && strcmp(e, "0") == 0);
    }();
    return on;
}

#ifdef __HIPCC__
// Range-checked global loads: a buffer descriptor over [base, base + bytes); any offset at or beyond `bytes` (such as
// kOob) reads as zero, so a guarded load needs no branch -- and without branches around its loads the compiler can count
// the outstanding ones exactly instead of waiting for all of them at the first use.
constexpr unsigned kOob = 0x80000000u;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_over(const float* base, long long bytes) {
    const unsigned n = bytes > 0x7fffffffll ? 0x7fffffffu : static_cast<unsigned>(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ f32x4_t buffer_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_offset) {
    return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, byte_offset, 0, 0));
}

// Workgroups are dispatched in linear order (x fastest) and handed to the 8 XCDs round-robin, and every
// XCD has its own L2. This returns the block index a kernel should WORK on so that consecutive linear
// indices (the workgroups that share a sample's operands) run on ONE XCD and find each other's data in
// its L2, instead of each fetching its own copy over the fabric.
__device__ __forceinline__ dim3 xcd_block() {
    const unsigned nx = gridDim.x, ny = gridDim.y;
    const unsigned total = nx * ny * gridDim.z;
    const unsigned id = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const unsigned q = total >> 3, rr = total & 7, xcd = id & 7;
    const unsigned nid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
    return dim3(nid % nx, (nid / nx) % ny, nid / (nx * ny));
}
#endif

}  // namespace ddpm

#define DDPM_REQUIRE(cond)                          \
    do {                                            \
        if (!(cond)) return ddpm::fail_invalid(#cond); \
    } while (0)

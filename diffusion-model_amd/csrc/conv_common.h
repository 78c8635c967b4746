// Definitions shared by the implicit-GEMM convolution kernels (conv_gemm.hip, conv3x3_halo.hip).
#pragma once
#include "common.h"

namespace ddpm {
namespace conv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int MODE_CONV3 = 0;        // 3x3, pad 1
constexpr int MODE_CONV1 = 1;        // 1x1
constexpr int MODE_CONVT = 2;        // 2x2 stride-2 transposed: GEMM + scatter
constexpr int MODE_CONVT_DGRAD = 3;  // data gradient of MODE_CONVT: 4 taps gathered from the (out_h x out_w) image

constexpr int BK = 32;
constexpr int LDS_LD = 36;             // padded row length in floats
constexpr int BM = 128;                // rows of a full tile (4 groups)
constexpr unsigned OOB = 0x80000000u;  // any offset >= the descriptor's num_records reads as zero
constexpr int kSlots = 512;            // resident workgroups: 256 CUs x 2 (LDS-limited)
#ifndef DDPM_ABLATE
#define DDPM_ABLATE 0                  // diagnostic builds only: 1 = no staging in the K loop, 2 = also no barrier, 3 = no MFMAs, 4 = loads but no LDS stores, 5 = LDS stores but no loads
#endif

struct GemmParams {
    const float* src1;
    const float* src2;
    const float* w;
    float* out;
    const float* scale;
    const float* shift;
    const float* residual;
    const float* temb;
    int batch, in_h, in_w;
    int win_y0, win_x0, win_h, win_w;
    int c1, ld1, c2, ld2, ctot;
    int n_total, cout;
    int ldo, ldr, act;
    int out_h, out_w, off_y, off_x;
    int m_total, kc, taps;
    int n_tiles, groups_total, gpb;
    long long src1_pixels, src2_pixels;  // pixels (rows of ld floats) in each source tensor
    unsigned w_bytes;
    int kc16;                            // 16-channel chunks per tap (split-bf16 path)
    int nbp;                             // 32-row blocks of the packed weights (split-bf16 path)
    int tile_groups, hrows;              // halo kernel: row groups per tile, LDS rows per buffer
};

static __device__ __forceinline__ float apply_act(float v, int act) {
    if (act == DDPM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DDPM_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}

// erf by Abramowitz & Stegun 7.1.26 (absolute error below 1.5e-7 in exact arithmetic; about 3e-7 with
// the hardware reciprocal and exponential): a third of the instructions of the library erff. The
// GELU built on it differs from the exact one by less than 2e-7 * |v| + 2e-7.
static __device__ __forceinline__ float erf_fast(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float q = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    q = __builtin_fmaf(q, t, 1.421413741f);
    q = __builtin_fmaf(q, t, -0.284496736f);
    q = __builtin_fmaf(q, t, 0.254829592f);
    q *= t;
    const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
    const float y = __builtin_fmaf(-q, e, 1.0f);
    return __builtin_copysignf(y, x);
}

static __device__ __forceinline__ float apply_act_fast(float v, int act) {
    if (act == DDPM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DDPM_ACT_GELU) {
        const float hv = 0.5f * v;
        return __builtin_fmaf(hv, erf_fast(v * 0.70710678118654752440f), hv);
    }
    return v;
}

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, long long bytes) {
    const unsigned n = bytes > 0x7fffffffll ? 0x7fffffffu : static_cast<unsigned>(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, n, 0x00020000);
}

static __device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int XK = 16;      // channels per K step
constexpr int XROW = 112;   // LDS bytes per tile row
constexpr int XFLUSH = 4;   // steps between folds into the master accumulator

static __device__ __forceinline__ void split_store(const f32x4 v, unsigned char* dst) {
    unsigned u[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u[e] = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(u[e] & 0xffff0000u);
        m[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(m[e] & 0xffff0000u);
        l[e] = __float_as_uint(r2);
    }
    // upper halves of two dwords -> one dword of two bf16 (element 0 in the low half)
    u32x2 hi, mid, lo;
    hi[0] = __builtin_amdgcn_perm(u[1], u[0], 0x07060302u);
    hi[1] = __builtin_amdgcn_perm(u[3], u[2], 0x07060302u);
    mid[0] = __builtin_amdgcn_perm(m[1], m[0], 0x07060302u);
    mid[1] = __builtin_amdgcn_perm(m[3], m[2], 0x07060302u);
    lo[0] = __builtin_amdgcn_perm(l[1], l[0], 0x07060302u);
    lo[1] = __builtin_amdgcn_perm(l[3], l[2], 0x07060302u);
    *reinterpret_cast<u32x2*>(dst) = hi;
    *reinterpret_cast<u32x2*>(dst + 32) = mid;
    *reinterpret_cast<u32x2*>(dst + 64) = lo;
}


// 3x3 convolution with the activation tile (plus halo) resident in LDS across the nine taps
// (conv3x3_halo.hip). Returns DDPM_OK after launching, or a positive value when the shape is outside
// what that kernel handles and the caller should use the generic one.
int launch_conv3x3_halo(GemmParams& p, hipStream_t s);

// 1x1 convolution with the whole packed weight matrix resident in LDS and the activations loaded straight
// into MFMA operand registers (conv1x1_resident.hip). Same return convention as launch_conv3x3_halo.
int launch_conv1x1_resident(GemmParams& p, hipStream_t s);

// DDPM_CONV1X1_RESIDENT=0 (read once) keeps every 1x1 convolution on the generic GEMM kernel: a diagnostic
// switch for A/B measurements.
inline bool resident_1x1() {
    static const bool on = [] {
        const char* e = getenv("DDPM_CONV1X1_RESIDENT");
        return !(e && strcmp(e, "0") == 0);
    }();
    return on;
}

}  // namespace conv
}  // namespace ddpm

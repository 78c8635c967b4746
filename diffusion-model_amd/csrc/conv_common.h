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
#ifndef DDPM_STORE_AUX
#define DDPM_STORE_AUX 2               // cache policy of the epilogues' output stores: 2 = non-temporal (0.3-0.7 % of a step against 0, the default policy; same bits)
#endif
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
    unsigned* range_flag;                // fp16-pair kernels: raised when an operand lies beyond fp16's range
    double* stat_partial;                // halo kernel, optional: per-channel partial sums of the values written, [block][sum | sum of squares][cout]
    int stat_blocks;                     // set by the launcher: number of blocks in stat_partial
    // launch_conv1x1_pairs only (the q|k|v projection of an inference-time attention block): columns from pair_from on
    // (v) go to pair_v as fp16-pair planes in the order the attention kernels' matrix operands take them, columns
    // pair_from / 2 .. pair_from - 1 (k) stay in `out` as one (hi | lo << 16) word per element
    unsigned char* pair_v;
    int pair_from, pair_c;
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

// ---- the fp16 pair scheme -------------------------------------------------------------------
// x = hi + lo * 2^-11 with hi = RN_fp16(x) and lo = RN_fp16((x - hi) * 2^11): 22-23 significant bits, the remainder
// scaled so that it keeps its precision whatever x's exponent (fp16 has only five exponent bits). A product a*b is
// ah*bh + (ah*bl + al*bh) * 2^-11: the hi*hi products accumulate in one MFMA accumulator, the two cross products in
// a second one that is scaled and added at the end; the dropped al*bl term is below 2^-22 of a*b, zero on average.
// The fp16 MFMA does not lose bits when it accumulates the way the bf16 one does (measured: no drift over 9216
// like-signed terms), so no master accumulator is needed. Operands beyond 65504 in magnitude are clipped to +-65536
// (fp16_saturate below); values below 2^-14 lose relative (not absolute) precision.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kLoScale = 2048.0f;
constexpr float kLoInv = 1.0f / 2048.0f;
constexpr int XROW_F16 = 80;   // LDS bytes per tile row: two planes of 16 fp16 and 16 bytes of padding (conflict-free 16-byte reads)

// MODE.FP16_OVFL = 1 for the calling wave (the MODE register is per-wave state, set afresh at every launch): a
// float -> fp16 conversion that overflows gives +-65504 instead of infinity (tools/probe/fp16_ovfl.hip); the conversion
// passes true infinities and NaNs through, but the matrix instructions of such a wave do not (see RangeWatch below). Every kernel that forms fp16 pairs calls this first, so that an operand beyond fp16's
// range is clipped to +-65536 (hi = 65504, remainder 65504 / 2048) instead of turning into inf - inf = NaN inside
// the matrix instructions: an untrained network's samples, which grow to 1e7 over a 1000-step loop, stay finite as
// they do in fp32.
static __device__ __forceinline__ void fp16_saturate() { __builtin_amdgcn_s_setreg(1 | (23 << 6), 1); }

static __device__ __forceinline__ void split2_pair(const f32x2 v, unsigned& hi, unsigned& lo) {
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 rem = (v - __builtin_convertvector(h, f32x2)) * kLoScale;
    const f16x2 l = __builtin_convertvector(rem, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// Range watch of the fp16-pair kernels: every thread keeps the largest magnitude it has turned into a pair (one
// three-operand maximum per two elements) and whether it met a NaN (one unordered compare per two elements, kept as a
// lane mask in scalar registers), and, when it leaves, raises the library's range flag if one of ITS operands was not
// representable: beyond fp16's largest finite value (clipped), infinite, or NaN. The last two matter because a wave
// with MODE.FP16_OVFL set does not propagate them: v_mfma_f32_32x32x16_f16 then treats a NaN operand as a finite
// number and clamps what an infinite one produces (tools/probe/nan_mfma.hip), where fp32 arithmetic would hand the
// NaN on. With the flag raised the caller repeats the work on three bf16 planes, which do propagate them.
struct RangeWatch {
    float amax = 0.f;
    bool nan = false;
};
#ifndef DDPM_NO_WATCH
#define DDPM_NO_WATCH 0   // diagnostic builds only: 1 = the kernels do not watch their operands (to price the watch)
#endif
static __device__ __forceinline__ void watch2(RangeWatch& w, const f32x2 v) {
    if (DDPM_NO_WATCH) return;
    w.amax = __builtin_fmaxf(w.amax, __builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])));
    w.nan |= __builtin_isunordered(v[0], v[1]);
}
// A workgroup that stages an operand which another workgroup of the same launch stages too (and watches) passes this
// instead: the splitting helpers are templates on the watch's type, and with this one they check nothing.
struct NoWatch {};
static __device__ __forceinline__ void watch2(NoWatch&, const f32x2) {}
static __device__ __forceinline__ RangeWatch merged(const RangeWatch& a, const NoWatch&) { return a; }
static __device__ __forceinline__ RangeWatch merged(const RangeWatch& a, const RangeWatch& b) {
    RangeWatch w;
    w.amax = __builtin_fmaxf(a.amax, b.amax);
    w.nan = a.nan || b.nan;
    return w;
}
static __device__ __forceinline__ void raise_range_flag(unsigned* flag, const RangeWatch& w) {
    if (w.amax > 65504.0f || w.nan) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// four consecutive channels of one tile row -> the two fp16 planes of that row (8 bytes each, 32 bytes apart)
template <class W>
static __device__ __forceinline__ void split_store_f16(const f32x4 v, unsigned char* dst, W& amax) {
    unsigned h0, l0, h1, l1;
    watch2(amax, f32x2{v[0], v[1]});
    watch2(amax, f32x2{v[2], v[3]});
    split2_pair(f32x2{v[0], v[1]}, h0, l0);
    split2_pair(f32x2{v[2], v[3]}, h1, l1);
    *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(dst + 32) = u32x2{l0, l1};
}

// 8 consecutive fp32 (two 16-byte loads) -> the two fp16 planes of an MFMA operand fragment
template <class W>
static __device__ __forceinline__ void split8_f16(const f32x4 v0, const f32x4 v1, u32x4& hi, u32x4& lo, W& amax) {
    unsigned h[4], l[4];
    watch2(amax, f32x2{v0[0], v0[1]});
    watch2(amax, f32x2{v0[2], v0[3]});
    watch2(amax, f32x2{v1[0], v1[1]});
    watch2(amax, f32x2{v1[2], v1[3]});
    split2_pair(f32x2{v0[0], v0[1]}, h[0], l[0]);
    split2_pair(f32x2{v0[2], v0[3]}, h[1], l[1]);
    split2_pair(f32x2{v1[0], v1[1]}, h[2], l[2]);
    split2_pair(f32x2{v1[2], v1[3]}, h[3], l[3]);
    hi = u32x4{h[0], h[1], h[2], h[3]};
    lo = u32x4{l[0], l[1], l[2], l[3]};
}

// One 32x32x16 MFMA on operand fragments held as four dwords, in either arithmetic.
template <bool F16>
static __device__ __forceinline__ f32x16 mfma16(const u32x4& a, const u32x4& b, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}


// 3x3 convolution with the activation tile (plus halo) resident in LDS across the nine taps
// (conv3x3_halo.hip). Returns DDPM_OK after launching, or a positive value when the shape is outside
// what that kernel handles and the caller should use the generic one.
int launch_conv3x3_halo(GemmParams& p, hipStream_t s);

// 1x1 convolution with the whole packed weight matrix resident in LDS and the activations loaded straight
// into MFMA operand registers (conv1x1_resident.hip). Same return convention as launch_conv3x3_halo.
int launch_conv1x1_resident(GemmParams& p, hipStream_t s);
// The same kernel with the pair epilogue (GemmParams::pair_v); 1 = not a shape it takes, nothing launched.
int launch_conv1x1_pairs(GemmParams& p, hipStream_t s);

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

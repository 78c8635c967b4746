// 1x1 convolution (a plain GEMM: rows = pixels, K = input channels, columns = output channels) for the
// shapes of the attention blocks at full resolution: many rows, K and N of at most a few hundred. Such a
// GEMM moves more bytes than it computes (M*K in, M*N out, six bf16 MFMAs per 16x32x32 of products), so the
// kernel is built around touching every byte once and keeping the matrix pipe fed without barriers:
//
//  * the WHOLE packed weight matrix (fragment-major, three bf16 planes, at most 120 KiB) is copied into LDS
//    once per workgroup; one workgroup of 8 waves per CU stays resident and walks a run of 32-row groups;
//  * activations never touch LDS: lane (row r, half h) of a wave loads the 8 consecutive channels that the
//    MFMA A-operand layout assigns to it straight from global memory (two 16-byte loads, four steps ahead),
//    splits them into the three bf16 planes in registers and feeds the MFMAs;
//  * the 8 waves are 4 row groups x 2 column halves; a wave owns up to three 32-column blocks, reads their
//    weight fragments from LDS plane by plane as the previous step releases them and issues the six cross
//    products term-major across its blocks, so consecutive MFMAs hit different accumulators; the products
//    are summed in the generic kernel's order, which makes the two kernels agree bit for bit;
//  * no barrier after the prologue: every wave runs at its own pace.
// Accuracy scheme as in conv_gemm.hip (six cross products, MFMA partial sums folded into a master
// accumulator with a round-to-nearest add every 64 input channels).
#include "conv_common.h"

namespace ddpm {
namespace conv {
namespace {

#ifndef DDPM_C1R_ABLATE
#define DDPM_C1R_ABLATE 0   // diagnostic builds only (wrong results): bit 0 no output stores, bit 1 activations loaded once per kernel, bit 2 no MFMAs
#endif
constexpr int kThreads = 512;
constexpr int kMaxWeightLds = 120 * 1024;

// 8 consecutive fp32 (two 16-byte loads) -> the three bf16 planes of an MFMA operand fragment.
__device__ __forceinline__ void split8(const f32x4 v0, const f32x4 v1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    unsigned u[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? v0[e] : v1[e - 4];
        u[e] = __float_as_uint(x);
        const float r1 = x - __uint_as_float(u[e] & 0xffff0000u);
        m[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(m[e] & 0xffff0000u);
        l[e] = __float_as_uint(r2);
    }
    u32x4 ph, pm, pl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ph[e] = __builtin_amdgcn_perm(u[2 * e + 1], u[2 * e], 0x07060302u);
        pm[e] = __builtin_amdgcn_perm(m[2 * e + 1], m[2 * e], 0x07060302u);
        pl[e] = __builtin_amdgcn_perm(l[2 * e + 1], l[2 * e], 0x07060302u);
    }
    hi = __builtin_bit_cast(bf16x8, ph);
    mid = __builtin_bit_cast(bf16x8, pm);
    lo = __builtin_bit_cast(bf16x8, pl);
}

// One wave's share: row groups g_begin + wr, + 4, ... of the workgroup's run, column blocks nb0 .. nb0 + NBW - 1.
template <int NBW>
__device__ __forceinline__ void wave_body(const GemmParams& p, const unsigned char* wsm, int wr, int nb0, int lane);

template <int NB0, int NB1>   // column blocks of the waves of the first / second column half
__global__ __launch_bounds__(kThreads, 1) void conv1x1_resident_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];   // [chunk][column block][plane][lane][16 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves w and w + 4 share a SIMD: one of each column half there, so every SIMD carries the same MFMA load
    const int wr = wave & 3, wn = wave >> 2;

    {
        // all loads first, then all stores: one round trip to L2 instead of one per piece
        const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
        constexpr int kPieces = kMaxWeightLds / 16 / kThreads;
        u32x4 tmp[kPieces];
#pragma unroll
        for (int i = 0; i < kPieces; ++i) tmp[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (tid + i * kThreads) * 16u, 0, 0);
#pragma unroll
        for (int i = 0; i < kPieces; ++i)
            if ((tid + i * kThreads) * 16u < p.w_bytes) *reinterpret_cast<u32x4*>(wsm + (tid + i * kThreads) * 16u) = tmp[i];
    }
    __syncthreads();
    // the two column halves hold different numbers of blocks: each gets its own straight-line code
    if (wn == 0) wave_body<NB0>(p, wsm, wr, 0, lane);
    else if (NB1 > 0) wave_body<(NB1 > 0 ? NB1 : 1)>(p, wsm, wr, NB0, lane);
}

template <int NBW>
__device__ __forceinline__ void wave_body(const GemmParams& p, const unsigned char* wsm, int wr, int nb0, int lane) {
    const int r = lane & 31, h = lane >> 5;
    const int g_begin = __builtin_amdgcn_readfirstlane(blockIdx.x * p.gpb);
    const int g_end = __builtin_amdgcn_readfirstlane(min(p.groups_total, g_begin + p.gpb));
    if (g_begin + wr >= g_end) return;

    const int kc16 = p.kc16;
    const int ntile = (g_end - g_begin - wr + 3) >> 2;        // row groups g_begin + wr, + 4, ...
    const int nsteps = ntile * kc16;

    // activations: descriptor from this workgroup's first row to the end of the tensor
    const long long row0 = static_cast<long long>(g_begin) * 32;
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.src1 + row0 * p.ld1, (p.src1_pixels - row0) * p.ld1 * 4ll);
    const unsigned ld4 = static_cast<unsigned>(p.ld1) * 4u;
    // (tile, chunk) of the next load to request
    int lt = 0, lk = 0;
    auto load_a = [&](f32x4 (&v)[2]) {
        const int row = (wr + 4 * lt) * 32 + r;               // relative to row0
        const bool live = lt < ntile && row0 + row < p.m_total;
        const int c = lk * XK + h * 8;
        const unsigned base = static_cast<unsigned>(row) * ld4 + static_cast<unsigned>(c) * 4u;
        v[0] = buf_load(rs_a, (live && c < p.ctot) ? base : OOB);
        v[1] = buf_load(rs_a, (live && c + 4 < p.ctot) ? base + 16u : OOB);
        if (++lk == kc16) {
            lk = 0;
            ++lt;
        }
    };
    // weight fragments of one plane of chunk kc for this wave's blocks
    const unsigned char* wbase = wsm + static_cast<unsigned>(nb0) * 3072u + lane * 16u;
    const unsigned chunk_stride = static_cast<unsigned>(p.nbp) * 3072u;
    // planes 1 and 2 have one register set, refilled for the next step as soon as this step's last term with
    // them has issued; plane 0 is used by the first and by the last term of a step, so it alternates between two
    bf16x8 bw[NBW][3], bw0[2][NBW];
    auto read_b = [&](int kc, int pl) {
#pragma unroll
        for (int j = 0; j < NBW; ++j)
            bw[j][pl] = *reinterpret_cast<const bf16x8*>(wbase + kc * chunk_stride + j * 3072u + pl * 1024u);
    };
    auto read_b0 = [&](bf16x8 (&dst)[NBW], int kc) {
#pragma unroll
        for (int j = 0; j < NBW; ++j) dst[j] = *reinterpret_cast<const bf16x8*>(wbase + kc * chunk_stride + j * 3072u);
    };

    f32x4 ra[4][2];                 // activations of the next four steps, in flight
#pragma unroll
    for (int i = 0; i < 4; ++i) load_a(ra[i]);
    read_b0(bw0[0], 0);
    read_b(0, 1);
    read_b(0, 2);

    f32x16 acc[NBW], master[NBW];
    int kc = 0, t = 0;
    auto zero = [](f32x16& v) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = 0.f;
    };
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        zero(acc[j]);
        zero(master[j]);
    }

    auto step = [&](f32x4 (&raw)[2], bf16x8 (&b0cur)[NBW], bf16x8 (&b0next)[NBW]) {
        bf16x8 a0, a1, a2;
        split8(raw[0], raw[1], a0, a1, a2);
        kc = __builtin_amdgcn_readfirstlane(kc);
        t = __builtin_amdgcn_readfirstlane(t);
        if (!(DDPM_C1R_ABLATE & 2)) load_a(raw);              // this register set now carries step s + 4
        const int kn = kc + 1 == kc16 ? 0 : kc + 1;
        auto term = [&](const bf16x8& a, int pb) {
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                const bf16x8& b = pb == 0 ? b0cur[j] : bw[j][pb];
                if (!(DDPM_C1R_ABLATE & 4)) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
                else acc[j][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, a)[0] ^ __builtin_bit_cast(u32x4, b)[1]);
            }
        };
        // The order of the six products (smallest first) and the fold interval are those of the generic GEMM
        // kernel, so that both kernels produce the same bits for the same rows (which kernel runs depends on the
        // number of rows, and a result must not depend on how a batch was cut).
        read_b0(b0next, kn);
        term(a2, 0);
        term(a0, 2);
        read_b(kn, 2);
        term(a1, 1);
        term(a1, 0);
        term(a0, 1);
        read_b(kn, 1);
        term(a0, 0);
        const bool last = kc + 1 == kc16;
        if ((kc & (XFLUSH - 1)) == XFLUSH - 1 || last) {
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                master[j] += acc[j];
                zero(acc[j]);
            }
        }
        if (last) {
            // ---- epilogue of row group g: bias / folded BatchNorm, residual, activation ----------------
            // (wave-uniform by construction; said explicitly so that the descriptors below are built from scalars)
            const int g = __builtin_amdgcn_readfirstlane(g_begin + wr + 4 * t);
            const long long m0 = static_cast<long long>(g) * 32;
            const long long rows_left = p.m_total - m0;
            const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out + m0 * p.ldo, rows_left * p.ldo * 4ll);
            const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.residual ? p.residual + m0 * p.ldr : p.out, p.residual ? rows_left * p.ldr * 4ll : 0ll);
            const unsigned ldo4 = static_cast<unsigned>(p.ldo) * 4u, ldr4 = static_cast<unsigned>(p.ldr) * 4u;
            int hq = 4 * h;
            asm volatile("" : "+v"(hq));   // keeps the per-element offsets below from being hoisted out of the step loop (registers)
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                const int n = (nb0 + j) * 32 + r;
                if (n < p.n_total) {
                    // every mode decision is taken once per block of 16 elements, not per element (the conditions
                    // are uniform, but a branch per element costs more than the arithmetic it skips)
                    const float sc = p.scale ? p.scale[n] : 1.0f;
                    const float sh = p.shift ? p.shift[n] : 0.0f;
                    const unsigned n4 = static_cast<unsigned>(n) * 4u;
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = master[j][e] * sc + sh;
                    if (p.residual) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const unsigned row = static_cast<unsigned>((e & 3) + 8 * (e >> 2) + hq);
                            v[e] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, row * ldr4 + n4, 0, 0));
                        }
                    }
                    if (p.act == DDPM_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    } else if (p.act == DDPM_ACT_GELU) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = apply_act_fast(v[e], DDPM_ACT_GELU);
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const unsigned row = static_cast<unsigned>((e & 3) + 8 * (e >> 2) + hq);
                        if (!(DDPM_C1R_ABLATE & 1) || v[e] == 1.2345f)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), rs_o, row * ldo4 + n4, 0, 0);
                    }
                }
                zero(master[j]);
            }
            kc = 0;
            ++t;
        } else {
            ++kc;
        }
    };

    // four steps per iteration: the four activation register sets are named, not indexed
    int s = 0;
    for (; s + 3 < nsteps; s += 4) {
        step(ra[0], bw0[0], bw0[1]);
        step(ra[1], bw0[1], bw0[0]);
        step(ra[2], bw0[0], bw0[1]);
        step(ra[3], bw0[1], bw0[0]);
    }
    if (s < nsteps) step(ra[0], bw0[0], bw0[1]);
    if (s + 1 < nsteps) step(ra[1], bw0[1], bw0[0]);
    if (s + 2 < nsteps) step(ra[2], bw0[0], bw0[1]);
}

template <int NB0, int NB1>
int launch_one(const GemmParams& p, int grid, int lds, hipStream_t s) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_resident_kernel<NB0, NB1>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMaxWeightLds);
    if (attr != hipSuccess) {
        set_error("conv1x1_resident_kernel: cannot reserve %d bytes of LDS: %s", kMaxWeightLds, hipGetErrorString(attr));
        return DDPM_E_LAUNCH;
    }
    conv1x1_resident_kernel<NB0, NB1><<<dim3(grid), dim3(kThreads), lds, s>>>(p);
    return check_launch("conv1x1_resident_kernel");
}

}  // namespace

int launch_conv1x1_resident(GemmParams& p, hipStream_t s) {
    // shapes left to the generic kernel: a second source, a time-embedding add, few rows (the weight copy
    // into LDS would not pay), more columns than two waves' worth of blocks, weights beyond the LDS budget
    if (p.c2 > 0 || p.temb || p.m_total < 16384 || (p.ctot & 3) != 0) return 1;
    p.kc16 = (p.ctot + XK - 1) / XK;
    p.nbp = (p.n_total + 31) / 32;
    if (p.nbp > 6) return 1;
    const long long wb = 3072ll * p.nbp * p.kc16;
    if (wb > kMaxWeightLds) return 1;
    if (static_cast<long long>(p.m_total) * p.ld1 * 4ll >= 0x7fffffffll) return 1;   // 32-bit offsets inside one descriptor
    p.w_bytes = static_cast<unsigned>(wb);
    p.groups_total = (p.m_total + 31) / 32;
    const int gpb = (p.groups_total + 255) / 256;      // one workgroup per CU; its four row waves take every fourth group
    p.gpb = gpb;
    const int grid = (p.groups_total + gpb - 1) / gpb;
    const int lds = static_cast<int>(wb);
    switch (p.nbp) {
        case 1: return launch_one<1, 0>(p, grid, lds, s);
        case 2: return launch_one<1, 1>(p, grid, lds, s);
        case 3: return launch_one<2, 1>(p, grid, lds, s);
        case 4: return launch_one<2, 2>(p, grid, lds, s);
        case 5: return launch_one<3, 2>(p, grid, lds, s);
        default: return launch_one<3, 3>(p, grid, lds, s);
    }
}

}  // namespace conv
}  // namespace ddpm

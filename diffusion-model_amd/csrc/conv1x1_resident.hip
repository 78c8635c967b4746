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
#include <type_traits>

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
__device__ __forceinline__ void split8(const f32x4 v0, const f32x4 v1, u32x4& hi, u32x4& mid, u32x4& lo) {
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
    hi = ph;
    mid = pm;
    lo = pl;
}

// One wave's share: row groups g_begin + wr, + 4, ... of the workgroup's run, column blocks nb0 .. nb0 + NBW - 1.
// WATCH: this wave watches the range of the activations it loads (F16); the two column halves load the same rows, so
// the first half does the checking for both.
// PAIRS (F16 only): the epilogue writes k as (hi | lo << 16) words and v as pair planes (GemmParams::pair_v) and
// watches the range of what it splits; every wave then has something to check.
template <int NBW, bool F16, bool WATCH, bool PAIRS>
__device__ __forceinline__ void wave_body(const GemmParams& p, const unsigned char* wsm, int wr, int nb0, int lane);

// With F16 the operands are fp16 pairs (conv_common.h): planes 0 and 1 of the packed weights, three MFMAs per step and
// block, the hi*hi products in `acc` and the cross products in `master`, scaled by 2^-11 and added in the epilogue.
template <int NB0, int NB1, bool F16, bool PAIRS = false>   // column blocks of the waves of the first / second column half
__global__ __launch_bounds__(kThreads, 1) void conv1x1_resident_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];   // [chunk][column block][plane][lane][16 B]
    if constexpr (F16) fp16_saturate();

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
    if (wn == 0) wave_body<NB0, F16, true, PAIRS>(p, wsm, wr, 0, lane);
    else if (NB1 > 0) wave_body<(NB1 > 0 ? NB1 : 1), F16, false, PAIRS>(p, wsm, wr, NB0, lane);
}

template <int NBW, bool F16, bool WATCH, bool PAIRS>
__device__ __forceinline__ void wave_body(const GemmParams& p, const unsigned char* wsm, int wr, int nb0, int lane) {
    const int r = lane & 31, h = lane >> 5;
    const int g_begin = __builtin_amdgcn_readfirstlane(blockIdx.x * p.gpb);
    const int g_end = __builtin_amdgcn_readfirstlane(min(p.groups_total, g_begin + p.gpb));
    if (g_begin + wr >= g_end) return;
    std::conditional_t<WATCH || PAIRS, RangeWatch, NoWatch> amax;   // F16: largest magnitude this thread has turned into an fp16 pair
    static_assert(F16 || !PAIRS, "the pair epilogue belongs to the fp16-pair arithmetic");

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
    u32x4 bw[NBW][3], bw0[2][NBW];
    auto read_b = [&](int kc, int pl) {
#pragma unroll
        for (int j = 0; j < NBW; ++j)
            bw[j][pl] = *reinterpret_cast<const u32x4*>(wbase + kc * chunk_stride + j * 3072u + pl * 1024u);
    };
    auto read_b0 = [&](u32x4 (&dst)[NBW], int kc) {
#pragma unroll
        for (int j = 0; j < NBW; ++j) dst[j] = *reinterpret_cast<const u32x4*>(wbase + kc * chunk_stride + j * 3072u);
    };

    f32x4 ra[4][2];                 // activations of the next four steps, in flight
#pragma unroll
    for (int i = 0; i < 4; ++i) load_a(ra[i]);
    read_b0(bw0[0], 0);
    read_b(0, 1);
    if constexpr (!F16) read_b(0, 2);

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

    auto step = [&](f32x4 (&raw)[2], u32x4 (&b0cur)[NBW], u32x4 (&b0next)[NBW]) {
        u32x4 a0, a1, a2;
        if constexpr (F16 && WATCH) split8_f16(raw[0], raw[1], a0, a1, amax);
        else if constexpr (F16) {
            NoWatch none;
            split8_f16(raw[0], raw[1], a0, a1, none);
        } else split8(raw[0], raw[1], a0, a1, a2);
        kc = __builtin_amdgcn_readfirstlane(kc);
        t = __builtin_amdgcn_readfirstlane(t);
        if (!(DDPM_C1R_ABLATE & 2)) load_a(raw);              // this register set now carries step s + 4
        const int kn = kc + 1 == kc16 ? 0 : kc + 1;
        auto term = [&](f32x16 (&c)[NBW], const u32x4& a, int pb) {
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                const u32x4& b = pb == 0 ? b0cur[j] : bw[j][pb];
                if (!(DDPM_C1R_ABLATE & 4)) c[j] = mfma16<F16>(a, b, c[j]);
                else c[j][0] += __builtin_bit_cast(float, a[0] ^ b[1]);
            }
        };
        // The order of the products (smallest first) and the fold interval are those of the generic GEMM
        // kernel, so that both kernels produce the same bits for the same rows (which kernel runs depends on the
        // number of rows, and a result must not depend on how a batch was cut).
        const bool last = kc + 1 == kc16;
        read_b0(b0next, kn);
        if constexpr (F16) {
            term(master, a1, 0);
            term(master, a0, 1);
            read_b(kn, 1);
            term(acc, a0, 0);
            if (last) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    master[j] = acc[j] + master[j] * kLoInv;
                    zero(acc[j]);
                }
            }
        } else {
            term(acc, a2, 0);
            term(acc, a0, 2);
            read_b(kn, 2);
            term(acc, a1, 1);
            term(acc, a1, 0);
            term(acc, a0, 1);
            read_b(kn, 1);
            term(acc, a0, 0);
            if ((kc & (XFLUSH - 1)) == XFLUSH - 1 || last) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    master[j] += acc[j];
                    zero(acc[j]);
                }
            }
        }
        if (last) {
            // ---- epilogue of row group g: bias / folded BatchNorm, residual, activation ----------------
            if (PAIRS && p.nbp > 3) __builtin_amdgcn_s_setprio(0);
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
                    if constexpr (PAIRS) {
                        // (no residual, no activation in this mode: the launcher checks)
                        if (n >= p.pair_from) {
                            // v: registers 8ck .. 8ck + 7 are rows 16ck + 4h + i and 16ck + 8 + 4h + i (i < 4) of the
                            // group, the eight keys that half h of a matrix operand takes from 16-key chunk ck: one
                            // 16-byte store per chunk and plane at [chunk][h][plane][channel]
                            const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(
                                p.pair_v + m0 * p.pair_c * 4ll, 0, static_cast<unsigned>(32 * p.pair_c * 4), 0x00020000);
                            const unsigned ch16 = static_cast<unsigned>(n - p.pair_from) * 16u;
                            const unsigned pc16 = static_cast<unsigned>(p.pair_c) * 16u;
#pragma unroll
                            for (int ck = 0; ck < 2; ++ck) {
                                u32x4 ph, pl;
                                split8_f16(f32x4{v[8 * ck], v[8 * ck + 1], v[8 * ck + 2], v[8 * ck + 3]},
                                           f32x4{v[8 * ck + 4], v[8 * ck + 5], v[8 * ck + 6], v[8 * ck + 7]}, ph, pl, amax);
                                const unsigned o = static_cast<unsigned>((ck * 2 + (hq >> 2)) * 2) * pc16 + ch16;
                                __builtin_amdgcn_raw_buffer_store_b128(ph, rs_v, o, 0, DDPM_STORE_AUX);
                                __builtin_amdgcn_raw_buffer_store_b128(pl, rs_v, o + pc16, 0, DDPM_STORE_AUX);
                            }
                        } else {
                            if (2 * n >= p.pair_from) {   // k: each element as (hi | lo << 16)
#pragma unroll
                                for (int e = 0; e < 16; e += 2) {
                                    unsigned hh, ll;
                                    watch2(amax, f32x2{v[e], v[e + 1]});
                                    split2_pair(f32x2{v[e], v[e + 1]}, hh, ll);
                                    v[e] = __builtin_bit_cast(float, __builtin_amdgcn_perm(ll, hh, 0x05040100u));
                                    v[e + 1] = __builtin_bit_cast(float, __builtin_amdgcn_perm(ll, hh, 0x07060302u));
                                }
                            }
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const unsigned row = static_cast<unsigned>((e & 3) + 8 * (e >> 2) + hq);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), rs_o, row * ldo4 + n4, 0, DDPM_STORE_AUX);
                            }
                        }
                        zero(master[j]);
                        continue;
                    }
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
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), rs_o, row * ldo4 + n4, 0, DDPM_STORE_AUX);
                    }
                }
                zero(master[j]);
            }
            if (PAIRS && p.nbp > 3) __builtin_amdgcn_s_setprio(2);
            kc = 0;
            ++t;
        } else {
            ++kc;
        }
    };

    // four steps per iteration: the four activation register sets are named, not indexed
    // (The pair epilogue is long vector work -- splitting 16 values per block. Raised priority for the waves that are in
    // their steps keeps their loads and matrix instructions from queueing behind it: 1-2 % of the 128-channel attention
    // block. With the plain epilogue, and with the pair epilogue of the three-block shape (64 channels), the same setting
    // measured 1 % slower, so only the five-block pair variant has it.)
    if (PAIRS && p.nbp > 3) __builtin_amdgcn_s_setprio(2);
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
    if constexpr (F16 && (WATCH || PAIRS)) raise_range_flag(p.range_flag, amax);
}

template <int NB0, int NB1, bool F16>
int launch_one_math(const GemmParams& p, int grid, int lds, hipStream_t s) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_resident_kernel<NB0, NB1, F16>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMaxWeightLds);
    if (attr != hipSuccess) {
        set_error("conv1x1_resident_kernel: cannot reserve %d bytes of LDS: %s", kMaxWeightLds, hipGetErrorString(attr));
        return DDPM_E_LAUNCH;
    }
    conv1x1_resident_kernel<NB0, NB1, F16><<<dim3(grid), dim3(kThreads), lds, s>>>(p);
    return check_launch("conv1x1_resident_kernel");
}

template <int NB0, int NB1>
int launch_one(const GemmParams& p, int grid, int lds, hipStream_t s) {
    if (f16_math() && !p.range_flag) return DDPM_E_LAUNCH;
    return f16_math() ? launch_one_math<NB0, NB1, true>(p, grid, lds, s) : launch_one_math<NB0, NB1, false>(p, grid, lds, s);
}

// ---- the same kernel for weight matrices that do not fit LDS (256 and 512 channels) ----------------------------
// The columns are cut into tiles of at most six 32-column blocks (grid.y; the workgroups of one row range run on one
// XCD and share the activations in its L2) and K into slabs of four 16-channel chunks. A slab of the tile's weights
// (at most 72 KiB) sits in one of two LDS buffers; all 512 threads request the next slab at the top of a slab, store it
// into the other buffer behind the slab's four steps, and one barrier per slab hands it over. A slab is four steps,
// which is also the depth of the activation prefetch ring and the fold interval. Same product order and folds as the
// resident kernel and the generic one: the three agree bit for bit.
constexpr int kSlabChunks = 4;

template <int NBW, bool F16, bool WATCH>   // WATCH: as in wave_body; here only the first column tile's first half checks
__device__ __forceinline__ void slab_wave_body(const GemmParams& p, unsigned char* wsm, int wr, int nb_in_tile, int nb_first, int nbt,
                                               int g_begin, int g_end, int lane, const __amdgpu_buffer_rsrc_t rs_w, unsigned tile_off) {
    const int tid = threadIdx.x;
    const int r = lane & 31, h = lane >> 5;
    const int kc16 = p.kc16, nslab = kc16 / kSlabChunks;
    const int groups = g_end - g_begin;
    const int max_tile = (groups + 3) >> 2;                    // row groups of the busiest wave: every wave runs that many rounds
    const int ntile = groups > wr ? (groups - wr + 3) >> 2 : 0;
    const unsigned slab_bytes = static_cast<unsigned>(kSlabChunks * nbt) * 3072u;
    const unsigned run_bytes = static_cast<unsigned>(nbt) * 3072u;          // one chunk of the tile: contiguous in the packed weights
    const unsigned chunk_stride_g = static_cast<unsigned>(p.nbp) * 3072u;

    // this thread's pieces of a slab: 16-byte piece i of chunk c of the slab comes from global offset
    // (slab * 4 + c) * chunk_stride_g + tile_off + i * 16
    constexpr int kPieces = (kSlabChunks * 6 * 3072 / 16 + kThreads - 1) / kThreads;   // 9 for six blocks
    u32x4 wreg[kPieces];
    auto load_slab = [&](int slab) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const unsigned q = (tid + i * kThreads) * 16u;                 // byte offset inside the slab's LDS image
            const unsigned c = q / run_bytes, o = q - c * run_bytes;
            const unsigned goff = (static_cast<unsigned>(slab) * kSlabChunks + c) * chunk_stride_g + tile_off + o;
            // columns past the last block of the matrix (a ragged last tile) lie in the next chunk's run: read zeros instead
            const bool ok = q < slab_bytes && tile_off + o < chunk_stride_g;
            wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, ok ? goff : OOB, 0, 0);
        }
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const unsigned q = (tid + i * kThreads) * 16u;
            if (q < slab_bytes) *reinterpret_cast<u32x4*>(wsm + buf * slab_bytes + q) = wreg[i];
        }
    };

    const long long row0 = static_cast<long long>(g_begin) * 32;
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.src1 + row0 * p.ld1, (p.src1_pixels - row0) * p.ld1 * 4ll);
    const unsigned ld4 = static_cast<unsigned>(p.ld1) * 4u;
    int lt = 0, lk = 0;
    auto load_a = [&](f32x4 (&v)[2]) {
        const int row = (wr + 4 * lt) * 32 + r;
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

    u32x4 bw[NBW][3], bw0[2][NBW];
    const unsigned lane_off = static_cast<unsigned>(nb_in_tile) * 3072u + lane * 16u;
    const unsigned char* wcur = wsm;            // current slab buffer + this wave's blocks
    auto read_b = [&](int c, int pl) {
#pragma unroll
        for (int j = 0; j < NBW; ++j) bw[j][pl] = *reinterpret_cast<const u32x4*>(wcur + c * run_bytes + j * 3072u + pl * 1024u);
    };
    auto read_b0 = [&](u32x4 (&dst)[NBW], int c) {
#pragma unroll
        for (int j = 0; j < NBW; ++j) dst[j] = *reinterpret_cast<const u32x4*>(wcur + c * run_bytes + j * 3072u);
    };

    f32x4 ra[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) load_a(ra[i]);

    f32x16 acc[NBW], master[NBW];
    auto zero = [](f32x16& v) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = 0.f;
    };
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        zero(acc[j]);
        zero(master[j]);
    }
    std::conditional_t<WATCH, RangeWatch, NoWatch> amax;   // F16: largest magnitude this thread has turned into an fp16 pair

    // step c (0..3) of the current slab
    auto step = [&](int c, f32x4 (&raw)[2], u32x4 (&b0cur)[NBW], u32x4 (&b0next)[NBW]) {
        u32x4 a0, a1, a2;
        if constexpr (F16) split8_f16(raw[0], raw[1], a0, a1, amax);
        else split8(raw[0], raw[1], a0, a1, a2);
        load_a(raw);
        auto term = [&](f32x16 (&cc)[NBW], const u32x4& a, int pb) {
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                const u32x4& b = pb == 0 ? b0cur[j] : bw[j][pb];
                cc[j] = mfma16<F16>(a, b, cc[j]);
            }
        };
        const bool more = c + 1 < kSlabChunks;      // the next step's fragments are in this slab
        if (more) read_b0(b0next, c + 1);
        if constexpr (F16) {
            term(master, a1, 0);
            term(master, a0, 1);
            if (more) read_b(c + 1, 1);
            term(acc, a0, 0);
        } else {
            term(acc, a2, 0);
            term(acc, a0, 2);
            if (more) read_b(c + 1, 2);
            term(acc, a1, 1);
            term(acc, a1, 0);
            term(acc, a0, 1);
            if (more) read_b(c + 1, 1);
            term(acc, a0, 0);
        }
    };

    load_slab(0);
    store_slab(0);
    __syncthreads();

    const int rounds = max_tile * nslab;
    int t = 0, sl = 0;                               // row-group round and slab of this iteration
    for (int q = 0; q < rounds; ++q) {
        const int buf = q & 1;
        const bool more = q + 1 < rounds;
        const int sl_next = sl + 1 == nslab ? 0 : sl + 1;
        if (more) load_slab(sl_next);
        if (t < ntile) {
            wcur = wsm + buf * slab_bytes + lane_off;
            read_b0(bw0[0], 0);
            read_b(0, 1);
            if constexpr (!F16) read_b(0, 2);
            step(0, ra[0], bw0[0], bw0[1]);
            step(1, ra[1], bw0[1], bw0[0]);
            step(2, ra[2], bw0[0], bw0[1]);
            step(3, ra[3], bw0[1], bw0[0]);
            if constexpr (!F16) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) {          // a slab is one fold interval
                    master[j] += acc[j];
                    zero(acc[j]);
                }
            } else if (sl + 1 == nslab) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    master[j] = acc[j] + master[j] * kLoInv;
                    zero(acc[j]);
                }
            }
            if (sl + 1 == nslab) {
                // ---- epilogue of row group g: bias / folded BatchNorm, residual, activation ----------------
                const int g = __builtin_amdgcn_readfirstlane(g_begin + wr + 4 * t);
                const long long m0 = static_cast<long long>(g) * 32;
                const long long rows_left = p.m_total - m0;
                const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out + m0 * p.ldo, rows_left * p.ldo * 4ll);
                const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.residual ? p.residual + m0 * p.ldr : p.out, p.residual ? rows_left * p.ldr * 4ll : 0ll);
                const unsigned ldo4 = static_cast<unsigned>(p.ldo) * 4u, ldr4 = static_cast<unsigned>(p.ldr) * 4u;
                int hq = 4 * h;
                asm volatile("" : "+v"(hq));   // keeps the per-element offsets below from being hoisted out of the loop (registers)
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    const int n = (nb_first + j) * 32 + r;
                    if (n < p.n_total) {
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
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), rs_o, row * ldo4 + n4, 0, DDPM_STORE_AUX);
                        }
                    }
                    zero(master[j]);
                }
            }
        }
        if (more) store_slab(buf ^ 1);
        __syncthreads();
        if (++sl == nslab) {
            sl = 0;
            ++t;
        }
    }
    if constexpr (F16 && WATCH) raise_range_flag(p.range_flag, amax);
}

template <int NB0, int NB1, bool F16>
__global__ __launch_bounds__(kThreads, 1) void conv1x1_slab_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];   // [2][chunk of the slab][block of the tile][plane][lane][16 B]
    if constexpr (F16) fp16_saturate();
    constexpr int NBT = NB0 + NB1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 3, wn = wave >> 2;
    // workgroups that share rows (the column tiles of one row range) take consecutive numbers on ONE XCD
    const int nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int q = nwg >> 3, rr = nwg & 7, xcd = id & 7;
        id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
    }
    const int ct = id % p.n_tiles;
    const int rw = id / p.n_tiles;
    const int g_begin = __builtin_amdgcn_readfirstlane(rw * p.gpb);
    const int g_end = __builtin_amdgcn_readfirstlane(min(p.groups_total, g_begin + p.gpb));
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
    const unsigned tile_off = static_cast<unsigned>(ct * NBT) * 3072u;
    if (g_begin >= g_end) return;       // (whole workgroup: no barrier is skipped by part of it)
    if (wn == 0 && ct == 0) slab_wave_body<NB0, F16, true>(p, wsm, wr, 0, ct * NBT, NBT, g_begin, g_end, lane, rs_w, tile_off);
    else if (wn == 0) slab_wave_body<NB0, F16, false>(p, wsm, wr, 0, ct * NBT, NBT, g_begin, g_end, lane, rs_w, tile_off);
    else slab_wave_body<(NB1 > 0 ? NB1 : 1), F16, false>(p, wsm, wr, NB0, ct * NBT + NB0, NBT, g_begin, g_end, lane, rs_w, tile_off);
}

template <int NB0, int NB1, bool F16>
int launch_slab_math(const GemmParams& p, int grid, int lds, hipStream_t s) {
    constexpr int kMax = 2 * kSlabChunks * 6 * 3072;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_slab_kernel<NB0, NB1, F16>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
    if (attr != hipSuccess) {
        set_error("conv1x1_slab_kernel: cannot reserve %d bytes of LDS: %s", kMax, hipGetErrorString(attr));
        return DDPM_E_LAUNCH;
    }
    conv1x1_slab_kernel<NB0, NB1, F16><<<dim3(grid), dim3(kThreads), lds, s>>>(p);
    return check_launch("conv1x1_slab_kernel");
}

template <int NB0, int NB1>
int launch_slab(const GemmParams& p, int grid, int lds, hipStream_t s) {
    if (f16_math() && !p.range_flag) return DDPM_E_LAUNCH;
    return f16_math() ? launch_slab_math<NB0, NB1, true>(p, grid, lds, s) : launch_slab_math<NB0, NB1, false>(p, grid, lds, s);
}

// Returns 1 when the shape is not this kernel's (the caller goes on to the generic one).
int launch_conv1x1_slab(GemmParams& p, hipStream_t s) {
    if (p.c2 > 0 || p.temb || p.m_total < 16384 || (p.ctot % (XK * kSlabChunks)) != 0) return 1;
    p.kc16 = p.ctot / XK;
    p.nbp = (p.n_total + 31) / 32;
    const long long wb = 3072ll * p.nbp * p.kc16;
    if (wb >= 0x7fffffffll) return 1;
    p.w_bytes = static_cast<unsigned>(wb);
    p.groups_total = (p.m_total + 31) / 32;
    // One workgroup per CU: 256 of them, shared between the column tiles; a wave takes every fourth row group of its
    // workgroup. The busiest SIMD issues (rounds of row groups) x (blocks per tile) MFMA sets, and every round reads its
    // rows once more (the "+ 1"): the fewest tiles that hold six blocks each, or one tile more when that is cheaper
    // (16 blocks: four tiles of four instead of three of six, which would leave a quarter of the CUs idle).
    int best_nt = 0, best_cost = 0, nbt = 0, gpb = 0;
    for (int nt = (p.nbp + 5) / 6; nt <= (p.nbp + 5) / 6 + 1 && nt * 2 <= p.nbp; ++nt) {
        const int b = (p.nbp + nt - 1) / nt;
        const int row_wgs = 256 / nt;
        const int g = ((p.groups_total + row_wgs - 1) / row_wgs + 3) & ~3;     // whole rounds: the four row waves stay level
        const int cost = (g / 4) * (b + 1);
        if (best_nt == 0 || cost < best_cost) {
            best_nt = nt;
            best_cost = cost;
            nbt = b;
            gpb = g;
        }
    }
    if (best_nt == 0) return 1;
    p.n_tiles = best_nt;
    p.gpb = gpb;
    // 32-bit offsets inside a workgroup's descriptors, which begin at ITS first row: its run of rows must stay below 2 GiB
    if (static_cast<long long>(gpb) * 32 * (p.ld1 > p.ldo ? p.ld1 : p.ldo) * 4ll >= 0x7fffffffll) return 1;
    const int grid = ((p.groups_total + gpb - 1) / gpb) * p.n_tiles;
    const int lds = 2 * kSlabChunks * nbt * 3072;
    switch (nbt) {
        case 2: return launch_slab<1, 1>(p, grid, lds, s);
        case 3: return launch_slab<2, 1>(p, grid, lds, s);
        case 4: return launch_slab<2, 2>(p, grid, lds, s);
        case 5: return launch_slab<3, 2>(p, grid, lds, s);
        default: return launch_slab<3, 3>(p, grid, lds, s);
    }
}

}  // namespace

int launch_conv1x1_resident(GemmParams& p, hipStream_t s) {
    // shapes left to the generic kernel: a second source, a time-embedding add, few rows (the weight copy
    // into LDS would not pay), more columns than two waves' worth of blocks, weights beyond the LDS budget
    if (p.c2 > 0 || p.temb || p.m_total < 16384 || (p.ctot & 3) != 0) return 1;
    p.kc16 = (p.ctot + XK - 1) / XK;
    p.nbp = (p.n_total + 31) / 32;
    const long long wb = 3072ll * p.nbp * p.kc16;
    if (p.nbp > 6 || wb > kMaxWeightLds) return launch_conv1x1_slab(p, s);
    p.w_bytes = static_cast<unsigned>(wb);
    p.groups_total = (p.m_total + 31) / 32;
    const int gpb = (p.groups_total + 255) / 256;      // one workgroup per CU; its four row waves take every fourth group
    p.gpb = gpb;
    // 32-bit offsets inside a workgroup's descriptors, which begin at ITS first row: its run of rows must stay below 2 GiB
    // (the tensor itself may be larger)
    if (static_cast<long long>(gpb) * 32 * (p.ld1 > p.ldo ? p.ld1 : p.ldo) * 4ll >= 0x7fffffffll) return 1;
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

namespace {
template <int NB0, int NB1>
int launch_pairs(const GemmParams& p, int grid, int lds, hipStream_t s) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_resident_kernel<NB0, NB1, true, true>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMaxWeightLds);
    if (attr != hipSuccess) {
        set_error("conv1x1_resident_kernel (pairs): cannot reserve %d bytes of LDS: %s", kMaxWeightLds, hipGetErrorString(attr));
        return DDPM_E_LAUNCH;
    }
    conv1x1_resident_kernel<NB0, NB1, true, true><<<dim3(grid), dim3(kThreads), lds, s>>>(p);
    return check_launch("conv1x1_resident_kernel (pairs)");
}
}  // namespace

// The q|k|v projection of an inference-time attention block with its k and v columns written as fp16 pairs. Takes what
// the resident kernel takes in the fp16-pair arithmetic, for the two shapes that have pair-consuming attention kernels
// behind them (q|k of 8 + 8 or 16 + 16 columns, v of 64 or 128): 3 or 5 column blocks. Whole row groups only.
int launch_conv1x1_pairs(GemmParams& p, hipStream_t s) {
    if (!f16_math() || p.c2 > 0 || p.temb || p.residual || p.scale || p.act != DDPM_ACT_NONE || p.m_total < 16384 || (p.ctot & 3) != 0) return 1;
    if ((p.m_total & 31) != 0 || !p.pair_v || p.n_total != p.pair_from + p.pair_c || p.ldo != p.pair_from) return 1;
    if (!p.range_flag) return DDPM_E_LAUNCH;
    p.kc16 = (p.ctot + XK - 1) / XK;
    p.nbp = (p.n_total + 31) / 32;
    const long long wb = 3072ll * p.nbp * p.kc16;
    if ((p.nbp != 3 && p.nbp != 5) || wb > kMaxWeightLds) return 1;
    p.w_bytes = static_cast<unsigned>(wb);
    p.groups_total = p.m_total / 32;
    const int gpb = (p.groups_total + 255) / 256;
    p.gpb = gpb;
    if (static_cast<long long>(gpb) * 32 * (p.ld1 > p.n_total ? p.ld1 : p.n_total) * 4ll >= 0x7fffffffll) return 1;
    const int grid = (p.groups_total + gpb - 1) / gpb;
    const int lds = static_cast<int>(wb);
    return p.nbp == 3 ? launch_pairs<2, 1>(p, grid, lds, s) : launch_pairs<3, 2>(p, grid, lds, s);
}

}  // namespace conv
}  // namespace ddpm

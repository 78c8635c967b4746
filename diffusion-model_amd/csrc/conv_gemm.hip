// Implicit-GEMM convolutions on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   rows    (M) = input pixels (b, y, x) of a window of the input image
//   columns (N) = output channels (x4 sub-positions for the 2x2 transposed conv)
//   depth   (K) = taps x input channels, walked in steps of 32 channels of one tap
//
// Work decomposition. Rows are handled in groups of 32 (one MFMA tile). The host gives every
// workgroup a contiguous run of `gpb` row groups of one column tile, sized so that the grid is one
// resident round of the chip (256 CUs x 2 workgroups); the workgroup walks its run in tiles of up to
// 4 groups (128 rows) and the last, shorter tile only issues the MFMAs of the groups it has. This
// removes the "3.06 rounds cost 4" quantisation of a fixed 128-row tiling.
//
// Per tile, each K-step stages a (<=128) x 32 tile of shifted, zero-padded activations and a
// BN x 32 tile of weights through LDS: buffer loads for step s+1 are issued into registers before
// the MFMAs of step s and written to the other LDS buffer after them, so one barrier per step
// suffices. All loads are branch-free: an out-of-image tap, a row or column past the end, or a
// channel past Cin gets the offset 0x80000000, which the buffer descriptor's range check turns into
// zeros. Per-row offsets are recomputed only when the tap changes. LDS rows are padded to 36 floats,
// which makes the 16-byte fragment reads (lane = row) conflict-free.
//
// Fragment mapping: lane l = (r = l & 31, h = l >> 5) reads four consecutive k at 4h..4h+3 of an
// 8-wide chunk; MFMA number e of the chunk multiplies element e of the A and B fragments, i.e.
// k = e (lanes h=0) and k = 4+e (lanes h=1). A and B use the same permutation, so the sum over
// the chunk is complete after four MFMAs. Accumulator: column = lane & 31,
// row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5). The 4 waves form a 2 x 2 grid; row group g of a
// tile belongs to wave row g & 1 (slot g >> 1), so a short tile keeps both wave rows busy.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "attention_small.h"
#include "conv_common.h"

using ddpm::as_stream;
using namespace ddpm::conv;

namespace {

template <int BN, int MODE, bool DUAL>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const GemmParams p) {
    constexpr int TN = BN / 64;  // 32-column MFMA tiles per wave (2 x 2 waves)
    constexpr int BR = BN / 32;  // weight rows loaded per thread

    __shared__ __attribute__((aligned(16))) float smem[2][(BM + BN) * LDS_LD];
    __shared__ int4 rowinfo[BM];  // (pixel index or -1, b, y, x) of every tile row

    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so give each XCD a
    // contiguous run of row ranges (neighbouring ranges share activation rows in that XCD's L2).
    const int nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int q = nwg >> 3, rr = nwg & 7, xcd = id & 7;
        id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
    }
    const int n_tile = id % p.n_tiles;
    const int m_block = id / p.n_tiles;
    const int n0 = n_tile * BN;
    const int g_begin = m_block * p.gpb;
    const int g_end = min(p.groups_total, g_begin + p.gpb);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform for the compiler
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int lrow = tid >> 3, lc4 = tid & 7;

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
    // weight rows of this thread: byte offset of (tap 0, row n, channel 0), or OOB past the last column
    unsigned wrow[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + lrow + 32 * i;
        wrow[i] = n < p.n_total ? static_cast<unsigned>(n) * p.ctot * 4u : OOB;
    }
    const unsigned tap_stride_w = static_cast<unsigned>(p.n_total) * p.ctot * 4u;
    const int steps = p.taps * p.kc;

    for (int g0 = g_begin; g0 < g_end; g0 += 4) {
        const int kg = min(4, g_end - g0);  // row groups of this tile (workgroup-uniform)
        const int m0 = g0 * 32;

        __syncthreads();  // the previous tile's epilogue still reads rowinfo
        if (tid < BM) {
            const int m = m0 + tid;
            int4 info = make_int4(-1, 0, 0, 0);
            if (tid < kg * 32 && m < p.m_total) {
                const int per = p.win_h * p.win_w;
                const int b = m / per;
                const int rem = m - b * per;
                const int wy = rem / p.win_w;
                const int y = p.win_y0 + wy, x = p.win_x0 + (rem - wy * p.win_w);
                info = make_int4((b * p.in_h + y) * p.in_w + x, b, y, x);
            }
            rowinfo[tid] = info;
        }
        __syncthreads();

        int4 ri[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ri[i] = rowinfo[lrow + 32 * i];

        // Activation descriptors based just before this tile's first pixel, so that offsets stay
        // small whatever the tensor size (transposed-conv gradient gathers: absolute offsets).
        long long pix0 = 0;
        if (MODE != MODE_CONVT_DGRAD) {
            const int first = __builtin_amdgcn_readfirstlane(rowinfo[0].x);
            pix0 = first - p.in_w - 1;
            if (pix0 < 0) pix0 = 0;
        }
        const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(p.src1 + pix0 * p.ld1, (p.src1_pixels - pix0) * p.ld1 * 4ll);
        const __amdgpu_buffer_rsrc_t rs2 = make_rsrc(p.src2 + pix0 * p.ld2, (p.src2_pixels - pix0) * p.ld2 * 4ll);
        const int pix0i = static_cast<int>(pix0);

        f32x16 acc[2][TN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        unsigned aoff1[4], aoff2[4];  // per-row byte offset of channel 0 for the current tap, or OOB
        auto set_tap = [&](int tap) {
            int dy = 0, dx = 0;
            if (MODE == MODE_CONV3) {
                dy = tap / 3 - 1;
                dx = tap - (tap / 3) * 3 - 1;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bool ok = ri[i].x >= 0;
                int rel;
                if (MODE == MODE_CONVT_DGRAD) {
                    // tap = a*2 + b: the output pixel (2y+a+off_y, 2x+b+off_x) this input pixel fed
                    const int oy = 2 * ri[i].z + (tap >> 1) + p.off_y, ox = 2 * ri[i].w + (tap & 1) + p.off_x;
                    ok = ok && oy >= 0 && oy < p.out_h && ox >= 0 && ox < p.out_w;
                    rel = (ri[i].y * p.out_h + oy) * p.out_w + ox;
                } else {
                    const int yy = ri[i].z + dy, xx = ri[i].w + dx;
                    ok = ok && yy >= 0 && yy < p.in_h && xx >= 0 && xx < p.in_w;
                    rel = ri[i].x - pix0i + dy * p.in_w + dx;
                }
                aoff1[i] = ok ? static_cast<unsigned>(rel) * p.ld1 * 4u : OOB;
                if (DUAL) aoff2[i] = ok ? static_cast<unsigned>(rel) * p.ld2 * 4u : OOB;
            }
        };

        f32x4 areg[4], breg[BR];
        // full = the tile has all 4 row groups (no per-group guards: straight-line code)
        auto load_a = [&](int kc, bool full) {
            const int c = kc * BK + lc4 * 4;
            const unsigned cb = c < p.ctot ? static_cast<unsigned>(c) * 4u : OOB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (full || i < kg) {
                    if (DUAL) {
                        // exactly one of the two loads is in range; the other returns zeros
                        const unsigned o1 = c < p.c1 ? aoff1[i] + cb : OOB;
                        const unsigned o2 = (c >= p.c1 && c < p.ctot) ? aoff2[i] + static_cast<unsigned>(c - p.c1) * 4u : OOB;
                        const f32x4 v1 = buf_load(rs1, o1), v2 = buf_load(rs2, o2);
                        areg[i] = v1 + v2;
                    } else {
                        areg[i] = buf_load(rs1, aoff1[i] + cb);
                    }
                }
            }
        };
        auto load_b = [&](int tap, int kc) {
            const int c = kc * BK + lc4 * 4;
            const unsigned cb = c < p.ctot ? static_cast<unsigned>(c) * 4u : OOB;
            const unsigned tb = static_cast<unsigned>(tap) * tap_stride_w;
#pragma unroll
            for (int i = 0; i < BR; ++i) breg[i] = buf_load(rs_w, wrow[i] + tb + cb);
        };
        auto store_lds = [&](int buf, bool full) {
            float* sA = smem[buf];
            float* sB = smem[buf] + BM * LDS_LD;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (full || i < kg) *reinterpret_cast<f32x4*>(sA + (lrow + 32 * i) * LDS_LD + lc4 * 4) = areg[i];
#pragma unroll
            for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(sB + (lrow + 32 * i) * LDS_LD + lc4 * 4) = breg[i];
        };

        int tap = 0, kc = 0;
        set_tap(0);
        load_a(0, false);
        load_b(0, 0);
        store_lds(0, false);
        __syncthreads();

        // The K loop is instantiated per SLOTS = row groups this wave owns in the tile (groups wm and
        // wm+2; 2 in a full tile), so that each variant keeps its accumulators in fixed registers.
        // Inside a step the staging of the next step is spread between the MFMA chunks (loads after
        // the first two chunks' fragments are requested, LDS stores before the last chunk), so that a
        // wave's own matrix work covers its staging instead of leaving a bubble before it.
        const int my_slots = (kg - wm + 1) >> 1;
        auto k_loop = [&](auto slots_tag, auto full_tag) {
            constexpr int SLOTS = decltype(slots_tag)::value;
            constexpr bool FULL = decltype(full_tag)::value;
            for (int s = 0; s < steps; ++s) {
                const int buf = s & 1;
                const bool more = s + 1 < steps;
                if (more) {
                    if (++kc == p.kc) {
                        kc = 0;
                        ++tap;
                        set_tap(tap);
                    }
                }
                if (SLOTS > 0 && DDPM_ABLATE != 3) {
                    const float* sA = smem[buf] + (wm * 32 + r) * LDS_LD + h * 4;
                    const float* sB = smem[buf] + BM * LDS_LD + (wn * TN * 32 + r) * LDS_LD + h * 4;
                    constexpr int SL = SLOTS > 0 ? SLOTS : 1;
                    f32x4 a[2][SL], b[2][TN];
                    auto fetch = [&](int kk, int which) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) b[which][j] = *reinterpret_cast<const f32x4*>(sB + j * 32 * LDS_LD + kk * 8);
#pragma unroll
                        for (int i = 0; i < SL; ++i) a[which][i] = *reinterpret_cast<const f32x4*>(sA + i * 64 * LDS_LD + kk * 8);
                    };
                    fetch(0, 0);
#pragma unroll
                    for (int kk = 0; kk < BK / 8; ++kk) {
                        const int cur = kk & 1;
                        if (kk + 1 < BK / 8) fetch(kk + 1, cur ^ 1);
                        if (more && DDPM_ABLATE != 1 && DDPM_ABLATE != 2) {
                            if (kk == 0 && DDPM_ABLATE != 5) load_a(kc, FULL);
                            if (kk == 1 && DDPM_ABLATE != 5) load_b(tap, kc);
                            if (kk == BK / 8 - 1 && DDPM_ABLATE != 4) store_lds(buf ^ 1, FULL);
                            if (kk == BK / 8 - 1 && DDPM_ABLATE == 4) {   // keep the loads alive without storing them
#pragma unroll
                                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(areg[i]));
#pragma unroll
                                for (int i = 0; i < BR; ++i) asm volatile("" ::"v"(breg[i]));
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int i = 0; i < SL; ++i)
#pragma unroll
                                for (int j = 0; j < TN; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][e], b[cur][j][e], acc[i][j], 0, 0, 0);
                    }
                } else if (more) {
                    load_a(kc, FULL);
                    load_b(tap, kc);
                    store_lds(buf ^ 1, FULL);
                }
                if (DDPM_ABLATE != 2) __syncthreads();
            }
        };
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        using T2 = std::integral_constant<int, 2>;
        if (kg == 4) {
            if (my_slots == 2) k_loop(T2{}, std::true_type{});     // always, for a full tile
            else k_loop(T1{}, std::true_type{});
        } else if (my_slots == 2) k_loop(T2{}, std::false_type{});
        else if (my_slots == 1) k_loop(T1{}, std::false_type{});
        else k_loop(T0{}, std::false_type{});

        // ---- epilogue ---------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + r;
            if (n >= p.n_total) continue;
            int co = n, qa = 0, qb = 0;
            if (MODE == MODE_CONVT) {
                const int q = n / p.cout;
                co = n - q * p.cout;
                qa = q >> 1;
                qb = q & 1;
            }
            const float sc = p.scale ? p.scale[co] : 1.0f;
            const float sh = p.shift ? p.shift[co] : 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i >= my_slots) continue;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (i * 2 + wm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const int4 info = rowinfo[row];
                    if (info.x < 0) continue;
                    float v = acc[i][j][e] * sc + sh;
                    if (MODE == MODE_CONVT) {
                        const int oy = 2 * info.z + qa + p.off_y, ox = 2 * info.w + qb + p.off_x;
                        if (oy < 0 || oy >= p.out_h || ox < 0 || ox >= p.out_w) continue;
                        const int64_t opix = (static_cast<int64_t>(info.y) * p.out_h + oy) * p.out_w + ox;
                        p.out[opix * p.ldo + co] = v;
                    } else {
                        const int64_t pix = info.x;
                        if (p.residual) v += p.residual[pix * p.ldr + n];
                        v = apply_act(v, p.act);
                        if (p.temb) v += p.temb[static_cast<int64_t>(info.y) * p.cout + n];
                        p.out[pix * p.ldo + n] = v;
                    }
                }
            }
        }
    }
}


// ---- split-bf16 variant ---------------------------------------------------------------------
// Every fp32 operand is written as the exact sum of three bf16 numbers (hi + mid + lo: truncation
// splits, 8 significant bits each, 24 in total), and a fp32 product a*b is evaluated as the six
// cross products of weight >= 2^-16 (hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi) on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the three dropped products are below 2^-23 of
// a*b, the size of one fp32 rounding. Six bf16 MFMAs of 32 cycles replace eight fp32 MFMAs of 64
// cycles per 16-deep slice: 2.67x the matrix throughput at fp32 accuracy.
//
// The bf16 MFMA truncates (rounds toward zero) when it accumulates, which biases long sums of
// like-signed terms; the K loop therefore accumulates at most 4 steps (64 input channels of one tap)
// in the MFMA accumulator and folds that partial sum into a master accumulator with a
// round-to-nearest fp32 add.
//
// Activations are split in registers on their way from global memory to LDS; weights arrive already
// split from the packing kernels: [tap][row][chunk of 16 channels][plane hi|mid|lo][16] bf16.
// LDS tile rows hold the three planes of one 16-deep slice (3 x 32 B) padded to 112 B, which makes the
// 16-byte fragment reads (lane = row) conflict-free. Loads run two steps ahead of the MFMAs: the data
// of step s+2 is requested during step s, split and written to LDS during step s+1.
//
// With F16 the operands are fp16 pairs (conv_common.h): planes hi | lo in the same 112-byte rows, three MFMAs per
// fragment pair, the hi*hi products in `acc` and the cross products in `master`, which is scaled by 2^-11 and added
// after the last step; no fold.
template <int BN, int MODE, bool DUAL, bool F16>
__global__ __launch_bounds__(256, 2) void conv_gemm_x3_kernel(const GemmParams p) {
    if constexpr (F16) fp16_saturate();
    RangeWatch amax;                         // F16: largest magnitude this thread has turned into an fp16 pair
    constexpr int TN = BN / 64;               // 32-column MFMA tiles per wave (2 x 2 waves)
    constexpr int NB = (BN * 6 + 255) / 256;  // 16-byte weight pieces per thread and step

    __shared__ __attribute__((aligned(16))) unsigned char smem[2][(BM + BN) * XROW];
    __shared__ int4 rowinfo[BM];
    // transposed conv: byte offset of every row's output pixel (2y + off_y, 2x + off_x) from the tile's first one, with
    // the four taps' validity in its low four bits (a pixel's stride is a multiple of 16 bytes)
    __shared__ __attribute__((aligned(16))) unsigned rowout[MODE == MODE_CONVT ? BM : 4];

    const int nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int q = nwg >> 3, rr = nwg & 7, xcd = id & 7;
        id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
    }
    const int n_tile = id % p.n_tiles;
    const int m_block = id / p.n_tiles;
    const int n0 = n_tile * BN;
    const int g_begin = m_block * p.gpb;
    const int g_end = min(p.groups_total, g_begin + p.gpb);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int lrow = tid >> 2, lq = tid & 3;   // activation staging: row (and row + 64), 4-channel quad

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
    // weight pieces of this thread: the tile's BN/32 row blocks are contiguous in the packed buffer,
    // 192 pieces of 16 bytes each (plane, fragment lane); piece (plane, lane) of a block lands in LDS
    // row lane & 31, plane, half lane >> 5
    unsigned wrow[NB];
    int wlds[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + 256 * i;
        const int blk = idx / 192, rem = idx - blk * 192;
        const int plane = rem >> 6, fl = rem & 63;
        const bool ok = blk < BN / 32 && (n0 >> 5) + blk < p.nbp;
        wrow[i] = ok ? (static_cast<unsigned>((n0 >> 5) + blk) * 3072u + rem * 16u) : OOB;
        wlds[i] = blk < BN / 32 ? (blk * 32 + (fl & 31)) * XROW + plane * 32 + (fl >> 5) * 16 : -1;
    }
    const unsigned chunk_stride_w = static_cast<unsigned>(p.nbp) * 3072u;            // one (tap, chunk)
    const unsigned tap_stride_w = chunk_stride_w * static_cast<unsigned>(p.kc16);
    const int steps = p.taps * p.kc16;

    for (int g0 = g_begin; g0 < g_end; g0 += 4) {
        const int kg = min(4, g_end - g0);
        const int m0 = g0 * 32;

        __syncthreads();
        if (tid < BM) {
            const int m = m0 + tid;
            int4 info = make_int4(-1, 0, 0, 0);
            if (tid < kg * 32 && m < p.m_total) {
                const int per = p.win_h * p.win_w;
                const int b = m / per;
                const int rem = m - b * per;
                const int wy = rem / p.win_w;
                const int y = p.win_y0 + wy, x = p.win_x0 + (rem - wy * p.win_w);
                info = make_int4((b * p.in_h + y) * p.in_w + x, b, y, x);
            }
            rowinfo[tid] = info;
        }
        __syncthreads();
        long long convt_ref = 0;   // MODE_CONVT: output pixel index behind offset 0 of rowout (the tile's first row, tap 0)
        if (MODE == MODE_CONVT) {
            const int4 first = rowinfo[0];
            convt_ref = (static_cast<long long>(__builtin_amdgcn_readfirstlane(first.y)) * p.out_h + 2 * __builtin_amdgcn_readfirstlane(first.z) + p.off_y) * p.out_w +
                        2 * __builtin_amdgcn_readfirstlane(first.w) + p.off_x;
            if (tid < BM) {
                const int4 info = rowinfo[tid];
                unsigned packed = 0u;
                if (info.x >= 0) {
                    const int oy0 = 2 * info.z + p.off_y, ox0 = 2 * info.w + p.off_x;
                    const long long pix = (static_cast<long long>(info.y) * p.out_h + oy0) * p.out_w + ox0;
                    unsigned mask = 0u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int oy = oy0 + (q >> 1), ox = ox0 + (q & 1);
                        if (oy >= 0 && oy < p.out_h && ox >= 0 && ox < p.out_w) mask |= 1u << q;
                    }
                    packed = static_cast<unsigned>((pix - convt_ref) * p.ldo * 4ll) | mask;
                }
                rowout[tid] = packed;
            }
            __syncthreads();
        }

        int4 ri[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) ri[i] = rowinfo[lrow + 64 * i];

        long long pix0 = 0;
        if (MODE != MODE_CONVT_DGRAD) {
            const int first = __builtin_amdgcn_readfirstlane(rowinfo[0].x);
            pix0 = first - p.in_w - 1;
            if (pix0 < 0) pix0 = 0;
        }
        const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(p.src1 + pix0 * p.ld1, (p.src1_pixels - pix0) * p.ld1 * 4ll);
        const __amdgpu_buffer_rsrc_t rs2 = make_rsrc(p.src2 + pix0 * p.ld2, (p.src2_pixels - pix0) * p.ld2 * 4ll);
        const int pix0i = static_cast<int>(pix0);

        unsigned aoff1[2], aoff2[2];
        auto set_tap = [&](int tap) {
            int dy = 0, dx = 0;
            if (MODE == MODE_CONV3) {
                dy = tap / 3 - 1;
                dx = tap - (tap / 3) * 3 - 1;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bool ok = ri[i].x >= 0;
                int rel;
                if (MODE == MODE_CONVT_DGRAD) {
                    const int oy = 2 * ri[i].z + (tap >> 1) + p.off_y, ox = 2 * ri[i].w + (tap & 1) + p.off_x;
                    ok = ok && oy >= 0 && oy < p.out_h && ox >= 0 && ox < p.out_w;
                    rel = (ri[i].y * p.out_h + oy) * p.out_w + ox;
                } else {
                    const int yy = ri[i].z + dy, xx = ri[i].w + dx;
                    ok = ok && yy >= 0 && yy < p.in_h && xx >= 0 && xx < p.in_w;
                    rel = ri[i].x - pix0i + dy * p.in_w + dx;
                }
                aoff1[i] = ok ? static_cast<unsigned>(rel) * p.ld1 * 4u : OOB;
                if (DUAL) aoff2[i] = ok ? static_cast<unsigned>(rel) * p.ld2 * 4u : OOB;
            }
        };

        f32x4 areg[2];
        u32x4 breg[NB];
        int ltap = 0, lkc = 0;   // (tap, chunk) of the next stage to request
        auto load_stage = [&]() {
            const int c = lkc * XK + lq * 4;
            const unsigned cb = c < p.ctot ? static_cast<unsigned>(c) * 4u : OOB;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (DUAL) {
                    const unsigned o1 = c < p.c1 ? aoff1[i] + cb : OOB;
                    const unsigned o2 = (c >= p.c1 && c < p.ctot) ? aoff2[i] + static_cast<unsigned>(c - p.c1) * 4u : OOB;
                    areg[i] = buf_load(rs1, o1) + buf_load(rs2, o2);
                } else {
                    areg[i] = buf_load(rs1, aoff1[i] + cb);
                }
            }
            const unsigned tb = static_cast<unsigned>(ltap) * tap_stride_w + static_cast<unsigned>(lkc) * chunk_stride_w;
#pragma unroll
            for (int i = 0; i < NB; ++i) breg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wrow[i] + tb, 0, 0);
            if (++lkc == p.kc16) {
                lkc = 0;
                ++ltap;
                if (ltap < p.taps) set_tap(ltap);
            }
        };
        auto store_stage = [&](int buf) {
            unsigned char* sA = smem[buf];
            unsigned char* sB = smem[buf] + BM * XROW;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (F16) split_store_f16(areg[i], sA + (lrow + 64 * i) * XROW + lq * 8, amax);
                else split_store(areg[i], sA + (lrow + 64 * i) * XROW + lq * 8);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (NB * 256 == BN * 6 || wlds[i] >= 0) *reinterpret_cast<u32x4*>(sB + wlds[i]) = breg[i];
        };

        f32x16 master[2][TN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) master[i][j][e] = 0.f;

        set_tap(0);
        load_stage();
        store_stage(0);
        if (steps > 1) load_stage();
        __syncthreads();

        const int my_slots = (kg - wm + 1) >> 1;
        auto k_loop = [&](auto slots_tag) {
            constexpr int SLOTS = decltype(slots_tag)::value;
            constexpr int SL = SLOTS > 0 ? SLOTS : 1;
            f32x16 acc[SL][TN];
#pragma unroll
            for (int i = 0; i < SL; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            for (int s = 0; s < steps; ++s) {
                const int buf = s & 1;
                if (SLOTS > 0) {
                    const unsigned char* sA = smem[buf] + (wm * 32 + r) * XROW + h * 16;
                    const unsigned char* sB = smem[buf] + BM * XROW + (wn * TN * 32 + r) * XROW + h * 16;
                    constexpr int NPL = F16 ? 2 : 3;
                    u32x4 a[SL][NPL], b[TN][NPL];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) b[j][pl] = *reinterpret_cast<const u32x4*>(sB + j * 32 * XROW + pl * 32);
#pragma unroll
                        for (int i = 0; i < SL; ++i) a[i][pl] = *reinterpret_cast<const u32x4*>(sA + i * 64 * XROW + pl * 32);
                    }
                    if constexpr (F16) {
                        auto term = [&](auto& c, int pa, int pb) {
#pragma unroll
                            for (int i = 0; i < SL; ++i)
#pragma unroll
                                for (int j = 0; j < TN; ++j) c[i][j] = mfma16<true>(a[i][pa], b[j][pb], c[i][j]);
                        };
                        term(master, 1, 0);
                        if (s + 1 < steps) store_stage(buf ^ 1);
                        if (s + 2 < steps) load_stage();
                        term(master, 0, 1);
                        term(acc, 0, 0);
                        if (s + 1 == steps) {
#pragma unroll
                            for (int i = 0; i < SL; ++i)
#pragma unroll
                                for (int j = 0; j < TN; ++j) master[i][j] = acc[i][j] + master[i][j] * kLoInv;
                        }
                    } else {
                        auto term = [&](int pa, int pb) {
#pragma unroll
                            for (int i = 0; i < SL; ++i)
#pragma unroll
                                for (int j = 0; j < TN; ++j) acc[i][j] = mfma16<false>(a[i][pa], b[j][pb], acc[i][j]);
                        };
                        // smallest products first
                        term(2, 0);
                        term(0, 2);
                        term(1, 1);
                        if (s + 1 < steps) store_stage(buf ^ 1);
                        if (s + 2 < steps) load_stage();
                        term(1, 0);
                        term(0, 1);
                        term(0, 0);
                        if ((s & (XFLUSH - 1)) == XFLUSH - 1 || s + 1 == steps) {
#pragma unroll
                            for (int i = 0; i < SL; ++i)
#pragma unroll
                                for (int j = 0; j < TN; ++j) {
                                    master[i][j] += acc[i][j];
#pragma unroll
                                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                                }
                        }
                    }
                } else {
                    if (s + 1 < steps) store_stage(buf ^ 1);
                    if (s + 2 < steps) load_stage();
                }
                __syncthreads();
            }
        };
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        using T2 = std::integral_constant<int, 2>;
        if (my_slots == 2) k_loop(T2{});
        else if (my_slots == 1) k_loop(T1{});
        else k_loop(T0{});

        // ---- epilogue ---------------------------------------------------------------------
        // Rows that are consecutive pixels of the tensor (every 1x1 conv, a 3x3 conv over the whole
        // image): stores and residual loads go through buffer descriptors that begin at the tile's first
        // pixel and end with the tensor, so rows past the last pixel are dropped (or read as zero) by the
        // range check and no element carries a row lookup, a branch or 64-bit address arithmetic.
        const bool linear_rows = MODE != MODE_CONVT && p.win_y0 == 0 && p.win_x0 == 0 && p.win_h == p.in_h && p.win_w == p.in_w;
        if (linear_rows) {
            const long long rows_left = p.m_total - m0;
            const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out + static_cast<int64_t>(m0) * p.ldo, rows_left * p.ldo * 4ll);
            const __amdgpu_buffer_rsrc_t rs_r =
                make_rsrc(p.residual ? p.residual + static_cast<int64_t>(m0) * p.ldr : p.out, p.residual ? rows_left * p.ldr * 4ll : 0ll);
            const unsigned ldo4 = static_cast<unsigned>(p.ldo) * 4u, ldr4 = static_cast<unsigned>(p.ldr) * 4u;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int n = n0 + (wn * TN + j) * 32 + r;
                asm volatile("" : "+v"(n));   // keeps the per-element offsets from being hoisted out of the tile loop
                if (n >= p.n_total) continue;
                const float sc = p.scale ? p.scale[n] : 1.0f;
                const float sh = p.shift ? p.shift[n] : 0.0f;
                const unsigned vo = static_cast<unsigned>((wm * 32 + 4 * h) * p.ldo + n) * 4u;
                const unsigned vr = static_cast<unsigned>((wm * 32 + 4 * h) * p.ldr + n) * 4u;
                // mode decisions once per 16 elements, each phase over all 16 (a branch or a waited-for load per
                // element costs more than the arithmetic)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (i >= my_slots) continue;
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = master[i][j][e] * sc + sh;
                    if (p.residual) {
                        float rv[16];
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const unsigned rel = static_cast<unsigned>(i * 64 + (e & 3) + 8 * (e >> 2));   // + wm * 32 + 4 * h
                            rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, vr + rel * ldr4, 0, 0));
                        }
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] += rv[e];
                    }
                    if (p.act == DDPM_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    } else if (p.act == DDPM_ACT_GELU) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = apply_act_fast(v[e], DDPM_ACT_GELU);
                    }
                    if (p.temb) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int4 info = rowinfo[i * 64 + (e & 3) + 8 * (e >> 2) + wm * 32 + 4 * h];
                            v[e] += info.x >= 0 ? p.temb[static_cast<int64_t>(info.y) * p.cout + n] : 0.f;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const unsigned rel = static_cast<unsigned>(i * 64 + (e & 3) + 8 * (e >> 2));
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), rs_o, vo + rel * ldo4, 0, DDPM_STORE_AUX);
                    }
                }
            }
            continue;
        }
        if (MODE == MODE_CONVT && (p.cout & 31) == 0) {
            // every column block lies inside one tap (cout is a multiple of 32): a row's output offset comes from the
            // tile's table, the tap and the channel add a per-block constant, and a tap that falls outside the cropped
            // image turns the offset into one the descriptor's range check drops. Same values as the general path below.
            const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out + convt_ref * p.ldo, 0x7fffffffll);
            unsigned tapoff[TN];
            float scj[TN], shj[TN];
            int qj[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int n = n0 + (wn * TN + j) * 32 + r;
                asm volatile("" : "+v"(n));
                const int q = n / p.cout, co = n - q * p.cout;
                qj[j] = n < p.n_total ? q : -1;      // -1: no such column, every store of the block is dropped
                tapoff[j] = static_cast<unsigned>(((q >> 1) * p.out_w + (q & 1)) * p.ldo + co) * 4u;
                scj[j] = (p.scale && n < p.n_total) ? p.scale[co] : 1.0f;
                shj[j] = (p.shift && n < p.n_total) ? p.shift[co] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i >= my_slots) continue;
                unsigned ro[16];
#pragma unroll
                for (int eq = 0; eq < 4; ++eq) {
                    const uint4 t4 = *reinterpret_cast<const uint4*>(&rowout[(i * 2 + wm) * 32 + 8 * eq + 4 * h]);
                    ro[eq * 4 + 0] = t4.x;
                    ro[eq * 4 + 1] = t4.y;
                    ro[eq * 4 + 2] = t4.z;
                    ro[eq * 4 + 3] = t4.w;
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float v = master[i][j][e] * scj[j] + shj[j];
                        const bool ok = qj[j] >= 0 && ((ro[e] >> (qj[j] & 3)) & 1u);
                        const unsigned off = ok ? (ro[e] & ~15u) + tapoff[j] : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o, off, 0, DDPM_STORE_AUX);
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + r;
            if (n >= p.n_total) continue;
            int co = n, qa = 0, qb = 0;
            if (MODE == MODE_CONVT) {
                const int q = n / p.cout;
                co = n - q * p.cout;
                qa = q >> 1;
                qb = q & 1;
            }
            const float sc = p.scale ? p.scale[co] : 1.0f;
            const float sh = p.shift ? p.shift[co] : 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i >= my_slots) continue;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (i * 2 + wm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const int4 info = rowinfo[row];
                    if (info.x < 0) continue;
                    float v = master[i][j][e] * sc + sh;
                    if (MODE == MODE_CONVT) {
                        const int oy = 2 * info.z + qa + p.off_y, ox = 2 * info.w + qb + p.off_x;
                        if (oy < 0 || oy >= p.out_h || ox < 0 || ox >= p.out_w) continue;
                        const int64_t opix = (static_cast<int64_t>(info.y) * p.out_h + oy) * p.out_w + ox;
                        p.out[opix * p.ldo + co] = v;
                    } else {
                        const int64_t pix = info.x;
                        if (p.residual) v += p.residual[pix * p.ldr + n];
                        v = apply_act_fast(v, p.act);
                        if (p.temb) v += p.temb[static_cast<int64_t>(info.y) * p.cout + n];
                        p.out[pix * p.ldo + n] = v;
                    }
                }
            }
        }
    }
    if constexpr (F16) raise_range_flag(p.range_flag, amax);
}

using ddpm::split_math;
using ddpm::f16_math;

template <int MODE>
int launch_gemm(GemmParams& p, hipStream_t s) {
    // Column tile and run length: minimise (row groups per workgroup) x (columns per tile), the time
    // of the busiest workgroup when the whole grid is one resident round. The 64-column tile stages
    // as many activation bytes for half the MFMA work, and measures ~35 % slower per FLOP, hence the handicap.
    const bool x3 = split_math();
    p.groups_total = (p.m_total + 31) / 32;
    p.kc16 = (p.ctot + XK - 1) / XK;
    p.nbp = (p.n_total + 31) / 32;
    const long long wb = x3 ? 3072ll * p.taps * p.nbp * p.kc16 : 4ll * p.taps * p.n_total * p.ctot;
    if (wb >= 0x7fffffffll) return ddpm::fail_invalid("weight tensor of 2 GiB or more");
    p.w_bytes = static_cast<unsigned>(wb);
    int best_bn = 0, best_gpb = 0;
    double best_cost = 0.0;
    for (int bn : {128, 64}) {
        const int nt = (p.n_total + bn - 1) / bn;
        int per_tile = kSlots / nt;
        if (per_tile < 1) per_tile = 1;
        const int gpb = (p.groups_total + per_tile - 1) / per_tile;
        const double cost = static_cast<double>(gpb) * bn * (bn == 64 ? 1.35 : 1.0);
        if (best_bn == 0 || cost < best_cost) {
            best_bn = bn;
            best_gpb = gpb;
            best_cost = cost;
        }
    }
    p.gpb = best_gpb;
    p.n_tiles = (p.n_total + best_bn - 1) / best_bn;
    const int64_t grid = static_cast<int64_t>((p.groups_total + p.gpb - 1) / p.gpb) * p.n_tiles;
    if (grid <= 0 || grid > 0x7fffffffll) return ddpm::fail_invalid("conv grid size");
    const dim3 g(static_cast<unsigned>(grid)), b(256);
    const bool dual = p.c2 > 0;
    if (x3) {
        const bool f16 = ddpm::f16_math();
        if (f16 && !p.range_flag) return DDPM_E_LAUNCH;
        if (best_bn == 128) {
            if (dual) f16 ? conv_gemm_x3_kernel<128, MODE, true, true><<<g, b, 0, s>>>(p) : conv_gemm_x3_kernel<128, MODE, true, false><<<g, b, 0, s>>>(p);
            else f16 ? conv_gemm_x3_kernel<128, MODE, false, true><<<g, b, 0, s>>>(p) : conv_gemm_x3_kernel<128, MODE, false, false><<<g, b, 0, s>>>(p);
        } else {
            if (dual) f16 ? conv_gemm_x3_kernel<64, MODE, true, true><<<g, b, 0, s>>>(p) : conv_gemm_x3_kernel<64, MODE, true, false><<<g, b, 0, s>>>(p);
            else f16 ? conv_gemm_x3_kernel<64, MODE, false, true><<<g, b, 0, s>>>(p) : conv_gemm_x3_kernel<64, MODE, false, false><<<g, b, 0, s>>>(p);
        }
        return ddpm::check_launch("conv_gemm_x3_kernel");
    }
    if (best_bn == 128) {
        if (dual) conv_gemm_kernel<128, MODE, true><<<g, b, 0, s>>>(p);
        else conv_gemm_kernel<128, MODE, false><<<g, b, 0, s>>>(p);
    } else {
        if (dual) conv_gemm_kernel<64, MODE, true><<<g, b, 0, s>>>(p);
        else conv_gemm_kernel<64, MODE, false><<<g, b, 0, s>>>(p);
    }
    return ddpm::check_launch("conv_gemm_kernel");
}

int validate_common(const ddpm_conv_desc* d, const float* src1, const float* src2, const float* w,
                    const ddpm_epilogue* ep, float* out) {
    DDPM_REQUIRE(d && src1 && w && ep && out);
    DDPM_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cout > 0);
    DDPM_REQUIRE(d->c1 > 0 && (d->c1 & 3) == 0 && (d->ld1 & 3) == 0 && d->ld1 >= d->c1);
    DDPM_REQUIRE(d->c2 >= 0 && (d->c2 & 3) == 0);
    DDPM_REQUIRE(d->c2 == 0 || (src2 && (d->ld2 & 3) == 0 && d->ld2 >= d->c2 && ddpm::aligned16(src2)));
    DDPM_REQUIRE(ddpm::aligned16(src1) && ddpm::aligned16(w));
    DDPM_REQUIRE(d->ldo >= d->cout);
    DDPM_REQUIRE(d->act >= DDPM_ACT_NONE && d->act <= DDPM_ACT_GELU);
    DDPM_REQUIRE(!ep->scale || ep->shift);  // a scale always comes with its shift
    DDPM_REQUIRE(!ep->residual || d->ldr >= d->cout);
    DDPM_REQUIRE(static_cast<int64_t>(d->batch) * d->in_h * d->in_w < (1ll << 31));
    return DDPM_OK;
}

void fill_common(GemmParams& p, const ddpm_conv_desc* d, const float* src1, const float* src2, const float* w,
                 const ddpm_epilogue* ep, float* out) {
    p.src1 = src1;
    p.src2 = d->c2 ? src2 : src1;
    p.w = w;
    p.out = out;
    p.scale = ep->scale;
    p.shift = ep->shift;
    p.residual = ep->residual;
    p.temb = ep->temb;
    p.batch = d->batch;
    p.in_h = d->in_h;
    p.in_w = d->in_w;
    p.win_y0 = 0;
    p.win_x0 = 0;
    p.win_h = d->in_h;
    p.win_w = d->in_w;
    p.c1 = d->c1;
    p.ld1 = d->ld1;
    p.c2 = d->c2;
    p.ld2 = d->c2 ? d->ld2 : d->ld1;
    p.ctot = d->c1 + d->c2;
    p.cout = d->cout;
    p.n_total = d->cout;
    p.ldo = d->ldo;
    p.ldr = d->ldr;
    p.act = d->act;
    p.out_h = d->in_h;
    p.out_w = d->in_w;
    p.off_y = 0;
    p.off_x = 0;
    p.stat_partial = nullptr;
    p.stat_blocks = 0;
    p.range_flag = f16_math() ? ddpm::range_flag() : nullptr;   // (null on failure: the launchers of the fp16-pair kernels refuse)
    p.kc = (p.ctot + BK - 1) / BK;
    p.m_total = d->batch * d->in_h * d->in_w;
    p.src1_pixels = static_cast<long long>(d->batch) * d->in_h * d->in_w;
    p.src2_pixels = p.src1_pixels;
}

// ---- weight repacking ---------------------------------------------------------------------
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, int cout, int cin, int flip, float* __restrict__ out) {
    const int64_t total = static_cast<int64_t>(cout) * cin * 9;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        if (!flip) {  // out[tap][co][ci] = w[co][ci][tap]
            const int ci = static_cast<int>(i % cin);
            const int co = static_cast<int>((i / cin) % cout);
            const int tap = static_cast<int>(i / (static_cast<int64_t>(cin) * cout));
            out[i] = w[(static_cast<int64_t>(co) * cin + ci) * 9 + tap];
        } else {  // out[tap][ci][co] = w[co][ci][8 - tap]
            const int co = static_cast<int>(i % cout);
            const int ci = static_cast<int>((i / cout) % cin);
            const int tap = static_cast<int>(i / (static_cast<int64_t>(cin) * cout));
            out[i] = w[(static_cast<int64_t>(co) * cin + ci) * 9 + (8 - tap)];
        }
    }
}

__global__ void pack_convT_kernel(const float* __restrict__ w, int cin, int cout, float* __restrict__ out) {
    const int64_t total = static_cast<int64_t>(cin) * cout * 4;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        // out[(q*cout + co)][ci] = w[ci][co][q],  q = a*2 + b
        const int ci = static_cast<int>(i % cin);
        const int co = static_cast<int>((i / cin) % cout);
        const int q = static_cast<int>(i / (static_cast<int64_t>(cin) * cout));
        out[i] = w[(static_cast<int64_t>(ci) * cout + co) * 4 + q];
    }
}

__global__ void pack_convT_dgrad_kernel(const float* __restrict__ w, int cin, int cout, float* __restrict__ out) {
    const int64_t total = static_cast<int64_t>(cin) * cout * 4;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        // out[q][ci][co] = w[ci][co][q]
        const int co = static_cast<int>(i % cout);
        const int ci = static_cast<int>((i / cout) % cin);
        const int q = static_cast<int>(i / (static_cast<int64_t>(cin) * cout));
        out[i] = w[(static_cast<int64_t>(ci) * cout + co) * 4 + q];
    }
}


// Element i of the fragment-major packed image (i >> 9 = the 3 KiB block, i & 511 = position in a plane): three
// bf16 planes, or -- fp16 pair scheme -- hi and scaled remainder in planes 0 and 1, plane 2 unused.
__device__ __forceinline__ void store_split_element(unsigned short* out, int64_t i, float v, int f16, unsigned* range_flag) {
    unsigned short* o = out + (i >> 9) * 1536 + (i & 511);
    if (f16) {
        RangeWatch w;                                       // a weight beyond fp16's range (or not a number) is reported like any operand
        watch2(w, f32x2{v, v});
        raise_range_flag(range_flag, w);
        const _Float16 h = static_cast<_Float16>(v);
        const _Float16 l = static_cast<_Float16>((v - static_cast<float>(h)) * kLoScale);
        o[0] = __builtin_bit_cast(unsigned short, h);
        o[512] = __builtin_bit_cast(unsigned short, l);
        o[1024] = 0;
        return;
    }
    const unsigned u = __float_as_uint(v);
    const float r1 = v - __uint_as_float(u & 0xffff0000u);
    const unsigned m = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(m & 0xffff0000u);
    o[0] = static_cast<unsigned short>(u >> 16);
    o[512] = static_cast<unsigned short>(m >> 16);
    o[1024] = static_cast<unsigned short>(__float_as_uint(r2) >> 16);
}

// Split-bf16 packing: logical weight W(tap, n, c) = w[base + tap*st + (n / nq)*sq + (n % nq)*sn + c*sc]
// -> out[tap][c / 16][n / 32][plane][lane][8] (bf16; planes hi | mid | lo), lane = 32*h + n % 32 holding
// channels 16*(c/16) + 8*h + 0..7: one (tap, chunk, row block, plane) is the 1 KiB MFMA operand
// fragment of a wave. Rows are padded with zeros to a multiple of 32, channels to a multiple of 16.
struct SplitPack {
    const float* w;
    unsigned short* out;
    int taps, rows, cin, kc16, nbp, nq, f16;
    long long base, st, sq, sn, sc;
    unsigned* range_flag;
};

__global__ void pack_split_kernel(const SplitPack a) {
    if (a.f16) fp16_saturate();
    const int64_t total = static_cast<int64_t>(a.taps) * a.kc16 * a.nbp * 512;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int j = static_cast<int>(i & 7);
        const int fl = static_cast<int>((i >> 3) & 63);
        int64_t rest = i >> 9;                                 // (tap*kc16 + kc)*nbp + nb
        const int nb = static_cast<int>(rest % a.nbp);
        rest /= a.nbp;
        const int kc = static_cast<int>(rest % a.kc16);
        const int tap = static_cast<int>(rest / a.kc16);
        const int n = nb * 32 + (fl & 31);
        const int c = kc * 16 + (fl >> 5) * 8 + j;
        float v = 0.f;
        if (c < a.cin && n < a.rows) v = a.w[a.base + tap * a.st + (n / a.nq) * a.sq + (n % a.nq) * a.sn + c * a.sc];
        store_split_element(a.out, i, v, a.f16, a.range_flag);
    }
}

// Fused q|k|v projection weights of the attention block in one launch: the [2*dqp + C][C] weight
// (q and k rows zero-padded from dq to dqp) packed for the forward 1x1 conv, its transpose packed for
// the data-gradient GEMM, and the fused bias.
struct QkvPack {
    const float *qw, *kw, *vw, *qb, *kb, *vb;
    float *out_f, *out_t, *bias;
    int c, dq, dqp, split, f16;
    unsigned* range_flag;
};

__device__ __forceinline__ float qkv_weight(const QkvPack& a, int n, int ch) {
    if (n < a.dqp) return n < a.dq ? a.qw[n * a.c + ch] : 0.f;
    if (n < 2 * a.dqp) return n - a.dqp < a.dq ? a.kw[(n - a.dqp) * a.c + ch] : 0.f;
    return a.vw[(n - 2 * a.dqp) * a.c + ch];
}

__global__ void pack_qkv_kernel(const QkvPack a) {
    if (a.f16) fp16_saturate();
    const int lq = 2 * a.dqp + a.c;
    const int64_t t0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x, stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    if (a.split) {
        // forward: rows = lq, depth = c
        const int nbp = (lq + 31) / 32, kc16 = (a.c + 15) / 16;
        for (int64_t i = t0; i < static_cast<int64_t>(kc16) * nbp * 512; i += stride) {
            const int j = static_cast<int>(i & 7), fl = static_cast<int>((i >> 3) & 63);
            const int nb = static_cast<int>((i >> 9) % nbp), kc = static_cast<int>((i >> 9) / nbp);
            const int n = nb * 32 + (fl & 31), ch = kc * 16 + (fl >> 5) * 8 + j;
            store_split_element(reinterpret_cast<unsigned short*>(a.out_f), i, (n < lq && ch < a.c) ? qkv_weight(a, n, ch) : 0.f, a.f16, a.range_flag);
        }
        // transpose: rows = c, depth = lq
        const int nbt = (a.c + 31) / 32, kct = (lq + 15) / 16;
        for (int64_t i = t0; i < static_cast<int64_t>(kct) * nbt * 512; i += stride) {
            const int j = static_cast<int>(i & 7), fl = static_cast<int>((i >> 3) & 63);
            const int nb = static_cast<int>((i >> 9) % nbt), kc = static_cast<int>((i >> 9) / nbt);
            const int ch = nb * 32 + (fl & 31), n = kc * 16 + (fl >> 5) * 8 + j;
            store_split_element(reinterpret_cast<unsigned short*>(a.out_t), i, (n < lq && ch < a.c) ? qkv_weight(a, n, ch) : 0.f, a.f16, a.range_flag);
        }
    } else {
        for (int64_t i = t0; i < static_cast<int64_t>(lq) * a.c; i += stride) {
            const int n = static_cast<int>(i / a.c), ch = static_cast<int>(i % a.c);
            const float v = qkv_weight(a, n, ch);
            a.out_f[i] = v;
            a.out_t[static_cast<int64_t>(ch) * lq + n] = v;
        }
    }
    for (int64_t n = t0; n < lq; n += stride) {
        float v;
        if (n < a.dqp) v = n < a.dq ? a.qb[n] : 0.f;
        else if (n < 2 * a.dqp) v = n - a.dqp < a.dq ? a.kb[n - a.dqp] : 0.f;
        else v = a.vb[n - 2 * a.dqp];
        a.bias[n] = v;
    }
}

SplitPack make_split_pack(const float* w, float* out, int taps, int rows, int cin, int nq, long long base, long long st,
                          long long sq, long long sn, long long sc) {
    SplitPack a;
    a.w = w;
    a.out = reinterpret_cast<unsigned short*>(out);
    a.taps = taps;
    a.rows = rows;
    a.cin = cin;
    a.kc16 = (cin + 15) / 16;
    a.nbp = (rows + 31) / 32;
    a.nq = nq;
    a.f16 = f16_math() ? 1 : 0;
    a.range_flag = a.f16 ? ddpm::range_flag() : nullptr;
    a.base = base;
    a.st = st;
    a.sq = sq;
    a.sn = sn;
    a.sc = sc;
    return a;
}

// Several weight tensors in one launch (blockIdx.y picks the tensor): a training step repacks every convolution weight
// after the optimizer has changed it, and forty launches of a few microseconds each cost more than the packing.
constexpr int kPackBatch = 16;
struct SplitPackBatch {
    SplitPack item[kPackBatch];
};
__global__ void pack_split_batch_kernel(const SplitPackBatch b) {
    const SplitPack& a = b.item[blockIdx.y];
    if (a.f16) fp16_saturate();
    const int64_t total = static_cast<int64_t>(a.taps) * a.kc16 * a.nbp * 512;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int j = static_cast<int>(i & 7);
        const int fl = static_cast<int>((i >> 3) & 63);
        int64_t rest = i >> 9;
        const int nb = static_cast<int>(rest % a.nbp);
        rest /= a.nbp;
        const int kc = static_cast<int>(rest % a.kc16);
        const int tap = static_cast<int>(rest / a.kc16);
        const int n = nb * 32 + (fl & 31);
        const int c = kc * 16 + (fl >> 5) * 8 + j;
        float v = 0.f;
        if (c < a.cin && n < a.rows) v = a.w[a.base + tap * a.st + (n / a.nq) * a.sq + (n % a.nq) * a.sn + c * a.sc];
        store_split_element(a.out, i, v, a.f16, a.range_flag);
    }
}

int launch_split_pack(const float* w, float* out, int taps, int rows, int cin, int nq, long long base, long long st,
                      long long sq, long long sn, long long sc, hipStream_t s) {
    SplitPack a;
    a.w = w;
    a.out = reinterpret_cast<unsigned short*>(out);
    a.taps = taps;
    a.rows = rows;
    a.cin = cin;
    a.kc16 = (cin + 15) / 16;
    a.nbp = (rows + 31) / 32;
    a.nq = nq;
    a.f16 = f16_math() ? 1 : 0;
    a.range_flag = a.f16 ? ddpm::range_flag() : nullptr;
    a.base = base;
    a.st = st;
    a.sq = sq;
    a.sn = sn;
    a.sc = sc;
    if (a.f16 && !a.range_flag) return DDPM_E_LAUNCH;
    const int64_t total = static_cast<int64_t>(taps) * a.kc16 * a.nbp * 512;
    const int grid = static_cast<int>((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    pack_split_kernel<<<grid, 256, 0, s>>>(a);
    return ddpm::check_launch("pack_split_kernel");
}

__global__ void copy_transpose_kernel(const float* __restrict__ w, int rows, int cin, int transpose, float* __restrict__ out) {
    const int64_t total = static_cast<int64_t>(rows) * cin;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int c = static_cast<int>(i % cin);
        const int n = static_cast<int>(i / cin);
        out[i] = transpose ? w[static_cast<int64_t>(c) * rows + n] : w[i];
    }
}

}  // namespace

extern "C" int64_t ddpm_packed_weight_floats(int taps, int rows, int cin) {
    if (taps <= 0 || rows <= 0 || cin <= 0) return 0;
    if (!split_math()) return static_cast<int64_t>(taps) * rows * cin;
    return static_cast<int64_t>(taps) * ((cin + 15) / 16) * ((rows + 31) / 32) * 768;   // 3 KiB per (tap, 16 channels, 32 rows)
}

extern "C" int ddpm_pack_conv1x1(const float* w, int rows, int cin, int transpose, float* out, void* stream) {
    DDPM_REQUIRE(w && out && rows > 0 && cin > 0);
    if (split_math())
        return transpose ? launch_split_pack(w, out, 1, rows, cin, rows, 0, 0, 0, 1, rows, as_stream(stream))
                         : launch_split_pack(w, out, 1, rows, cin, rows, 0, 0, 0, cin, 1, as_stream(stream));
    const int64_t total = static_cast<int64_t>(rows) * cin;
    const int grid = static_cast<int>((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    copy_transpose_kernel<<<grid, 256, 0, as_stream(stream)>>>(w, rows, cin, transpose, out);
    return ddpm::check_launch("copy_transpose_kernel");
}

extern "C" int ddpm_pack_qkv(const float* qw, const float* kw, const float* vw, const float* qb, const float* kb, const float* vb,
                             int c, int dq, int dqp, float* w_packed, float* wt_packed, float* bias, void* stream) {
    DDPM_REQUIRE(qw && kw && vw && qb && kb && vb && w_packed && wt_packed && bias);
    DDPM_REQUIRE(c > 0 && dq > 0 && dqp >= dq);
    QkvPack a;
    a.qw = qw; a.kw = kw; a.vw = vw; a.qb = qb; a.kb = kb; a.vb = vb;
    a.out_f = w_packed; a.out_t = wt_packed; a.bias = bias;
    a.c = c; a.dq = dq; a.dqp = dqp;
    a.split = split_math() ? 1 : 0;
    a.f16 = f16_math() ? 1 : 0;
    a.range_flag = a.f16 ? ddpm::range_flag() : nullptr;
    if (a.f16 && !a.range_flag) return DDPM_E_LAUNCH;
    const int64_t total = static_cast<int64_t>(2 * dqp + c + 31) * (c + 31);
    const int grid = static_cast<int>((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    pack_qkv_kernel<<<grid, 256, 0, as_stream(stream)>>>(a);
    return ddpm::check_launch("pack_qkv_kernel");
}

extern "C" int ddpm_conv3x3_fwd(const ddpm_conv_desc* d, const float* src1, const float* src2, const float* w_packed,
                                const ddpm_epilogue* ep, float* out, void* stream) {
    if (int rc = validate_common(d, src1, src2, w_packed, ep, out)) return rc;
    GemmParams p;
    fill_common(p, d, src1, src2, w_packed, ep, out);
    p.taps = 9;
    if (split_math()) {
        const int rc = launch_conv3x3_halo(p, as_stream(stream));
        if (rc <= 0) return rc;   // launched, or failed; positive = shape for the generic kernel
    }
    return launch_gemm<MODE_CONV3>(p, as_stream(stream));
}

extern "C" int ddpm_conv3x3_fwd_stats(const ddpm_conv_desc* d, const float* src1, const float* src2, const float* w_packed,
                                      const ddpm_epilogue* ep, float* out, double* partial, int* nblocks, void* stream) {
    if (int rc = validate_common(d, src1, src2, w_packed, ep, out)) return rc;
    DDPM_REQUIRE(partial && nblocks && (d->cout & 3) == 0 && d->cout <= 1024);
    GemmParams p;
    fill_common(p, d, src1, src2, w_packed, ep, out);
    p.taps = 9;
    if (split_math()) {
        p.stat_partial = partial;
        const int rc = launch_conv3x3_halo(p, as_stream(stream));
        if (rc < 0) return rc;
        if (rc == 0) {
            *nblocks = p.stat_blocks;
            return DDPM_OK;
        }
        p.stat_partial = nullptr;
    }
    // shapes of the generic kernel: the convolution, then the statistics' own pass over its output
    if (int rc = launch_gemm<MODE_CONV3>(p, as_stream(stream))) return rc;
    return ddpm_bn_partials(out, d->ldo, static_cast<int64_t>(d->batch) * d->in_h * d->in_w, d->cout, partial, nblocks, stream);
}

extern "C" int ddpm_conv1x1_fwd(const ddpm_conv_desc* d, const float* src1, const float* src2, const float* w_packed,
                                const ddpm_epilogue* ep, float* out, void* stream) {
    if (int rc = validate_common(d, src1, src2, w_packed, ep, out)) return rc;
    GemmParams p;
    fill_common(p, d, src1, src2, w_packed, ep, out);
    p.taps = 1;
    if (split_math() && resident_1x1()) {
        const int rc = launch_conv1x1_resident(p, as_stream(stream));
        if (rc <= 0) return rc;   // launched, or failed; positive = shape for the generic kernel
    }
    return launch_gemm<MODE_CONV1>(p, as_stream(stream));
}

extern "C" int ddpm_self_attention_infer(const float* x, int ldx, const float* wqkv_packed, const float* bias_qkv, const float* gamma,
                                         const float* temb, float* out, int ldo, float* workspace, int batch, int n, int c, int dq,
                                         void* stream) {
    DDPM_REQUIRE(x && wqkv_packed && bias_qkv && gamma && out && workspace && batch > 0 && n > 0 && c > 0 && dq > 0);
    DDPM_REQUIRE(static_cast<int64_t>(batch) * n < (1ll << 31) && ddpm::aligned16(workspace));
    ddpm_conv_desc d = {};
    d.batch = batch;
    d.in_h = n;
    d.in_w = 1;
    d.c1 = c;
    d.ld1 = ldx;
    d.cout = 2 * dq + c;
    d.ldo = d.cout;
    d.act = DDPM_ACT_NONE;
    d.out_h = n;
    d.out_w = 1;
    ddpm_epilogue ep = {};
    ep.shift = bias_qkv;
    if (split_math() && resident_1x1() && ddpm::attention_fwd_pairs_fits(batch, n, c, dq)) {
        if (int rc = validate_common(&d, x, nullptr, wqkv_packed, &ep, workspace)) return rc;
        GemmParams p;
        fill_common(p, &d, x, nullptr, wqkv_packed, &ep, workspace);
        p.taps = 1;
        p.ldo = 2 * dq;
        p.pair_from = 2 * dq;
        p.pair_c = c;
        p.pair_v = reinterpret_cast<unsigned char*>(workspace + static_cast<int64_t>(batch) * n * 2 * dq);
        const int rc = launch_conv1x1_pairs(p, as_stream(stream));
        if (rc < 0) return rc;
        if (rc == 0)
            return ddpm::attention_fwd_pairs(workspace, p.pair_v, x, ldx, gamma, temb, out, ldo, batch, n, c, dq, as_stream(stream));
    }
    if (int rc = ddpm_conv1x1_fwd(&d, x, nullptr, wqkv_packed, &ep, workspace, stream)) return rc;
    return ddpm_attention_fwd(workspace, x, ldx, gamma, out, ldo, nullptr, nullptr, temb, batch, n, c, dq, stream);
}

extern "C" int ddpm_convT2x2_fwd(const ddpm_conv_desc* d, const float* src, const float* w_packed, const ddpm_epilogue* ep,
                                 float* out, void* stream) {
    if (int rc = validate_common(d, src, nullptr, w_packed, ep, out)) return rc;
    DDPM_REQUIRE(d->c2 == 0 && d->out_h > 0 && d->out_w > 0);
    DDPM_REQUIRE(!ep->residual && !ep->temb && d->act == DDPM_ACT_NONE);
    DDPM_REQUIRE(static_cast<int64_t>(d->batch) * d->out_h * d->out_w < (1ll << 31));
    GemmParams p;
    fill_common(p, d, src, nullptr, w_packed, ep, out);
    p.taps = 1;
    p.n_total = 4 * d->cout;
    p.out_h = d->out_h;
    p.out_w = d->out_w;
    p.off_y = d->off_y;
    p.off_x = d->off_x;
    // Only input pixels with at least one surviving output are enumerated as GEMM rows.
    int y0 = d->in_h, y1 = -1, x0 = d->in_w, x1 = -1;
    for (int y = 0; y < d->in_h; ++y)
        if ((2 * y + d->off_y >= 0 && 2 * y + d->off_y < d->out_h) || (2 * y + 1 + d->off_y >= 0 && 2 * y + 1 + d->off_y < d->out_h)) {
            if (y < y0) y0 = y;
            y1 = y;
        }
    for (int x = 0; x < d->in_w; ++x)
        if ((2 * x + d->off_x >= 0 && 2 * x + d->off_x < d->out_w) || (2 * x + 1 + d->off_x >= 0 && 2 * x + 1 + d->off_x < d->out_w)) {
            if (x < x0) x0 = x;
            x1 = x;
        }
    if (y1 < y0 || x1 < x0) return DDPM_OK;  // everything cropped away
    p.win_y0 = y0;
    p.win_x0 = x0;
    p.win_h = y1 - y0 + 1;
    p.win_w = x1 - x0 + 1;
    p.m_total = d->batch * p.win_h * p.win_w;
    return launch_gemm<MODE_CONVT>(p, as_stream(stream));
}

extern "C" int ddpm_pack_conv3x3(const float* w, int cout, int cin, int flip, float* out, void* stream) {
    DDPM_REQUIRE(w && out && cout > 0 && cin > 0);
    if (split_math())   // rows = cout, depth = cin; flipped: rows = cin, depth = cout, tap mirrored
        return flip ? launch_split_pack(w, out, 9, cin, cout, cin, 8, -1, 0, 9, 9ll * cin, as_stream(stream))
                    : launch_split_pack(w, out, 9, cout, cin, cout, 0, 1, 0, 9ll * cin, 9, as_stream(stream));
    const int64_t total = static_cast<int64_t>(cout) * cin * 9;
    const int grid = static_cast<int>((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    pack_conv3x3_kernel<<<grid, 256, 0, as_stream(stream)>>>(w, cout, cin, flip, out);
    return ddpm::check_launch("pack_conv3x3_kernel");
}

extern "C" int ddpm_pack_conv3x3_batch(int n, const float* const* w, const int* cout, const int* cin, const int* flip, float* const* out,
                                       void* stream) {
    DDPM_REQUIRE(n >= 0 && (n == 0 || (w && cout && cin && flip && out)));
    for (int i = 0; i < n; ++i) DDPM_REQUIRE(w[i] && out[i] && cout[i] > 0 && cin[i] > 0);
    if (!split_math()) {
        for (int i = 0; i < n; ++i)
            if (int rc = ddpm_pack_conv3x3(w[i], cout[i], cin[i], flip[i], out[i], stream)) return rc;
        return DDPM_OK;
    }
    for (int i0 = 0; i0 < n; i0 += kPackBatch) {
        const int cnt = n - i0 < kPackBatch ? n - i0 : kPackBatch;
        SplitPackBatch b;
        int64_t most = 0;
        for (int k = 0; k < kPackBatch; ++k) {
            const int i = i0 + (k < cnt ? k : 0);      // unused slots repeat the first item (never selected: gridDim.y = cnt)
            b.item[k] = flip[i] ? make_split_pack(w[i], out[i], 9, cin[i], cout[i], cin[i], 8, -1, 0, 9, 9ll * cin[i])
                                : make_split_pack(w[i], out[i], 9, cout[i], cin[i], cout[i], 0, 1, 0, 9ll * cin[i], 9);
            if (b.item[k].f16 && !b.item[k].range_flag) return DDPM_E_LAUNCH;
            const int64_t total = 9ll * b.item[k].kc16 * b.item[k].nbp * 512;
            if (k < cnt && total > most) most = total;
        }
        const int gx = static_cast<int>((most + 255) / 256 > 1024 ? 1024 : (most + 255) / 256);
        pack_split_batch_kernel<<<dim3(gx, cnt), 256, 0, as_stream(stream)>>>(b);
    }
    return ddpm::check_launch("pack_split_batch_kernel");
}

extern "C" int ddpm_pack_convT2x2(const float* w, int cin, int cout, float* out, void* stream) {
    DDPM_REQUIRE(w && out && cout > 0 && cin > 0);
    if (split_math())   // row q*cout + co, depth ci: w[ci][co][q]
        return launch_split_pack(w, out, 1, 4 * cout, cin, cout, 0, 0, 1, 4, 4ll * cout, as_stream(stream));
    const int64_t total = static_cast<int64_t>(cout) * cin * 4;
    const int grid = static_cast<int>((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    pack_convT_kernel<<<grid, 256, 0, as_stream(stream)>>>(w, cin, cout, out);
    return ddpm::check_launch("pack_convT_kernel");
}

extern "C" int ddpm_pack_convT2x2_dgrad(const float* w, int cin, int cout, float* out, void* stream) {
    DDPM_REQUIRE(w && out && cout > 0 && cin > 0);
    if (split_math())   // tap q, row ci, depth co: w[ci][co][q]
        return launch_split_pack(w, out, 4, cin, cout, cin, 0, 1, 0, 4ll * cout, 4, as_stream(stream));
    const int64_t total = static_cast<int64_t>(cout) * cin * 4;
    const int grid = static_cast<int>((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    pack_convT_dgrad_kernel<<<grid, 256, 0, as_stream(stream)>>>(w, cin, cout, out);
    return ddpm::check_launch("pack_convT_dgrad_kernel");
}

// d->in_h/in_w: geometry of the transposed conv's INPUT (= of the gradient produced, `out`);
// d->out_h/out_w/off_*: geometry/placement of its output (= of `dy`, c1 = its channels, ld1 its stride);
// d->cout: channels of the gradient produced.
extern "C" int ddpm_convT2x2_dgrad(const ddpm_conv_desc* d, const float* dy, const float* w_packed, const ddpm_epilogue* ep,
                                   float* out, void* stream) {
    if (int rc = validate_common(d, dy, nullptr, w_packed, ep, out)) return rc;
    DDPM_REQUIRE(d->c2 == 0 && d->out_h > 0 && d->out_w > 0 && !ep->temb);
    DDPM_REQUIRE(static_cast<int64_t>(d->batch) * d->out_h * d->out_w < (1ll << 31));
    GemmParams p;
    fill_common(p, d, dy, nullptr, w_packed, ep, out);
    p.taps = 4;
    p.out_h = d->out_h;
    p.out_w = d->out_w;
    p.off_y = d->off_y;
    p.off_x = d->off_x;
    // the gather addresses dy with absolute 32-bit byte offsets
    p.src1_pixels = static_cast<long long>(d->batch) * d->out_h * d->out_w;
    p.src2_pixels = p.src1_pixels;
    DDPM_REQUIRE(p.src1_pixels * d->ld1 * 4ll < 0x7fffffffll);
    return launch_gemm<MODE_CONVT_DGRAD>(p, as_stream(stream));
}

// Weight gradients of the convolutions as split-K GEMMs on the fp32 matrix cores:
//
//   dW[tap][m][n] = sum over pixels p of  A[pA(p,tap)][m] * B[pB(p,tap)][n]
//
//   3x3 conv : A = dY (m = output channel) at p,            B = X (n = input channel) at p + tap offset
//   1x1 conv : A = dY at p,                                  B = X at p
//   convT 2x2: A = dY at (2y+a+off_y, 2x+b+off_x), tap=(a,b), B = X at p           (m = cout, n = cin)
//
// The reduction axis (pixels) is the slow axis of both NHWC operands, so tiles are staged in LDS
// K-major ([pixel][channel], a plain copy of 32 pixel rows) and the MFMA fragments are read with
// conflict-free 4-byte reads (lane = channel). Pixels are split over workgroups; every workgroup
// adds its partial tile into the packed result [tap][M][N] with fp32 atomics (contiguous along n),
// which the caller has zeroed. A second tiny kernel permutes the packed result into torch's
// parameter layout.
#include <cstdlib>
#include "conv_common.h"

using ddpm::as_stream;
using ddpm::conv::f16x2;
using ddpm::conv::f16x8;
using ddpm::conv::f32x2;
using ddpm::conv::kLoInv;
using ddpm::conv::kLoScale;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_CONV3 = 0;
constexpr int WG_CONV1 = 1;
constexpr int WG_CONVT = 2;
constexpr int BKP = 32;  // pixels per K-step

struct WgradParams {
    const float* a;  // dY
    const float* b;  // X
    float* out;      // packed [taps][m_total][n_total]
    int batch, h, w;          // image the reduction enumerates (= X's geometry)
    int a_h, a_w;             // dY's geometry (differs for convT)
    int m_total, lda, n_total, ldb;
    int off_y, off_x, taps;
    int k_total, k_chunk, splits;
    int m_tiles, n_tiles;
    int work;                 // workgroups with work (the grid is rounded up to a multiple of 8)
    int tap_groups;           // work items along the tap axis: taps, or ceil(taps / 2) when a tile holds two taps
    unsigned* range_flag;     // fp16-pair kernel: raised when an operand lies beyond fp16's range
};

constexpr unsigned OOB = 0x80000000u;  // offsets at or beyond the buffer descriptor's range read as zero

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, long long bytes) {
    const unsigned n = bytes > 0x7fffffffll ? 0x7fffffffu : (bytes < 0 ? 0u : static_cast<unsigned>(bytes));
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

template <int BT, int MODE>  // BT x BT output tile (channels x channels)
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int TT = BT / 64;           // 32x32 MFMA tiles per wave per dimension
    constexpr int F4R = BT / 4;           // float4 per tile row
    constexpr int LPT = BKP * F4R / 256;  // float4 loads per thread per operand
    constexpr int RSTEP = 256 / F4R;      // rows between a thread's consecutive loads
    __shared__ __attribute__((aligned(16))) float sA[2][BKP * BT];
    __shared__ __attribute__((aligned(16))) float sB[2][BKP * BT];

    // Workgroups are handed to the 8 XCDs round-robin and each XCD has its own L2: renumber so that an
    // XCD runs CONSECUTIVE work items, i.e. the taps and channel tiles that read the same pixels share
    // one L2 instead of each fetching its own copy over the fabric.
    int id = static_cast<int>((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    if (id >= p.work) return;
    const int nt = id % p.n_tiles; id /= p.n_tiles;
    const int mt = id % p.m_tiles; id /= p.m_tiles;
    const int tap = id % p.taps;
    const int split = id / p.taps;
    const int m0 = mt * BT, n0 = nt * BT;
    const int k_begin = split * p.k_chunk;
    const int k_end = min(p.k_total, k_begin + p.k_chunk);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    int dy = 0, dx = 0;
    if (MODE == WG_CONV3) {
        dy = tap / 3 - 1;
        dx = tap % 3 - 1;
    }
    const int per = p.h * p.w;

    // Branch-free loads through buffer descriptors based just before this workgroup's first pixel
    // (small 32-bit offsets whatever the tensor size); rows or taps outside the image, pixels past
    // the chunk and channels past the tile get an out-of-range offset and read as zeros. The
    // transposed-conv gather addresses dY by absolute offsets (the host checks it is < 2 GiB).
    const long long pixels = static_cast<long long>(p.batch) * per;
    long long base_pix = static_cast<long long>(k_begin) - p.w - 1;
    if (base_pix < 0) base_pix = 0;
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.b + base_pix * p.ldb, (pixels - base_pix) * p.ldb * 4ll);
    const long long a_pixels = static_cast<long long>(p.batch) * p.a_h * p.a_w;
    const long long a_base = MODE == WG_CONVT ? 0 : base_pix;
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a + a_base * p.lda, (a_pixels - a_base) * p.lda * 4ll);
    const int base_i = static_cast<int>(base_pix);

    // this thread's tile rows are row0 + RSTEP*i; its channel quad is fixed
    const int row0 = tid / F4R, c4 = tid % F4R;
    const unsigned ca = m0 + c4 * 4 < p.m_total ? static_cast<unsigned>(m0 + c4 * 4) * 4u : OOB;
    const unsigned cb = n0 + c4 * 4 < p.n_total ? static_cast<unsigned>(n0 + c4 * 4) * 4u : OOB;
    // (sample, pixel-in-image) of each row, advanced by BKP per step without divisions;
    // y = rem / w by multiplication with ceil(2^32 / w), exact for rem * w < 2^32.
    const unsigned inv_w = static_cast<unsigned>((0x100000000ull + p.w - 1) / p.w);
    int rb[LPT], rrem[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int pk = k_begin + row0 + RSTEP * i;
        rb[i] = pk / per;
        rrem[i] = pk - rb[i] * per;
    }

    f32x16 acc[TT][TT];
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 areg[LPT], breg[LPT];
    auto load_global = [&](int k0) {   // must be called with k0 = k_begin, k_begin + BKP, ... in order
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int pk = k0 + row0 + RSTEP * i;
            const int rem = rrem[i], bi = rb[i];
            const int y = p.w == 1 ? rem : static_cast<int>(__umulhi(static_cast<unsigned>(rem), inv_w));
            const int x = rem - y * p.w;
            const bool in_chunk = pk < k_end;
            unsigned oa, ob;
            if (MODE == WG_CONVT) {
                const int oy = 2 * y + (tap >> 1) + p.off_y, ox = 2 * x + (tap & 1) + p.off_x;
                const bool ok = in_chunk && oy >= 0 && oy < p.a_h && ox >= 0 && ox < p.a_w;
                oa = ok ? static_cast<unsigned>((bi * p.a_h + oy) * p.a_w + ox) * p.lda * 4u + ca : OOB;
                ob = in_chunk ? static_cast<unsigned>(pk - base_i) * p.ldb * 4u + cb : OOB;
            } else {
                oa = in_chunk ? static_cast<unsigned>(pk - base_i) * p.lda * 4u + ca : OOB;
                const int yy = y + dy, xx = x + dx;
                const bool ok = in_chunk && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
                ob = ok ? static_cast<unsigned>(pk - base_i + dy * p.w + dx) * p.ldb * 4u + cb : OOB;
            }
            areg[i] = buf_load(rs_a, oa);
            breg[i] = buf_load(rs_b, ob);
            // advance this row by one K-step
            int nr = rem + BKP;
            int nb = bi;
            while (nr >= per) {
                nr -= per;
                ++nb;
            }
            rrem[i] = nr;
            rb[i] = nb;
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int f = tid + 256 * i;
            *reinterpret_cast<f32x4*>(&sA[buf][f * 4]) = areg[i];   // [row][c4*4] with row stride BT == f*4
            *reinterpret_cast<f32x4*>(&sB[buf][f * 4]) = breg[i];
        }
    };

    const int steps = (k_end - k_begin + BKP - 1) / BKP;
    if (steps > 0) {
        load_global(k_begin);
        store_lds(0);
    }
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < steps;
        if (more) load_global(k_begin + (s + 1) * BKP);
        const float* pa = &sA[buf][wm * (BT / 2) + r];
        const float* pb = &sB[buf][wn * (BT / 2) + r];
#pragma unroll
        for (int ks = 0; ks < BKP / 2; ++ks) {
            const int k = 2 * ks + h;
            float a[TT], b[TT];
#pragma unroll
            for (int i = 0; i < TT; ++i) a[i] = pa[k * BT + i * 32];
#pragma unroll
            for (int j = 0; j < TT; ++j) b[j] = pb[k * BT + j * 32];
            if (ks == BKP / 2 - 4 && more) store_lds(buf ^ 1);   // before the last MFMAs, not after them
#pragma unroll
            for (int i = 0; i < TT; ++i)
#pragma unroll
                for (int j = 0; j < TT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    float* outp = p.out + static_cast<int64_t>(tap) * p.m_total * p.n_total;
#pragma unroll
    for (int j = 0; j < TT; ++j) {
        const int n = n0 + wn * (BT / 2) + j * 32 + r;
        if (n >= p.n_total) continue;
#pragma unroll
        for (int i = 0; i < TT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BT / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < p.m_total) atomicAdd(outp + static_cast<int64_t>(m) * p.n_total + n, acc[i][j][e]);
            }
    }
}


// ---- split-bf16 variant (128 x 128 tiles) -------------------------------------------------------
// Same decomposition, on v_mfma_f32_32x32x16_bf16: both operands are split into three bf16 planes
// (x = hi + mid + lo exactly) on their way from global memory to LDS and a product is the sum of its
// six cross terms of weight >= 2^-16, accumulated in fp32 (dropped terms below 2^-23 of the product).
// The LDS image keeps the natural [pixel][channel] order; the MFMA wants 8 consecutive PIXELS of one
// channel per lane, which the transposing read ds_read_b64_tr_b16 delivers (4 pixel rows x 16
// channels per 16-lane group; rows of 256 + 64 bytes keep the four rows of a 32-lane half on
// different banks). The MFMA accumulator is folded into a master accumulator with a round-to-nearest
// add every 8 steps (128 pixels), because the bf16 MFMA truncates when it accumulates.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int XKP = 16;                  // pixels per K-step
constexpr int XROWB = 128 * 2 + 64;      // bytes of one pixel row of one plane
constexpr int XPLANE = XKP * XROWB;
constexpr int XFOLD = 8;

__device__ __forceinline__ void split4_store(const f32x4 v, unsigned char* dst) {
    unsigned u[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u[e] = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(u[e] & 0xffff0000u);
        m[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(m[e] & 0xffff0000u);
        l[e] = __float_as_uint(r2);
    }
    u32x2 hi, mid, lo;
    hi[0] = __builtin_amdgcn_perm(u[1], u[0], 0x07060302u);
    hi[1] = __builtin_amdgcn_perm(u[3], u[2], 0x07060302u);
    mid[0] = __builtin_amdgcn_perm(m[1], m[0], 0x07060302u);
    mid[1] = __builtin_amdgcn_perm(m[3], m[2], 0x07060302u);
    lo[0] = __builtin_amdgcn_perm(l[1], l[0], 0x07060302u);
    lo[1] = __builtin_amdgcn_perm(l[3], l[2], 0x07060302u);
    *reinterpret_cast<u32x2*>(dst) = hi;
    *reinterpret_cast<u32x2*>(dst + XPLANE) = mid;
    *reinterpret_cast<u32x2*>(dst + 2 * XPLANE) = lo;
}

template <class W>
__device__ __forceinline__ void split4_store_f16(const f32x4 v, unsigned char* dst, W& amax) {
    unsigned h0, l0, h1, l1;
    ddpm::conv::watch2(amax, f32x2{v[0], v[1]});
    ddpm::conv::watch2(amax, f32x2{v[2], v[3]});
    ddpm::conv::split2_pair(f32x2{v[0], v[1]}, h0, l0);
    ddpm::conv::split2_pair(f32x2{v[2], v[3]}, h1, l1);
    *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(dst + XPLANE) = u32x2{l0, l1};
}

// PAIR (3x3 only): a layer with 64 channels on one side would leave half of the 128 x 128 tile empty. Instead the
// tile's two 64-wide halves on that side take TWO CONSECUTIVE TAPS of the same 64 channels (five work items along the
// tap axis instead of nine, the odd one half empty). PAIR = 1: 64 input channels, the halves of the B tile are X at
// the two taps' offsets. PAIR = 2: 64 output channels; the B tile must be common to both halves, so the sum runs over
// X's pixels q with A = dY at q - offset of the half's tap (the same products, enumerated from the other side).
//
// F16: both operands as fp16 pairs instead (hi, and the remainder scaled by 2^11; conv_common.h): two planes, three
// MFMAs per fragment pair -- hi*hi into `acc`, the two cross products into `master`, which is scaled by 2^-11 and
// added at the end -- and no fold (the fp16 MFMA accumulates without drift).
//
// WATCH (F16): this workgroup watches the range of the operands it stages. The same pixels of dY and X are staged by
// every tap group and by every channel tile of the other operand, so the kernel below picks the workgroups whose
// loads together cover both tensors once and gives only those the checking code.
template <int MODE, int PAIR, bool F16, bool WATCH>
__device__ __forceinline__ void conv_wgrad_x3_body(const WgradParams& p, int id, unsigned char* sA_mem, unsigned char* sB_mem) {
    static_assert(PAIR == 0 || MODE == WG_CONV3, "two taps per tile exist for the 3x3 convolution only");
    if constexpr (F16) ddpm::conv::fp16_saturate();
    std::conditional_t<WATCH, ddpm::conv::RangeWatch, ddpm::conv::NoWatch> amax;   // F16: largest magnitude this thread has turned into an fp16 pair
    constexpr int BT = 128;
    constexpr int NPL = F16 ? 2 : 3;      // operand planes
    constexpr int F4R = BT / 4;           // float4 per tile row
    constexpr int LPT = XKP * F4R / 256;  // float4 loads per thread per operand (2)
    constexpr int RSTEP = 256 / F4R;      // rows between a thread's consecutive loads (8)
    unsigned char (*sA)[NPL * XPLANE] = reinterpret_cast<unsigned char (*)[NPL * XPLANE]>(sA_mem);   // [2]
    unsigned char (*sB)[NPL * XPLANE] = reinterpret_cast<unsigned char (*)[NPL * XPLANE]>(sB_mem);   // [2]

    const int nt = id % p.n_tiles; id /= p.n_tiles;
    const int mt = id % p.m_tiles; id /= p.m_tiles;
    const int tg = id % p.tap_groups;
    const int split = id / p.tap_groups;
    const int m0 = mt * BT, n0 = nt * BT;
    const int k_begin = split * p.k_chunk;
    const int k_end = min(p.k_total, k_begin + p.k_chunk);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = tid / F4R, c4 = tid % F4R;
    // the tap this THREAD loads for (its column half decides with PAIR), and the tap this WAVE accumulates
    const int tap = PAIR ? 2 * tg + (c4 >> 4) : tg;
    const int wave_tap = PAIR == 1 ? 2 * tg + wn : (PAIR == 2 ? 2 * tg + wm : tg);
    const bool tap_ok = tap < p.taps;
    // a layer with at most 64 channels on BOTH sides runs as PAIR = 1 with the lower half of the tile's rows empty: the
    // waves of that half stage their share of the operands but issue no matrix instructions
    const bool idle_wave = PAIR == 1 && __builtin_amdgcn_readfirstlane(wm * 64 >= p.m_total ? 1 : 0) != 0;
    int dy = 0, dx = 0;
    if (MODE == WG_CONV3) {
        dy = tap / 3 - 1;
        dx = tap % 3 - 1;
        if (PAIR == 2) {   // dY is the shifted operand, by the opposite offset
            dy = -dy;
            dx = -dx;
        }
    }
    const int per = p.h * p.w;

    const long long pixels = static_cast<long long>(p.batch) * per;
    long long base_pix = static_cast<long long>(k_begin) - p.w - 1;
    if (base_pix < 0) base_pix = 0;
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.b + base_pix * p.ldb, (pixels - base_pix) * p.ldb * 4ll);
    const long long a_pixels = static_cast<long long>(p.batch) * p.a_h * p.a_w;
    const long long a_base = MODE == WG_CONVT ? 0 : base_pix;
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a + a_base * p.lda, (a_pixels - a_base) * p.lda * 4ll);
    const int base_i = static_cast<int>(base_pix);

    const int cma = PAIR == 2 ? (c4 & 15) * 4 : m0 + c4 * 4;    // this thread's first channel of A and of B
    const int cnb = PAIR == 1 ? (c4 & 15) * 4 : n0 + c4 * 4;
    const unsigned ca = cma < p.m_total && (PAIR != 2 || tap_ok) ? static_cast<unsigned>(cma) * 4u : OOB;
    const unsigned cb = cnb < p.n_total && (PAIR != 1 || tap_ok) ? static_cast<unsigned>(cnb) * 4u : OOB;
    // Per-thread walk over its pixels (one every XKP per step), kept as (x, y) inside the image plus
    // running byte offsets, so that a step costs adds, compares and selects only: advancing XKP pixels
    // is q16 rows and r16 columns, with one carry each into the row and the image.
    const int r16 = XKP % p.w, q16 = XKP / p.w, ri = q16 % p.h, qi = q16 / p.h;
    int px[LPT], py[LPT], pkv[LPT];
    unsigned oa_run[LPT], ob_run[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int pk = k_begin + row0 + RSTEP * i;
        const int bi = pk / per, rem = pk - bi * per;
        py[i] = rem / p.w;
        px[i] = rem - py[i] * p.w;
        pkv[i] = pk;
        if (MODE == WG_CONVT) {
            const int oy = 2 * py[i] + (tap >> 1) + p.off_y, ox = 2 * px[i] + (tap & 1) + p.off_x;
            oa_run[i] = static_cast<unsigned>((bi * p.a_h + oy) * p.a_w + ox) * p.lda * 4u + ca;
            ob_run[i] = static_cast<unsigned>(pk - base_i) * p.ldb * 4u + cb;
        } else if (PAIR == 2) {
            oa_run[i] = static_cast<unsigned>(pk - base_i + dy * p.w + dx) * p.lda * 4u + ca;
            ob_run[i] = static_cast<unsigned>(pk - base_i) * p.ldb * 4u + cb;
        } else {
            oa_run[i] = static_cast<unsigned>(pk - base_i) * p.lda * 4u + ca;
            ob_run[i] = static_cast<unsigned>(pk - base_i + dy * p.w + dx) * p.ldb * 4u + cb;
        }
    }
    // offset steps: the linear ones, and for the transposed conv the (2y, 2x) walk over the larger image
    const unsigned a_step = MODE == WG_CONVT ? static_cast<unsigned>((qi * p.a_h + 2 * ri) * p.a_w + 2 * r16) * p.lda * 4u
                                             : static_cast<unsigned>(XKP) * p.lda * 4u;
    const unsigned a_cx = static_cast<unsigned>(2 * p.a_w - 2 * p.w) * p.lda * 4u;                    // a column carry
    const unsigned a_cy = static_cast<unsigned>(p.a_h * p.a_w - 2 * p.h * p.a_w) * p.lda * 4u;        // a row carry
    const unsigned b_step = static_cast<unsigned>(XKP) * p.ldb * 4u;

    f32x16 acc[2][2], master[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                master[i][j][e] = 0.f;
            }

    // Two register sets: the loads of step s + 2 are in flight while step s + 1 is split into LDS and
    // step s is multiplied, so a global load has a whole step to arrive. Pixels past the chunk read
    // as zero (out-of-range buffer offsets), which lets the loop run an even number of branch-free steps.
    f32x4 areg[2][LPT], breg[2][LPT];
    auto load_global = [&](int set) {   // the n-th call loads pixels k_begin + n * XKP ...
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int x = px[i], y = py[i];
            const bool in_chunk = pkv[i] < k_end;
            bool ok;
            if (MODE == WG_CONVT) {
                const int oy = 2 * y + (tap >> 1) + p.off_y, ox = 2 * x + (tap & 1) + p.off_x;
                ok = in_chunk & (static_cast<unsigned>(oy) < static_cast<unsigned>(p.a_h)) & (static_cast<unsigned>(ox) < static_cast<unsigned>(p.a_w));
                areg[set][i] = buf_load(rs_a, ok ? oa_run[i] : OOB);
                breg[set][i] = buf_load(rs_b, in_chunk ? ob_run[i] : OOB);
            } else {
                const int yy = y + dy, xx = x + dx;
                ok = in_chunk & (static_cast<unsigned>(yy) < static_cast<unsigned>(p.h)) & (static_cast<unsigned>(xx) < static_cast<unsigned>(p.w));
                areg[set][i] = buf_load(rs_a, (PAIR == 2 ? ok : in_chunk) ? oa_run[i] : OOB);
                breg[set][i] = buf_load(rs_b, (PAIR == 2 ? in_chunk : ok) ? ob_run[i] : OOB);
            }
            int nx = x + r16;
            const bool cx = nx >= p.w;
            nx -= cx ? p.w : 0;
            int ny = y + ri + (cx ? 1 : 0);
            const bool cy = ny >= p.h;
            ny -= cy ? p.h : 0;
            px[i] = nx;
            py[i] = ny;
            pkv[i] += XKP;
            ob_run[i] += b_step;
            if (MODE == WG_CONVT) oa_run[i] += a_step + (cx ? a_cx : 0u) + (cy ? a_cy : 0u);
            else oa_run[i] += a_step;
        }
    };
    auto store_lds = [&](int set, int buf) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int off = (row0 + RSTEP * i) * XROWB + c4 * 8;
            if constexpr (F16) {
                split4_store_f16(areg[set][i], &sA[buf][off], amax);
                split4_store_f16(breg[set][i], &sB[buf][off], amax);
            } else {
                split4_store(areg[set][i], &sA[buf][off]);
                split4_store(breg[set][i], &sB[buf][off]);
            }
        }
    };

    // transposing reads: lane 4q + c of a 16-lane group names pixel row q (of the lane half's 4j-th
    // quartet) and channels 4c .. 4c+3 of the group's 16; the lane then holds ITS channel for those 4 pixels
    const int tlane = (8 * h + ((lane & 15) >> 2)) * XROWB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    auto frag = [&](const unsigned char* tile, int block, int pl) -> s16x8 {
        const unsigned char* src = tile + pl * XPLANE + block * 64 + tlane;
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(src));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(src + 4 * XROWB));
        return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    // One K-step from LDS buffer `set`: multiply it, split step + 1 (register set set ^ 1) into the other
    // buffer, start the loads of step + 2 into register set `set`. The splitting arithmetic is cut into
    // twelve stages of about eight instructions and one stage follows every other matrix instruction,
    // so that it issues while the matrix pipe works; the scheduling barriers keep that order.
    auto step = [&](int set, int s) {
        const unsigned char* tA = sA[set];
        const unsigned char* tB = sB[set];
        s16x8 a[2][NPL], b[2][NPL];
        auto reads = [&](int pa, int pb) {
            a[0][pa] = frag(tA, wm * 2, pa);
            a[1][pa] = frag(tA, wm * 2 + 1, pa);
            b[0][pb] = frag(tB, wn * 2, pb);
            b[1][pb] = frag(tB, wn * 2 + 1, pb);
        };
        auto src = [&](int v) -> f32x4 { return v < 2 ? areg[set ^ 1][v] : breg[set ^ 1][v - 2]; };
        auto dst_of = [&](int v) -> unsigned char* {
            return (v < 2 ? &sA[set ^ 1][0] : &sB[set ^ 1][0]) + (row0 + RSTEP * (v & 1)) * XROWB + c4 * 8;
        };
#ifndef DDPM_WG_ABLATE
#define DDPM_WG_ABLATE 0   // diagnostic builds only: 1 no split stages, 2 no global loads, 4 no barrier, 8 no matrix ops, 16 no fragment reads (fp16 pair), 32 no atomic adds
#endif
        if constexpr (F16) {
            // split state of the four float4 of step + 1 (two of each operand): eight stages, one behind each of the
            // first eight matrix instructions
            unsigned hi2[4][2];
            f32x2 rem[4][2];
            auto stage = [&](int st) {
                const int v = st >> 1;
                if ((st & 1) == 0) {
                    const f32x4 x = src(v);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x2 xv = {x[2 * q], x[2 * q + 1]};
                        ddpm::conv::watch2(amax, xv);
                        const f16x2 hh = __builtin_convertvector(xv, f16x2);
                        hi2[v][q] = __builtin_bit_cast(unsigned, hh);
                        rem[v][q] = (xv - __builtin_convertvector(hh, f32x2)) * kLoScale;
                    }
                } else {
                    unsigned lo2[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) lo2[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(rem[v][q], f16x2));
                    unsigned char* dst = dst_of(v);
                    *reinterpret_cast<u32x2*>(dst) = u32x2{hi2[v][0], hi2[v][1]};
                    *reinterpret_cast<u32x2*>(dst + XPLANE) = u32x2{lo2[0], lo2[1]};
                }
            };
            auto mfma = [&](int k) {
                const int t = k >> 2, i = (k >> 1) & 1, j = k & 1;
                const f16x8 av = __builtin_bit_cast(f16x8, a[i][t == 0 ? 1 : 0]), bv = __builtin_bit_cast(f16x8, b[j][t == 1 ? 1 : 0]);
                if (DDPM_WG_ABLATE & 8) master[i][j][k & 15] += static_cast<float>(av[0]) + static_cast<float>(bv[1]);
                else if (t < 2) master[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, master[i][j], 0, 0, 0);   // lo*hi, hi*lo
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[i][j], 0, 0, 0);               // hi*hi
            };
            if (!(DDPM_WG_ABLATE & 16)) {
                reads(1, 0);
                reads(0, 1);
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        a[q][pl] = __builtin_bit_cast(s16x8, areg[set][q]);
                        b[q][pl] = __builtin_bit_cast(s16x8, breg[set][pl]);
                    }
            }
            if (!(DDPM_WG_ABLATE & 2)) load_global(set);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                if (!idle_wave) mfma(k);
                if (k < 8 && !(DDPM_WG_ABLATE & 1)) stage(k);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // split state of the four float4 of step + 1 (two of each operand)
            unsigned u[4][4], m[4][4], l[4][4];
            float r1[4][4];
            auto stage = [&](int st) {
                const int v = st / 3, ph = st % 3;
                if (ph == 0) {
                    const f32x4 x = src(v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        u[v][e] = __float_as_uint(x[e]);
                        r1[v][e] = x[e] - __uint_as_float(u[v][e] & 0xffff0000u);
                    }
                } else if (ph == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        m[v][e] = __float_as_uint(r1[v][e]);
                        l[v][e] = __float_as_uint(r1[v][e] - __uint_as_float(m[v][e] & 0xffff0000u));
                    }
                } else {
                    u32x2 hi, mid, lo;
                    hi[0] = __builtin_amdgcn_perm(u[v][1], u[v][0], 0x07060302u);
                    hi[1] = __builtin_amdgcn_perm(u[v][3], u[v][2], 0x07060302u);
                    mid[0] = __builtin_amdgcn_perm(m[v][1], m[v][0], 0x07060302u);
                    mid[1] = __builtin_amdgcn_perm(m[v][3], m[v][2], 0x07060302u);
                    lo[0] = __builtin_amdgcn_perm(l[v][1], l[v][0], 0x07060302u);
                    lo[1] = __builtin_amdgcn_perm(l[v][3], l[v][2], 0x07060302u);
                    unsigned char* dst = dst_of(v);
                    *reinterpret_cast<u32x2*>(dst) = hi;
                    *reinterpret_cast<u32x2*>(dst + XPLANE) = mid;
                    *reinterpret_cast<u32x2*>(dst + 2 * XPLANE) = lo;
                }
            };
            auto mfma = [&](int k) {
                const int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
                const int t = k >> 2, i = (k >> 1) & 1, j = k & 1;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][pa[t]]), __builtin_bit_cast(bf16x8, b[j][pb[t]]), acc[i][j], 0, 0, 0);
            };
            reads(2, 0);
            reads(0, 2);
            reads(1, 1);
            if (!(DDPM_WG_ABLATE & 2)) load_global(set);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                if (idle_wave) {
                } else if (!(DDPM_WG_ABLATE & 8) || p.taps == 77) mfma(k);
                else acc[(k >> 1) & 1][k & 1][k & 15] += static_cast<float>(a[(k >> 1) & 1][k % 3][0]) + static_cast<float>(b[k & 1][k % 3][1]);
                if ((k & 1) && (!(DDPM_WG_ABLATE & 1) || p.taps == 77)) stage(k >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(DDPM_WG_ABLATE & 4)) __syncthreads();
    };

    const int steps = (k_end - k_begin + XKP - 1) / XKP;
    if (steps > 0) {
        load_global(0);
        store_lds(0, 0);
        load_global(1);
    }
    __syncthreads();
    for (int s = 0; s < steps;) {
        const int fold_end = min(steps, s + XFOLD);
        for (; s < fold_end; s += 2) {   // an odd last step is followed by one of zeros
            step(0, s);
            step(1, s + 1);
        }
        if constexpr (!F16) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    master[i][j] += acc[i][j];
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                }
        }
    }
    if constexpr (F16) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) master[i][j] = acc[i][j] + master[i][j] * kLoInv;
    }

    if constexpr (F16 && WATCH) ddpm::conv::raise_range_flag(p.range_flag, amax);
    if (wave_tap >= p.taps) return;
    float* outp = p.out + static_cast<int64_t>(wave_tap) * p.m_total * p.n_total;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = (PAIR == 1 ? 0 : n0 + wn * 64) + j * 32 + r;
        if (n >= p.n_total) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (PAIR == 2 ? 0 : m0 + wm * 64) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < p.m_total && (!(DDPM_WG_ABLATE & 32) || master[i][j][e] == 1.2345f)) atomicAdd(outp + static_cast<int64_t>(m) * p.n_total + n, master[i][j][e]);
            }
    }
}

template <int MODE, int PAIR, bool F16>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3_kernel(const WgradParams p) {
    constexpr int NPL = F16 ? 2 : 3;
    __shared__ __attribute__((aligned(16))) unsigned char sA[2 * NPL * XPLANE];
    __shared__ __attribute__((aligned(16))) unsigned char sB[2 * NPL * XPLANE];
    // Workgroups are handed to the 8 XCDs round-robin and each XCD has its own L2: renumber so that an
    // XCD runs CONSECUTIVE work items, i.e. the taps and channel tiles that read the same pixels share
    // one L2 instead of each fetching its own copy over the fabric.
    const int id = static_cast<int>((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    if (id >= p.work) return;
    bool watch = false;
    if constexpr (F16) {
        // A (dY) is loaded unshifted and B (X) shifted by the tap, except with PAIR = 2 (the other way round) and for the
        // transposed convolution (every tap gathers its own quarter of dY, X is unshifted). The unshifted operand is
        // checked by the first tap group, the shifted one by the centre tap (whose window is the chunk itself) or, where
        // taps share a tile, by all of them; each by the first channel tile of the other operand only.
        const int nt = id % p.n_tiles, mt = (id / p.n_tiles) % p.m_tiles, tg = (id / (p.n_tiles * p.m_tiles)) % p.tap_groups;
        bool wa, wb;
        if (MODE == WG_CONVT || PAIR == 2) {
            wa = nt == 0;
            wb = tg == 0 && mt == 0;
        } else if (PAIR == 1) {
            wa = tg == 0 && nt == 0;
            wb = mt == 0;
        } else {
            wa = tg == 0 && nt == 0;
            wb = tg == p.taps / 2 && mt == 0;
        }
        watch = wa || wb;
    }
    if (watch) conv_wgrad_x3_body<MODE, PAIR, F16, F16>(p, id, sA, sB);
    else conv_wgrad_x3_body<MODE, PAIR, F16, false>(p, id, sA, sB);
}

// The split-K partial sums meet in `out` through atomic adds, so it starts at zero. A plain kernel of
// 16-byte stores: the runtime's own fill (hipMemsetAsync) took 70 us on average for these 0.1-9 MB
// buffers when it shared the chip with the data-gradient kernels of the other stream.
__global__ __launch_bounds__(256) void clear_kernel(float* __restrict__ out, int64_t n) {
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    const int64_t n4 = n >> 2;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<int64_t>(gridDim.x) * 256)
        reinterpret_cast<f32x4*>(out)[i] = z;
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[(n4 << 2) + threadIdx.x] = 0.f;
}

int clear_accumulator(float* out, int64_t n, hipStream_t s) {
    if (!ddpm::aligned16(out)) {
        if (hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * static_cast<size_t>(n), s); e != hipSuccess) {
            ddpm::set_error("hipMemsetAsync: %s", hipGetErrorString(e));
            return DDPM_E_LAUNCH;
        }
        return 0;
    }
    const int64_t want = (n / 4 + 1023) / 1024;   // four stores per thread
    clear_kernel<<<static_cast<unsigned>(want < 1 ? 1 : (want > 2048 ? 2048 : want)), 256, 0, s>>>(out, n);
    return 0;
}

template <int MODE>
int launch_wgrad(WgradParams& p, hipStream_t s) {
    // 64 channels on one side of a 3x3 layer: two taps share a tile (see the kernel); 64 (or fewer) on both sides:
    // the same with half of the tile's rows empty, still ahead of the fp32 kernel's 64 x 64 tiles
    static const bool pairs_on = [] {   // DDPM_WGRAD_PAIR=0: diagnostic switch for A/B measurements
        const char* e = getenv("DDPM_WGRAD_PAIR");
        return !(e && strcmp(e, "0") == 0);
    }();
    const bool small_pair = MODE == WG_CONV3 && ddpm::split_math() && pairs_on && p.m_total <= 64 && p.n_total <= 64 && p.m_total > 32 &&
                            p.n_total > 32;
    const bool big = p.m_total > 64 || p.n_total > 64 || small_pair;
    const int bt = big ? 128 : 64;
    p.m_tiles = (p.m_total + bt - 1) / bt;
    p.n_tiles = (p.n_total + bt - 1) / bt;
    const int pair = (MODE == WG_CONV3 && big && ddpm::split_math() && pairs_on) ? (p.n_total <= 64 ? 1 : (p.m_total <= 64 ? 2 : 0)) : 0;
    p.tap_groups = pair ? (p.taps + 1) / 2 : p.taps;
    const int64_t tiles = static_cast<int64_t>(p.m_tiles) * p.n_tiles * p.tap_groups;
    // Pixel splits: fill the chip with whole rounds of workgroups. The 128-wide tile fits 2
    // workgroups per CU (64 KB of LDS each), the 64-wide one 4: aim at two rounds of the resident
    // capacity and never exceed it by a few stragglers, which would cost a whole extra round.
    const int64_t slots = 256 * (big ? 2 : 4);
    static const int rounds = [] {   // diagnostic override of the number of resident rounds the splits aim at
        const char* e = getenv("DDPM_WGRAD_ROUNDS");
        return e ? atoi(e) : 2;
    }();
    int64_t splits = (rounds * slots) / tiles;
    const int64_t max_splits = (p.k_total + 255) / 256;      // at least 8 K-steps per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int64_t chunk = (p.k_total + splits - 1) / splits;
    chunk = (chunk + BKP - 1) / BKP * BKP;   // a multiple of both kernels' K-steps (32 and 16 pixels)
    splits = (p.k_total + chunk - 1) / chunk;
    p.k_chunk = static_cast<int>(chunk);
    p.splits = static_cast<int>(splits);
    if (tiles * splits > 0x7ffffff0ll) return ddpm::fail_invalid("wgrad grid size");
    p.work = static_cast<int>(tiles * splits);
    const int64_t grid = (tiles * splits + 7) / 8 * 8;
    if (int rc = clear_accumulator(p.out, static_cast<int64_t>(p.taps) * p.m_total * p.n_total, s)) return rc;
    const bool f16 = ddpm::f16_math();
    p.range_flag = f16 ? ddpm::range_flag() : nullptr;
    if (f16 && !p.range_flag) return DDPM_E_LAUNCH;
    if constexpr (MODE == WG_CONV3) {
        if (pair == 1) {
            if (f16) conv_wgrad_x3_kernel<MODE, 1, true><<<static_cast<unsigned>(grid), 256, 0, s>>>(p);
            else conv_wgrad_x3_kernel<MODE, 1, false><<<static_cast<unsigned>(grid), 256, 0, s>>>(p);
            return ddpm::check_launch("conv_wgrad_x3_kernel");
        }
        if (pair == 2) {
            if (f16) conv_wgrad_x3_kernel<MODE, 2, true><<<static_cast<unsigned>(grid), 256, 0, s>>>(p);
            else conv_wgrad_x3_kernel<MODE, 2, false><<<static_cast<unsigned>(grid), 256, 0, s>>>(p);
            return ddpm::check_launch("conv_wgrad_x3_kernel");
        }
    }
    if (big && f16)
        conv_wgrad_x3_kernel<MODE, 0, true><<<static_cast<unsigned>(grid), 256, 0, s>>>(p);
    else if (big && ddpm::split_math())
        conv_wgrad_x3_kernel<MODE, 0, false><<<static_cast<unsigned>(grid), 256, 0, s>>>(p);
    else if (big)
        conv_wgrad_kernel<128, MODE><<<static_cast<unsigned>(grid), 256, 0, s>>>(p);
    else
        conv_wgrad_kernel<64, MODE><<<static_cast<unsigned>(grid), 256, 0, s>>>(p);
    return ddpm::check_launch("conv_wgrad_kernel");
}

// packed [taps][m][n] -> torch parameter layouts
__global__ void unpack_conv_kernel(const float* __restrict__ packed, int taps, int m, int n, float* __restrict__ out) {
    // out[m][n][tap]  (Conv2d weight [cout][cin][kh][kw])
    const int64_t total = static_cast<int64_t>(taps) * m * n;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int tap = static_cast<int>(i % taps);
        const int64_t mn = i / taps;
        out[i] = packed[static_cast<int64_t>(tap) * m * n + mn];
    }
}

__global__ void unpack_convT_kernel(const float* __restrict__ packed, int m /*cout*/, int n /*cin*/, float* __restrict__ out) {
    // out[ci][co][q]  (ConvTranspose2d weight [cin][cout][2][2]) from packed[q][co][ci]
    const int64_t total = static_cast<int64_t>(4) * m * n;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int q = static_cast<int>(i % 4);
        const int co = static_cast<int>((i / 4) % m);
        const int ci = static_cast<int>(i / (4ll * m));
        out[i] = packed[(static_cast<int64_t>(q) * m + co) * n + ci];
    }
}

int validate(const float* dy, int lda, int m, const float* x, int ldb, int n, float* packed, float* grad) {
    DDPM_REQUIRE(dy && x && packed && grad && m > 0 && n > 0);
    DDPM_REQUIRE((m & 3) == 0 && (n & 3) == 0 && (lda & 3) == 0 && (ldb & 3) == 0 && lda >= m && ldb >= n);
    DDPM_REQUIRE(ddpm::aligned16(dy) && ddpm::aligned16(x) && ddpm::aligned16(packed));
    return DDPM_OK;
}

int grid_1d(int64_t total) { return static_cast<int>((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256); }

}  // namespace

extern "C" int ddpm_conv3x3_wgrad(const float* dy, int ldy, int cout, const float* x, int ldx, int cin, int batch, int h, int w,
                                  float* packed, float* grad_w, void* stream) {
    if (int rc = validate(dy, ldy, cout, x, ldx, cin, packed, grad_w)) return rc;
    DDPM_REQUIRE(batch > 0 && h > 0 && w > 0 && static_cast<int64_t>(batch) * h * w < (1ll << 31));
    WgradParams p{};
    p.a = dy; p.b = x; p.out = packed;
    p.batch = batch; p.h = h; p.w = w; p.a_h = h; p.a_w = w;
    p.m_total = cout; p.lda = ldy; p.n_total = cin; p.ldb = ldx;
    p.taps = 9; p.k_total = batch * h * w;
    if (int rc = launch_wgrad<WG_CONV3>(p, as_stream(stream))) return rc;
    unpack_conv_kernel<<<grid_1d(9ll * cout * cin), 256, 0, as_stream(stream)>>>(packed, 9, cout, cin, grad_w);
    return ddpm::check_launch("unpack_conv_kernel");
}

extern "C" int ddpm_conv1x1_wgrad(const float* dy, int ldy, int cout, const float* x, int ldx, int cin, int64_t rows, float* grad_w,
                                  void* stream) {
    if (int rc = validate(dy, ldy, cout, x, ldx, cin, grad_w, grad_w)) return rc;
    DDPM_REQUIRE(rows > 0 && rows < (1ll << 31));
    WgradParams p{};
    p.a = dy; p.b = x; p.out = grad_w;      // [cout][cin] is already the parameter layout
    p.batch = 1; p.h = static_cast<int>(rows); p.w = 1; p.a_h = p.h; p.a_w = 1;
    p.m_total = cout; p.lda = ldy; p.n_total = cin; p.ldb = ldx;
    p.taps = 1; p.k_total = static_cast<int>(rows);
    return launch_wgrad<WG_CONV1>(p, as_stream(stream));
}

// dy: gradient w.r.t. the padded/cropped output image [batch][out_h][out_w] (channels cout, stride ldy);
// x: the transposed conv's input [batch][h][w] (cin channels). grad_w: [cin][cout][2][2].
extern "C" int ddpm_convT2x2_wgrad(const float* dy, int ldy, int cout, int out_h, int out_w, int off_y, int off_x, const float* x,
                                   int ldx, int cin, int batch, int h, int w, float* packed, float* grad_w, void* stream) {
    if (int rc = validate(dy, ldy, cout, x, ldx, cin, packed, grad_w)) return rc;
    DDPM_REQUIRE(batch > 0 && h > 0 && w > 0 && out_h > 0 && out_w > 0);
    DDPM_REQUIRE(static_cast<int64_t>(batch) * h * w < (1ll << 31) && static_cast<int64_t>(batch) * out_h * out_w < (1ll << 31));
    WgradParams p{};
    p.a = dy; p.b = x; p.out = packed;
    p.batch = batch; p.h = h; p.w = w; p.a_h = out_h; p.a_w = out_w;
    p.m_total = cout; p.lda = ldy; p.n_total = cin; p.ldb = ldx;
    p.off_y = off_y; p.off_x = off_x;
    p.taps = 4; p.k_total = batch * h * w;
    DDPM_REQUIRE(static_cast<int64_t>(batch) * out_h * out_w * ldy * 4ll < 0x7fffffffll);   // dY is gathered by 32-bit byte offsets
    if (int rc = launch_wgrad<WG_CONVT>(p, as_stream(stream))) return rc;
    unpack_convT_kernel<<<grid_1d(4ll * cout * cin), 256, 0, as_stream(stream)>>>(packed, cout, cin, grad_w);
    return ddpm::check_launch("unpack_convT_kernel");
}

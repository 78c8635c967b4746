// Implicit-GEMM convolutions on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   rows    (M) = input pixels (b, y, x) of a window of the input image
//   columns (N) = output channels (x4 sub-positions for the 2x2 transposed conv)
//   depth   (K) = taps x input channels, walked in steps of 32 channels of one tap
//
// A 256-thread workgroup (4 waves) owns a BM x BN output tile. Each K-step stages a BM x 32 tile
// of (shifted, zero-padded) activations and a BN x 32 tile of weights through LDS: global loads
// for step s+1 are issued into registers before the MFMAs of step s and written to the other LDS
// buffer after them, so one barrier per step suffices. Rows of both tiles are padded to 36 floats,
// which makes the 16-byte fragment reads (lane = row) conflict-free.
//
// Fragment mapping: lane l = (r = l & 31, h = l >> 5) reads four consecutive k at 4h..4h+3 of an
// 8-wide chunk; MFMA number e of the chunk multiplies element e of the A and B fragments, i.e.
// k = e (lanes h=0) and k = 4+e (lanes h=1). A and B use the same permutation, so the sum over
// the chunk is complete after four MFMAs. Accumulator: column = lane & 31,
// row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
#include "common.h"

using ddpm::as_stream;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MODE_CONV3 = 0;  // 3x3, pad 1
constexpr int MODE_CONV1 = 1;  // 1x1
constexpr int MODE_CONVT = 2;  // 2x2 stride-2 transposed: GEMM + scatter
constexpr int MODE_CONVT_DGRAD = 3;  // data gradient of MODE_CONVT: 4 taps gathered from the (out_h x out_w) image

constexpr int BK = 32;
constexpr int LDS_LD = 36;  // padded row length in floats

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
    int m_tiles, n_tiles;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == DDPM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DDPM_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}

template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const GemmParams p) {
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int AR = BM / 32;  // A rows loaded per thread
    constexpr int BR = BN / 32;  // B rows loaded per thread
    static_assert(WM * WN == 4, "4 waves per workgroup");

    __shared__ __attribute__((aligned(16))) float smem[2][(BM + BN) * LDS_LD];
    __shared__ int4 rowinfo[BM];  // (pixel index or -1, b, y, x) of every tile row

    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, so give each XCD a
    // contiguous run of tiles (neighbouring tiles share activation rows in that XCD's L2).
    const int nwg = p.m_tiles * p.n_tiles;
    int id = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int n_tile = id % p.n_tiles;
    const int m_tile = id / p.n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int lrow = tid >> 3, lc4 = tid & 7;

    if (tid < BM) {
        const int m = m0 + tid;
        int4 info = make_int4(-1, 0, 0, 0);
        if (m < p.m_total) {
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

    int4 ri[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) ri[i] = rowinfo[lrow + 32 * i];

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 areg[AR], breg[BR];
    const int steps = p.taps * p.kc;

    auto load_global = [&](int tap, int kc) {
        int dy = 0, dx = 0;
        if (MODE == MODE_CONV3) {
            dy = tap / 3 - 1;
            dx = tap - (tap / 3) * 3 - 1;
        }
        const int c = kc * BK + lc4 * 4;
        const bool c_ok = c < p.ctot;
        const bool first = c < p.c1;
        const float* base = first ? p.src1 : p.src2;
        const int ld = first ? p.ld1 : p.ld2;
        const int cc = first ? c : c - p.c1;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (MODE == MODE_CONVT_DGRAD) {
                // tap = a*2 + b: the output pixel (2y+a+off_y, 2x+b+off_x) this input pixel fed
                const int oy = 2 * ri[i].z + (tap >> 1) + p.off_y, ox = 2 * ri[i].w + (tap & 1) + p.off_x;
                if (c_ok && ri[i].x >= 0 && oy >= 0 && oy < p.out_h && ox >= 0 && ox < p.out_w) {
                    const int64_t pix = (static_cast<int64_t>(ri[i].y) * p.out_h + oy) * p.out_w + ox;
                    v = *reinterpret_cast<const f32x4*>(base + pix * ld + cc);
                }
            } else {
                const int yy = ri[i].z + dy, xx = ri[i].w + dx;
                const bool ok = c_ok && ri[i].x >= 0 && yy >= 0 && yy < p.in_h && xx >= 0 && xx < p.in_w;
                if (ok) {
                    const int64_t pix = static_cast<int64_t>(ri[i].x) + dy * p.in_w + dx;
                    v = *reinterpret_cast<const f32x4*>(base + pix * ld + cc);
                }
            }
            areg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int n = n0 + lrow + 32 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c_ok && n < p.n_total)
                v = *reinterpret_cast<const f32x4*>(p.w + (static_cast<int64_t>(tap) * p.n_total + n) * p.ctot + c);
            breg[i] = v;
        }
    };
    auto store_lds = [&](int buf) {
        float* sA = smem[buf];
        float* sB = smem[buf] + BM * LDS_LD;
#pragma unroll
        for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(sA + (lrow + 32 * i) * LDS_LD + lc4 * 4) = areg[i];
#pragma unroll
        for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(sB + (lrow + 32 * i) * LDS_LD + lc4 * 4) = breg[i];
    };

    int tap = 0, kc = 0;
    load_global(0, 0);
    store_lds(0);
    __syncthreads();

    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < steps;
        if (more) {
            if (++kc == p.kc) {
                kc = 0;
                ++tap;
            }
            load_global(tap, kc);
        }
        const float* sA = smem[buf] + (wm * TM * 32 + r) * LDS_LD + h * 4;
        const float* sB = smem[buf] + BM * LDS_LD + (wn * TN * 32 + r) * LDS_LD + h * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(sA + i * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(sB + j * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        if (more) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue -------------------------------------------------------------------------
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
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
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

template <int MODE>
int launch_gemm(GemmParams& p, hipStream_t s) {
    // Tile choice: the narrowest BN that wastes the fewest padded columns (128 preferred), and the
    // small 64x64 tile when 128-row tiles would leave most of the 256 CUs without a workgroup.
    const int pad128 = (p.n_total + 127) / 128 * 128, pad64 = (p.n_total + 63) / 64 * 64;
    int bm = 128, bn = pad128 <= pad64 ? 128 : 64;
    const int64_t blocks = static_cast<int64_t>((p.m_total + bm - 1) / bm) * ((p.n_total + bn - 1) / bn);
    if (blocks < 256) bm = bn = 64;
    p.m_tiles = (p.m_total + bm - 1) / bm;
    p.n_tiles = (p.n_total + bn - 1) / bn;
    const int64_t grid = static_cast<int64_t>(p.m_tiles) * p.n_tiles;
    if (grid <= 0 || grid > 0x7fffffffll) return ddpm::fail_invalid("conv grid size");
    const dim3 g(static_cast<unsigned>(grid)), b(256);
    if (bm == 128 && bn == 128)
        conv_gemm_kernel<128, 128, 2, 2, MODE><<<g, b, 0, s>>>(p);
    else if (bm == 128)
        conv_gemm_kernel<128, 64, 2, 2, MODE><<<g, b, 0, s>>>(p);
    else
        conv_gemm_kernel<64, 64, 2, 2, MODE><<<g, b, 0, s>>>(p);
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
    p.kc = (p.ctot + BK - 1) / BK;
    p.m_total = d->batch * d->in_h * d->in_w;
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

}  // namespace

extern "C" int ddpm_conv3x3_fwd(const ddpm_conv_desc* d, const float* src1, const float* src2, const float* w_packed,
                                const ddpm_epilogue* ep, float* out, void* stream) {
    if (int rc = validate_common(d, src1, src2, w_packed, ep, out)) return rc;
    GemmParams p;
    fill_common(p, d, src1, src2, w_packed, ep, out);
    p.taps = 9;
    return launch_gemm<MODE_CONV3>(p, as_stream(stream));
}

extern "C" int ddpm_conv1x1_fwd(const ddpm_conv_desc* d, const float* src1, const float* src2, const float* w_packed,
                                const ddpm_epilogue* ep, float* out, void* stream) {
    if (int rc = validate_common(d, src1, src2, w_packed, ep, out)) return rc;
    GemmParams p;
    fill_common(p, d, src1, src2, w_packed, ep, out);
    p.taps = 1;
    return launch_gemm<MODE_CONV1>(p, as_stream(stream));
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
    const int64_t total = static_cast<int64_t>(cout) * cin * 9;
    const int grid = static_cast<int>((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    pack_conv3x3_kernel<<<grid, 256, 0, as_stream(stream)>>>(w, cout, cin, flip, out);
    return ddpm::check_launch("pack_conv3x3_kernel");
}

extern "C" int ddpm_pack_convT2x2(const float* w, int cin, int cout, float* out, void* stream) {
    DDPM_REQUIRE(w && out && cout > 0 && cin > 0);
    const int64_t total = static_cast<int64_t>(cout) * cin * 4;
    const int grid = static_cast<int>((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    pack_convT_kernel<<<grid, 256, 0, as_stream(stream)>>>(w, cin, cout, out);
    return ddpm::check_launch("pack_convT_kernel");
}

extern "C" int ddpm_pack_convT2x2_dgrad(const float* w, int cin, int cout, float* out, void* stream) {
    DDPM_REQUIRE(w && out && cout > 0 && cin > 0);
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
    return launch_gemm<MODE_CONVT_DGRAD>(p, as_stream(stream));
}

// Small backward kernels around the GEMM-shaped ones: max-pool/skip merge, the two edge
// convolutions, windowed column sums (transposed-conv bias), time/label embedding Linears.
// Reductions over pixels accumulate in float64 per workgroup and are finished in a fixed order.
#include "common.h"

using ddpm::as_stream;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 512;

struct Lanes {
    int c4n, rows_per_pass, c4, roff;
    bool active;
};
__device__ __forceinline__ Lanes lanes_for(int c) {
    Lanes l;
    l.c4n = c >> 2;
    l.rows_per_pass = kThreads / l.c4n;
    if (l.rows_per_pass < 1) l.rows_per_pass = 1;
    l.c4 = threadIdx.x % l.c4n;
    l.roff = threadIdx.x / l.c4n;
    l.active = l.roff < l.rows_per_pass;
    return l;
}
__device__ __forceinline__ void block_reduce8(double v[8], const Lanes& l, double* sh) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[threadIdx.x * 8 + e] = l.active ? v[e] : 0.0;
    __syncthreads();
    if (l.roff == 0) {
        for (int r = 1; r < l.rows_per_pass; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += sh[(r * l.c4n + l.c4) * 8 + e];
    }
}


// Deterministic finishing sum: a 256-thread workgroup finishes 16 consecutive outputs; thread
// (li = tid & 15, sl = tid >> 4) adds every 16th partial, the 16 slices are then added in order.
// Returns the total in the threads with sl == 0.
__device__ __forceinline__ double sliced_sum(const double* __restrict__ partial, int nblk, int64_t stride, int64_t offset, bool valid,
                                             double* sh /* [256] */) {
    const int li = threadIdx.x & 15, sl = threadIdx.x >> 4;
    double acc = 0.0;
    if (valid)
        for (int b = sl; b < nblk; b += 16) acc += partial[static_cast<int64_t>(b) * stride + offset];
    __syncthreads();
    sh[sl * 16 + li] = acc;
    __syncthreads();
    double tot = 0.0;
    if (sl == 0)
        for (int k = 0; k < 16; ++k) tot += sh[k * 16 + li];
    return tot;
}

int blocks_for(int64_t rows, int c) {
    const int rpp = kThreads / (c / 4) > 0 ? kThreads / (c / 4) : 1;
    int64_t nb = (rows + static_cast<int64_t>(rpp) * 8 - 1) / (static_cast<int64_t>(rpp) * 8);
    if (nb < 1) nb = 1;
    if (nb > kMaxBlocks) nb = kMaxBlocks;
    return static_cast<int>(nb);
}

// ---- max-pool + skip backward: dx = dskip + scatter(dpooled at the recorded argmax) ----------
__global__ __launch_bounds__(kThreads) void pool_skip_bwd_kernel(const float* __restrict__ dskip, int lds_, const float* __restrict__ dpooled,
                                                                  int ldp, const uint8_t* __restrict__ argmax, float* __restrict__ dx,
                                                                  int lddx, int batch, int h, int w, int c) {
    const int c4n = c >> 2;
    const int ch = (h + 1) >> 1, cw = (w + 1) >> 1, ph = h >> 1, pw = w >> 1;
    const int64_t total = static_cast<int64_t>(batch) * ch * cw * c4n;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int c4 = static_cast<int>(idx % c4n);
        int64_t r = idx / c4n;
        const int cx = static_cast<int>(r % cw);
        r /= cw;
        const int cy = static_cast<int>(r % ch);
        const int b = static_cast<int>(r / ch);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        uchar4 am = make_uchar4(255, 255, 255, 255);
        if (cy < ph && cx < pw) {
            const int64_t pp = (static_cast<int64_t>(b) * ph + cy) * pw + cx;
            g = *reinterpret_cast<const float4*>(dpooled + pp * ldp + c4 * 4);
            am = *reinterpret_cast<const uchar4*>(argmax + pp * c + c4 * 4);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int y = cy * 2 + (q >> 1), xx = cx * 2 + (q & 1);
            if (y < h && xx < w) {
                const int64_t pix = (static_cast<int64_t>(b) * h + y) * w + xx;
                float4 d = *reinterpret_cast<const float4*>(dskip + pix * lds_ + c4 * 4);
                if (am.x == q) d.x += g.x;
                if (am.y == q) d.y += g.y;
                if (am.z == q) d.z += g.z;
                if (am.w == q) d.w += g.w;
                *reinterpret_cast<float4*>(dx + pix * lddx + c4 * 4) = d;
            }
        }
    }
}

// ---- final 1x1 conv backward ------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void final_bwd_data_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                                   float* __restrict__ dx, int lddx, int batch, int hw, int c, int cout) {
    const int c4n = c >> 2;
    const int64_t total = static_cast<int64_t>(batch) * hw * c4n;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int c4 = static_cast<int>(idx % c4n);
        const int64_t pix = idx / c4n;
        const int64_t b = pix / hw, q = pix % hw;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int co = 0; co < cout; ++co) {
            const float g = dout[(b * cout + co) * hw + q];
            const float4 wv = *reinterpret_cast<const float4*>(w + static_cast<int64_t>(co) * c + c4 * 4);
            acc.x += g * wv.x; acc.y += g * wv.y; acc.z += g * wv.z; acc.w += g * wv.w;
        }
        *reinterpret_cast<float4*>(dx + pix * lddx + c4 * 4) = acc;
    }
}

// partial[blk][0..c) = sum_r g[r]*x[r][:], partial[blk][c] = sum_r g[r], for one output channel co
__global__ __launch_bounds__(kThreads) void final_bwd_weight_partial_kernel(const float* __restrict__ dout, const float* __restrict__ x, int ld,
                                                                             int batch, int hw, int c, int cout, int co,
                                                                             double* __restrict__ partial) {
    __shared__ double sh[kThreads * 8];
    const Lanes l = lanes_for(c);
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t rows = static_cast<int64_t>(batch) * hw;
    if (l.active) {
        for (int64_t r = static_cast<int64_t>(blockIdx.x) * l.rows_per_pass + l.roff; r < rows;
             r += static_cast<int64_t>(gridDim.x) * l.rows_per_pass) {
            const int64_t b = r / hw, q = r % hw;
            const float g = dout[(b * cout + co) * hw + q];
            const float4 a = *reinterpret_cast<const float4*>(x + r * ld + l.c4 * 4);
            v[0] += static_cast<double>(g) * a.x; v[1] += static_cast<double>(g) * a.y;
            v[2] += static_cast<double>(g) * a.z; v[3] += static_cast<double>(g) * a.w;
            if (l.c4 == 0) v[4] += g;
        }
    }
    block_reduce8(v, l, sh);
    if (l.roff == 0 && l.active) {
#pragma unroll
        for (int e = 0; e < 4; ++e) partial[static_cast<int64_t>(blockIdx.x) * (c + 4) + l.c4 * 4 + e] = v[e];
        if (l.c4 == 0) partial[static_cast<int64_t>(blockIdx.x) * (c + 4) + c] = v[4];
    }
}

__global__ __launch_bounds__(kThreads) void final_bwd_weight_final_kernel(const double* __restrict__ partial, int nblk, int c, int co,
                                                                           float* __restrict__ dw, float* __restrict__ db) {
    __shared__ double sh[kThreads];
    const int i = blockIdx.x * 16 + (threadIdx.x & 15);        // i == c is the bias slot
    const bool valid = i <= c;
    const double s = sliced_sum(partial, nblk, c + 4, i, valid, sh);
    if ((threadIdx.x >> 4) != 0 || !valid) return;
    if (i < c) dw[static_cast<int64_t>(co) * c + i] = static_cast<float>(s);
    else db[co] = static_cast<float>(s);
}

// ---- first 3x3 conv weight gradient (NCHW image with few channels) ---------------------------
// One launch per input channel ci; partial[blk][tap][c].
__global__ __launch_bounds__(kThreads) void first_wgrad_partial_kernel(const float* __restrict__ dy, int ld, const float* __restrict__ x,
                                                                        int batch, int cin, int ci, int h, int w, int c,
                                                                        double* __restrict__ partial) {
    __shared__ double sh[kThreads * 8];
    const Lanes l = lanes_for(c);
    double v[36];
#pragma unroll
    for (int e = 0; e < 36; ++e) v[e] = 0.0;
    const unsigned rows = static_cast<unsigned>(batch) * h * w;       // below 2^31: checked by the caller (32-bit divisions below)
    if (l.active) {
        for (unsigned r = blockIdx.x * l.rows_per_pass + l.roff; r < rows; r += gridDim.x * l.rows_per_pass) {
            const unsigned row = r / static_cast<unsigned>(w);
            const int xx = static_cast<int>(r - row * w);
            const unsigned b = row / static_cast<unsigned>(h);
            const int yy = static_cast<int>(row - b * h);
            const float* plane = x + (static_cast<int64_t>(b) * cin + ci) * h * w;
            const float4 g = *reinterpret_cast<const float4*>(dy + static_cast<int64_t>(r) * ld + l.c4 * 4);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int y2 = yy + tap / 3 - 1, x2 = xx + tap % 3 - 1;
                const float xv = (y2 >= 0 && y2 < h && x2 >= 0 && x2 < w) ? plane[y2 * w + x2] : 0.f;
                v[tap * 4 + 0] += static_cast<double>(g.x) * xv; v[tap * 4 + 1] += static_cast<double>(g.y) * xv;
                v[tap * 4 + 2] += static_cast<double>(g.z) * xv; v[tap * 4 + 3] += static_cast<double>(g.w) * xv;
            }
        }
    }
#pragma unroll
    for (int pair = 0; pair < 5; ++pair) {           // taps (0,1) (2,3) (4,5) (6,7) (8,-)
        double t8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t8[e] = (pair * 8 + e < 36) ? v[pair * 8 + e] : 0.0;
        block_reduce8(t8, l, sh);
        if (l.roff == 0 && l.active) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int tap = pair * 2 + (e >> 2);
                if (tap < 9) partial[(static_cast<int64_t>(blockIdx.x) * 9 + tap) * c + l.c4 * 4 + (e & 3)] = t8[e];
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void first_wgrad_final_kernel(const double* __restrict__ partial, int nblk, int cin, int ci, int c,
                                                                      float* __restrict__ dw) {
    // dw[co][ci][tap]
    __shared__ double sh[kThreads];
    const int i = blockIdx.x * 16 + (threadIdx.x & 15);
    const bool valid = i < 9 * c;
    const double s = sliced_sum(partial, nblk, 9 * static_cast<int64_t>(c), i, valid, sh);
    if ((threadIdx.x >> 4) != 0 || !valid) return;
    const int tap = i / c, co = i % c;
    dw[(static_cast<int64_t>(co) * cin + ci) * 9 + tap] = static_cast<float>(s);
}

// ---- windowed column sum: out[c] = sum over batch and a pixel window of x ---------------------
__global__ __launch_bounds__(kThreads) void colsum_window_partial_kernel(const float* __restrict__ x, int ld, int batch, int img_h, int img_w,
                                                                          int y0, int x0, int win_h, int win_w, int c,
                                                                          double* __restrict__ partial) {
    __shared__ double sh[kThreads * 8];
    const Lanes l = lanes_for(c);
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t rows = static_cast<int64_t>(batch) * win_h * win_w;
    if (l.active) {
        for (int64_t r = static_cast<int64_t>(blockIdx.x) * l.rows_per_pass + l.roff; r < rows;
             r += static_cast<int64_t>(gridDim.x) * l.rows_per_pass) {
            const int wx = static_cast<int>(r % win_w);
            const int wy = static_cast<int>((r / win_w) % win_h);
            const int64_t b = r / (static_cast<int64_t>(win_w) * win_h);
            const int64_t pix = (b * img_h + y0 + wy) * img_w + x0 + wx;
            const float4 a = *reinterpret_cast<const float4*>(x + pix * ld + l.c4 * 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        }
    }
    block_reduce8(v, l, sh);
    if (l.roff == 0 && l.active) {
#pragma unroll
        for (int e = 0; e < 4; ++e) partial[static_cast<int64_t>(blockIdx.x) * c + l.c4 * 4 + e] = v[e];
    }
}

__global__ __launch_bounds__(kThreads) void vec_final_kernel(const double* __restrict__ partial, int nblk, int c, float* __restrict__ out) {
    __shared__ double sh[kThreads];
    const int i = blockIdx.x * 16 + (threadIdx.x & 15);
    const bool valid = i < c;
    const double s = sliced_sum(partial, nblk, c, i, valid, sh);
    if ((threadIdx.x >> 4) == 0 && valid) out[i] = static_cast<float>(s);
}

// ---- time / label embedding Linears backward -------------------------------------------------
struct EmbedBwd {
    int batch, emb_dim, n_labels, n_classes;
    const float* emb;          // [B][E] sinusoidal embedding saved by the forward
    const int64_t* y;          // labels or null
    const float* label_table;  // [classes][E]
    int ch[4];
    const float* dtemb[4];     // [B][Ck]
    const float* lw[4];        // label Linear weights (forward values)
    float* dw[4];              // [Ck][E]
    float* db[4];              // [Ck]
    float* dlw[4];
    float* dlb[4];
    float* dtable;             // [classes][E], zeroed by the caller
};

// grid (ceil(max Ck*E / 256), 4): one thread per weight element
__global__ __launch_bounds__(kThreads) void embed_bwd_weight_kernel(const EmbedBwd a) {
    const int k = blockIdx.y;
    const int E = a.emb_dim, C = a.ch[k];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * E) return;
    const int co = i / E, e = i % E;
    const float* g = a.dtemb[k];
    double s = 0.0, sl = 0.0, sb = 0.0;
    // eight samples' operands are requested before any is used (the loop is bound by load latency);
    // the sums still run over the batch in order
    constexpr int U = 8;
    for (int b0 = 0; b0 < a.batch; b0 += U) {
        float gv[U], ev[U], lv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + u < a.batch ? b0 + u : a.batch - 1;
            gv[u] = g[static_cast<int64_t>(b) * C + co];
            ev[u] = a.emb[static_cast<int64_t>(b) * E + e];
            lv[u] = 0.f;
            if (a.n_labels > 0) {
                const int64_t cls = a.y[a.n_labels == 1 ? 0 : b];
                if (cls >= 0 && cls < a.n_classes) lv[u] = a.label_table[cls * E + e];   // out-of-range labels contribute nothing
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (b0 + u >= a.batch) break;
            s += static_cast<double>(gv[u]) * ev[u];
            if (a.n_labels > 0) sl += static_cast<double>(gv[u]) * lv[u];
            sb += gv[u];
        }
    }
    a.dw[k][i] = static_cast<float>(s);
    if (e == 0) a.db[k][co] = static_cast<float>(sb);
    if (a.n_labels > 0) {
        a.dlw[k][i] = static_cast<float>(sl);
        if (e == 0) a.dlb[k][co] = static_cast<float>(sb);
    }
}

// grid (B): d label_embedding row = sum_k sum_co dtemb_k[b][co] * LW_k[co][:], scattered by class
__global__ __launch_bounds__(kThreads) void embed_bwd_table_kernel(const EmbedBwd a) {
    const int b = blockIdx.x, E = a.emb_dim;
    const int64_t cls = a.y[a.n_labels == 1 ? 0 : b];
    if (cls < 0 || cls >= a.n_classes) return;             // no row to add to
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < 4; ++k) {
            const int C = a.ch[k];
            const float* g = a.dtemb[k] + static_cast<int64_t>(b) * C;
            const float* w = a.lw[k];
            for (int co = 0; co < C; ++co) s += g[co] * w[static_cast<int64_t>(co) * E + e];
        }
        atomicAdd(a.dtable + cls * E + e, s);
    }
}

}  // namespace

extern "C" int ddpm_pool_skip_bwd(const float* dskip, int ldds, const float* dpooled, int ldp, const uint8_t* argmax, float* dx, int lddx,
                                  int batch, int h, int w, int c, void* stream) {
    DDPM_REQUIRE(dskip && dpooled && argmax && dx && batch > 0 && h > 1 && w > 1 && c > 0 && (c & 3) == 0);
    DDPM_REQUIRE((ldds & 3) == 0 && (ldp & 3) == 0 && (lddx & 3) == 0 && ldds >= c && ldp >= c && lddx >= c);
    DDPM_REQUIRE(ddpm::aligned16(dskip) && ddpm::aligned16(dpooled) && ddpm::aligned16(dx) && (reinterpret_cast<uintptr_t>(argmax) & 3) == 0);
    const int64_t total = static_cast<int64_t>(batch) * ((h + 1) / 2) * ((w + 1) / 2) * (c / 4);
    int64_t grid = (total + kThreads - 1) / kThreads;
    if (grid > 256 * 32) grid = 256 * 32;
    pool_skip_bwd_kernel<<<static_cast<int>(grid), kThreads, 0, as_stream(stream)>>>(dskip, ldds, dpooled, ldp, argmax, dx, lddx, batch, h,
                                                                                   w, c);
    return ddpm::check_launch("pool_skip_bwd_kernel");
}

extern "C" int ddpm_conv1x1_final_bwd(const float* dout_nchw, const float* x, int ldx, const float* w, double* workspace, float* dx, int lddx,
                                      float* dw, float* db, int batch, int h, int w_, int c, int cout, void* stream) {
    DDPM_REQUIRE(dout_nchw && x && w && workspace && dx && dw && db && batch > 0 && h > 0 && w_ > 0 && c > 0 && cout > 0);
    DDPM_REQUIRE((c & 3) == 0 && c <= 1024 && (ldx & 3) == 0 && (lddx & 3) == 0 && ldx >= c && lddx >= c);
    DDPM_REQUIRE(ddpm::aligned16(x) && ddpm::aligned16(dx) && ddpm::aligned16(w));
    hipStream_t s = as_stream(stream);
    const int hw = h * w_;
    const int64_t total = static_cast<int64_t>(batch) * hw * (c / 4);
    int64_t grid = (total + kThreads - 1) / kThreads;
    if (grid > 256 * 32) grid = 256 * 32;
    final_bwd_data_kernel<<<static_cast<int>(grid), kThreads, 0, s>>>(dout_nchw, w, dx, lddx, batch, hw, c, cout);
    const int nb = blocks_for(static_cast<int64_t>(batch) * hw, c);
    for (int co = 0; co < cout; ++co) {
        final_bwd_weight_partial_kernel<<<nb, kThreads, 0, s>>>(dout_nchw, x, ldx, batch, hw, c, cout, co, workspace);
        final_bwd_weight_final_kernel<<<(c + 1 + 15) / 16, kThreads, 0, s>>>(workspace, nb, c, co, dw, db);
    }
    return ddpm::check_launch("conv1x1_final_bwd");
}

extern "C" int ddpm_conv3x3_first_wgrad(const float* dy, int ldy, const float* x_nchw, double* workspace, float* dw, int batch, int cin, int h,
                                        int w_, int cout, void* stream) {
    DDPM_REQUIRE(dy && x_nchw && workspace && dw && batch > 0 && cin > 0 && h > 0 && w_ > 0 && cout > 0);
    DDPM_REQUIRE((cout & 3) == 0 && cout <= 1024 && (ldy & 3) == 0 && ldy >= cout && ddpm::aligned16(dy));
    DDPM_REQUIRE(static_cast<int64_t>(batch) * h * w_ < (1ll << 31));
    hipStream_t s = as_stream(stream);
    int nb = blocks_for(static_cast<int64_t>(batch) * h * w_, cout);
    for (int ci = 0; ci < cin; ++ci) {
        first_wgrad_partial_kernel<<<nb, kThreads, 0, s>>>(dy, ldy, x_nchw, batch, cin, ci, h, w_, cout, workspace);
        first_wgrad_final_kernel<<<(9 * cout + 15) / 16, kThreads, 0, s>>>(workspace, nb, cin, ci, cout, dw);
    }
    return ddpm::check_launch("conv3x3_first_wgrad");
}

extern "C" int ddpm_colsum_window(const float* x, int ld, int batch, int img_h, int img_w, int y0, int x0, int win_h, int win_w, int c,
                                  double* workspace, float* out, void* stream) {
    DDPM_REQUIRE(x && workspace && out && batch > 0 && c > 0 && (c & 3) == 0 && c <= 1024 && (ld & 3) == 0 && ld >= c);
    DDPM_REQUIRE(y0 >= 0 && x0 >= 0 && win_h > 0 && win_w > 0 && y0 + win_h <= img_h && x0 + win_w <= img_w && ddpm::aligned16(x));
    hipStream_t s = as_stream(stream);
    const int nb = blocks_for(static_cast<int64_t>(batch) * win_h * win_w, c);
    colsum_window_partial_kernel<<<nb, kThreads, 0, s>>>(x, ld, batch, img_h, img_w, y0, x0, win_h, win_w, c, workspace);
    vec_final_kernel<<<(c + 15) / 16, kThreads, 0, s>>>(workspace, nb, c, out);
    return ddpm::check_launch("colsum_window");
}

extern "C" int ddpm_time_embed_bwd(const ddpm_embed_bwd_args* a, void* stream) {
    DDPM_REQUIRE(a && a->batch > 0 && a->emb_dim >= 4 && a->emb);
    DDPM_REQUIRE(a->n_labels == 0 || a->n_labels == 1 || a->n_labels == a->batch);
    EmbedBwd d;
    d.batch = a->batch;
    d.emb_dim = a->emb_dim / 2 * 2;
    d.n_labels = a->n_labels;
    d.n_classes = a->n_classes;
    DDPM_REQUIRE(a->n_labels == 0 || a->n_classes > 0);
    d.emb = a->emb;
    d.y = a->y;
    d.label_table = a->label_table;
    d.dtable = a->dlabel_table;
    int cmax = 0;
    for (int k = 0; k < 4; ++k) {
        DDPM_REQUIRE(a->ch[k] > 0 && a->dtemb[k] && a->dw[k] && a->db[k]);
        DDPM_REQUIRE(a->n_labels == 0 || (a->lw[k] && a->dlw[k] && a->dlb[k] && a->y && a->label_table && a->dlabel_table));
        d.ch[k] = a->ch[k];
        d.dtemb[k] = a->dtemb[k];
        d.lw[k] = a->lw[k];
        d.dw[k] = a->dw[k];
        d.db[k] = a->db[k];
        d.dlw[k] = a->dlw[k];
        d.dlb[k] = a->dlb[k];
        if (a->ch[k] > cmax) cmax = a->ch[k];
    }
    hipStream_t s = as_stream(stream);
    const int gx = (cmax * d.emb_dim + kThreads - 1) / kThreads;
    embed_bwd_weight_kernel<<<dim3(gx, 4), kThreads, 0, s>>>(d);
    if (a->n_labels > 0) embed_bwd_table_kernel<<<a->batch, kThreads, 0, s>>>(d);
    return ddpm::check_launch("time_embed_bwd");
}

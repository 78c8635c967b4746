// Train-mode BatchNorm2d fused with activation / residual / time-embedding add, forward and
// backward, on NHWC fp32 tensors. All reductions over (batch, pixels) accumulate in float64 in a
// fixed order (per-block partials, then one finishing workgroup), so results are deterministic.
//
//   forward : stats(x) -> mean, invstd (+ running statistics update)
//             out = act((x-mean)*invstd*w + b [+ residual]) [+ temb]
//   backward: dz = dout * act'(z)             z = (x-mean)*invstd*w + b [+ residual]
//             dw = sum dz*xhat, db = sum dz
//             dx = w*invstd * (dz - db/M - xhat*dw/M),   dresidual = dz
#include <mutex>

#include "common.h"

using ddpm::as_stream;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPartialBlocks = 1024;  // blocks of one per-channel reduction (4 per CU)
constexpr int kMaxColsumBlocks = 8192;    // (group, chunk) blocks of one grouped column sum

// erf by Abramowitz & Stegun 7.1.26 (absolute error 1.5e-7): rcp, four fma, one exp2. `e` returns
// exp(-x*x), which the GELU gradient needs as well.
__device__ __forceinline__ float erf_poly(float x, float& e) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float q = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    q = __builtin_fmaf(q, t, 1.421413741f);
    q = __builtin_fmaf(q, t, -0.284496736f);
    q = __builtin_fmaf(q, t, 0.254829592f);
    q *= t;
    e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
    return __builtin_copysignf(__builtin_fmaf(-q, e, 1.0f), x);
}

__device__ __forceinline__ float act_fwd(float z, int act) {
    if (act == DDPM_ACT_RELU) return z > 0.f ? z : 0.f;
    if (act == DDPM_ACT_GELU) {
        float e;
        const float hz = 0.5f * z;
        return __builtin_fmaf(hz, erf_poly(z * 0.70710678118654752440f, e), hz);
    }
    return z;
}

__device__ __forceinline__ float act_grad(float z, int act) {
    if (act == DDPM_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == DDPM_ACT_GELU) {
        float e;   // exp(-z*z/2)
        const float cdf = __builtin_fmaf(0.5f, erf_poly(z * 0.70710678118654752440f, e), 0.5f);
        return __builtin_fmaf(z * 0.39894228040143267794f, e, cdf);
    }
    return 1.f;
}

// Thread layout shared by the reduction kernels: c4 = tid % (C/4) picks 4 channels, the remaining
// bits pick a row offset; a workgroup covers rows_per_pass rows per iteration.
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

// Reduces v[0..7] (per thread) over the threads that share c4; thread with roff == 0 holds the sum.
__device__ __forceinline__ void block_reduce8(double v[8], const Lanes& l, double* sh /* [kThreads][8] */) {
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[threadIdx.x * 8 + e] = l.active ? v[e] : 0.0;
    __syncthreads();
    if (l.roff == 0) {
        for (int r = 1; r < l.rows_per_pass; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += sh[(r * l.c4n + l.c4) * 8 + e];
    }
}


// Deterministic finishing sum: a 256-thread workgroup finishes kFin consecutive outputs; thread
// (li = tid % kFin, sl = tid / kFin) adds every kSlices-th partial, the slices are then added in
// order. Few outputs per workgroup and many slices: the kernel is a chain of dependent loads, so its
// time is the number of partials per slice. Returns the total in the threads with sl == 0.
constexpr int kFin = 4;
constexpr int kSlices = kThreads / kFin;
__device__ __forceinline__ double sliced_sum(const double* __restrict__ partial, int nblk, int64_t stride, int64_t offset, bool valid,
                                             double* sh /* [256] */) {
    const int li = threadIdx.x % kFin, sl = threadIdx.x / kFin;
    double acc = 0.0;
    if (valid) {
        int b = sl;
        for (; b + 3 * kSlices < nblk; b += 4 * kSlices) {   // four loads in flight, added in block order
            const double v0 = partial[static_cast<int64_t>(b) * stride + offset];
            const double v1 = partial[static_cast<int64_t>(b + kSlices) * stride + offset];
            const double v2 = partial[static_cast<int64_t>(b + 2 * kSlices) * stride + offset];
            const double v3 = partial[static_cast<int64_t>(b + 3 * kSlices) * stride + offset];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; b < nblk; b += kSlices) acc += partial[static_cast<int64_t>(b) * stride + offset];
    }
    __syncthreads();
    sh[sl * kFin + li] = acc;
    __syncthreads();
    double tot = 0.0;
    if (sl == 0)
        for (int k = 0; k < kSlices; ++k) tot += sh[k * kFin + li];
    return tot;
}

// partial[(blk*2 + which)*C + c]
__global__ __launch_bounds__(kThreads) void bn_stats_partial_kernel(const float* __restrict__ x, int ld, int64_t rows, int c,
                                                                     double* __restrict__ partial) {
    __shared__ double sh[kThreads * 8];
    const Lanes l = lanes_for(c);
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l.active) {
        for (int64_t r = static_cast<int64_t>(blockIdx.x) * l.rows_per_pass + l.roff; r < rows;
             r += static_cast<int64_t>(gridDim.x) * l.rows_per_pass) {
            const float4 a = *reinterpret_cast<const float4*>(x + r * ld + l.c4 * 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
            v[4] += static_cast<double>(a.x) * a.x; v[5] += static_cast<double>(a.y) * a.y;
            v[6] += static_cast<double>(a.z) * a.z; v[7] += static_cast<double>(a.w) * a.w;
        }
    }
    block_reduce8(v, l, sh);
    if (l.roff == 0 && l.active) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            partial[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * c + l.c4 * 4 + e] = v[e];
            partial[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * c + l.c4 * 4 + e] = v[4 + e];
        }
    }
}

__global__ __launch_bounds__(kThreads) void bn_stats_final_kernel(const double* __restrict__ partial, int nblk, int64_t rows, int c,
                                                                   float eps, float momentum, float* __restrict__ mean,
                                                                   float* __restrict__ invstd, float* __restrict__ running_mean,
                                                                   float* __restrict__ running_var) {
    __shared__ double sh[kThreads];
    const int ch = blockIdx.x * kFin + (threadIdx.x % kFin);
    const bool valid = ch < c;
    const double s = sliced_sum(partial, nblk, 2 * static_cast<int64_t>(c), ch, valid, sh);
    const double ss = sliced_sum(partial, nblk, 2 * static_cast<int64_t>(c), c + ch, valid, sh);
    if (threadIdx.x / kFin != 0 || !valid) return;
    const double m = s / static_cast<double>(rows);
    double var = ss / static_cast<double>(rows) - m * m;
    if (var < 0.0) var = 0.0;
    mean[ch] = static_cast<float>(m);
    invstd[ch] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    if (running_mean) {
        const double unbiased = rows > 1 ? var * static_cast<double>(rows) / static_cast<double>(rows - 1) : var;
        running_mean[ch] = static_cast<float>((1.0 - momentum) * running_mean[ch] + momentum * m);
        running_var[ch] = static_cast<float>((1.0 - momentum) * running_var[ch] + momentum * unbiased);
    }
}

// (the activation is a template parameter in the three kernels below: a run-time switch per element costs these
// memory-bound loops a branch per value)
template <int act>
__global__ __launch_bounds__(kThreads) void bn_apply_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ w,
                                                                 const float* __restrict__ b, const float* __restrict__ res, int ldr,
                                                                 const float* __restrict__ temb, float* __restrict__ out,
                                                                 int ldo, int64_t rows, int hw, int c) {
    const int c4n = c >> 2;
    const int64_t total = rows * c4n;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int c4 = static_cast<int>(idx % c4n);
        const int64_t r = idx / c4n;
        const int ch = c4 * 4;
        const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + ch);
        const float4 mu = *reinterpret_cast<const float4*>(mean + ch);
        const float4 is = *reinterpret_cast<const float4*>(invstd + ch);
        const float4 wv = *reinterpret_cast<const float4*>(w + ch);
        const float4 bv = *reinterpret_cast<const float4*>(b + ch);
        float z[4] = {(xv.x - mu.x) * is.x * wv.x + bv.x, (xv.y - mu.y) * is.y * wv.y + bv.y, (xv.z - mu.z) * is.z * wv.z + bv.z,
                      (xv.w - mu.w) * is.w * wv.w + bv.w};
        if (res) {
            const float4 rv = *reinterpret_cast<const float4*>(res + r * ldr + ch);
            z[0] += rv.x; z[1] += rv.y; z[2] += rv.z; z[3] += rv.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = act_fwd(z[e], act);
        if (temb) {
            const float4 tv = *reinterpret_cast<const float4*>(temb + (r / hw) * c + ch);
            z[0] += tv.x; z[1] += tv.y; z[2] += tv.z; z[3] += tv.w;
        }
        *reinterpret_cast<float4*>(out + r * ldo + ch) = make_float4(z[0], z[1], z[2], z[3]);
    }
}

// partial[(blk*2 + which)*C + c]: which 0 = sum dz, 1 = sum dz*xhat
template <int act>
__global__ __launch_bounds__(kThreads) void bn_bwd_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dout,
                                                                   int ldd, const float* __restrict__ res, int ldr,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const float* __restrict__ w, const float* __restrict__ b,
                                                                   int64_t rows, int c, double* __restrict__ partial) {
    __shared__ double sh[kThreads * 8];
    const Lanes l = lanes_for(c);
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l.active) {
        const int ch = l.c4 * 4;
        const float4 mu = *reinterpret_cast<const float4*>(mean + ch);
        const float4 is = *reinterpret_cast<const float4*>(invstd + ch);
        const float4 wv = *reinterpret_cast<const float4*>(w + ch);
        const float4 bv = *reinterpret_cast<const float4*>(b + ch);
        const float mu_[4] = {mu.x, mu.y, mu.z, mu.w}, is_[4] = {is.x, is.y, is.z, is.w};
        const float w_[4] = {wv.x, wv.y, wv.z, wv.w}, b_[4] = {bv.x, bv.y, bv.z, bv.w};
        for (int64_t r = static_cast<int64_t>(blockIdx.x) * l.rows_per_pass + l.roff; r < rows;
             r += static_cast<int64_t>(gridDim.x) * l.rows_per_pass) {
            const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + ch);
            const float4 dv = *reinterpret_cast<const float4*>(dout + r * ldd + ch);
            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (res) rv = *reinterpret_cast<const float4*>(res + r * ldr + ch);
            const float x_[4] = {xv.x, xv.y, xv.z, xv.w}, d_[4] = {dv.x, dv.y, dv.z, dv.w}, r_[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (x_[e] - mu_[e]) * is_[e];
                const float z = xh * w_[e] + b_[e] + r_[e];
                const float dz = d_[e] * act_grad(z, act);
                v[e] += dz;
                v[4 + e] += static_cast<double>(dz) * xh;
            }
        }
    }
    block_reduce8(v, l, sh);
    if (l.roff == 0 && l.active) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            partial[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * c + l.c4 * 4 + e] = v[e];
            partial[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * c + l.c4 * 4 + e] = v[4 + e];
        }
    }
}

// dw/db: written (accumulate == 0) or added to (accumulate != 0); sums[2][C] keeps (db, dw) for the apply pass.
__global__ __launch_bounds__(kThreads) void bn_bwd_final_kernel(const double* __restrict__ partial, int nblk, int c, int accumulate,
                                                                 float* __restrict__ dw, float* __restrict__ db,
                                                                 float* __restrict__ sums) {
    __shared__ double sh[kThreads];
    const int ch = blockIdx.x * kFin + (threadIdx.x % kFin);
    const bool valid = ch < c;
    const double s = sliced_sum(partial, nblk, 2 * static_cast<int64_t>(c), ch, valid, sh);
    const double sx = sliced_sum(partial, nblk, 2 * static_cast<int64_t>(c), c + ch, valid, sh);
    if (threadIdx.x / kFin != 0 || !valid) return;
    sums[ch] = static_cast<float>(s);
    sums[c + ch] = static_cast<float>(sx);
    if (accumulate) {
        db[ch] += static_cast<float>(s);
        dw[ch] += static_cast<float>(sx);
    } else {
        db[ch] = static_cast<float>(s);
        dw[ch] = static_cast<float>(sx);
    }
}

// dx (and dresidual = dz). colsum_partial[(blk)*C + c] = sum over this block's rows of dx (the
// gradient of the conv bias that precedes the BatchNorm: zero up to rounding).
template <int act>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dout,
                                                                 int ldd, const float* __restrict__ res, int ldr,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ w, const float* __restrict__ b,
                                                                 const float* __restrict__ sums, int64_t rows, int64_t stat_rows,
                                                                 int c, float* __restrict__ dx, int lddx, float* __restrict__ dres,
                                                                 int lddr, double* __restrict__ colsum_partial) {
    __shared__ double sh[kThreads * 8];
    const Lanes l = lanes_for(c);
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l.active) {
        const int ch = l.c4 * 4;
        const float inv_m = 1.0f / static_cast<float>(stat_rows);   // rows the statistics were taken over (all ranks' rows when synchronised)
        float mu_[4], is_[4], w_[4], b_[4], kb[4], kw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mu_[e] = mean[ch + e]; is_[e] = invstd[ch + e]; w_[e] = w[ch + e]; b_[e] = b[ch + e];
            kb[e] = sums[ch + e] * inv_m;        // mean of dz
            kw[e] = sums[c + ch + e] * inv_m;    // mean of dz*xhat
        }
        for (int64_t r = static_cast<int64_t>(blockIdx.x) * l.rows_per_pass + l.roff; r < rows;
             r += static_cast<int64_t>(gridDim.x) * l.rows_per_pass) {
            const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + ch);
            const float4 dv = *reinterpret_cast<const float4*>(dout + r * ldd + ch);
            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (res) rv = *reinterpret_cast<const float4*>(res + r * ldr + ch);
            const float x_[4] = {xv.x, xv.y, xv.z, xv.w}, d_[4] = {dv.x, dv.y, dv.z, dv.w}, r_[4] = {rv.x, rv.y, rv.z, rv.w};
            float o[4], dzv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (x_[e] - mu_[e]) * is_[e];
                const float z = xh * w_[e] + b_[e] + r_[e];
                const float dz = d_[e] * act_grad(z, act);
                dzv[e] = dz;
                o[e] = (w_[e] * is_[e]) * (dz - kb[e] - xh * kw[e]);
                v[e] += o[e];
            }
            *reinterpret_cast<float4*>(dx + r * lddx + ch) = make_float4(o[0], o[1], o[2], o[3]);
            if (dres) *reinterpret_cast<float4*>(dres + r * lddr + ch) = make_float4(dzv[0], dzv[1], dzv[2], dzv[3]);
        }
    }
    if (colsum_partial) {
        block_reduce8(v, l, sh);
        if (l.roff == 0 && l.active) {
#pragma unroll
            for (int e = 0; e < 4; ++e) colsum_partial[static_cast<int64_t>(blockIdx.x) * c + l.c4 * 4 + e] = v[e];
        }
    }
}

// The two-quantity reductions above, finished to RAW float64 sums instead of statistics: sums[0..c) and sums[c..2c)
// (sum x | sum x^2, or sum dz | sum dz*xhat). What a data-parallel job all-reduces across ranks before finalising.
__global__ __launch_bounds__(kThreads) void pair_sums_final_kernel(const double* __restrict__ partial, int nblk, int c,
                                                                    double* __restrict__ sums) {
    __shared__ double sh[kThreads];
    const int ch = blockIdx.x * kFin + (threadIdx.x % kFin);
    const bool valid = ch < c;
    const double s = sliced_sum(partial, nblk, 2 * static_cast<int64_t>(c), ch, valid, sh);
    const double ss = sliced_sum(partial, nblk, 2 * static_cast<int64_t>(c), c + ch, valid, sh);
    if (threadIdx.x / kFin != 0 || !valid) return;
    sums[ch] = s;
    sums[c + ch] = ss;
}

// mean / invstd / running statistics from (possibly all-reduced) sums over total_rows rows: the arithmetic of
// bn_stats_final_kernel, one thread per channel.
__global__ __launch_bounds__(kThreads) void bn_finalize_kernel(const double* __restrict__ sums, int64_t rows, int c, float eps, float momentum,
                                                                float* __restrict__ mean, float* __restrict__ invstd,
                                                                float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const double m = sums[ch] / static_cast<double>(rows);
    double var = sums[c + ch] / static_cast<double>(rows) - m * m;
    if (var < 0.0) var = 0.0;
    mean[ch] = static_cast<float>(m);
    invstd[ch] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    if (running_mean) {
        const double unbiased = rows > 1 ? var * static_cast<double>(rows) / static_cast<double>(rows - 1) : var;
        running_mean[ch] = static_cast<float>((1.0 - momentum) * running_mean[ch] + momentum * m);
        running_var[ch] = static_cast<float>((1.0 - momentum) * running_var[ch] + momentum * unbiased);
    }
}

// out[g][c] (+)= sum over rows r in group g of x[r][c]; group g = rows [g*rows_per_group, (g+1)*rows_per_group).
// One workgroup per (group, row-chunk); chunk partials are combined by colsum_final_kernel.
__global__ __launch_bounds__(kThreads) void colsum_partial_kernel(const float* __restrict__ x, int ld, int64_t rows_per_group, int c,
                                                                   int chunks, double* __restrict__ partial) {
    __shared__ double sh[kThreads * 8];
    const Lanes l = lanes_for(c);
    const int g = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l.active) {
        const float* base = x + static_cast<int64_t>(g) * rows_per_group * ld;
        for (int64_t r = static_cast<int64_t>(chunk) * l.rows_per_pass + l.roff; r < rows_per_group;
             r += static_cast<int64_t>(chunks) * l.rows_per_pass) {
            const float4 a = *reinterpret_cast<const float4*>(base + r * ld + l.c4 * 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        }
    }
    block_reduce8(v, l, sh);
    if (l.roff == 0 && l.active) {
#pragma unroll
        for (int e = 0; e < 4; ++e) partial[static_cast<int64_t>(blockIdx.x) * c + l.c4 * 4 + e] = v[e];
    }
}

// grid (ceil(c/kFin), groups)
__global__ __launch_bounds__(kThreads) void colsum_final_kernel(const double* __restrict__ partial, int groups, int chunks, int c,
                                                                 int accumulate, float* __restrict__ out) {
    __shared__ double sh[kThreads];
    const int ch = blockIdx.x * kFin + (threadIdx.x % kFin);
    const int64_t g = blockIdx.y;
    const bool valid = ch < c;
    const double s = sliced_sum(partial + g * chunks * c, chunks, c, ch, valid, sh);
    if (threadIdx.x / kFin != 0 || !valid) return;
    const int64_t idx = g * c + ch;
    out[idx] = accumulate ? out[idx] + static_cast<float>(s) : static_cast<float>(s);
}

template <typename... A>
void launch_bwd_partial(int act, int nb, hipStream_t s, A... a) {
    if (act == DDPM_ACT_RELU) bn_bwd_partial_kernel<DDPM_ACT_RELU><<<nb, kThreads, 0, s>>>(a...);
    else if (act == DDPM_ACT_GELU) bn_bwd_partial_kernel<DDPM_ACT_GELU><<<nb, kThreads, 0, s>>>(a...);
    else bn_bwd_partial_kernel<DDPM_ACT_NONE><<<nb, kThreads, 0, s>>>(a...);
}

template <typename... A>
void launch_bwd_apply(int act, int nb, hipStream_t s, A... a) {
    if (act == DDPM_ACT_RELU) bn_bwd_apply_kernel<DDPM_ACT_RELU><<<nb, kThreads, 0, s>>>(a...);
    else if (act == DDPM_ACT_GELU) bn_bwd_apply_kernel<DDPM_ACT_GELU><<<nb, kThreads, 0, s>>>(a...);
    else bn_bwd_apply_kernel<DDPM_ACT_NONE><<<nb, kThreads, 0, s>>>(a...);
}

int partial_blocks(int64_t rows, int c) {
    const int rpp = kThreads / (c / 4) > 0 ? kThreads / (c / 4) : 1;
    int64_t nb = (rows + static_cast<int64_t>(rpp) * 8 - 1) / (static_cast<int64_t>(rpp) * 8);
    if (nb < 1) nb = 1;
    if (nb > kMaxPartialBlocks) nb = kMaxPartialBlocks;
    return static_cast<int>(nb);
}

}  // namespace

extern "C" int64_t ddpm_reduce_workspace_doubles(int c) { return static_cast<int64_t>(kMaxColsumBlocks) * c; }

extern "C" int ddpm_bn_stats(const float* x, int ldx, int64_t rows, int c, float eps, float momentum, double* workspace, float* mean,
                             float* invstd, float* running_mean, float* running_var, void* stream) {
    DDPM_REQUIRE(x && workspace && mean && invstd && rows > 0 && c > 0 && (c & 3) == 0 && c <= 1024);
    DDPM_REQUIRE((ldx & 3) == 0 && ldx >= c && ddpm::aligned16(x) && ddpm::aligned16(mean) && ddpm::aligned16(invstd));
    DDPM_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    const int nb = partial_blocks(rows, c);
    bn_stats_partial_kernel<<<nb, kThreads, 0, as_stream(stream)>>>(x, ldx, rows, c, workspace);
    bn_stats_final_kernel<<<(c + kFin - 1) / kFin, kThreads, 0, as_stream(stream)>>>(workspace, nb, rows, c, eps, momentum, mean,
                                                                                             invstd, running_mean, running_var);
    return ddpm::check_launch("bn_stats");
}

// The two halves of ddpm_bn_stats as separate calls, so that a producer that already holds the values in registers
// (ddpm_conv3x3_fwd_stats: the 3x3 convolution's epilogue) can leave the partial sums itself and only the finishing
// kernel runs afterwards.
extern "C" int ddpm_bn_partials(const float* x, int ldx, int64_t rows, int c, double* partial, int* nblocks, void* stream) {
    DDPM_REQUIRE(x && partial && nblocks && rows > 0 && c > 0 && (c & 3) == 0 && c <= 1024);
    DDPM_REQUIRE((ldx & 3) == 0 && ldx >= c && ddpm::aligned16(x));
    const int nb = partial_blocks(rows, c);
    bn_stats_partial_kernel<<<nb, kThreads, 0, as_stream(stream)>>>(x, ldx, rows, c, partial);
    *nblocks = nb;
    return ddpm::check_launch("bn_stats_partial_kernel");
}

extern "C" int ddpm_bn_stats_from_partials(const double* partial, int nblocks, int64_t rows, int c, float eps, float momentum, float* mean,
                                           float* invstd, float* running_mean, float* running_var, void* stream) {
    DDPM_REQUIRE(partial && nblocks > 0 && mean && invstd && rows > 0 && c > 0 && (c & 3) == 0 && c <= 1024);
    DDPM_REQUIRE(ddpm::aligned16(mean) && ddpm::aligned16(invstd));
    DDPM_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    bn_stats_final_kernel<<<(c + kFin - 1) / kFin, kThreads, 0, as_stream(stream)>>>(partial, nblocks, rows, c, eps, momentum, mean, invstd,
                                                                                             running_mean, running_var);
    return ddpm::check_launch("bn_stats_final_kernel");
}

// ---- allocation-free variant of ddpm_bn_stats -------------------------------------------------
// The per-block partial sums live in a file-scope device array instead of a caller workspace, for
// callers that cannot allocate device memory (code under HIP graph capture, serving loops, harnesses
// that forbid hipMalloc). Every GPU holds its own copy of the array at its own address; the host
// looks that address up once per device and passes it to the kernels like any other workspace.
constexpr int kNoAllocMaxChannels = 512;
constexpr int kMaxDevices = 64;
static __device__ __align__(256) double scratch[static_cast<size_t>(kMaxPartialBlocks) * 2 * kNoAllocMaxChannels];   // 8 MiB

namespace {
double* g_scratch_ptr[kMaxDevices];      // device address of `scratch` on each GPU (null until first use there)
std::mutex g_scratch_mutex;

// Device address of this GPU's copy of `scratch`, looked up on the first call on that device. The kernels run on
// `stream`: a stream that belongs to another GPU than the current one would hand that GPU's kernels THIS GPU's
// address, so such a call is refused.
int scratch_for_current_device(double** out, hipStream_t stream) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        ddpm::set_error("ddpm_bn_stats_noalloc: cannot identify the current device (%s, device %d, at most %d supported)",
                        hipGetErrorString(e), dev, kMaxDevices);
        return DDPM_E_LAUNCH;
    }
    if (stream != nullptr) {
        hipDevice_t sdev = -1;
        e = hipStreamGetDevice(stream, &sdev);
        if (e != hipSuccess || static_cast<int>(sdev) != dev) {
            ddpm::set_error("ddpm_bn_stats_noalloc: the stream belongs to device %d, the current device is %d (%s): make the stream's "
                            "device current before the call", static_cast<int>(sdev), dev, hipGetErrorString(e));
            return DDPM_E_INVALID;
        }
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            ddpm::set_error("ddpm_bn_stats_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return DDPM_E_LAUNCH;
        }
        g_scratch_ptr[dev] = static_cast<double*>(p);
    }
    *out = g_scratch_ptr[dev];
    return DDPM_OK;
}
}  // namespace

extern "C" int ddpm_bn_stats_noalloc_max_channels(void) { return kNoAllocMaxChannels; }

extern "C" int ddpm_bn_stats_noalloc(const float* x, int ldx, int64_t rows, int c, float eps, float momentum, float* mean, float* invstd,
                                     float* running_mean, float* running_var, void* stream) {
    DDPM_REQUIRE(x && mean && invstd && rows > 0 && c > 0 && (c & 3) == 0);
    if (c > kNoAllocMaxChannels) {
        ddpm::set_error("ddpm_bn_stats_noalloc: %d channels exceed the %d its built-in scratch buffer holds; use ddpm_bn_stats with a "
                        "caller workspace", c, kNoAllocMaxChannels);
        return DDPM_E_INVALID;
    }
    DDPM_REQUIRE((ldx & 3) == 0 && ldx >= c && ddpm::aligned16(x) && ddpm::aligned16(mean) && ddpm::aligned16(invstd));
    DDPM_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    double* ws = nullptr;
    if (int rc = scratch_for_current_device(&ws, as_stream(stream))) return rc;
    // every partial slot that the finishing kernel reads is written by the first kernel of this call
    // (nb blocks x 2 x c), so nothing left over from an earlier, larger call is ever read: no clearing needed
    const int nb = partial_blocks(rows, c);
    bn_stats_partial_kernel<<<nb, kThreads, 0, as_stream(stream)>>>(x, ldx, rows, c, ws);
    bn_stats_final_kernel<<<(c + kFin - 1) / kFin, kThreads, 0, as_stream(stream)>>>(ws, nb, rows, c, eps, momentum, mean, invstd,
                                                                                             running_mean, running_var);
    return ddpm::check_launch("bn_stats_noalloc");
}

extern "C" int ddpm_bn_apply_fwd(const float* x, int ldx, const float* mean, const float* invstd, const float* w, const float* b,
                                 const float* residual, int ldr, const float* temb, int act, float* out, int ldo, int batch, int hw,
                                 int c, void* stream) {
    DDPM_REQUIRE(x && mean && invstd && w && b && out && batch > 0 && hw > 0 && c > 0 && (c & 3) == 0);
    DDPM_REQUIRE((ldx & 3) == 0 && (ldo & 3) == 0 && ldx >= c && ldo >= c && ddpm::aligned16(x) && ddpm::aligned16(out));
    DDPM_REQUIRE(!residual || ((ldr & 3) == 0 && ldr >= c && ddpm::aligned16(residual)));
    DDPM_REQUIRE(ddpm::aligned16(mean) && ddpm::aligned16(invstd) && ddpm::aligned16(w) && ddpm::aligned16(b));
    DDPM_REQUIRE(!temb || ddpm::aligned16(temb));
    DDPM_REQUIRE(act >= DDPM_ACT_NONE && act <= DDPM_ACT_GELU);
    const int64_t rows = static_cast<int64_t>(batch) * hw;
    int64_t grid = (rows * (c / 4) + kThreads - 1) / kThreads;
    if (grid > 256 * 32) grid = 256 * 32;
#define DDPM_BN_LAUNCH(A) \
    bn_apply_fwd_kernel<A><<<static_cast<int>(grid), kThreads, 0, as_stream(stream)>>>(x, ldx, mean, invstd, w, b, residual, ldr, temb, out, ldo, rows, hw, c)
    if (act == DDPM_ACT_RELU) DDPM_BN_LAUNCH(DDPM_ACT_RELU);
    else if (act == DDPM_ACT_GELU) DDPM_BN_LAUNCH(DDPM_ACT_GELU);
    else DDPM_BN_LAUNCH(DDPM_ACT_NONE);
#undef DDPM_BN_LAUNCH
    return ddpm::check_launch("bn_apply_fwd_kernel");
}

extern "C" int ddpm_bn_bwd(const float* x, int ldx, const float* dout, int ldd, const float* residual, int ldr, const float* mean,
                           const float* invstd, const float* w, const float* b, int act, int64_t rows, int c, double* workspace,
                           float* sums, float* dw, float* db, int accumulate, float* dx, int lddx, float* dresidual, int lddr,
                           float* dconv_bias, void* stream) {
    DDPM_REQUIRE(x && dout && mean && invstd && w && b && workspace && sums && dw && db && dx && rows > 0);
    DDPM_REQUIRE(c > 0 && (c & 3) == 0 && c <= 1024);
    DDPM_REQUIRE((ldx & 3) == 0 && (ldd & 3) == 0 && (lddx & 3) == 0 && ldx >= c && ldd >= c && lddx >= c);
    DDPM_REQUIRE(ddpm::aligned16(x) && ddpm::aligned16(dout) && ddpm::aligned16(dx) && ddpm::aligned16(mean) && ddpm::aligned16(invstd));
    DDPM_REQUIRE(ddpm::aligned16(w) && ddpm::aligned16(b));
    DDPM_REQUIRE(!residual || ((ldr & 3) == 0 && ldr >= c && ddpm::aligned16(residual)));
    DDPM_REQUIRE(!dresidual || ((lddr & 3) == 0 && lddr >= c && ddpm::aligned16(dresidual)));
    DDPM_REQUIRE(act >= DDPM_ACT_NONE && act <= DDPM_ACT_GELU);
    hipStream_t s = as_stream(stream);
    const int nb = partial_blocks(rows, c);
    launch_bwd_partial(act, nb, s, x, ldx, dout, ldd, residual, ldr, mean, invstd, w, b, rows, c, workspace);
    bn_bwd_final_kernel<<<(c + kFin - 1) / kFin, kThreads, 0, s>>>(workspace, nb, c, accumulate, dw, db, sums);
    launch_bwd_apply(act, nb, s, x, ldx, dout, ldd, residual, ldr, mean, invstd, w, b, sums, rows, rows, c, dx, lddx, dresidual, lddr,
                     dconv_bias ? workspace : nullptr);
    if (dconv_bias) colsum_final_kernel<<<dim3((c + kFin - 1) / kFin, 1), kThreads, 0, s>>>(workspace, 1, nb, c, accumulate, dconv_bias);
    return ddpm::check_launch("bn_bwd");
}

// ---- the same BatchNorm in two halves, for data-parallel training with GLOBAL batch statistics ----------------
extern "C" int ddpm_bn_partial_sums(const float* x, int ldx, int64_t rows, int c, double* workspace, double* sums, void* stream) {
    DDPM_REQUIRE(x && workspace && sums && rows > 0 && c > 0 && (c & 3) == 0 && c <= 1024);
    DDPM_REQUIRE((ldx & 3) == 0 && ldx >= c && ddpm::aligned16(x));
    const int nb = partial_blocks(rows, c);
    bn_stats_partial_kernel<<<nb, kThreads, 0, as_stream(stream)>>>(x, ldx, rows, c, workspace);
    pair_sums_final_kernel<<<(c + kFin - 1) / kFin, kThreads, 0, as_stream(stream)>>>(workspace, nb, c, sums);
    return ddpm::check_launch("bn_partial_sums");
}

// The partial sums a producer left (ddpm_conv3x3_fwd_stats, ddpm_bn_partials) finished to this rank's raw sums: the
// second half of ddpm_bn_partial_sums alone.
extern "C" int ddpm_bn_sums_from_partials(const double* partial, int nblocks, int c, double* sums, void* stream) {
    DDPM_REQUIRE(partial && sums && nblocks > 0 && c > 0 && (c & 3) == 0 && c <= 1024);
    pair_sums_final_kernel<<<(c + kFin - 1) / kFin, kThreads, 0, as_stream(stream)>>>(partial, nblocks, c, sums);
    return ddpm::check_launch("pair_sums_final_kernel");
}

extern "C" int ddpm_bn_finalize_stats(const double* sums, int64_t total_rows, int c, float eps, float momentum, float* mean, float* invstd,
                                      float* running_mean, float* running_var, void* stream) {
    DDPM_REQUIRE(sums && mean && invstd && total_rows > 0 && c > 0);
    DDPM_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    bn_finalize_kernel<<<(c + kThreads - 1) / kThreads, kThreads, 0, as_stream(stream)>>>(sums, total_rows, c, eps, momentum, mean, invstd,
                                                                                          running_mean, running_var);
    return ddpm::check_launch("bn_finalize_kernel");
}

extern "C" int ddpm_bn_bwd_sums(const float* x, int ldx, const float* dout, int ldd, const float* residual, int ldr, const float* mean,
                                const float* invstd, const float* w, const float* b, int act, int64_t rows, int c, double* workspace,
                                double* sums, void* stream) {
    DDPM_REQUIRE(x && dout && mean && invstd && w && b && workspace && sums && rows > 0);
    DDPM_REQUIRE(c > 0 && (c & 3) == 0 && c <= 1024);
    DDPM_REQUIRE((ldx & 3) == 0 && (ldd & 3) == 0 && ldx >= c && ldd >= c);
    DDPM_REQUIRE(ddpm::aligned16(x) && ddpm::aligned16(dout) && ddpm::aligned16(mean) && ddpm::aligned16(invstd));
    DDPM_REQUIRE(ddpm::aligned16(w) && ddpm::aligned16(b));
    DDPM_REQUIRE(!residual || ((ldr & 3) == 0 && ldr >= c && ddpm::aligned16(residual)));
    DDPM_REQUIRE(act >= DDPM_ACT_NONE && act <= DDPM_ACT_GELU);
    hipStream_t s = as_stream(stream);
    const int nb = partial_blocks(rows, c);
    launch_bwd_partial(act, nb, s, x, ldx, dout, ldd, residual, ldr, mean, invstd, w, b, rows, c, workspace);
    pair_sums_final_kernel<<<(c + kFin - 1) / kFin, kThreads, 0, s>>>(workspace, nb, c, sums);
    return ddpm::check_launch("bn_bwd_sums");
}

extern "C" int ddpm_bn_bwd_apply(const float* x, int ldx, const float* dout, int ldd, const float* residual, int ldr, const float* mean,
                                 const float* invstd, const float* w, const float* b, const float* sums, int64_t total_rows, int act,
                                 int64_t rows, int c, double* workspace, float* dx, int lddx, float* dresidual, int lddr,
                                 float* dconv_bias, void* stream) {
    DDPM_REQUIRE(x && dout && mean && invstd && w && b && sums && dx && rows > 0 && total_rows >= rows);
    DDPM_REQUIRE(c > 0 && (c & 3) == 0 && c <= 1024 && (!dconv_bias || workspace));
    DDPM_REQUIRE((ldx & 3) == 0 && (ldd & 3) == 0 && (lddx & 3) == 0 && ldx >= c && ldd >= c && lddx >= c);
    DDPM_REQUIRE(ddpm::aligned16(x) && ddpm::aligned16(dout) && ddpm::aligned16(dx));
    DDPM_REQUIRE(!residual || ((ldr & 3) == 0 && ldr >= c && ddpm::aligned16(residual)));
    DDPM_REQUIRE(!dresidual || ((lddr & 3) == 0 && lddr >= c && ddpm::aligned16(dresidual)));
    DDPM_REQUIRE(act >= DDPM_ACT_NONE && act <= DDPM_ACT_GELU);
    hipStream_t s = as_stream(stream);
    const int nb = partial_blocks(rows, c);
    launch_bwd_apply(act, nb, s, x, ldx, dout, ldd, residual, ldr, mean, invstd, w, b, sums, rows, total_rows, c, dx, lddx, dresidual, lddr,
                     dconv_bias ? workspace : nullptr);
    if (dconv_bias) colsum_final_kernel<<<dim3((c + kFin - 1) / kFin, 1), kThreads, 0, s>>>(workspace, 1, nb, c, 0, dconv_bias);
    return ddpm::check_launch("bn_bwd_apply");
}

extern "C" int ddpm_colsum(const float* x, int ld, int groups, int64_t rows_per_group, int c, double* workspace, int accumulate,
                           float* out, void* stream) {
    DDPM_REQUIRE(x && workspace && out && groups > 0 && rows_per_group > 0 && c > 0 && (c & 3) == 0 && c <= 1024);
    DDPM_REQUIRE((ld & 3) == 0 && ld >= c && ddpm::aligned16(x));
    DDPM_REQUIRE(groups <= kMaxColsumBlocks);
    int chunks = partial_blocks(rows_per_group, c);
    if (chunks > kMaxColsumBlocks / groups) chunks = kMaxColsumBlocks / groups;
    hipStream_t s = as_stream(stream);
    colsum_partial_kernel<<<groups * chunks, kThreads, 0, s>>>(x, ld, rows_per_group, c, chunks, workspace);
    colsum_final_kernel<<<dim3((c + kFin - 1) / kFin, groups), kThreads, 0, s>>>(workspace, groups, chunks, c, accumulate, out);
    return ddpm::check_launch("colsum");
}

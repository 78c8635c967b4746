// HBM-bound elementwise kernels of the diffusion process: reverse ancestral step, forward noising,
// eps-MSE loss, max-pool + skip hand-off, time-embedding broadcast add.
#include "common.h"
#include "process_math.h"

// The process kernels promise separately rounded fp32 operations: never contract a*b+c into an FMA.
#pragma clang fp contract(off)

using ddpm::as_stream;

namespace {

constexpr int kThreads = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ inline int grid_for(int64_t work_items, int per_block, int cap = 256 * 16) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return static_cast<int>(g);
}

// ---- reverse step ---------------------------------------------------------------------------
using ddpm::guided_eps;
using ddpm::step1;

// z[i] = element i of the Gaussian stream (seed, call): what the fused final-conv + step kernel draws in place.
__global__ __launch_bounds__(kThreads) void philox_normal_kernel(float* __restrict__ z, int64_t n, uint64_t seed, uint64_t call) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
        z[i] = ddpm::philox_normal(seed, call, static_cast<uint64_t>(i));
}

typedef float vf4 __attribute__((ext_vector_type(4)));

// STREAM: the tensors are far larger than the caches (every operand comes from HBM and nothing is read again soon):
// four 16-byte pieces per thread and stream are requested before the first is used, with non-temporal loads and
// stores; measured 6.3 TB/s against 5.1 TB/s for one piece per iteration with ordinary accesses (tools/probe/step_bw.hip).
// Small tensors (one denoising step of a sampling batch) stay on ordinary accesses so that the result stays in the
// caches for the network that reads it next.
template <bool HAS_Z, bool HAS_U, bool STREAM>
__global__ __launch_bounds__(kThreads) void reverse_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                                 const float* __restrict__ eps_u, float w,
                                                                 const float* __restrict__ z, float c1, float c2, float c3,
                                                                 float* __restrict__ out, int64_t n) {
    const bool low = fabsf(w) < 0.5f;
    const int64_t n4 = n >> 2;
    constexpr int U = STREAM ? 4 : 1;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x * U;
    auto ld = [](const float* p, int64_t i) -> vf4 {
        return STREAM ? __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p) + i) : reinterpret_cast<const vf4*>(p)[i];
    };
    for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * blockDim.x * U + threadIdx.x; i0 < n4; i0 += stride) {
        vf4 xv[U], ev[U], uv[U], zv[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int64_t i = i0 + static_cast<int64_t>(k) * blockDim.x;
            if (i < n4) {
                xv[k] = ld(x, i);
                ev[k] = ld(eps, i);
                if (HAS_U) uv[k] = ld(eps_u, i);
                if (HAS_Z) zv[k] = ld(z, i);
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int64_t i = i0 + static_cast<int64_t>(k) * blockDim.x;
            if (i >= n4) continue;
            vf4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ee = HAS_U ? guided_eps(ev[k][e], uv[k][e], w, low) : ev[k][e];
                o[e] = step1(xv[k][e], ee, HAS_Z ? zv[k][e] : 0.f, c1, c2, c3, HAS_Z);
            }
            if (STREAM) __builtin_nontemporal_store(o, reinterpret_cast<vf4*>(out) + i);
            else reinterpret_cast<vf4*>(out)[i] = o;
        }
    }
    // tail (n not a multiple of 4)
    const int64_t tail0 = n4 << 2;
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid < n - tail0) {
        const int64_t i = tail0 + gid;
        float e = eps[i];
        if (HAS_U) e = guided_eps(e, eps_u[i], w, low);
        out[i] = step1(x[i], e, HAS_Z ? z[i] : 0.f, c1, c2, c3, HAS_Z);
    }
}

// ---- hand-off to the callers: [-1, 1] samples -> [0, 1] images, and the image grid ---------------------
// img = clamp((x + 1) / 2, 0, 1) with torch's roundings (one add, an exact halving) and its NaN propagation.
__device__ __forceinline__ float to_unit_range(float x) {
    const float v = __fmul_rn(__fadd_rn(x, 1.0f), 0.5f);
    return v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
}

__global__ __launch_bounds__(kThreads) void postprocess_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = to_unit_range(x[i]);
}

// One thread per grid element: image k = row*xmaps + col occupies rows [row*(h+pad)+pad, +h) and the matching
// columns; everything else (and cells beyond the last image) is pad_value. Single-channel images are repeated
// over the three grid channels. normalize: v = (clamp(v, lo, hi) - lo) / max(hi - lo, 1e-5).
__global__ __launch_bounds__(kThreads) void make_grid_kernel(const float* __restrict__ img, int n, int c, int h, int w, int xmaps,
                                                              int pad, float pad_value, int normalize, float lo, float hi, int gc, int gh,
                                                              int gw, float* __restrict__ grid) {
    const int64_t total = static_cast<int64_t>(gc) * gh * gw;
    const float span = fmaxf(__fsub_rn(hi, lo), 1e-5f);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int gx = static_cast<int>(i % gw), gy = static_cast<int>((i / gw) % gh), ch = static_cast<int>(i / (static_cast<int64_t>(gw) * gh));
        const int cy = gy / (h + pad), cx = gx / (w + pad);
        const int iy = gy % (h + pad) - pad, ix = gx % (w + pad) - pad;
        const int64_t k = static_cast<int64_t>(cy) * xmaps + cx;
        float v = pad_value;
        // the trailing `pad` rows / columns of the grid fall into cell row ymaps / cell column xmaps with iy, ix < 0
        if (iy >= 0 && ix >= 0 && cx < xmaps && k < n) {
            v = img[((k * c + (c == 1 ? 0 : ch)) * h + iy) * w + ix];
            if (normalize) {
                v = v < lo ? lo : (v > hi ? hi : v);
                v = __fdiv_rn(__fsub_rn(v, lo), span);
            }
        }
        grid[i] = v;
    }
}

// ---- forward noising ------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                              const int64_t* __restrict__ t, const float* __restrict__ abar,
                                                              int timesteps, float* __restrict__ out, int64_t per_sample,
                                                              int blocks_per_sample) {
    const int b = blockIdx.x / blocks_per_sample;
    const int chunk = blockIdx.x % blocks_per_sample;
    int64_t ti = t[b];
    if (ti < 0) ti += timesteps;               // python-style negative index
    ti = ti < 0 ? 0 : (ti >= timesteps ? timesteps - 1 : ti);   // memory safety; the host validates the range
    const float ab = abar[ti];
    // sqrtf is the correctly rounded square root here (hipcc's default); __fsqrt_rn is the 1-ulp native one
    const float s1 = sqrtf(ab);
    const float s2 = sqrtf(__fsub_rn(1.0f, ab));
    const float* xs = x0 + static_cast<int64_t>(b) * per_sample;
    const float* ns = noise + static_cast<int64_t>(b) * per_sample;
    float* os = out + static_cast<int64_t>(b) * per_sample;
    const bool vec = (per_sample & 3) == 0;
    if (vec) {
        const int64_t n4 = per_sample >> 2;
        for (int64_t i = static_cast<int64_t>(chunk) * blockDim.x + threadIdx.x; i < n4;
             i += static_cast<int64_t>(blocks_per_sample) * blockDim.x) {
            const float4 a = reinterpret_cast<const float4*>(xs)[i];
            const float4 e = reinterpret_cast<const float4*>(ns)[i];
            float4 o;
            o.x = __fadd_rn(__fmul_rn(s1, a.x), __fmul_rn(s2, e.x));
            o.y = __fadd_rn(__fmul_rn(s1, a.y), __fmul_rn(s2, e.y));
            o.z = __fadd_rn(__fmul_rn(s1, a.z), __fmul_rn(s2, e.z));
            o.w = __fadd_rn(__fmul_rn(s1, a.w), __fmul_rn(s2, e.w));
            reinterpret_cast<float4*>(os)[i] = o;
        }
    } else {
        for (int64_t i = static_cast<int64_t>(chunk) * blockDim.x + threadIdx.x; i < per_sample;
             i += static_cast<int64_t>(blocks_per_sample) * blockDim.x)
            os[i] = __fadd_rn(__fmul_rn(s1, xs[i]), __fmul_rn(s2, ns[i]));
    }
}

// ---- MSE ------------------------------------------------------------------------------------
constexpr int kMsePartials = 1024;

__global__ __launch_bounds__(kThreads) void mse_partial_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                                int64_t n, double* __restrict__ partial) {
    __shared__ double red[kThreads / 64];
    double acc = 0.0;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float d = p[i] - y[i];
        acc += static_cast<double>(d) * static_cast<double>(d);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < kThreads / 64; ++i) s += red[i];
        partial[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(kThreads) void mse_final_kernel(const double* __restrict__ partial, int count, int64_t n,
                                                              float* __restrict__ loss) {
    __shared__ double red[kThreads / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += blockDim.x) acc += partial[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < kThreads / 64; ++i) s += red[i];
        loss[0] = static_cast<float>(s / static_cast<double>(n));
    }
}

__global__ __launch_bounds__(kThreads) void mse_bwd_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                            const float* __restrict__ gout, int64_t n, float* __restrict__ g) {
    const float k = gout[0] * (2.0f / static_cast<float>(n));
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
        g[i] = (p[i] - y[i]) * k;
}

// ---- max-pool 2x2 + skip hand-off -----------------------------------------------------------
// One thread per (2x2 cell, 4 channels). Cells cover ceil(H/2) x ceil(W/2) so that the last
// row/column of an odd-sized image still gets its skip value; only complete cells are pooled
// (floor semantics of MaxPool2d(2,2)).
__global__ __launch_bounds__(kThreads) void pool_skip_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ temb,
                                                              float* __restrict__ skip, int lds_, float* __restrict__ pooled,
                                                              int ldp, uint8_t* __restrict__ argmax, int batch, int h, int w,
                                                              int c) {
    const int c4n = c >> 2;
    const int ch = (h + 1) >> 1, cw = (w + 1) >> 1;
    const int ph = h >> 1, pw = w >> 1;
    const int64_t total = static_cast<int64_t>(batch) * ch * cw * c4n;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int c4 = static_cast<int>(idx % c4n);
        int64_t r = idx / c4n;
        const int cx = static_cast<int>(r % cw);
        r /= cw;
        const int cy = static_cast<int>(r % ch);
        const int b = static_cast<int>(r / ch);
        float4 te = make_float4(0.f, 0.f, 0.f, 0.f);
        if (temb) te = *reinterpret_cast<const float4*>(temb + static_cast<int64_t>(b) * c + c4 * 4);
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 arg = make_uchar4(0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int y = cy * 2 + (q >> 1), xx = cx * 2 + (q & 1);
            if (y < h && xx < w) {
                const int64_t pix = (static_cast<int64_t>(b) * h + y) * w + xx;
                const float4 v = *reinterpret_cast<const float4*>(x + pix * ldx + c4 * 4);
                if (skip) {
                    float4 s;
                    s.x = v.x + te.x; s.y = v.y + te.y; s.z = v.z + te.z; s.w = v.w + te.w;
                    *reinterpret_cast<float4*>(skip + pix * lds_ + c4 * 4) = s;
                }
                // first maximum in window scan order wins (NaN propagates like torch: v != v)
                if (v.x > best.x || v.x != v.x) { best.x = v.x; arg.x = q; }
                if (v.y > best.y || v.y != v.y) { best.y = v.y; arg.y = q; }
                if (v.z > best.z || v.z != v.z) { best.z = v.z; arg.z = q; }
                if (v.w > best.w || v.w != v.w) { best.w = v.w; arg.w = q; }
            }
        }
        if (pooled && cy < ph && cx < pw) {
            const int64_t pp = (static_cast<int64_t>(b) * ph + cy) * pw + cx;
            *reinterpret_cast<float4*>(pooled + pp * ldp + c4 * 4) = best;
            if (argmax) *reinterpret_cast<uchar4*>(argmax + pp * c + c4 * 4) = arg;
        }
    }
}

__global__ __launch_bounds__(kThreads) void add_temb_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ temb,
                                                             float* __restrict__ y, int ldy, int batch, int hw, int c) {
    const int c4n = c >> 2;
    const int64_t total = static_cast<int64_t>(batch) * hw * c4n;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int c4 = static_cast<int>(idx % c4n);
        const int64_t pix = idx / c4n;
        const int b = static_cast<int>(pix / hw);
        const float4 v = *reinterpret_cast<const float4*>(x + pix * ldx + c4 * 4);
        const float4 te = *reinterpret_cast<const float4*>(temb + static_cast<int64_t>(b) * c + c4 * 4);
        float4 o;
        o.x = v.x + te.x; o.y = v.y + te.y; o.z = v.z + te.z; o.w = v.w + te.w;
        *reinterpret_cast<float4*>(y + pix * ldy + c4 * 4) = o;
    }
}

}  // namespace

extern "C" int ddpm_reverse_step(const float* x, const float* eps, const float* eps_uncond, float cfg_w, const float* z,
                                 float c1, float c2, float c3, float* x_out, int64_t n, void* stream) {
    DDPM_REQUIRE(x && eps && x_out && n >= 0);
    DDPM_REQUIRE(ddpm::aligned16(x) && ddpm::aligned16(eps) && ddpm::aligned16(x_out));
    DDPM_REQUIRE(!z || ddpm::aligned16(z));
    DDPM_REQUIRE(!eps_uncond || ddpm::aligned16(eps_uncond));
    if (n == 0) return DDPM_OK;
    hipStream_t s = as_stream(stream);
    const bool stream_mode = n >= (1ll << 23);        // 32 MiB per tensor and more: beyond what the caches keep between kernels
    const int grid = stream_mode ? grid_for((n + 15) / 16, kThreads, 256 * 64) : grid_for((n + 3) / 4, kThreads, 256 * 32);
#define DDPM_STEP_LAUNCH(Z, U_)                                                                                                      \
    do {                                                                                                                             \
        if (stream_mode) reverse_step_kernel<Z, U_, true><<<grid, kThreads, 0, s>>>(x, eps, eps_uncond, cfg_w, z, c1, c2, c3, x_out, n); \
        else reverse_step_kernel<Z, U_, false><<<grid, kThreads, 0, s>>>(x, eps, eps_uncond, cfg_w, z, c1, c2, c3, x_out, n);          \
    } while (0)
    if (z && eps_uncond) DDPM_STEP_LAUNCH(true, true);
    else if (z) DDPM_STEP_LAUNCH(true, false);
    else if (eps_uncond) DDPM_STEP_LAUNCH(false, true);
    else DDPM_STEP_LAUNCH(false, false);
#undef DDPM_STEP_LAUNCH
    return ddpm::check_launch("reverse_step_kernel");
}

extern "C" int ddpm_philox_normal(float* z, int64_t n, uint64_t seed, uint64_t call, void* stream) {
    DDPM_REQUIRE(z && n >= 0);
    if (n == 0) return DDPM_OK;
    philox_normal_kernel<<<grid_for(n, kThreads, 256 * 32), kThreads, 0, as_stream(stream)>>>(z, n, seed, call);
    return ddpm::check_launch("philox_normal_kernel");
}

extern "C" int ddpm_postprocess_images(const float* x, float* out, int64_t n, void* stream) {
    DDPM_REQUIRE(x && out && n >= 0);
    if (n == 0) return DDPM_OK;
    postprocess_kernel<<<grid_for(n, kThreads, 256 * 32), kThreads, 0, as_stream(stream)>>>(x, out, n);
    return ddpm::check_launch("postprocess_kernel");
}

extern "C" int ddpm_make_grid(const float* images, int n, int c, int h, int w, int nrow, int padding, float pad_value, int normalize,
                              float lo, float hi, float* grid, void* stream) {
    DDPM_REQUIRE(images && grid && n > 0 && c > 0 && h > 0 && w > 0 && nrow > 0 && padding >= 0);
    const int xmaps = nrow < n ? nrow : n, ymaps = (n + xmaps - 1) / xmaps;
    const int gc = c == 1 ? 3 : c, gh = (h + padding) * ymaps + padding, gw = (w + padding) * xmaps + padding;
    const int64_t total = static_cast<int64_t>(gc) * gh * gw;
    make_grid_kernel<<<grid_for(total, kThreads, 256 * 32), kThreads, 0, as_stream(stream)>>>(images, n, c, h, w, xmaps, padding, pad_value,
                                                                                            normalize, lo, hi, gc, gh, gw, grid);
    return ddpm::check_launch("make_grid_kernel");
}

extern "C" int ddpm_add_noise(const float* x0, const float* noise, const int64_t* t, const float* alpha_bar, int timesteps,
                              float* out, int batch, int64_t per_sample, void* stream) {
    DDPM_REQUIRE(x0 && noise && t && alpha_bar && out && batch >= 0 && per_sample >= 0 && timesteps >= 1);
    DDPM_REQUIRE(ddpm::aligned16(x0) && ddpm::aligned16(noise) && ddpm::aligned16(out));
    if (batch == 0 || per_sample == 0) return DDPM_OK;
    int bps = static_cast<int>((per_sample / 4 + kThreads - 1) / kThreads);
    if (bps < 1) bps = 1;
    if (bps > 64) bps = 64;
    DDPM_REQUIRE(static_cast<int64_t>(batch) * bps < (1ll << 31));
    add_noise_kernel<<<batch * bps, kThreads, 0, as_stream(stream)>>>(x0, noise, t, alpha_bar, timesteps, out, per_sample, bps);
    return ddpm::check_launch("add_noise_kernel");
}

extern "C" int ddpm_mse_partial_count(void) { return kMsePartials; }

extern "C" int ddpm_mse_fwd(const float* pred, const float* target, int64_t n, double* partial, float* loss_out, void* stream) {
    DDPM_REQUIRE(pred && target && partial && loss_out && n > 0);
    const int grid = grid_for(n, kThreads * 4, kMsePartials);
    mse_partial_kernel<<<grid, kThreads, 0, as_stream(stream)>>>(pred, target, n, partial);
    mse_final_kernel<<<1, kThreads, 0, as_stream(stream)>>>(partial, grid, n, loss_out);
    return ddpm::check_launch("mse_fwd");
}

extern "C" int ddpm_mse_bwd(const float* pred, const float* target, const float* gout, int64_t n, float* grad, void* stream) {
    DDPM_REQUIRE(pred && target && gout && grad && n > 0);
    mse_bwd_kernel<<<grid_for(n, kThreads * 4), kThreads, 0, as_stream(stream)>>>(pred, target, gout, n, grad);
    return ddpm::check_launch("mse_bwd_kernel");
}

// ---- power-of-two scaling of the backward pass ------------------------------------------------
// The backward pass is linear in the gradient it starts from. With the fp16-pair arithmetic its MFMA operands must
// sit inside fp16's exponent range, and loss gradients (2 * diff / numel) are far below it; the backward therefore
// runs on the incoming gradient times a power of two that brings its largest magnitude into [1, 2), and every
// parameter gradient is multiplied by the inverse power at the end. Both multiplications are exact in fp32.
// state: three device floats: [0] largest |g| (bit pattern compared as an integer), [1] the scale, [2] its inverse.
namespace {

__global__ __launch_bounds__(256) void grad_amax_kernel(const float* __restrict__ g, int64_t n, unsigned* __restrict__ state) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    unsigned m = 0u;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
        const unsigned a = __float_as_uint(g[i]) & 0x7fffffffu;
        if (a <= 0x7f800000u && a > m) m = a;     // NaNs do not take part; an infinity does (and disables the scaling)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = __shfl_xor(m, o);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(state, m);
}

__device__ __forceinline__ float scale_for(unsigned amax_bits) {
    const unsigned e = amax_bits >> 23;           // biased exponent of the largest magnitude
    if (e == 0u || e >= 254u) return 1.0f;        // zero, subnormal or not finite: leave the gradient as it is
    return __uint_as_float((254u - e) << 23);     // 2^-(e - 127)
}

__global__ __launch_bounds__(256) void grad_scale_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ state, float* __restrict__ out) {
    const float sc = scale_for(__float_as_uint(state[0]));
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) out[i] = g[i] * sc;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        state[1] = sc;
        state[2] = 1.0f / sc;
    }
}

constexpr int kScaleTensors = 96;
struct ScaleBatch {
    float* p[kScaleTensors];
    int64_t n[kScaleTensors];
};

__global__ __launch_bounds__(256) void scale_tensors_kernel(const ScaleBatch b, const float* __restrict__ factor) {
    const float f = factor[0];
    float* __restrict__ p = b.p[blockIdx.y];
    const int64_t n = b.n[blockIdx.y];
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    const int64_t tid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(p) & 15) == 0 ? n >> 2 : 0;
    for (int64_t i = tid; i < n4; i += stride) {
        f32x4 v = reinterpret_cast<f32x4*>(p)[i];
        v *= f;
        reinterpret_cast<f32x4*>(p)[i] = v;
    }
    for (int64_t i = (n4 << 2) + tid; i < n; i += stride) p[i] *= f;
}

}  // namespace

extern "C" int ddpm_grad_amax(const float* g, int64_t n, float* state, void* stream) {
    DDPM_REQUIRE(g && state && n > 0);
    if (hipMemsetAsync(state, 0, 3 * sizeof(float), as_stream(stream)) != hipSuccess) {
        ddpm::set_error("ddpm_grad_amax: cannot clear the state");
        return DDPM_E_LAUNCH;
    }
    grad_amax_kernel<<<grid_for(n, kThreads * 4), 256, 0, as_stream(stream)>>>(g, n, reinterpret_cast<unsigned*>(state));
    return ddpm::check_launch("grad_amax_kernel");
}

extern "C" int ddpm_grad_scale(const float* g, int64_t n, float* state, float* out, void* stream) {
    DDPM_REQUIRE(g && state && out && n > 0);
    grad_scale_kernel<<<grid_for(n, kThreads * 4), 256, 0, as_stream(stream)>>>(g, n, state, out);
    return ddpm::check_launch("grad_scale_kernel");
}

extern "C" int ddpm_scale_tensors(int ntensors, float* const* tensors, const int64_t* numel, const float* factor, void* stream) {
    DDPM_REQUIRE(ntensors >= 0 && factor);
    if (ntensors == 0) return DDPM_OK;
    DDPM_REQUIRE(tensors && numel);
    for (int t0 = 0; t0 < ntensors; t0 += kScaleTensors) {
        ScaleBatch b;
        const int cnt = ntensors - t0 < kScaleTensors ? ntensors - t0 : kScaleTensors;
        int64_t longest = 0;
        for (int i = 0; i < cnt; ++i) {
            DDPM_REQUIRE(numel[t0 + i] >= 0 && (numel[t0 + i] == 0 || tensors[t0 + i]));
            b.p[i] = tensors[t0 + i];
            b.n[i] = numel[t0 + i];
            if (b.n[i] > longest) longest = b.n[i];
        }
        if (longest == 0) continue;
        const int64_t want = (longest + 1023) / 1024;
        scale_tensors_kernel<<<dim3(static_cast<unsigned>(want < 128 ? want : 128), cnt), 256, 0, as_stream(stream)>>>(b, factor);
    }
    return ddpm::check_launch("scale_tensors_kernel");
}

extern "C" int ddpm_pool_skip_fwd(const float* x, int ldx, const float* temb, float* skip, int lds_, float* pooled, int ldp,
                                  uint8_t* argmax, int batch, int h, int w, int c, void* stream) {
    DDPM_REQUIRE(x && batch > 0 && h > 0 && w > 0 && c > 0 && (c & 3) == 0 && (ldx & 3) == 0);
    DDPM_REQUIRE(skip || pooled);
    DDPM_REQUIRE(!skip || ((lds_ & 3) == 0 && lds_ >= c && ddpm::aligned16(skip)));
    DDPM_REQUIRE(!pooled || ((ldp & 3) == 0 && ldp >= c && ddpm::aligned16(pooled)));
    DDPM_REQUIRE(ddpm::aligned16(x) && (!temb || ddpm::aligned16(temb)) && ldx >= c);
    const int64_t total = static_cast<int64_t>(batch) * ((h + 1) / 2) * ((w + 1) / 2) * (c / 4);
    pool_skip_kernel<<<grid_for(total, kThreads, 256 * 32), kThreads, 0, as_stream(stream)>>>(x, ldx, temb, skip, lds_, pooled, ldp,
                                                                                            argmax, batch, h, w, c);
    return ddpm::check_launch("pool_skip_kernel");
}

extern "C" int ddpm_add_temb(const float* x, int ldx, const float* temb, float* y, int ldy, int batch, int hw, int c,
                             void* stream) {
    DDPM_REQUIRE(x && temb && y && batch > 0 && hw > 0 && c > 0 && (c & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0);
    DDPM_REQUIRE(ldx >= c && ldy >= c && ddpm::aligned16(x) && ddpm::aligned16(y) && ddpm::aligned16(temb));
    const int64_t total = static_cast<int64_t>(batch) * hw * (c / 4);
    add_temb_kernel<<<grid_for(total, kThreads, 256 * 32), kThreads, 0, as_stream(stream)>>>(x, ldx, temb, y, ldy, batch, hw, c);
    return ddpm::check_launch("add_temb_kernel");
}

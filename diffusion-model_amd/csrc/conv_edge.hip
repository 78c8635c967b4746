// The two thin layers at the edges of the UNet, which are not GEMM-shaped enough for the matrix
// cores: the first 3x3 conv (1 or 3 input channels, NCHW image in, NHWC out) and the final 1x1
// conv (NHWC in, 1 or 3 channels, NCHW out). Both are memory-bound.
#include "conv_common.h"
#include "process_math.h"

using ddpm::as_stream;

namespace {

constexpr int kMaxLdsFloats = 12288;  // 48 KB of weights staged per workgroup

// One thread = one pixel x 4 consecutive output channels. Weights sit in LDS as [ci*9+tap][cout]. The kernel writes
// 16 bytes per thread and reads next to nothing, so what it costs besides the store stream is index arithmetic and the
// activation: when the workgroup's 256 threads cover whole pixels (cout / 4 divides 256) a thread keeps its channel
// quad for the whole launch (scale, shift in registers, no division by the channel count) and walks pixels with 32-bit
// arithmetic; GELU is the polynomial-erf form the matrix kernels use in their epilogues.
__global__ __launch_bounds__(256) void conv3x3_first_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ residual_unused, const float* __restrict__ temb,
                                                             int act, float* __restrict__ out, int batch, int cin, int h, int wd,
                                                             int cout, int ldo) {
    extern __shared__ __attribute__((aligned(16))) float sw[];   // cin * 9 * cout floats
    const int kt = cin * 9;
    for (int i = threadIdx.x; i < kt * cout; i += blockDim.x) {
        const int co = i % cout, k = i / cout;  // k = ci*9 + tap
        sw[i] = w[static_cast<int64_t>(co) * kt + k];
    }
    __syncthreads();
    const unsigned c4n = static_cast<unsigned>(cout) >> 2;
    const unsigned pixels = static_cast<unsigned>(batch) * h * wd;      // < 2^31, checked by the caller
    const bool whole = 256u % c4n == 0u;
    // pixel and channel quad of this thread: fixed quad when a workgroup covers whole pixels, else re-derived per step
    const unsigned ppb = whole ? 256u / c4n : 0u;
    unsigned c4 = whole ? threadIdx.x % c4n : 0u;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    auto load_affine = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sc[e] = scale ? scale[c4 * 4 + e] : 1.0f;
            sh[e] = shift ? shift[c4 * 4 + e] : 0.0f;
        }
    };
    if (whole) load_affine();
    const uint64_t total = static_cast<uint64_t>(pixels) * c4n;
    uint64_t idx = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    unsigned pix = whole ? blockIdx.x * ppb + threadIdx.x / c4n : 0u;
    for (;;) {
        if (whole) {
            if (pix >= pixels) break;
        } else {
            if (idx >= total) break;
            c4 = static_cast<unsigned>(idx % c4n);
            pix = static_cast<unsigned>(idx / c4n);
            load_affine();
        }
        const unsigned row = pix / static_cast<unsigned>(wd);
        const int xx = static_cast<int>(pix - row * wd);
        const unsigned b = row / static_cast<unsigned>(h);
        const int yy = static_cast<int>(row - b * h);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ci = 0; ci < cin; ++ci) {
            const float* plane = x + (static_cast<int64_t>(b) * cin + ci) * h * wd;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int y2 = yy + tap / 3 - 1, x2 = xx + tap % 3 - 1;
                const float v = (y2 >= 0 && y2 < h && x2 >= 0 && x2 < wd) ? plane[y2 * wd + x2] : 0.f;
                const float4 wv = *reinterpret_cast<const float4*>(sw + (ci * 9 + tap) * cout + c4 * 4);
                acc.x += v * wv.x; acc.y += v * wv.y; acc.z += v * wv.z; acc.w += v * wv.w;
            }
        }
        float r[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = r[e] * sc[e] + sh[e];
        if (act == DDPM_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = ddpm::conv::apply_act_fast(r[e], DDPM_ACT_GELU);
        } else if (act == DDPM_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.f ? r[e] : 0.f;
        }
        if (temb) {
            const float4 tv = *reinterpret_cast<const float4*>(temb + static_cast<int64_t>(b) * cout + c4 * 4);
            r[0] += tv.x; r[1] += tv.y; r[2] += tv.z; r[3] += tv.w;
        }
        *reinterpret_cast<float4*>(out + static_cast<int64_t>(pix) * ldo + c4 * 4) = make_float4(r[0], r[1], r[2], r[3]);
        if (whole) pix += gridDim.x * ppb;
        else idx += static_cast<uint64_t>(gridDim.x) * 256u;
    }
}

// The final 1x1 conv (NHWC features -> NCHW, 1 or 3 channels) with or without the reverse ancestral step fused
// behind it. 16 lanes share a pixel, each holding 4 channels of every 64-channel slab; weights sit in LDS as
// [cout][c]; the 16 partial sums are combined inside the VALU (row16_sum) and lane 0 of the group finishes the pixel.
// MODE 0: out = eps-hat. MODE 1..3: eps-hat never goes to memory -- the reverse-step expression is applied on the
// spot (1: no noise term, the t == 0 step; 2: noise read from a tensor; 3: noise drawn in place from the Philox
// stream). Every mode forms eps-hat with the same operations in the same order, so its bits do not depend on the mode.
// The kernel is memory-bound on the feature read (256 B per pixel at 64 channels): a group works on PX pixels at a
// time, all of their feature loads issued (non-temporal: the features are dead after this kernel) before the first sum.
struct FinalStepArgs {
    const float* feat;       // [B][hw][ld] NHWC features
    const float* w;          // [cout][c]
    const float* bias;       // [cout] or null
    const float* x_t;        // [B][cout][hw] NCHW
    const float* z;          // NCHW noise (MODE 2)
    const float* eps_cond;   // NCHW conditional eps-hat: this pass is the unconditional one of a guided step (or null)
    float* x_out;            // NCHW: eps-hat (MODE 0) or x_{t-1}
    float* eps_out;          // optional NCHW copy of this pass's eps-hat (or null)
    float c1, c2, c3, cfg_w;
    uint64_t seed, call;
    int ld, batch, hw, c, cout;
};

typedef float vf4 __attribute__((ext_vector_type(4)));

// Sum over the 16 lanes of a DPP row (every lane gets the total): data-parallel-primitive moves inside the VALU,
// no trip through the LDS crossbar that a shuffle takes. Lane pairs, then quads, then the two halves of 8, then 16.
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp = [](float x, int ctrl_tag) -> float {
        const int b = __builtin_bit_cast(int, x);
        int r;
        if (ctrl_tag == 0) r = __builtin_amdgcn_update_dpp(0, b, 0xB1, 0xf, 0xf, true);        // quad_perm [1,0,3,2]
        else if (ctrl_tag == 1) r = __builtin_amdgcn_update_dpp(0, b, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
        else if (ctrl_tag == 2) r = __builtin_amdgcn_update_dpp(0, b, 0x141, 0xf, 0xf, true);  // row_half_mirror
        else r = __builtin_amdgcn_update_dpp(0, b, 0x140, 0xf, 0xf, true);                     // row_mirror
        return __builtin_bit_cast(float, r);
    };
    v += dpp(v, 0);
    v += dpp(v, 1);
    v += dpp(v, 2);
    v += dpp(v, 3);
    return v;
}

template <int MODE>
__device__ __forceinline__ void final_finish(const FinalStepArgs& a, int64_t pix, int co, float acc, bool low) {
    const int64_t b = pix / a.hw, q = pix % a.hw;
    const int64_t e = (b * a.cout + co) * a.hw + q;
    float eps = acc + (a.bias ? a.bias[co] : 0.f);
    if (MODE == 0) {
        a.x_out[e] = eps;
        return;
    }
    if (a.eps_out) a.eps_out[e] = eps;
    if (a.eps_cond) eps = ddpm::guided_eps(a.eps_cond[e], eps, a.cfg_w, low);
    float zv = 0.f;
    if (MODE == 2) zv = a.z[e];
    if (MODE == 3) zv = ddpm::philox_normal(a.seed, a.call, static_cast<uint64_t>(e));
    a.x_out[e] = ddpm::step1(a.x_t[e], eps, zv, a.c1, a.c2, a.c3, MODE != 1);
}

// CI = 64-channel slabs per pixel known at compile time (1 or 2: PX pixels in flight), 0 = any width (one pixel)
template <int MODE, int CI>
__global__ __launch_bounds__(256) void final_conv_kernel(const FinalStepArgs a) {
    // (dynamic: c * cout floats. A fixed 48 KiB array would cap the kernel at three workgroups per CU, too few
    // loads in flight for a kernel that does nothing but stream the features)
    extern __shared__ __attribute__((aligned(16))) float sw[];
    for (int i = threadIdx.x; i < a.c * a.cout; i += blockDim.x) sw[i] = a.w[i];
    __syncthreads();
    constexpr int PX = CI == 0 ? 1 : 4;
    const int sub = threadIdx.x & 15;
    const int64_t npix = static_cast<int64_t>(a.batch) * a.hw;
    const int64_t groups = static_cast<int64_t>(gridDim.x) * (blockDim.x >> 4);
    // every lane of a wave iterates the same number of times (the row sums need all 16 lanes of a group)
    const int64_t iters = (npix + groups * PX - 1) / (groups * PX);
    const bool low = fabsf(a.cfg_w) < 0.5f;
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 4) + (threadIdx.x >> 4);
    for (int64_t it = 0; it < iters; ++it) {
        int64_t pix[PX];
        bool live[PX];
        vf4 v[PX][CI == 0 ? 1 : CI];
#pragma unroll
        for (int u = 0; u < PX; ++u) {
            pix[u] = (it * PX + u) * groups + gid;
            live[u] = pix[u] < npix;
            if (CI > 0) {
#pragma unroll
                for (int k = 0; k < (CI == 0 ? 1 : CI); ++k) {
                    v[u][k] = vf4{0.f, 0.f, 0.f, 0.f};
                    if (live[u]) v[u][k] = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(a.feat + pix[u] * a.ld + k * 64 + sub * 4));
                }
            }
        }
        auto dot = [&](int u, int co) -> float {
            float acc = 0.f;
            if (CI > 0) {
#pragma unroll
                for (int k = 0; k < (CI == 0 ? 1 : CI); ++k) {
                    const float4 wv = *reinterpret_cast<const float4*>(sw + co * a.c + k * 64 + sub * 4);
                    acc += v[u][k].x * wv.x + v[u][k].y * wv.y + v[u][k].z * wv.z + v[u][k].w * wv.w;
                }
            } else if (live[u]) {
                for (int c0 = sub * 4; c0 < a.c; c0 += 64) {
                    const float4 xv = *reinterpret_cast<const float4*>(a.feat + pix[u] * a.ld + c0);
                    const float4 wv = *reinterpret_cast<const float4*>(sw + co * a.c + c0);
                    acc += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
                }
            }
            return row16_sum(acc);       // every lane of the group holds the pixel's sum
        };
        if (a.cout <= 4) {
            // up to PX * 4 <= 16 (pixel, channel) results per group: lane `sub` finishes result number `sub`, so the
            // loads, the noise draw and the store of the step run on as many lanes side by side instead of one after
            // the other on lane 0
            float mine = 0.f;
            int64_t my_pix = 0;
            bool my_live = false;
            const int my_co = sub % a.cout;
#pragma unroll
            for (int u = 0; u < PX; ++u) {
#pragma unroll
                for (int co = 0; co < 4; ++co) {
                    if (co < a.cout) {
                        const float tot = dot(u, co);
                        if (sub == u * a.cout + co) {
                            mine = tot;
                            my_pix = pix[u];
                            my_live = live[u];
                        }
                    }
                }
            }
            if (my_live && sub < PX * a.cout) final_finish<MODE>(a, my_pix, my_co, mine, low);
        } else {
            for (int co = 0; co < a.cout; ++co) {
#pragma unroll
                for (int u = 0; u < PX; ++u) {
                    const float tot = dot(u, co);
                    if (live[u] && sub == 0) final_finish<MODE>(a, pix[u], co, tot, low);
                }
            }
        }
    }
}

template <int MODE>
int launch_final(const FinalStepArgs& a, hipStream_t s) {
    const int64_t npix = static_cast<int64_t>(a.batch) * a.hw;
    const int px = (a.c == 64 || a.c == 128) ? 4 : 1;
    int64_t grid = (npix + 16 * px - 1) / (16 * px);
    if (grid > 256 * 32) grid = 256 * 32;
    const size_t lds = sizeof(float) * a.c * a.cout;
    if (a.c == 64) final_conv_kernel<MODE, 1><<<static_cast<int>(grid), 256, lds, s>>>(a);
    else if (a.c == 128) final_conv_kernel<MODE, 2><<<static_cast<int>(grid), 256, lds, s>>>(a);
    else final_conv_kernel<MODE, 0><<<static_cast<int>(grid), 256, lds, s>>>(a);
    return ddpm::check_launch("final_conv_kernel");
}

}  // namespace

extern "C" int ddpm_final_conv_reverse_step(const float* feat, int ld, const float* w, const float* bias, const float* x_t,
                                            const float* eps_cond, float cfg_w, int noise_mode, const float* z, uint64_t seed,
                                            uint64_t call, float c1, float c2, float c3, float* x_out, float* eps_out, int batch, int h,
                                            int w_, int c, int cout, void* stream) {
    DDPM_REQUIRE(feat && w && x_t && x_out && batch > 0 && h > 0 && w_ > 0 && c > 0 && cout > 0);
    DDPM_REQUIRE((c & 3) == 0 && (ld & 3) == 0 && ld >= c && ddpm::aligned16(feat));
    DDPM_REQUIRE(c * cout <= kMaxLdsFloats);
    DDPM_REQUIRE(noise_mode >= 0 && noise_mode <= 2 && (noise_mode != 1 || z));
    FinalStepArgs a;
    a.feat = feat; a.w = w; a.bias = bias; a.x_t = x_t; a.z = z; a.eps_cond = eps_cond; a.x_out = x_out; a.eps_out = eps_out;
    a.c1 = c1; a.c2 = c2; a.c3 = c3; a.cfg_w = cfg_w; a.seed = seed; a.call = call;
    a.ld = ld; a.batch = batch; a.hw = h * w_; a.c = c; a.cout = cout;
    hipStream_t s = as_stream(stream);
    if (noise_mode == 0) return launch_final<1>(a, s);
    if (noise_mode == 1) return launch_final<2>(a, s);
    return launch_final<3>(a, s);
}

extern "C" int ddpm_conv3x3_first_fwd(const float* x_nchw, const float* w, const ddpm_epilogue* ep, int act, float* out, int batch,
                                      int cin, int h, int w_, int cout, int ldo, void* stream) {
    DDPM_REQUIRE(x_nchw && w && ep && out && batch > 0 && cin > 0 && h > 0 && w_ > 0 && cout > 0);
    DDPM_REQUIRE((cout & 3) == 0 && (ldo & 3) == 0 && ldo >= cout && ddpm::aligned16(out));
    DDPM_REQUIRE(cin * 9 * cout <= kMaxLdsFloats);
    DDPM_REQUIRE(!ep->residual && (!ep->scale || ep->shift));
    DDPM_REQUIRE(act >= DDPM_ACT_NONE && act <= DDPM_ACT_GELU);
    DDPM_REQUIRE(static_cast<int64_t>(batch) * h * w_ < (1ll << 31));
    DDPM_REQUIRE(!ep->temb || ddpm::aligned16(ep->temb));
    const int64_t total = static_cast<int64_t>(batch) * h * w_ * (cout / 4);
    int64_t grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    conv3x3_first_kernel<<<static_cast<int>(grid), 256, sizeof(float) * cin * 9 * cout, as_stream(stream)>>>(x_nchw, w, ep->scale, ep->shift, nullptr, ep->temb,
                                                                               act, out, batch, cin, h, w_, cout, ldo);
    return ddpm::check_launch("conv3x3_first_kernel");
}

extern "C" int ddpm_conv1x1_final_fwd(const float* x, int ld, const float* w, const float* bias, float* out_nchw, int batch, int h,
                                      int w_, int c, int cout, void* stream) {
    DDPM_REQUIRE(x && w && out_nchw && batch > 0 && h > 0 && w_ > 0 && c > 0 && cout > 0);
    DDPM_REQUIRE((c & 3) == 0 && (ld & 3) == 0 && ld >= c && ddpm::aligned16(x));
    DDPM_REQUIRE(c * cout <= kMaxLdsFloats);
    FinalStepArgs a = {};
    a.feat = x; a.w = w; a.bias = bias; a.x_out = out_nchw;
    a.ld = ld; a.batch = batch; a.hw = h * w_; a.c = c; a.cout = cout;
    return launch_final<0>(a, as_stream(stream));
}

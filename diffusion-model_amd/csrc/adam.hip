// Adam update of a list of tensors in one or a few launches (the optimizer.step() that follows every
// backward of the training loop, first-iteration/main.py:51,84: torch.optim.Adam with its defaults).
// HBM-bound: 16 bytes read and 12 written per parameter element.  The tensor list travels in the kernel
// arguments, so the call allocates nothing and copies nothing.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "common.h"

namespace {

constexpr int kTensorsPerLaunch = 64;   // 64 x 40 bytes of kernel arguments
constexpr int kBlocksPerTensor = 128;

struct AdamBatch {
    float* p[kTensorsPerLaunch];
    const float* g[kTensorsPerLaunch];
    float* m[kTensorsPerLaunch];
    float* v[kTensorsPerLaunch];
    int64_t n[kTensorsPerLaunch];
};

struct AdamScalars {
    float b1, b2, one_minus_b1, one_minus_b2, step_size, bc2_sqrt, eps, weight_decay;
};

// the arithmetic of torch.optim.Adam: m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2);
// p.addcdiv_(m, sqrt(v)/sqrt(1-b2^t) + eps, -lr/(1-b1^t))
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamScalars& s) {
    if (s.weight_decay != 0.f) g = fmaf(s.weight_decay, p, g);
    m = fmaf(s.one_minus_b1, g - m, m);
    v = fmaf(s.one_minus_b2 * g, g, v * s.b2);
    const float denom = sqrtf(v) / s.bc2_sqrt + s.eps;
    p = fmaf(-s.step_size, m / denom, p);
}

using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(256) void adam_kernel(const AdamBatch b, const AdamScalars s) {
    const int t = blockIdx.y;
    const int64_t n = b.n[t];
    float* __restrict__ p = b.p[t];
    const float* __restrict__ g = b.g[t];
    float* __restrict__ m = b.m[t];
    float* __restrict__ v = b.v[t];
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    const int64_t tid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t n4 = vec ? n >> 2 : 0;
    for (int64_t i = tid; i < n4; i += stride) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pv[e], me = mv[e], ve = vv[e];
            adam_elem(pe, gv[e], me, ve, s);
            pv[e] = pe;
            mv[e] = me;
            vv[e] = ve;
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    for (int64_t i = (n4 << 2) + tid; i < n; i += stride) {
        float pv = p[i], mv = m[i], vv = v[i];
        adam_elem(pv, g[i], mv, vv, s);
        p[i] = pv;
        m[i] = mv;
        v[i] = vv;
    }
}

}  // namespace

extern "C" int ddpm_adam_step(int ntensors, float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const int64_t* numel, double lr, double beta1, double beta2, double eps,
                              double weight_decay, int64_t step, void* stream) {
    DDPM_REQUIRE(ntensors >= 0 && step >= 1 && lr >= 0.0 && eps >= 0.0);
    DDPM_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && weight_decay >= 0.0);
    if (ntensors == 0) return 0;
    DDPM_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel);
    // the scalars as torch.optim.Adam forms them: in double, rounded to float where they meet the tensors
    const double bc1 = 1.0 - std::pow(beta1, static_cast<double>(step));
    const double bc2 = 1.0 - std::pow(beta2, static_cast<double>(step));
    AdamScalars s;
    s.b1 = static_cast<float>(beta1);
    s.b2 = static_cast<float>(beta2);
    s.one_minus_b1 = static_cast<float>(1.0 - beta1);
    s.one_minus_b2 = static_cast<float>(1.0 - beta2);
    s.step_size = static_cast<float>(lr / bc1);
    s.bc2_sqrt = static_cast<float>(std::sqrt(bc2));
    s.eps = static_cast<float>(eps);
    s.weight_decay = static_cast<float>(weight_decay);
    hipStream_t st = ddpm::as_stream(stream);
    for (int t0 = 0; t0 < ntensors; t0 += kTensorsPerLaunch) {
        AdamBatch b;
        const int cnt = ntensors - t0 < kTensorsPerLaunch ? ntensors - t0 : kTensorsPerLaunch;
        int64_t longest = 0;
        for (int i = 0; i < cnt; ++i) {
            DDPM_REQUIRE(numel[t0 + i] >= 0);
            DDPM_REQUIRE(numel[t0 + i] == 0 || (params[t0 + i] && grads[t0 + i] && exp_avg[t0 + i] && exp_avg_sq[t0 + i]));
            b.p[i] = params[t0 + i];
            b.g[i] = grads[t0 + i];
            b.m[i] = exp_avg[t0 + i];
            b.v[i] = exp_avg_sq[t0 + i];
            b.n[i] = numel[t0 + i];
            if (b.n[i] > longest) longest = b.n[i];
        }
        if (longest == 0) continue;
        const int64_t want = (longest + 1023) / 1024;   // one float4 per thread per trip
        const int gx = static_cast<int>(want < kBlocksPerTensor ? want : kBlocksPerTensor);
        adam_kernel<<<dim3(gx, cnt), 256, 0, st>>>(b, s);
    }
    return ddpm::check_launch("adam_kernel");
}

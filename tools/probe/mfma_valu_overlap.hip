// Probe: do VALU instructions issue while the matrix pipe executes v_mfma_f32_32x32x16_bf16?
// Each wave runs ITER rounds of 4 independent MFMAs interleaved with K independent v_fma_f32.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int MF>
__global__ __launch_bounds__(256) void probe(float* out, int iters, long long* cycles) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(threadIdx.x * 0.001f + e);
        b[e] = (__bf16)(threadIdx.x * 0.002f - e);
    }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = threadIdx.x + e;
    const float c1 = out[0], c2 = out[1];
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (MF) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K / 4; ++k) v[(m * (K / 4) + k) & 7] = __builtin_fmaf(v[(m * (K / 4) + k) & 7], c1, c2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int e = 0; e < 8; ++e) s += v[e];
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int K, int MF>
void run(const char* name, int wgs_per_cu) {
    float* out;
    long long* cyc;
    const int blocks = 256 * wgs_per_cu;
    hipMalloc(&out, sizeof(float) * (2 + blocks * 256));
    hipMalloc(&cyc, 8);
    hipMemset(out, 0, sizeof(float) * (2 + blocks * 256));
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<K, MF><<<blocks, 256>>>(out, 100, cyc);
    hipEventRecord(e0);
    probe<K, MF><<<blocks, 256>>>(out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_round_ns = ms * 1e6 / iters;
    printf("%-28s waves/SIMD %d: %.1f ns per round (4 MFMA%s + %d VALU per wave)\n", name, wgs_per_cu, per_round_ns, MF ? "" : " [off]", K);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0, 1>("mfma only", w);
        run<16, 0>("valu 16 only", w);
        run<32, 0>("valu 32 only", w);
        run<16, 1>("mfma + 16 valu", w);
        run<32, 1>("mfma + 32 valu", w);
        run<64, 1>("mfma + 64 valu", w);
    }
    return 0;
}

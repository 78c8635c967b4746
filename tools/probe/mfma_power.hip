// Probe: sustained rate of v_mfma_f32_32x32x16_bf16 when the operands change from one instruction to
// the next (8 operand register sets of pseudo-random data) against constant operands: the chip's clock
// under dense matrix work depends on how much the multiplier inputs toggle.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NSETS>
__global__ __launch_bounds__(256) void probe(float* out, const unsigned* seed, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 a[NSETS], b[NSETS];
    for (int s = 0; s < NSETS; ++s)
        for (int e = 0; e < 4; ++e) {
            // bf16 pairs with exponents near 1.0 and random mantissas/signs
            unsigned x = seed[(threadIdx.x * 37 + s * 11 + e) & 1023], y = seed[(threadIdx.x * 53 + s * 7 + e + 512) & 1023];
            a[s][e] = (x & 0x807f807fu) | 0x3f003f00u;
            b[s][e] = (y & 0x807f807fu) | 0x3f003f00u;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[s % NSETS]), __builtin_bit_cast(bf16x8, b[(s * 3) % NSETS]),
                                                                 acc[s & 3], 0, 0, 0);
        }
        if ((it & 63) == 63)   // keep the accumulators finite
            for (int i = 0; i < 4; ++i)
                for (int e = 0; e < 16; ++e) acc[i][e] *= 1e-3f;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NSETS>
void run(const char* name, int wgs_per_cu, float* out, unsigned* seed) {
    const int blocks = 256 * wgs_per_cu, iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<NSETS><<<blocks, 256>>>(out, seed, 2000);
    hipEventRecord(e0);
    probe<NSETS><<<blocks, 256>>>(out, seed, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_mfma_ns = ms * 1e6 / (iters * 8.0 * wgs_per_cu);
    printf("%-22s waves/SIMD %d: %.2f ns per MFMA per SIMD  = %.0f TFLOP/s bf16 chip-wide (%.2f GHz if 32 clocks each)\n", name, wgs_per_cu, per_mfma_ns,
           32768.0 / per_mfma_ns * 1024 / 1e3, 32.0 / per_mfma_ns);
}

int main() {
    float* out;
    unsigned* seed;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 4);
    hipMalloc(&seed, 4096);
    unsigned h[1024];
    unsigned x = 12345u;
    for (int i = 0; i < 1024; ++i) {
        x = x * 1664525u + 1013904223u;
        h[i] = x ^ (x >> 13);
    }
    hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
    for (int w = 1; w <= 2; ++w) {
        run<1>("constant operands", w, out, seed);
        run<8>("8 operand sets", w, out, seed);
    }
    return 0;
}

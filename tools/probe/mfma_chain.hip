// Probe: rate of v_mfma_f32_32x32x16_bf16 when consecutive instructions accumulate into the SAME registers
// (a dependent chain) against 2 and 4 accumulators taken in turn, at one and two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void probe(float* out, const unsigned* seed, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 a[4], b[4];
    for (int s = 0; s < 4; ++s)
        for (int e = 0; e < 4; ++e) {
            unsigned x = seed[(threadIdx.x * 37 + s * 11 + e) & 1023], y = seed[(threadIdx.x * 53 + s * 7 + e + 512) & 1023];
            a[s][e] = (x & 0x807f807fu) | 0x3f003f00u;
            b[s][e] = (y & 0x807f807fu) | 0x3f003f00u;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
            acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[s & 3]), __builtin_bit_cast(bf16x8, b[(s * 3) & 3]),
                                                                    acc[s % NACC], 0, 0, 0);
        if ((it & 63) == 63)
            for (int i = 0; i < NACC; ++i)
                for (int e = 0; e < 16; ++e) acc[i][e] *= 1e-3f;
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int wgs_per_cu, float* out, unsigned* seed) {
    const int blocks = 256 * wgs_per_cu, iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<NACC><<<blocks, 256>>>(out, seed, 2000);
    hipEventRecord(e0);
    probe<NACC><<<blocks, 256>>>(out, seed, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("accumulators %d  waves/SIMD %d: %.2f ns per MFMA per SIMD\n", NACC, wgs_per_cu, ms * 1e6 / (iters * 8.0 * wgs_per_cu));
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
        run<1>(w, out, seed);
        run<2>(w, out, seed);
        run<4>(w, out, seed);
    }
    return 0;
}

// Probe for the two-plane fp16 scheme: x = hi + lo * 2^-11 with hi = RN_fp16(x), lo = RN_fp16((x - hi) * 2^11);
// a * b ~ ah*bh + (ah*bl + al*bh) * 2^-11 on v_mfma_f32_32x32x16_f16 (three products instead of the six of the
// three-plane bf16 scheme; the dropped al*bl term is below 2^-22 of a*b).
//  Part 1 (accuracy against fp64, C = A * B^T): variant 0 fp32 MFMA; 1 three-plane bf16, fold every 144; 2 fp16 pair,
//  two accumulators (hi*hi and the cross terms), no fold; 3 the same with the hi*hi accumulator folded into a master
//  accumulator every 144 elements of K; 4 fold every 576. Operands at several magnitudes (fp16's narrow exponent range).
//  Part 2 (rate): a loop of independent v_mfma_f32_32x32x16_f16 on changing pseudo-random operands, as mfma_power.hip
//  does for bf16.
// Build: hipcc --offload-arch=gfx950 -O3 -o f16x2_probe f16x2_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ inline void split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    const unsigned u = __float_as_uint(x);
    const float r1 = x - __uint_as_float(u & 0xffff0000u);
    const unsigned u1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
    hi = u >> 16;
    mid = u1 >> 16;
    lo = __float_as_uint(r2) >> 16;
}

__device__ inline void split2(float x, _Float16& hi, _Float16& lo) {
    hi = static_cast<_Float16>(x);
    lo = static_cast<_Float16>((x - static_cast<float>(hi)) * 2048.0f);
}

template <int VARIANT>
__global__ void gemm_kernel(const float* A, const float* B, float* C, int M, int N, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    f32x16 acc, accx, master;
    for (int e = 0; e < 16; ++e) acc[e] = accx[e] = master[e] = 0.f;
    const int fold = VARIANT == 4 ? 576 : 144;
    if (VARIANT == 0) {
        for (int k = 0; k < K; k += 2) {
            const float a = A[(m0 + r) * (size_t)K + k + h], b = B[(n0 + r) * (size_t)K + k + h];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    } else if (VARIANT == 1) {
        for (int k = 0; k < K; k += 16) {
            s16x8 ah, am, al, bh, bm, bl;
            for (int j = 0; j < 8; ++j) {
                unsigned short x, y, z;
                split3(A[(m0 + r) * (size_t)K + k + 8 * h + j], x, y, z);
                ah[j] = x; am[j] = y; al[j] = z;
                split3(B[(n0 + r) * (size_t)K + k + 8 * h + j], x, y, z);
                bh[j] = x; bm[j] = y; bl[j] = z;
            }
            auto mm = [&](s16x8 a, s16x8 b) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
            };
            mm(al, bh); mm(ah, bl); mm(am, bm); mm(am, bh); mm(ah, bm); mm(ah, bh);
            if ((k + 16) % fold == 0)
                for (int e = 0; e < 16; ++e) { master[e] += acc[e]; acc[e] = 0.f; }
        }
        for (int e = 0; e < 16; ++e) acc[e] += master[e];
    } else {
        for (int k = 0; k < K; k += 16) {
            f16x8 ah, al, bh, bl;
            for (int j = 0; j < 8; ++j) {
                _Float16 x, y;
                split2(A[(m0 + r) * (size_t)K + k + 8 * h + j], x, y);
                ah[j] = x; al[j] = y;
                split2(B[(n0 + r) * (size_t)K + k + 8 * h + j], x, y);
                bh[j] = x; bl[j] = y;
            }
            accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accx, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            if (VARIANT >= 3 && (k + 16) % fold == 0)
                for (int e = 0; e < 16; ++e) { master[e] += acc[e]; acc[e] = 0.f; }
        }
        for (int e = 0; e < 16; ++e) acc[e] = (master[e] + acc[e]) + accx[e] * (1.0f / 2048.0f);
    }
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        C[(m0 + row) * (size_t)N + n0 + r] = acc[e];
    }
}

template <int F16, int NSETS>
__global__ __launch_bounds__(256) void rate_kernel(float* out, const unsigned* seed, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 a[NSETS], b[NSETS];
    for (int s = 0; s < NSETS; ++s)
        for (int e = 0; e < 4; ++e) {
            const unsigned x = seed[(threadIdx.x * 37 + s * 11 + e) & 1023], y = seed[(threadIdx.x * 53 + s * 7 + e + 512) & 1023];
            if (F16) {   // fp16 pairs in [0.5, 2) with random mantissas and signs
                a[s][e] = (x & 0x87ff87ffu) | 0x38003800u;
                b[s][e] = (y & 0x87ff87ffu) | 0x38003800u;
            } else {     // bf16 pairs with exponents near 1.0
                a[s][e] = (x & 0x807f807fu) | 0x3f003f00u;
                b[s][e] = (y & 0x807f807fu) | 0x3f003f00u;
            }
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (F16)
                acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[s % NSETS]), __builtin_bit_cast(f16x8, b[(s * 3) % NSETS]), acc[s & 3], 0, 0, 0);
            else
                acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[s % NSETS]), __builtin_bit_cast(bf16x8, b[(s * 3) % NSETS]), acc[s & 3], 0, 0, 0);
        }
        if ((it & 63) == 63)
            for (int i = 0; i < 4; ++i)
                for (int e = 0; e < 16; ++e) acc[i][e] *= 1e-3f;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int F16, int NSETS>
void run_rate(const char* name, int wgs_per_cu, float* out, unsigned* seed) {
    const int blocks = 256 * wgs_per_cu, iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate_kernel<F16, NSETS><<<blocks, 256>>>(out, seed, 2000);
    hipEventRecord(e0);
    rate_kernel<F16, NSETS><<<blocks, 256>>>(out, seed, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_mfma_ns = ms * 1e6 / (iters * 8.0 * wgs_per_cu);
    printf("%-28s waves/SIMD %d: %.2f ns per MFMA per SIMD = %.0f TFLOP/s chip-wide (%.2f GHz if 32 clocks each)\n", name, wgs_per_cu, per_mfma_ns,
           32768.0 / per_mfma_ns * 1024 / 1e3, 32.0 / per_mfma_ns);
}

int main() {
    const int M = 128, N = 128;
    const float mags[4] = {1.0f, 1.0f / 1024, 1.0f / 1048576, 1024.0f};   // operand A magnitude; B is 0.05 times A's distribution
    for (int K : {64, 576, 4608, 9216}) {
        for (int mi = 0; mi < 4; ++mi) {
            if (mi > 0 && K != 576) continue;
            for (int dist = 0; dist < 2; ++dist) {
                std::vector<float> A((size_t)M * K), B((size_t)N * K);
                srand(1);
                auto rnd = [&]() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
                for (auto& v : A) v = (dist ? fabsf(rnd()) : rnd()) * mags[mi];
                for (auto& v : B) v = (dist ? fabsf(rnd()) : rnd()) * 0.05f;
                std::vector<double> ref((size_t)M * N);
                for (int m = 0; m < M; ++m)
                    for (int n = 0; n < N; ++n) {
                        double s = 0;
                        for (int k = 0; k < K; ++k) s += (double)A[(size_t)m * K + k] * B[(size_t)n * K + k];
                        ref[(size_t)m * N + n] = s;
                    }
                float *dA, *dB, *dC;
                hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
                hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
                hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
                std::vector<float> C((size_t)M * N);
                for (int v = 0; v < 5; ++v) {
                    dim3 g(M / 32, N / 32);
                    if (v == 0) gemm_kernel<0><<<g, 64>>>(dA, dB, dC, M, N, K);
                    if (v == 1) gemm_kernel<1><<<g, 64>>>(dA, dB, dC, M, N, K);
                    if (v == 2) gemm_kernel<2><<<g, 64>>>(dA, dB, dC, M, N, K);
                    if (v == 3) gemm_kernel<3><<<g, 64>>>(dA, dB, dC, M, N, K);
                    if (v == 4) gemm_kernel<4><<<g, 64>>>(dA, dB, dC, M, N, K);
                    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
                    double maxerr = 0, rms = 0, scale = 0, bias = 0;
                    for (size_t i = 0; i < C.size(); ++i) {
                        const double e = C[i] - ref[i];
                        maxerr = fmax(maxerr, fabs(e));
                        rms += e * e;
                        bias += e;
                        scale = fmax(scale, fabs(ref[i]));
                    }
                    printf("K=%5d |A|~%.0e dist=%d variant=%d  max|err|/max|ref| = %.3e  rms/max = %.3e  mean err/max = %+.3e\n", K, mags[mi], dist, v,
                           maxerr / scale, sqrt(rms / C.size()) / scale, bias / C.size() / scale);
                }
                hipFree(dA); hipFree(dB); hipFree(dC);
            }
        }
    }
    float* out;
    unsigned* seed;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 4);
    hipMalloc(&seed, 4096);
    unsigned hs[1024];
    unsigned x = 12345u;
    for (int i = 0; i < 1024; ++i) {
        x = x * 1664525u + 1013904223u;
        hs[i] = x ^ (x >> 13);
    }
    hipMemcpy(seed, hs, 4096, hipMemcpyHostToDevice);
    for (int w = 1; w <= 2; ++w) {
        run_rate<0, 8>("bf16, 8 operand sets", w, out, seed);
        run_rate<1, 8>("fp16, 8 operand sets", w, out, seed);
        run_rate<1, 1>("fp16, constant operands", w, out, seed);
    }
    return 0;
}

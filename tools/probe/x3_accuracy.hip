// Accuracy probe: C = A * B^T (A: MxK, B: NxK, fp32) computed three ways and compared with fp64:
//  (1) v_mfma_f32_32x32x2_f32, (2) three-way bf16 split, 6 cross products on v_mfma_f32_32x32x16_bf16,
//  (3) the same with only 3 products (hi*hi, hi*mid, mid*hi) for scale,
//  (4), (5) scheme (2) as the library's kernels run it: the MFMA partial sum is folded into a master accumulator with a
//  round-to-nearest fp32 add every 64 (generic GEMM kernel) or 144 (halo 3x3 kernel: nine taps of a 16-channel chunk)
//  elements of K, because the bf16 MFMA truncates when it accumulates.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    const unsigned u = __float_as_uint(x);
    const float fh = __uint_as_float(u & 0xffff0000u);
    const float r1 = x - fh;
    const unsigned u1 = __float_as_uint(r1);
    const float fm = __uint_as_float(u1 & 0xffff0000u);
    const float r2 = r1 - fm;
    hi = u >> 16;
    mid = u1 >> 16;
    lo = __float_as_uint(r2) >> 16;
}

template <int VARIANT>
__global__ void gemm_kernel(const float* A, const float* B, float* C, int M, int N, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    f32x16 acc, master;
    for (int e = 0; e < 16; ++e) acc[e] = master[e] = 0.f;
    const int fold = VARIANT == 3 ? 64 : 144;
    if (VARIANT == 0) {
        for (int k = 0; k < K; k += 2) {
            const float a = A[(m0 + r) * (size_t)K + k + h], b = B[(n0 + r) * (size_t)K + k + h];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    } else {
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
            if (VARIANT == 2) {
                mm(am, bh); mm(ah, bm); mm(ah, bh);
            } else {  // small terms first
                mm(al, bh); mm(ah, bl); mm(am, bm); mm(am, bh); mm(ah, bm); mm(ah, bh);
            }
            if (VARIANT >= 3 && (k + 16) % fold == 0) {
                for (int e = 0; e < 16; ++e) { master[e] += acc[e]; acc[e] = 0.f; }
            }
        }
        if (VARIANT >= 3)
            for (int e = 0; e < 16; ++e) acc[e] += master[e];
    }
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        C[(m0 + row) * (size_t)N + n0 + r] = acc[e];
    }
}

int main(int argc, char** argv) {
    const int M = 256, N = 256;
    for (int K : {576, 2304, 4608, 9216}) {
        for (int dist = 0; dist < 2; ++dist) {
            std::vector<float> A((size_t)M * K), B((size_t)N * K);
            srand(1);
            auto rnd = [&]() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
            for (auto& v : A) v = dist ? fabsf(rnd()) : rnd();     // dist 1: all positive (no cancellation), like post-ReLU
            for (auto& v : B) v = dist ? fabsf(rnd()) * 0.05f : rnd() * 0.05f;
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
                // also the error of a sequential fp32 CPU sum for reference on v==0 line
                printf("K=%5d dist=%d variant=%d  max|err|/max|ref| = %.3e  rms/max = %.3e  mean err/max = %+.3e\n", K, dist, v,
                       maxerr / scale, sqrt(rms / C.size()) / scale, bias / C.size() / scale);
            }
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    }
    return 0;
}

// Probe: the fp16-pair split written with v_fma_mixlo_f16 / v_fma_mixhi_f16 (4 vector instructions per two elements)
// against the plain sequence (6): same bits for every input? Random values over many exponents, values around fp16's
// limits, subnormal remainders, infinities and NaN, with MODE.FP16_OVFL set as in the kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const f32x2* in, unsigned* ref, unsigned* got, int n) {
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f32x2 v = in[i];
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 rem = (v - __builtin_convertvector(h, f32x2)) * 2048.0f;
    const f16x2 l = __builtin_convertvector(rem, f16x2);
    ref[2 * i] = __builtin_bit_cast(unsigned, h);
    ref[2 * i + 1] = __builtin_bit_cast(unsigned, l);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    const f32x2 s = v * 2048.0f;
    const float c = -2048.0f;
    unsigned lm;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lm) : "v"(hb), "s"(c), "v"(s[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lm) : "v"(hb), "s"(c), "v"(s[1]));
    got[2 * i] = hb;
    got[2 * i + 1] = lm;
}

int main() {
    std::vector<float> v;
    srand(1);
    auto rnd = [] { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
    for (int e = -40; e <= 20; ++e)
        for (int k = 0; k < 20000; ++k) v.push_back(ldexpf(rnd(), e));
    const float edge[] = {0.f, -0.f, 65504.f, 65519.9f, 65520.f, 65536.f, 1e5f, -1e5f, 3e38f, 6.1e-5f, 6.0e-5f, 5.96e-8f, 1e-7f, 1e-10f, 1e-30f,
                          __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), 2047.9999f, 2048.f, 0.33333334f, 1.0f + 1.f / 2048, 1.0f + 1.f / 4096};
    for (float x : edge) { v.push_back(x); v.push_back(-x); v.push_back(x); v.push_back(1.0f); }
    if (v.size() & 1) v.push_back(0.f);
    const int n = (int)v.size() / 2;
    f32x2* din; unsigned *dref, *dgot;
    hipMalloc(&din, n * 8); hipMalloc(&dref, n * 8); hipMalloc(&dgot, n * 8);
    hipMemcpy(din, v.data(), n * 8, hipMemcpyHostToDevice);
    probe<<<(n + 255) / 256, 256>>>(din, dref, dgot, n);
    std::vector<unsigned> r(2 * n), g(2 * n);
    hipMemcpy(r.data(), dref, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(g.data(), dgot, n * 8, hipMemcpyDeviceToHost);
    long diff = 0, nan_only = 0;
    for (int i = 0; i < n; ++i)
        if (r[2 * i + 1] != g[2 * i + 1]) {
            // both NaN with different payloads is not a difference that matters
            auto isnan16 = [](unsigned b) { return (b & 0x7c00) == 0x7c00 && (b & 0x3ff); };
            const unsigned a = r[2 * i + 1], b = g[2 * i + 1];
            if (isnan16(a & 0xffff) == isnan16(b & 0xffff) && isnan16(a >> 16) == isnan16(b >> 16) &&
                (isnan16(a & 0xffff) || (a & 0xffff) == (b & 0xffff)) && (isnan16(a >> 16) || (a >> 16) == (b >> 16))) { ++nan_only; continue; }
            if (diff < 12) printf("inputs %.9g %.9g: hi %08x  lo plain %08x  lo mix %08x\n", v[2 * i], v[2 * i + 1], r[2 * i], a, b);
            ++diff;
        }
    printf("%d pairs: %ld differ (%ld more differ only in NaN payload)\n", n, diff, nan_only);
    return diff != 0;
}

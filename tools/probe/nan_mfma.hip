// Probe: does a NaN operand survive the fp16-pair path? (conversion with MODE.FP16_OVFL set, then v_mfma_f32_32x32x16_f16)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(float bad, float* out, unsigned* bits, int set) {
    if (set) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)1.0f; b[e] = (_Float16)0.5f; }
    const f32x2 v = {lane == 3 ? bad : 1.0f, 1.0f};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    a[0] = h[0];
    if (lane == 3) bits[0] = __builtin_bit_cast(unsigned, h);
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    // row 3 of the product involves lane 3's A elements
    for (int e = 0; e < 16; ++e) out[lane * 16 + e] = c[e];
}

int main() {
    float* dout; unsigned* dbits;
    hipMalloc(&dout, 64 * 16 * 4); hipMalloc(&dbits, 4);
    const float vals[3] = {__builtin_nanf(""), __builtin_inff(), 1e6f};
    for (int set = 0; set < 2; ++set)
        for (float bad : vals) {
            probe<<<1, 64>>>(bad, dout, dbits, set);
            float o[64 * 16]; unsigned bb;
            hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
            hipMemcpy(&bb, dbits, 4, hipMemcpyDeviceToHost);
            int nn = 0, ni = 0; float mx = 0;
            for (float x : o) { if (x != x) ++nn; else if (__builtin_isinf(x)) ++ni; else if (x > mx) mx = x; }
            printf("FP16_OVFL=%d operand %g -> fp16 bits 0x%04x; MFMA outputs: %d NaN, %d inf, largest finite %g\n", set, bad, bb & 0xffff, nn, ni, mx);
        }
    return 0;
}

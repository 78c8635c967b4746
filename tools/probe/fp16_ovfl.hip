// Probe: MODE.FP16_OVFL (bit 23 of the MODE hardware register). With the bit set, a float -> fp16 conversion that
// overflows gives +-65504 instead of infinity (true infinities and NaNs pass through). Prints the conversions of a few
// values with the bit clear and set.
// Build: hipcc --offload-arch=gfx950 -O3 -o fp16_ovfl fp16_ovfl.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const float* in, unsigned* out, int n, int set) {
    if (set) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);   // hwreg(HW_REG_MODE, offset 23, size 1) = 1
    const int i = threadIdx.x;
    if (i < n) {
        const f32x2 v = {in[i], -in[i]};
        const f16x2 h = __builtin_convertvector(v, f16x2);
        const float back = static_cast<float>(h[0]);
        const f16x2 l = __builtin_convertvector((v - __builtin_convertvector(h, f32x2)) * 2048.0f, f16x2);
        out[3 * i] = __builtin_bit_cast(unsigned, h);
        out[3 * i + 1] = __builtin_bit_cast(unsigned, back);
        out[3 * i + 2] = __builtin_bit_cast(unsigned, l);
    }
}

int main() {
    const float h[8] = {1.0f, 65504.0f, 65519.0f, 65520.0f, 1.0e5f, 3.0e38f, __builtin_inff(), __builtin_nanf("")};
    float* din;
    unsigned* dout;
    hipMalloc(&din, sizeof(h));
    hipMalloc(&dout, 8 * 3 * 4);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    for (int set = 0; set < 2; ++set) {
        probe<<<1, 64>>>(din, dout, 8, set);
        unsigned o[24];
        hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
        for (int i = 0; i < 8; ++i)
            printf("FP16_OVFL=%d  x = %-12g -> fp16 pair (x, -x) 0x%08x, back to float %g, remainder pair 0x%08x\n", set, h[i], o[3 * i],
                   __builtin_bit_cast(float, o[3 * i + 1]), o[3 * i + 2]);
    }
    return 0;
}

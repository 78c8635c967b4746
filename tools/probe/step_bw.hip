// Bandwidth probe for the reverse-step kernel's access pattern (3 streams in, 1 out, 16 B per element):
// grid shape, f4 loads per thread and non-temporal hints, at 2^26 elements (every operand from HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ x, const f4* __restrict__ e, const f4* __restrict__ z,
                                         f4* __restrict__ o, long n4, float c1, float c2, float c3) {
    const long stride = (long)gridDim.x * blockDim.x * U;
    for (long i0 = (long)blockIdx.x * blockDim.x * U + threadIdx.x; i0 < n4; i0 += stride) {
        f4 xv[U], ev[U], zv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + (long)u * blockDim.x;
            if (i < n4) {
                if (NT) {
                    xv[u] = __builtin_nontemporal_load(x + i);
                    ev[u] = __builtin_nontemporal_load(e + i);
                    zv[u] = __builtin_nontemporal_load(z + i);
                } else {
                    xv[u] = x[i]; ev[u] = e[i]; zv[u] = z[i];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + (long)u * blockDim.x;
            if (i < n4) {
                f4 r;
                r.x = c1 * (xv[u].x - c2 * ev[u].x) + c3 * zv[u].x;
                r.y = c1 * (xv[u].y - c2 * ev[u].y) + c3 * zv[u].y;
                r.z = c1 * (xv[u].z - c2 * ev[u].z) + c3 * zv[u].z;
                r.w = c1 * (xv[u].w - c2 * ev[u].w) + c3 * zv[u].w;
                if (NT) __builtin_nontemporal_store(r, o + i);
                else o[i] = r;
            }
        }
    }
}

template <int U, bool NT>
void run(const char* tag, int grid, f4* x, f4* e, f4* z, f4* o, long n4) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k<U, NT><<<grid, 256>>>(x, e, z, o, n4, 1.01f, 0.02f, 0.1f);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) k<U, NT><<<grid, 256>>>(x, e, z, o, n4, 1.01f, 0.02f, 0.1f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    ms /= 20;
    printf("%-28s grid %7d: %.4f ms  %.0f GB/s\n", tag, grid, ms, 64.0 * n4 / (ms * 1e-3) / 1e9);
}

int main() {
    const long n4 = 1l << 24;
    f4 *x, *e, *z, *o;
    hipMalloc(&x, n4 * 16); hipMalloc(&e, n4 * 16); hipMalloc(&z, n4 * 16); hipMalloc(&o, n4 * 16);
    hipMemset(x, 0, n4 * 16); hipMemset(e, 0, n4 * 16); hipMemset(z, 0, n4 * 16);
    for (int grid : {2048, 4096, 8192, 16384, 65536}) {
        run<1, false>("U=1", grid, x, e, z, o, n4);
        run<2, false>("U=2", grid, x, e, z, o, n4);
        run<4, false>("U=4", grid, x, e, z, o, n4);
        run<1, true>("U=1 nontemporal", grid, x, e, z, o, n4);
        run<2, true>("U=2 nontemporal", grid, x, e, z, o, n4);
        run<4, true>("U=4 nontemporal", grid, x, e, z, o, n4);
    }
    return 0;
}

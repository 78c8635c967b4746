// Shared host-side helpers for the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/ddpm_mi355x.h"

namespace ddpm {

void set_error(const char* fmt, ...);

inline int fail_invalid(const char* what) {
    set_error("invalid argument: %s", what);
    return DDPM_E_INVALID;
}

// Checks the launch itself (not completion: nothing here synchronises).
inline int check_launch(const char* kernel) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", kernel, hipGetErrorString(e));
        return DDPM_E_LAUNCH;
    }
    return DDPM_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kMaxGridX = 1 << 20;

// DDPM_CONV_MATH=fp32 in the environment (read once) selects the plain fp32 MFMA kernels (and the
// fp32 packed-weight format) instead of the split-bf16 ones: a diagnostic switch for A/B measurements.
inline bool split_math() {
    static const bool on = [] {
        const char* e = getenv("DDPM_CONV_MATH");
        return !(e && strcmp(e, "fp32") == 0);
    }();
    return on;
}

// DDPM_ATTN_SMALL=0 (read once) routes feature maps of at most 64 pixels through the tiled attention
// kernels instead of the one-workgroup-per-sample ones: a diagnostic switch for A/B measurements.
inline bool small_attention() {
    static const bool on = [] {
        const char* e = getenv("DDPM_ATTN_SMALL");
        return !(e && strcmp(e, "0") == 0);
    }();
    return on;
}

#ifdef __HIPCC__
// Workgroups are dispatched in linear order (x fastest) and handed to the 8 XCDs round-robin, and every
// XCD has its own L2. This returns the block index a kernel should WORK on so that consecutive linear
// indices (the workgroups that share a sample's operands) run on ONE XCD and find each other's data in
// its L2, instead of each fetching its own copy over the fabric.
__device__ __forceinline__ dim3 xcd_block() {
    const unsigned nx = gridDim.x, ny = gridDim.y;
    const unsigned total = nx * ny * gridDim.z;
    const unsigned id = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const unsigned q = total >> 3, rr = total & 7, xcd = id & 7;
    const unsigned nid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
    return dim3(nid % nx, (nid / nx) % ny, nid / (nx * ny));
}
#endif

}  // namespace ddpm

#define DDPM_REQUIRE(cond)                          \
    do {                                            \
        if (!(cond)) return ddpm::fail_invalid(#cond); \
    } while (0)

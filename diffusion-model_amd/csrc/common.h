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

}  // namespace ddpm

#define DDPM_REQUIRE(cond)                          \
    do {                                            \
        if (!(cond)) return ddpm::fail_invalid(#cond); \
    } while (0)

// Process-wide state of the library: the arithmetic of the MFMA-fed kernels (switchable at run time) and the
// per-GPU "operand left fp16's range" flag that the fp16-pair kernels raise.
#include <atomic>
#include <mutex>

#include "common.h"

// Word 0: set to 1 by any fp16-pair kernel (or weight-packing kernel) that met an operand beyond +-65504, i.e. one
// that the pair representation clips. Every GPU holds its own copy of the array at its own address.
static __device__ __align__(256) unsigned range_flag_words[64];

namespace ddpm {
namespace {

std::atomic<int> g_math{-1};

int math_from_env() {
    const char* e = getenv("DDPM_CONV_MATH");
    if (e && strcmp(e, "fp32") == 0) return static_cast<int>(kMathFp32);
    if (e && strcmp(e, "bf16x3") == 0) return static_cast<int>(kMathBf16x3);
    return static_cast<int>(kMathF16x2);
}

constexpr int kMaxDevices = 64;
unsigned* g_flag_ptr[kMaxDevices];   // device address of range_flag_words on each GPU (null until first use there)
std::mutex g_flag_mutex;

}  // namespace

int conv_math() {
    int m = g_math.load(std::memory_order_relaxed);
    if (m < 0) {
        m = math_from_env();
        g_math.store(m, std::memory_order_relaxed);
    }
    return m;
}

unsigned* range_flag() {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        set_error("range flag: cannot identify the current device (%s, device %d, at most %d supported)", hipGetErrorString(e), dev,
                  kMaxDevices);
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_flag_mutex);
    if (!g_flag_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(range_flag_words));
        if (e != hipSuccess || !p) {
            set_error("range flag: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return nullptr;
        }
        g_flag_ptr[dev] = static_cast<unsigned*>(p);
    }
    return g_flag_ptr[dev];
}

}  // namespace ddpm

extern "C" int ddpm_conv_math(void) { return ddpm::conv_math(); }

extern "C" int ddpm_set_conv_math(int mode) {
    DDPM_REQUIRE(mode == ddpm::kMathFp32 || mode == ddpm::kMathBf16x3 || mode == ddpm::kMathF16x2);
    ddpm::g_math.store(mode, std::memory_order_relaxed);
    return DDPM_OK;
}

extern "C" int ddpm_range_flag_read(uint32_t* host_dst, int clear, void* stream) {
    unsigned* flag = ddpm::range_flag();
    if (!flag) return DDPM_E_LAUNCH;
    hipStream_t s = ddpm::as_stream(stream);
    if (host_dst) {
        if (hipError_t e = hipMemcpyAsync(host_dst, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, s); e != hipSuccess) {
            ddpm::set_error("ddpm_range_flag_read: hipMemcpyAsync: %s", hipGetErrorString(e));
            return DDPM_E_LAUNCH;
        }
    }
    if (clear) {
        if (hipError_t e = hipMemsetAsync(flag, 0, sizeof(uint32_t), s); e != hipSuccess) {
            ddpm::set_error("ddpm_range_flag_read: hipMemsetAsync: %s", hipGetErrorString(e));
            return DDPM_E_LAUNCH;
        }
    }
    return DDPM_OK;
}

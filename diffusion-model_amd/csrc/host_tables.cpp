// Host-side tables of the diffusion process: beta schedules, alpha / alpha_bar and the
// per-step coefficients. Pure C++ (no GPU work).
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

namespace ddpm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ddpm

extern "C" const char* ddpm_last_error(void) { return ddpm::g_err; }
extern "C" int ddpm_abi_version(void) { return 5; }

// numpy.linspace(start, stop, n) in float64: step = (stop-start)/(n-1), y[i] = i*step + start,
// last element forced to stop.
static void linspace(double start, double stop, int n, double* y) {
    if (n == 1) {
        y[0] = start;
        return;
    }
    const double step = (stop - start) / static_cast<double>(n - 1);
    for (int i = 0; i < n; ++i) y[i] = static_cast<double>(i) * step + start;
    y[n - 1] = stop;
}

extern "C" int ddpm_schedule_betas(int kind, int timesteps, double beta_start, double beta_end,
                                   float* betas) {
    DDPM_REQUIRE(timesteps >= 1 && betas != nullptr);
    const int T = timesteps;
    std::vector<double> b(T);
    if (kind == 0) {
        linspace(beta_start, beta_end, T, b.data());
    } else if (kind == 1) {
        linspace(std::pow(beta_start, 0.5), std::pow(beta_end, 0.5), T, b.data());
        for (int i = 0; i < T; ++i) b[i] = b[i] * b[i];
    } else if (kind == 2) {
        // Nichol-Dhariwal cosine schedule, s = 0.008, clipped to [0, 0.999].
        const double s = 0.008, pi = 3.141592653589793;
        std::vector<double> ac(T + 1);
        for (int i = 0; i <= T; ++i) {
            const double u = ((static_cast<double>(i) / static_cast<double>(T)) + s) / (1.0 + s) * pi * 0.5;
            const double c = std::cos(u);
            ac[i] = c * c;
        }
        const double a0 = ac[0];
        for (int i = 0; i <= T; ++i) ac[i] = ac[i] / a0;
        for (int i = 0; i < T; ++i) {
            double v = 1.0 - (ac[i + 1] / ac[i]);
            if (v < 0.0) v = 0.0;
            if (v > 0.999) v = 0.999;
            b[i] = v;
        }
    } else {
        ddpm::set_error("Invalid scheduler type: %d", kind);
        return DDPM_E_INVALID;
    }
    for (int i = 0; i < T; ++i) betas[i] = static_cast<float>(b[i]);
    return DDPM_OK;
}

extern "C" int ddpm_schedule_tables(int timesteps, const float* betas, float* alphas, float* alpha_bar) {
    DDPM_REQUIRE(timesteps >= 1 && betas && alphas && alpha_bar);
    double acc = 1.0;
    for (int i = 0; i < timesteps; ++i) {
        const float a = 1.0f - betas[i];
        alphas[i] = a;
        acc *= static_cast<double>(a);
        alpha_bar[i] = static_cast<float>(acc);
    }
    return DDPM_OK;
}

extern "C" int ddpm_step_coefficients(float beta_t, float alpha_t, float alpha_bar_t, float c[3]) {
    DDPM_REQUIRE(c != nullptr);
    volatile float s_a = std::sqrt(alpha_t);
    volatile float one_m_ab = 1.0f - alpha_bar_t;
    volatile float s_ab = std::sqrt(one_m_ab);
    volatile float one_m_a = 1.0f - alpha_t;
    c[0] = 1.0f / s_a;
    c[1] = one_m_a / s_ab;
    c[2] = std::sqrt(beta_t);
    return DDPM_OK;
}

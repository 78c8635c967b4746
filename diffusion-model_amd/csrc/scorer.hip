// Scoring of generated images with the small MNIST classifier: what follows the classifier's logits in
// the reference's metric scripts (softmax, per-split Inception-Score terms, mean / covariance of the
// logits for the Frechet distance). The classifier itself runs on the convolution / pooling entry points;
// these are the reductions over its [n][k] output (k = 10 classes), all tiny and latency-bound.
#include "common.h"

using ddpm::as_stream;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxClasses = 16;          // k*k threads finish the covariance
constexpr int kMaxCovBlocks = 256;

// probs[i][:] = softmax(logits[i][:]) in fp32 (max-subtracted), one thread per row.
__global__ __launch_bounds__(kThreads) void softmax_rows_kernel(const float* __restrict__ logits, int64_t n, int k,
                                                                 float* __restrict__ probs) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* row = logits + i * k;
    float v[kMaxClasses];
    float mx = -INFINITY;
    for (int j = 0; j < k; ++j) {
        v[j] = row[j];
        mx = fmaxf(mx, v[j]);
    }
    float s = 0.f;
    for (int j = 0; j < k; ++j) {
        v[j] = expf(v[j] - mx);
        s += v[j];
    }
    for (int j = 0; j < k; ++j) probs[i * k + j] = v[j] / s;
}

// One workgroup per split of `part` consecutive rows: py = column mean of the split's probabilities,
// kl_i = sum_j p_ij * (log p_ij - log py_j), score = exp(mean_i kl_i). Sums run in float64 in a fixed
// order; the elementwise terms are fp32 as in the reference (a probability that underflowed to 0 gives
// 0 * -inf = NaN there too).
__global__ __launch_bounds__(kThreads) void inception_split_kernel(const float* __restrict__ probs, int64_t part, int k,
                                                                    float* __restrict__ scores) {
    __shared__ double red[kThreads];
    __shared__ float logpy[kMaxClasses];
    const float* p = probs + static_cast<int64_t>(blockIdx.x) * part * k;
    for (int j = 0; j < k; ++j) {
        double acc = 0.0;
        for (int64_t i = threadIdx.x; i < part; i += kThreads) acc += p[i * k + j];
        red[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int t = 0; t < kThreads; ++t) s += red[t];
            logpy[j] = logf(static_cast<float>(s / static_cast<double>(part)));
        }
        __syncthreads();
    }
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < part; i += kThreads) {
        float kl = 0.f;
        for (int j = 0; j < k; ++j) {
            const float pv = p[i * k + j];
            kl += pv * (logf(pv) - logpy[j]);
        }
        acc += kl;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int t = 0; t < kThreads; ++t) s += red[t];
        scores[blockIdx.x] = expf(static_cast<float>(s / static_cast<double>(part)));
    }
}

// partial[blk][j] = sum over this block's rows of x[i][j]
__global__ __launch_bounds__(kThreads) void colsum_rows_kernel(const float* __restrict__ x, int64_t n, int k, double* __restrict__ partial) {
    __shared__ double red[kThreads];
    for (int j = 0; j < k; ++j) {
        double acc = 0.0;
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kThreads)
            acc += x[i * k + j];
        red[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int t = 0; t < kThreads; ++t) s += red[t];
            partial[blockIdx.x * k + j] = s;
        }
        __syncthreads();
    }
}

// Every block first finishes the mean (fixed order over the partials), then accumulates its rows'
// centred outer products: covp[blk][a][b] = sum_i (x_ia - mean_a) * (x_ib - mean_b); thread (a, b).
__global__ __launch_bounds__(kThreads) void cov_partial_kernel(const float* __restrict__ x, int64_t n, int k, const double* __restrict__ sum_partial,
                                                                int nblk_sum, double* __restrict__ mean_out, double* __restrict__ covp) {
    __shared__ double mean[kMaxClasses];
    __shared__ float rows[64][kMaxClasses];
    if (threadIdx.x < k) {
        double s = 0.0;
        for (int b = 0; b < nblk_sum; ++b) s += sum_partial[b * k + threadIdx.x];
        mean[threadIdx.x] = s / static_cast<double>(n);
        if (blockIdx.x == 0) mean_out[threadIdx.x] = mean[threadIdx.x];
    }
    __syncthreads();
    const int a = threadIdx.x / k, b = threadIdx.x % k;
    const bool active = threadIdx.x < k * k;
    double acc = 0.0;
    for (int64_t r0 = static_cast<int64_t>(blockIdx.x) * 64; r0 < n; r0 += static_cast<int64_t>(gridDim.x) * 64) {
        const int cnt = static_cast<int>(n - r0 < 64 ? n - r0 : 64);
        for (int e = threadIdx.x; e < cnt * k; e += kThreads) rows[e / k][e % k] = x[r0 * k + e];
        __syncthreads();
        if (active)
            for (int r = 0; r < cnt; ++r) acc += (static_cast<double>(rows[r][a]) - mean[a]) * (static_cast<double>(rows[r][b]) - mean[b]);
        __syncthreads();
    }
    if (active) covp[static_cast<int64_t>(blockIdx.x) * k * k + threadIdx.x] = acc;
}

__global__ __launch_bounds__(kThreads) void cov_final_kernel(const double* __restrict__ covp, int nblk, int64_t n, int k, double* __restrict__ cov) {
    if (threadIdx.x >= k * k) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += covp[static_cast<int64_t>(b) * k * k + threadIdx.x];
    cov[threadIdx.x] = s / static_cast<double>(n - 1);       // unbiased, as numpy.cov
}

}  // namespace

extern "C" int ddpm_softmax_rows(const float* logits, int64_t n, int k, float* probs, void* stream) {
    DDPM_REQUIRE(logits && probs && n > 0 && k > 0 && k <= kMaxClasses);
    const int64_t grid = (n + kThreads - 1) / kThreads;
    DDPM_REQUIRE(grid <= ddpm::kMaxGridX);
    softmax_rows_kernel<<<static_cast<int>(grid), kThreads, 0, as_stream(stream)>>>(logits, n, k, probs);
    return ddpm::check_launch("softmax_rows_kernel");
}

extern "C" int ddpm_inception_split_scores(const float* probs, int64_t n, int k, int splits, float* scores, void* stream) {
    DDPM_REQUIRE(probs && scores && n > 0 && k > 0 && k <= kMaxClasses && splits > 0 && splits <= 65535);
    DDPM_REQUIRE(n / splits > 0);                   // the reference's n // splits rows per split; an empty split has no score
    inception_split_kernel<<<splits, kThreads, 0, as_stream(stream)>>>(probs, n / splits, k, scores);
    return ddpm::check_launch("inception_split_kernel");
}

extern "C" int64_t ddpm_mean_cov_workspace_doubles(int k) { return static_cast<int64_t>(kMaxCovBlocks) * (k + k * k); }

extern "C" int ddpm_mean_cov(const float* x, int64_t n, int k, double* workspace, double* mean, double* cov, void* stream) {
    DDPM_REQUIRE(x && workspace && mean && cov && n > 1 && k > 0 && k <= kMaxClasses);
    hipStream_t s = as_stream(stream);
    int64_t nb = (n + 1023) / 1024;
    if (nb > kMaxCovBlocks) nb = kMaxCovBlocks;
    double* sum_partial = workspace;
    double* covp = workspace + static_cast<int64_t>(kMaxCovBlocks) * k;
    colsum_rows_kernel<<<static_cast<int>(nb), kThreads, 0, s>>>(x, n, k, sum_partial);
    cov_partial_kernel<<<static_cast<int>(nb), kThreads, 0, s>>>(x, n, k, sum_partial, static_cast<int>(nb), mean, covp);
    cov_final_kernel<<<1, kThreads, 0, s>>>(covp, static_cast<int>(nb), n, k, cov);
    return ddpm::check_launch("mean_cov");
}

// Timestep / label conditioning: sinusoidal embedding, then the four time Linears (plus the label
// Embedding and four label Linears when labels are given). One workgroup per (four samples, Linear).
#include <cmath>

#include "common.h"

namespace {

constexpr int kMaxEmb = 1024;

struct EmbedDev {
    int batch, emb_dim, n_labels, t_is_float, n_classes;
    const void* t;
    const int64_t* y;
    const float* label_table;
    int ch[4];
    const float* w[4];
    const float* b[4];
    const float* lw[4];
    const float* lb[4];
    float* out[4];
    float* emb_out;
    float log_step;  // fp32: log(10000) / (half - 1)
};

// A workgroup takes kSamples consecutive samples of one Linear, so that a weight row fetched from L2 serves all of them
// (one sample per workgroup re-read the 1 MB of weights for every sample: 60 us at batch 512, all of it L2 traffic).
constexpr int kSamples = 4;

__global__ __launch_bounds__(256) void time_embed_kernel(const EmbedDev a) {
    __shared__ float emb[kSamples][kMaxEmb];
    __shared__ float lemb[kSamples][kMaxEmb];
    __shared__ int bad[kSamples];
    const int b0 = blockIdx.x * kSamples, which = blockIdx.y;
    const int ns = min(kSamples, a.batch - b0);
    const int E = a.emb_dim, half = E >> 1;
    const bool has_label = a.n_labels > 0;
    for (int idx = threadIdx.x; idx < ns * 2 * half; idx += blockDim.x) {
        const int s = idx / (2 * half), i = idx - s * 2 * half;
        const int b = b0 + s;
        const float tv = a.t_is_float ? static_cast<const float*>(a.t)[b]
                                      : static_cast<float>(static_cast<const int64_t*>(a.t)[b]);
        const int k = i < half ? i : i - half;
        // frequency exp(-k * log_step): the exponent is formed in fp32 as the reference does; the
        // exponential itself is taken in double and rounded once.
        const float ex = -(static_cast<float>(k) * a.log_step);
        const float freq = static_cast<float>(exp(static_cast<double>(ex)));
        const float arg = tv * freq;
        const float v = i < half ? sinf(arg) : cosf(arg);
        emb[s][i] = v;
        if (which == 0 && a.emb_out) a.emb_out[static_cast<int64_t>(b) * E + i] = v;
    }
    if (has_label) {
        for (int idx = threadIdx.x; idx < ns * E; idx += blockDim.x) {
            const int s = idx / E, i = idx - s * E;
            const int64_t cls = a.y[a.n_labels == 1 ? 0 : b0 + s];
            const bool bad_label = cls < 0 || cls >= a.n_classes;      // never dereferenced: the outputs of this sample become NaN
            lemb[s][i] = bad_label ? 0.f : a.label_table[cls * E + i];
            if (i == 0) bad[s] = bad_label ? 1 : 0;
        }
    } else if (threadIdx.x < kSamples) {
        bad[threadIdx.x] = 0;
    }
    __syncthreads();
    const int C = a.ch[which];
    const float* w = a.w[which];
    const float* bias = a.b[which];
    const float* lw = a.lw[which];
    const float* lb = a.lb[which];
    // One output channel per thread; its weight row is read as 16-byte pieces, eight in flight, and added in channel
    // order for every sample (the sum's order, and with it the result's bits, does not depend on how the row is fetched
    // or on how many samples share it).
    const bool vec = (E & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 && (!has_label || (reinterpret_cast<uintptr_t>(lw) & 15) == 0);
    auto dot = [&](const float* __restrict__ row, const float (*x)[kMaxEmb], int n, float (&acc)[kSamples]) {
#pragma unroll
        for (int s = 0; s < kSamples; ++s) acc[s] = 0.f;
        int i = 0;
        if (vec) {
            for (; i + 32 <= n; i += 32) {
                float4 r[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) r[q] = *reinterpret_cast<const float4*>(row + i + 4 * q);
#pragma unroll
                for (int s = 0; s < kSamples; ++s)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        acc[s] += r[q].x * x[s][i + 4 * q];
                        acc[s] += r[q].y * x[s][i + 4 * q + 1];
                        acc[s] += r[q].z * x[s][i + 4 * q + 2];
                        acc[s] += r[q].w * x[s][i + 4 * q + 3];
                    }
            }
        }
        for (; i < n; ++i) {
            const float rv = row[i];
#pragma unroll
            for (int s = 0; s < kSamples; ++s) acc[s] += rv * x[s][i];
        }
    };
    for (int co = threadIdx.x; co < C; co += blockDim.x) {
        float acc[kSamples], l[kSamples];
        dot(w + static_cast<int64_t>(co) * E, emb, 2 * half, acc);
        if (has_label) dot(lw + static_cast<int64_t>(co) * E, lemb, E, l);
#pragma unroll
        for (int s = 0; s < kSamples; ++s) {
            if (s >= ns) continue;
            float v = acc[s] + bias[co];
            if (has_label) v += l[s] + lb[co];
            a.out[which][static_cast<int64_t>(b0 + s) * C + co] = bad[s] ? __builtin_nanf("") : v;
        }
    }
}

__global__ __launch_bounds__(256) void sinusoidal_kernel(const void* t, int t_is_float, int batch, int E, float log_step,
                                                          float* __restrict__ out) {
    const int half = E >> 1;
    const int64_t total = static_cast<int64_t>(batch) * 2 * half;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int i = static_cast<int>(idx % (2 * half));
        const int b = static_cast<int>(idx / (2 * half));
        const float tv = t_is_float ? static_cast<const float*>(t)[b] : static_cast<float>(static_cast<const int64_t*>(t)[b]);
        const int k = i < half ? i : i - half;
        const float ex = -(static_cast<float>(k) * log_step);
        const float arg = tv * static_cast<float>(exp(static_cast<double>(ex)));
        out[idx] = i < half ? sinf(arg) : cosf(arg);
    }
}

// rows 1 .. batch - 1 of the four outputs (and of the embedding itself) = row 0
struct RowCopy {
    float* out[5];
    int width[5];
    int batch;
};
__global__ __launch_bounds__(256) void replicate_row_kernel(const RowCopy a) {
    float* __restrict__ p = a.out[blockIdx.y];
    if (!p) return;
    const int w = a.width[blockIdx.y];
    const int64_t total = static_cast<int64_t>(a.batch - 1) * w;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * 256)
        p[w + i] = p[i % w];
}

static float log_step_for(int emb_dim) {
    volatile float lg = std::log(10000.0f);
    return lg / static_cast<float>(emb_dim / 2 - 1);
}

}  // namespace

extern "C" int ddpm_sinusoidal_embedding(const void* t, int t_is_float, int batch, int emb_dim, float* out, void* stream) {
    DDPM_REQUIRE(t && out && batch > 0 && emb_dim >= 4);
    const int64_t total = static_cast<int64_t>(batch) * (emb_dim / 2) * 2;
    const int grid = static_cast<int>((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    sinusoidal_kernel<<<grid, 256, 0, ddpm::as_stream(stream)>>>(t, t_is_float, batch, emb_dim, log_step_for(emb_dim), out);
    return ddpm::check_launch("sinusoidal_kernel");
}

extern "C" int ddpm_time_embed(const ddpm_embed_args* a, void* stream) {
    DDPM_REQUIRE(a && a->batch > 0 && a->t);
    DDPM_REQUIRE(a->emb_dim >= 4 && (a->emb_dim & 1) == 0 && a->emb_dim <= kMaxEmb);
    DDPM_REQUIRE(a->n_labels == 0 || a->n_labels == 1 || a->n_labels == a->batch);
    DDPM_REQUIRE(a->n_labels == 0 || a->n_classes > 0);
    EmbedDev d;
    d.n_classes = a->n_classes;
    d.batch = a->batch;
    d.emb_dim = a->emb_dim;
    d.n_labels = a->n_labels;
    d.t_is_float = a->t_is_float;
    d.t = a->t;
    d.y = a->y;
    d.label_table = a->label_table;
    d.emb_out = a->emb_out;
    for (int k = 0; k < 4; ++k) {
        DDPM_REQUIRE(a->ch[k] > 0 && a->w[k] && a->b[k] && a->out[k]);
        DDPM_REQUIRE(a->n_labels == 0 || (a->lw[k] && a->lb[k] && a->y && a->label_table));
        d.ch[k] = a->ch[k];
        d.w[k] = a->w[k];
        d.b[k] = a->b[k];
        d.lw[k] = a->lw[k];
        d.lb[k] = a->lb[k];
        d.out[k] = a->out[k];
    }
    d.log_step = log_step_for(a->emb_dim);
    if (a->uniform && a->batch > 1) {
        DDPM_REQUIRE(a->n_labels <= 1);
        d.batch = 1;
        time_embed_kernel<<<dim3(1, 4), 256, 0, ddpm::as_stream(stream)>>>(d);
        RowCopy rc;
        rc.batch = a->batch;
        for (int k = 0; k < 4; ++k) {
            rc.out[k] = a->out[k];
            rc.width[k] = a->ch[k];
        }
        rc.out[4] = a->emb_out;
        rc.width[4] = a->emb_dim;
        const int64_t most = static_cast<int64_t>(a->batch - 1) * (a->ch[3] > a->ch[0] ? a->ch[3] : a->ch[0]);
        const int gx = static_cast<int>((most + 1023) / 1024 > 512 ? 512 : (most + 1023) / 1024);
        replicate_row_kernel<<<dim3(gx < 1 ? 1 : gx, 5), 256, 0, ddpm::as_stream(stream)>>>(rc);
        return ddpm::check_launch("time_embed_kernel (uniform)");
    }
    time_embed_kernel<<<dim3((a->batch + kSamples - 1) / kSamples, 4), 256, 0, ddpm::as_stream(stream)>>>(d);
    return ddpm::check_launch("time_embed_kernel");
}

// Self-attention of a feature map with at most 64 pixels (the UNet's bottleneck: 3x3 to 7x7 pixels with
// 512 channels).  The whole score matrix fits in LDS, so a workgroup owns one sample: q, k, v (and dY in
// the backward) are read from HBM exactly once, the value channels pass through LDS a slice at a time,
// every product is a plain fp32 FMA chain and nothing is recomputed per channel slice.  The tiled kernels
// in attention.hip / attention_bwd.hip spend most of their time on idle waves and per-lane row reads at these sizes.
//
// Same operands and results as ddpm_attention_fwd / ddpm_attention_bwd (reference: SelfAttention.forward,
// unet.py:37-54, and its autograd).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "attention_small.h"
#include "common.h"

namespace {

constexpr int kMaxPixels = 64;
constexpr int kSLD = kMaxPixels + 1;   // score row stride (floats)

// Where these kernels beat the tiled ones (measured at 512 channels, batch 256-512): up to 32 pixels both
// ways, 4-7x; the backward up to 64 pixels when a workgroup may take the whole LDS (49 pixels: 130 against
// 300 us at batch 256; 64 pixels: 400 against 595 us at batch 512); the forward at 49 and 64 pixels is
// faster on the tiled kernels (49 against 62 us, 112 against 200 us).
constexpr int kFwdPixels = 32, kBwdPixels = 64;
int lds_limit(int n) { return (n <= 32 ? 80 : 160) * 1024; }   // small maps: two workgroups per CU

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct SmallFwd {
    const float *qkv, *x, *gamma, *temb;
    float *out, *attn_raw, *lse;
    int n, c, dqp, lq, ldx, ldo, cc;   // cc: value channels resident in LDS at a time
};

struct SmallBwd {
    const float *qkv, *dy, *lse, *e, *gamma;
    float* dqkv;
    int n, c, dqp, lq, ldd, cc;
};

// rows of q and k go to LDS with an odd stride (lanes that differ in the row hit different banks)
__device__ __forceinline__ void stage_qk(const float* __restrict__ rows, int n, int lq, int dq, float* sQ, float* sK) {
    const int qld = dq + 1, per_row = dq >> 1;
    for (int idx = threadIdx.x; idx < n * per_row; idx += 256) {
        const int i = idx / per_row, col = (idx - i * per_row) << 2;
        const f32x4 v = *reinterpret_cast<const f32x4*>(rows + static_cast<int64_t>(i) * lq + col);
        float* dst = col < dq ? sQ + i * qld + col : sK + i * qld + (col - dq);
        dst[0] = v[0];
        dst[1] = v[1];
        dst[2] = v[2];
        dst[3] = v[3];
    }
}

// dst[i][0..cc) = scale * src[i][0..cc) for the n rows of one sample
// (a workgroup is alone on its CU, one wave per SIMD: four loads are requested before the first is used, or every
// iteration would wait out a whole memory latency)
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, int64_t ld, int n, int cc, float scale, float* dst, int dld) {
    const int c4 = cc >> 2, total = n * c4;
    for (int idx0 = threadIdx.x; idx0 < total; idx0 += 4 * 256) {
        f32x4 v[4];
        int i[4], ch[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = idx0 + u * 256;
            i[u] = idx / c4;
            ch[u] = (idx - i[u] * c4) << 2;
            if (idx < total) v[u] = *reinterpret_cast<const f32x4*>(src + i[u] * ld + ch[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (idx0 + u * 256 >= total) break;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[u][e] *= scale;
            *reinterpret_cast<f32x4*>(dst + i[u] * dld + ch[u]) = v[u];
        }
    }
}

__device__ __forceinline__ float qk_dot(const float* q, const float* k, int dq) {
    float acc = 0.f;
    for (int d = 0; d < dq; ++d) acc = fmaf(q[d], k[d], acc);
    return acc;
}

__global__ __launch_bounds__(256) void attn_small_fwd_kernel(const SmallFwd p) {
    extern __shared__ __align__(16) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = p.n, c = p.c, cc = p.cc, dq = p.dqp, qld = dq + 1;
    float* sV = lds;                 // [n][cc]
    float* sQ = sV + n * cc;         // [n][dq+1]
    float* sK = sQ + n * qld;
    float* sS = sK + n * qld;        // [n][65]
    const int64_t pix0 = static_cast<int64_t>(b) * n;
    const float* rows = p.qkv + pix0 * p.lq;
    stage_qk(rows, n, p.lq, dq, sQ, sK);
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += 256) {
        const int i = idx / n, j = idx - i * n;
        sS[i * kSLD + j] = qk_dot(sQ + i * qld, sK + j * qld, dq);
    }
    __syncthreads();
    if (tid < n) {
        float* row = sS + tid * kSLD;
        float m = row[0];
        for (int j = 1; j < n; ++j) m = fmaxf(m, row[j]);
        float l = 0.f;
        for (int j = 0; j < n; ++j) {
            row[j] = expf(row[j] - m);
            l += row[j];
        }
        const float inv = 1.f / l;
        for (int j = 0; j < n; ++j) row[j] *= inv;
        if (p.lse) p.lse[pix0 + tid] = m + logf(l);
    }
    const float g = p.gamma[0];
    const int c4 = cc >> 2;
    for (int c0 = 0; c0 < c; c0 += cc) {
        __syncthreads();   // the scores are final; the previous chunk of V is no longer read
        stage_rows(rows + 2 * dq + c0, p.lq, n, cc, 1.f, sV, cc);
        __syncthreads();
        for (int idx = tid; idx < n * c4; idx += 256) {
            const int i = idx / c4, cl = (idx - i * c4) << 2, ch = c0 + cl;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            int j = 0;
            for (; j + 3 < n; j += 4) {
                float w[4];
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    w[u] = sS[i * kSLD + j + u];
                    v[u] = *reinterpret_cast<const f32x4*>(sV + (j + u) * cc + cl);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(w[u], v[u][e], acc[e]);
            }
            for (; j < n; ++j) {
                const float w = sS[i * kSLD + j];
                const f32x4 v = *reinterpret_cast<const f32x4*>(sV + j * cc + cl);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(w, v[e], acc[e]);
            }
            const int64_t pix = pix0 + i;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + pix * p.ldx + ch);
            f32x4 te = {0.f, 0.f, 0.f, 0.f};
            if (p.temb) te = *reinterpret_cast<const f32x4*>(p.temb + static_cast<int64_t>(b) * c + ch);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (g * acc[e] + xv[e]) + te[e];
            if (p.attn_raw) *reinterpret_cast<f32x4*>(p.attn_raw + pix * c + ch) = acc;
            *reinterpret_cast<f32x4*>(p.out + pix * p.ldo + ch) = y;
        }
    }
}

__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const SmallBwd p) {
    extern __shared__ __align__(16) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = p.n, c = p.c, cc = p.cc, dq = p.dqp, qld = dq + 1;
    const int vld = cc + 8;          // 8 lanes share a (query, key) pair below: a row shift of 8 banks keeps four pairs apart
    float* sV = lds;                 // [n][cc+8]
    float* sG = sV + n * vld;        // gamma * dY, [n][cc+8]
    float* sQ = sG + n * vld;
    float* sK = sQ + n * qld;
    float* sP = sK + n * qld;        // softmax probabilities [n][65]
    float* sT = sP + n * kSLD;       // dP, then dS [n][65]
    float* sL = sT + n * kSLD;       // lse [n]
    float* sD = sL + n;              // gamma * rowsum(dY . O) [n]
    const float g = p.gamma[0];
    const int64_t pix0 = static_cast<int64_t>(b) * n;
    const float* rows = p.qkv + pix0 * p.lq;
    float* drows = p.dqkv + pix0 * p.lq;
    stage_qk(rows, n, p.lq, dq, sQ, sK);
    if (tid < n) {
        sL[tid] = p.lse[pix0 + tid];
        sD[tid] = g * p.e[pix0 + tid];
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += 256) {
        const int i = idx / n, j = idx - i * n;
        sP[i * kSLD + j] = expf(qk_dot(sQ + i * qld, sK + j * qld, dq) - sL[i]);
        sT[i * kSLD + j] = 0.f;
    }
    const int grp = tid >> 3, l8 = tid & 7, c4 = cc >> 2;
    const int pairs = n * n;
    const int nb4 = (n + 3) >> 2, blocks = nb4 * nb4, blocks_up = (blocks + 31) & ~31;
    for (int c0 = 0; c0 < c; c0 += cc) {
        __syncthreads();
        stage_rows(rows + 2 * dq + c0, p.lq, n, cc, 1.f, sV, vld);
        stage_rows(p.dy + pix0 * p.ldd + c0, p.ldd, n, cc, g, sG, vld);
        __syncthreads();
        // dP[i][j] += (gamma dY)[i] . V[j] over this chunk: eight lanes per 4 x 4 block of (query, key) pairs, each over
        // every eighth group of four channels, so that eight row reads feed 64 FMAs; the block -> lane map is the same
        // in every chunk, so lane 0 of the group owns the running sums. Rows past n repeat row n - 1 and are dropped.
        for (int blk = grp; blk < blocks_up; blk += 32) {
            const bool live = blk < blocks;
            const int bi = live ? blk / nb4 : 0, bj = live ? blk - bi * nb4 : 0;
            const float* ga[4];
            const float* va[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ga[u] = sG + min(4 * bi + u, n - 1) * vld;
                va[u] = sV + min(4 * bj + u, n - 1) * vld;
            }
            float acc[4][4];
#pragma unroll
            for (int ui = 0; ui < 4; ++ui)
#pragma unroll
                for (int uj = 0; uj < 4; ++uj) acc[ui][uj] = 0.f;
            for (int ch = l8 << 2; ch < cc; ch += 32) {
                f32x4 a[4], v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = *reinterpret_cast<const f32x4*>(ga[u] + ch);
                    v[u] = *reinterpret_cast<const f32x4*>(va[u] + ch);
                }
#pragma unroll
                for (int ui = 0; ui < 4; ++ui)
#pragma unroll
                    for (int uj = 0; uj < 4; ++uj) {
                        float t = acc[ui][uj];
                        t = fmaf(a[ui][0], v[uj][0], t);
                        t = fmaf(a[ui][1], v[uj][1], t);
                        t = fmaf(a[ui][2], v[uj][2], t);
                        t = fmaf(a[ui][3], v[uj][3], t);
                        acc[ui][uj] = t;
                    }
            }
#pragma unroll
            for (int ui = 0; ui < 4; ++ui)
#pragma unroll
                for (int uj = 0; uj < 4; ++uj) {
                    float t = acc[ui][uj];
                    t += __shfl_xor(t, 1);
                    t += __shfl_xor(t, 2);
                    t += __shfl_xor(t, 4);
                    const int i = 4 * bi + ui, j = 4 * bj + uj;
                    if (live && l8 == 0 && i < n && j < n) sT[i * kSLD + j] += t;
                }
        }
        // dV = P^T (gamma dY) for this chunk: a thread owns four keys x four channels
        for (int idx = tid; idx < nb4 * c4; idx += 256) {
            const int jb = idx / c4, cl = (idx - jb * c4) << 2;
            int jj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) jj[u] = min(4 * jb + u, n - 1);
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < n; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sG + i * vld + cl);
                float w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = sP[i * kSLD + jj[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[u][e] = fmaf(w[u], v[e], acc[u][e]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (4 * jb + u < n)
                    *reinterpret_cast<f32x4*>(drows + static_cast<int64_t>(4 * jb + u) * p.lq + 2 * dq + c0 + cl) = acc[u];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < pairs; idx += 256) {
        const int i = idx / n, j = idx - i * n;
        sT[i * kSLD + j] = sP[i * kSLD + j] * (sT[i * kSLD + j] - sD[i]);
    }
    __syncthreads();
    // dQ = dS K and dK = dS^T Q
    for (int idx = tid; idx < 2 * n * dq; idx += 256) {
        const bool is_k = idx >= n * dq;
        const int r = is_k ? idx - n * dq : idx;
        const int i = r / dq, d = r - i * dq;
        float acc = 0.f;
        if (!is_k) {
            for (int j = 0; j < n; ++j) acc = fmaf(sT[i * kSLD + j], sK[j * qld + d], acc);
        } else {
            for (int j = 0; j < n; ++j) acc = fmaf(sT[j * kSLD + i], sQ[j * qld + d], acc);
        }
        drows[static_cast<int64_t>(i) * p.lq + (is_k ? dq : 0) + d] = acc;
    }
}

template <typename K>
bool allow_lds(K kernel, int bytes) {
    return bytes <= 64 * 1024 ||
           hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}

int fwd_lds_bytes(int n, int cc, int dq) { return 4 * (n * cc + 2 * n * (dq + 1) + n * kSLD); }
int bwd_lds_bytes(int n, int cc, int dq) { return 4 * (2 * n * (cc + 8) + 2 * n * (dq + 1) + 2 * n * kSLD + 2 * n); }

// the widest slice of the value channels (c, c/2, c/4, ...; a multiple of 4) whose working set fits; 0 if none
template <typename F>
int chunk_channels(int n, int c, int dq, int max_pixels, F bytes) {
    if (n > max_pixels) return 0;
    int cc = c;
    while (bytes(n, cc, dq) > lds_limit(n)) {
        if (cc % 8 != 0) return 0;
        cc /= 2;
    }
    return cc;
}

}  // namespace

namespace ddpm {

bool attention_small_fwd_fits(int n, int c, int dq) { return chunk_channels(n, c, dq, kFwdPixels, fwd_lds_bytes) > 0; }
bool attention_small_bwd_fits(int n, int c, int dq) { return chunk_channels(n, c, dq, kBwdPixels, bwd_lds_bytes) > 0; }

int attention_small_fwd(const float* qkv, const float* x, int ldx, const float* gamma, const float* temb, float* out, int ldo,
                        float* attn_raw, float* lse, int batch, int n, int c, int dq, hipStream_t s) {
    const int cc = chunk_channels(n, c, dq, kFwdPixels, fwd_lds_bytes);
    if (cc <= 0) return fail_invalid("attention_small_fwd: shape does not fit one workgroup");
    const int lds = fwd_lds_bytes(n, cc, dq);
    if (!allow_lds(attn_small_fwd_kernel, lds)) return fail_invalid("LDS size for small attention forward");
    SmallFwd p{qkv, x, gamma, temb, out, attn_raw, lse, n, c, dq, 2 * dq + c, ldx, ldo, cc};
    attn_small_fwd_kernel<<<batch, 256, lds, s>>>(p);
    return check_launch("attn_small_fwd");
}

int attention_small_bwd(const float* qkv, const float* dy, int ldd, const float* lse, const float* e, const float* gamma, float* dqkv,
                        int batch, int n, int c, int dq, hipStream_t s) {
    const int cc = chunk_channels(n, c, dq, kBwdPixels, bwd_lds_bytes);
    if (cc <= 0) return fail_invalid("attention_small_bwd: shape does not fit one workgroup");
    const int lds = bwd_lds_bytes(n, cc, dq);
    if (!allow_lds(attn_small_bwd_kernel, lds)) return fail_invalid("LDS size for small attention backward");
    SmallBwd p{qkv, dy, lse, e, gamma, dqkv, n, c, dq, 2 * dq + c, ldd, cc};
    attn_small_bwd_kernel<<<batch, 256, lds, s>>>(p);
    return check_launch("attn_small_bwd");
}

}  // namespace ddpm

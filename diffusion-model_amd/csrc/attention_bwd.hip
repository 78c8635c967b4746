// Backward of the flash-style self-attention (see attention.hip), recomputing the probabilities
// from the saved row log-sum-exp. With  y = gamma*O + x,  O = P V,  P = softmax(Q K^T):
//
//   E_i   = sum_c dY[i][c] * O[i][c]            dgamma = sum_i E_i,   D_i = gamma * E_i
//   dO    = gamma * dY
//   dV[j] = sum_i P_ij dO[i]                    dP_ij = dO[i] . V[j]
//   dS_ij = P_ij (dP_ij - D_i)
//   dQ[i] = sum_j dS_ij K[j]                    dK[j] = sum_i dS_ij Q[i]
//
// Two MFMA kernels avoid atomics: `dkv` gives every wave 32 keys and walks the queries (tiles in
// "lane = key, register = query" orientation, so the query-axis contractions for dV and dK take the
// score registers directly as MFMA B operands); `dq` gives every wave 32 queries and walks the keys
// in the transposed orientation. The x-gradient pass-through (dx += dY) is the caller's.
#include "attention_small.h"
#include "common.h"
#include "conv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TQ = 32;

__device__ __forceinline__ int rowmap(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

struct BwdParams {
    const float* qkv;    // [B][N][lq]
    const float* dy;     // [B][N][.] stride ldd
    const float* o;      // [B][N][C] attention output before the gate
    const float* lse;    // [B][N]
    const float* e;      // [B][N]  rowwise dY.O
    const float* gamma;
    float* dqkv;         // [B][N][lq]
    int n, c, dqp, lq, ldd;
    unsigned* range_flag;   // fp16-pair kernels: raised when an operand lies beyond fp16's range
};

// E[b][i] = dY[i].O[i]; partial sums of E per workgroup for dgamma. A row takes c / 4 lanes (16 bytes each) up to a
// whole wave, so a wave works on 64 / (c / 4) rows at once: with 128 channels one row per wave would leave half of
// the lanes without a load.
__global__ __launch_bounds__(256) void attn_bwd_pre_kernel(const float* __restrict__ dy, int ldd, const float* __restrict__ o, int c,
                                                            int64_t rows, float* __restrict__ e, double* __restrict__ partial) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int lpr = 64;                                   // lanes per row: the power of two that covers c / 4, at most 64
    while (lpr > 1 && (lpr >> 1) * 4 >= c) lpr >>= 1;
    const int rpw = 64 / lpr;                       // rows per wave
    const int sub = lane & (lpr - 1), rw = lane / lpr;
    double acc = 0.0;
    for (int64_t r0 = (static_cast<int64_t>(blockIdx.x) * 4 + wave) * rpw; r0 < rows; r0 += static_cast<int64_t>(gridDim.x) * 4 * rpw) {
        const int64_t r = r0 + rw;
        float s = 0.f;
        if (r < rows) {
            for (int ch = sub * 4; ch < c; ch += lpr * 4) {
                const float4 a = *reinterpret_cast<const float4*>(dy + r * ldd + ch);
                const float4 b = *reinterpret_cast<const float4*>(o + r * c + ch);
                s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
            }
        }
        for (int off = lpr >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (sub == 0 && r < rows) {
            e[r] = s;
            acc += s;
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);      // fixed tree: the sum does not depend on timing
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void attn_bwd_dgamma_kernel(const double* __restrict__ partial, int n, float* __restrict__ dgamma) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) dgamma[0] = static_cast<float>(red[0]);
}

// ---- dK, dV: wave = 32 keys, loop over query tiles -------------------------------------------
// grid (ceil(N/128), ceil(C/CS), B). Every channel-slice block recomputes S and the full dP; only
// slice 0 accumulates dK.
template <int DQ, int NCT>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const BwdParams p) {
    const dim3 bid = blockIdx;   // launch order (measured: the XCD-grouped order is slower for this kernel's short key runs)
    constexpr int CS = 32 * NCT;
    constexpr int QLD = DQ + 4;
    constexpr int DT = (DQ + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int DLD = p.c + 4;
    float* sQ = smem;                  // [TQ][QLD]
    float* sDO = sQ + TQ * QLD;        // [TQ][DLD]   gamma * dY
    float* sL = sDO + TQ * DLD;        // [TQ] lse
    float* sD = sL + TQ;               // [TQ] D_i

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = bid.z;
    const int cbase = bid.y * CS;
    const int j0 = (bid.x * 4 + wave) * 32;
    const bool wave_active = j0 < p.n;
    const int kj = j0 + r;  // this lane's key
    const float g = p.gamma[0];
    const float* base = p.qkv + static_cast<int64_t>(b) * p.n * p.lq;
    const float* dyb = p.dy + static_cast<int64_t>(b) * p.n * p.ldd;

    // K fragment (B operand of S = Q K^T): element e of chunk kk is K[kj][8kk + 4h + e]
    f32x4 kf[DQ / 8];
#pragma unroll
    for (int kk = 0; kk < DQ / 8; ++kk) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kj < p.n) v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(kj) * p.lq + p.dqp + kk * 8 + h * 4);
        kf[kk] = v;
    }
    const float* vrow = base + static_cast<int64_t>(kj < p.n ? kj : 0) * p.lq + 2 * p.dqp;

    f32x16 dv[NCT], dk[DT];
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) dv[t][e] = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) dk[t][e] = 0.f;

    const int qtiles = (p.n + TQ - 1) / TQ;
    const int c4n = p.c >> 2;
    for (int t = 0; t < qtiles; ++t) {
        const int i0 = t * TQ;
        __syncthreads();
        for (int f = tid; f < TQ * (DQ / 4); f += 256) {
            const int row = f / (DQ / 4), c4 = f % (DQ / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i0 + row < p.n) v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(i0 + row) * p.lq + c4 * 4);
            *reinterpret_cast<f32x4*>(sQ + row * QLD + c4 * 4) = v;
        }
        for (int f = tid; f < TQ * c4n; f += 256) {
            const int row = f / c4n, c4 = f % c4n;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i0 + row < p.n) v = *reinterpret_cast<const f32x4*>(dyb + static_cast<int64_t>(i0 + row) * p.ldd + c4 * 4);
            *reinterpret_cast<f32x4*>(sDO + row * DLD + c4 * 4) = v * g;
        }
        if (tid < TQ) {
            const bool ok = i0 + tid < p.n;
            sL[tid] = ok ? p.lse[static_cast<int64_t>(b) * p.n + i0 + tid] : 0.f;
            sD[tid] = ok ? g * p.e[static_cast<int64_t>(b) * p.n + i0 + tid] : 0.f;
        }
        __syncthreads();
        if (!wave_active) continue;

        // S[i][j]: lane = key j, register = query i = rowmap(reg, h)
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < DQ / 8; ++kk) {
            const f32x4 qf = *reinterpret_cast<const f32x4*>(sQ + r * QLD + kk * 8 + h * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[e], kf[kk][e], s, 0, 0, 0);
        }
        // dP[i][j] = dO[i] . V[j]   (contraction over all C channels)
        f32x16 dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) dp[e] = 0.f;
        for (int kk = 0; kk < p.c / 8; ++kk) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sDO + r * DLD + kk * 8 + h * 4);
            const f32x4 vv = *reinterpret_cast<const f32x4*>(vrow + kk * 8 + h * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], vv[e], dp, 0, 0, 0);
        }
        if ((p.c & 7) != 0) {  // C = 4 (mod 8): last half chunk, only the h = 0 lanes carry data
            const int kk = p.c / 8;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (h == 0) {
                a = *reinterpret_cast<const f32x4*>(sDO + r * DLD + kk * 8);
                vv = *reinterpret_cast<const f32x4*>(vrow + kk * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], vv[e], dp, 0, 0, 0);
        }
        // P and dS (in place: s -> P, dp -> dS)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int il = rowmap(e, h);
            const bool ok = (i0 + il < p.n) && (kj < p.n);
            const float pe = ok ? expf(s[e] - sL[il]) : 0.f;
            s[e] = pe;
            dp[e] = pe * (dp[e] - sD[il]);
        }
        // dV^T[c][j] += sum_i dO[i][c] P[i][j]
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int ch = cbase + ct * 32 + r;
            const bool ch_ok = ch < p.c;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float a = ch_ok ? sDO[rowmap(e, h) * DLD + ch] : 0.f;
                dv[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[e], dv[ct], 0, 0, 0);
            }
        }
        // dK^T[d][j] += sum_i Q[i][d] dS[i][j]
        if (bid.y == 0) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = dt * 32 + r;
                const bool d_ok = d < DQ;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float a = d_ok ? sQ[rowmap(e, h) * QLD + d] : 0.f;
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dp[e], dk[dt], 0, 0, 0);
                }
            }
        }
    }

    if (!wave_active || kj >= p.n) return;
    float* drow = p.dqkv + (static_cast<int64_t>(b) * p.n + kj) * p.lq;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int ch = cbase + ct * 32 + gq * 8 + h * 4;
            if (ch >= p.c) continue;
            f32x4 v = {dv[ct][gq * 4], dv[ct][gq * 4 + 1], dv[ct][gq * 4 + 2], dv[ct][gq * 4 + 3]};
            *reinterpret_cast<f32x4*>(drow + 2 * p.dqp + ch) = v;
        }
    if (bid.y == 0) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int d = dt * 32 + gq * 8 + h * 4;
                if (d >= DQ) continue;
                f32x4 v = {dk[dt][gq * 4], dk[dt][gq * 4 + 1], dk[dt][gq * 4 + 2], dk[dt][gq * 4 + 3]};
                *reinterpret_cast<f32x4*>(drow + p.dqp + d) = v;
            }
    }
}

// ---- dQ: wave = 32 queries, loop over key tiles ------------------------------------------------
template <int DQ>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const BwdParams p) {
    const dim3 bid = ddpm::xcd_block();   // the block to work on: neighbours in launch order share an XCD's L2
    constexpr int KLD = DQ + 4;
    constexpr int DT = (DQ + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int VLD = p.c + 4;
    float* sK = smem;             // [32][KLD]
    float* sV = sK + 32 * KLD;    // [32][VLD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = bid.z;
    const int i0 = (bid.x * 4 + wave) * 32;
    const bool wave_active = i0 < p.n;
    const int qi = i0 + r;
    const bool q_ok = qi < p.n;
    const float g = p.gamma[0];
    const float* base = p.qkv + static_cast<int64_t>(b) * p.n * p.lq;
    const float* dyrow = p.dy + (static_cast<int64_t>(b) * p.n + (q_ok ? qi : 0)) * p.ldd;

    f32x4 qf[DQ / 8];
#pragma unroll
    for (int kk = 0; kk < DQ / 8; ++kk) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q_ok) v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(qi) * p.lq + kk * 8 + h * 4);
        qf[kk] = v;
    }
    const float lse_i = q_ok ? p.lse[static_cast<int64_t>(b) * p.n + qi] : 0.f;
    const float d_i = q_ok ? g * p.e[static_cast<int64_t>(b) * p.n + qi] : 0.f;

    f32x16 dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[t][e] = 0.f;

    const int ktiles = (p.n + 31) / 32;
    const int c4n = p.c >> 2;
    for (int t = 0; t < ktiles; ++t) {
        const int j0 = t * 32;
        __syncthreads();
        for (int f = tid; f < 32 * (DQ / 4); f += 256) {
            const int row = f / (DQ / 4), c4 = f % (DQ / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j0 + row < p.n) v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(j0 + row) * p.lq + p.dqp + c4 * 4);
            *reinterpret_cast<f32x4*>(sK + row * KLD + c4 * 4) = v;
        }
        for (int f = tid; f < 32 * c4n; f += 256) {
            const int row = f / c4n, c4 = f % c4n;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j0 + row < p.n) v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(j0 + row) * p.lq + 2 * p.dqp + c4 * 4);
            *reinterpret_cast<f32x4*>(sV + row * VLD + c4 * 4) = v;
        }
        __syncthreads();
        if (!wave_active) continue;

        // S^T[j][i]: lane = query, register = key rowmap(reg, h)
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < DQ / 8; ++kk) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(sK + r * KLD + kk * 8 + h * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[kk][e], s, 0, 0, 0);
        }
        // dP^T[j][i] = V[j] . dO[i]
        f32x16 dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) dp[e] = 0.f;
        for (int kk = 0; kk < p.c / 8; ++kk) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sV + r * VLD + kk * 8 + h * 4);
            f32x4 dd = *reinterpret_cast<const f32x4*>(dyrow + kk * 8 + h * 4);
            dd = dd * g;
#pragma unroll
            for (int e = 0; e < 4; ++e) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], dd[e], dp, 0, 0, 0);
        }
        if ((p.c & 7) != 0) {
            const int kk = p.c / 8;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, dd = {0.f, 0.f, 0.f, 0.f};
            if (h == 0) {
                a = *reinterpret_cast<const f32x4*>(sV + r * VLD + kk * 8);
                dd = *reinterpret_cast<const f32x4*>(dyrow + kk * 8);
                dd = dd * g;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], dd[e], dp, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool ok = q_ok && (j0 + rowmap(e, h) < p.n);
            const float pe = ok ? expf(s[e] - lse_i) : 0.f;
            dp[e] = pe * (dp[e] - d_i);
        }
        // dQ^T[d][i] += sum_j K[j][d] dS[i][j]
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 32 + r;
            const bool d_ok = d < DQ;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float a = d_ok ? sK[rowmap(e, h) * KLD + d] : 0.f;
                dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dp[e], dq[dt], 0, 0, 0);
            }
        }
    }
    if (!wave_active || !q_ok) return;
    float* drow = p.dqkv + (static_cast<int64_t>(b) * p.n + qi) * p.lq;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int d = dt * 32 + gq * 8 + h * 4;
            if (d >= DQ) continue;
            f32x4 v = {dq[dt][gq * 4], dq[dt][gq * 4 + 1], dq[dt][gq * 4 + 2], dq[dt][gq * 4 + 3]};
            *reinterpret_cast<f32x4*>(drow + d) = v;
        }
}

// ---- fast paths for the heavy layers (C = 64 or 128 channels, d_qk = C/8) ------------------------
// Same mathematics as the generic kernels above, but the operand that is fixed for a wave's lifetime
// (its 32 V rows in dkv, its 32 dO rows in dq) lives in registers, and the tiles that stream past
// (Q/dO/lse/D, resp. K/V) are double-buffered in LDS with the next tile's global loads issued before
// the MFMAs of the current one, so one barrier per tile suffices and no load sits on the critical path.
template <int DQ, int CT, int WPS>   // WPS = waves per SIMD the register budget is sized for
__global__ __launch_bounds__(256, WPS) void attn_bwd_dkv_fast_kernel(const BwdParams p) {
    const dim3 bid = ddpm::xcd_block();   // the block to work on: neighbours in launch order share an XCD's L2
    constexpr int NCT = CT / 32;
    constexpr int QLD = DQ + 4, DLD = CT + 4;
    constexpr int DOPT = TQ * CT / 4 / 256;   // dO float4 per thread
    constexpr int QF4 = TQ * DQ / 4;          // Q float4 per tile (<= 256)
    __shared__ __attribute__((aligned(16))) float sQ[2][TQ * QLD];
    __shared__ __attribute__((aligned(16))) float sDO[2][TQ * DLD];
    __shared__ float sL[2][TQ], sD[2][TQ];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int b = bid.z;
    const int j0 = (bid.x * 4 + wave) * 32;
    const bool wave_active = j0 < p.n;
    const int kj = j0 + r;
    const float g = p.gamma[0];
    const float* base = p.qkv + static_cast<int64_t>(b) * p.n * p.lq;
    const float* dyb = p.dy + static_cast<int64_t>(b) * p.n * p.ldd;
    const float* lseb = p.lse + static_cast<int64_t>(b) * p.n;
    const float* eb = p.e + static_cast<int64_t>(b) * p.n;

    f32x4 kf[DQ / 8], vf[CT / 8];
#pragma unroll
    for (int kk = 0; kk < DQ / 8; ++kk) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kj < p.n) v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(kj) * p.lq + DQ + kk * 8 + h * 4);
        kf[kk] = v;
    }
#pragma unroll
    for (int kk = 0; kk < CT / 8; ++kk) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kj < p.n) v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(kj) * p.lq + 2 * DQ + kk * 8 + h * 4);
        vf[kk] = v;
    }

    f32x16 dv[NCT], dk;
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) dv[t][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[e] = 0.f;

    f32x4 doreg[DOPT], qreg;
    float lreg = 0.f, dreg = 0.f;
    auto load_tile = [&](int i0) {
#pragma unroll
        for (int i = 0; i < DOPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (CT / 4), c4 = f % (CT / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i0 + row < p.n) v = *reinterpret_cast<const f32x4*>(dyb + static_cast<int64_t>(i0 + row) * p.ldd + c4 * 4);
            doreg[i] = v * g;
        }
        if (tid < QF4) {
            const int row = tid / (DQ / 4), c4 = tid % (DQ / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i0 + row < p.n) v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(i0 + row) * p.lq + c4 * 4);
            qreg = v;
        }
        if (tid < TQ) {
            const bool ok = i0 + tid < p.n;
            lreg = ok ? lseb[i0 + tid] : 0.f;
            dreg = ok ? g * eb[i0 + tid] : 0.f;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < DOPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (CT / 4), c4 = f % (CT / 4);
            *reinterpret_cast<f32x4*>(&sDO[buf][row * DLD + c4 * 4]) = doreg[i];
        }
        if (tid < QF4) {
            const int row = tid / (DQ / 4), c4 = tid % (DQ / 4);
            *reinterpret_cast<f32x4*>(&sQ[buf][row * QLD + c4 * 4]) = qreg;
        }
        if (tid < TQ) {
            sL[buf][tid] = lreg;
            sD[buf][tid] = dreg;
        }
    };

    const int qtiles = (p.n + TQ - 1) / TQ;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < qtiles; ++t) {
        const int buf = t & 1, i0 = t * TQ;
        const bool more = t + 1 < qtiles;
        if (more) load_tile(i0 + TQ);
        if (wave_active) {
            const float* q = sQ[buf];
            const float* dO = sDO[buf];
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < DQ / 8; ++kk) {
                const f32x4 qf = *reinterpret_cast<const f32x4*>(q + r * QLD + kk * 8 + h * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[e], kf[kk][e], s, 0, 0, 0);
            }
#pragma unroll
            for (int kk = 0; kk < CT / 8; ++kk) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(dO + r * DLD + kk * 8 + h * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], vf[kk][e], dp, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int il = rowmap(e, h);
                const bool ok = (i0 + il < p.n) && (kj < p.n);
                const float pe = ok ? expf(s[e] - sL[buf][il]) : 0.f;
                s[e] = pe;
                dp[e] = pe * (dp[e] - sD[buf][il]);
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    dv[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(dO[rowmap(e, h) * DLD + ct * 32 + r], s[e], dv[ct], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float a = r < DQ ? q[rowmap(e, h) * QLD + r] : 0.f;
                dk = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dp[e], dk, 0, 0, 0);
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    if (!wave_active || kj >= p.n) return;
    float* drow = p.dqkv + (static_cast<int64_t>(b) * p.n + kj) * p.lq;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int ch = ct * 32 + gq * 8 + h * 4;
            f32x4 v = {dv[ct][gq * 4], dv[ct][gq * 4 + 1], dv[ct][gq * 4 + 2], dv[ct][gq * 4 + 3]};
            *reinterpret_cast<f32x4*>(drow + 2 * DQ + ch) = v;
        }
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int d = gq * 8 + h * 4;
        if (d >= DQ) continue;
        f32x4 v = {dk[gq * 4], dk[gq * 4 + 1], dk[gq * 4 + 2], dk[gq * 4 + 3]};
        *reinterpret_cast<f32x4*>(drow + DQ + d) = v;
    }
}

template <int DQ, int CT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_fast_kernel(const BwdParams p) {
    const dim3 bid = ddpm::xcd_block();   // the block to work on: neighbours in launch order share an XCD's L2
    constexpr int KLD = DQ + 4, VLD = CT + 4;
    constexpr int VPT = 32 * CT / 4 / 256;
    constexpr int KF4 = 32 * DQ / 4;
    __shared__ __attribute__((aligned(16))) float sK[2][32 * KLD];
    __shared__ __attribute__((aligned(16))) float sV[2][32 * VLD];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int b = bid.z;
    const int i0 = (bid.x * 4 + wave) * 32;
    const bool wave_active = i0 < p.n;
    const int qi = i0 + r;
    const bool q_ok = qi < p.n;
    const float g = p.gamma[0];
    const float* base = p.qkv + static_cast<int64_t>(b) * p.n * p.lq;

    f32x4 qf[DQ / 8], dof[CT / 8];
#pragma unroll
    for (int kk = 0; kk < DQ / 8; ++kk) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q_ok) v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(qi) * p.lq + kk * 8 + h * 4);
        qf[kk] = v;
    }
#pragma unroll
    for (int kk = 0; kk < CT / 8; ++kk) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q_ok) v = *reinterpret_cast<const f32x4*>(p.dy + (static_cast<int64_t>(b) * p.n + qi) * p.ldd + kk * 8 + h * 4);
        dof[kk] = v * g;
    }
    const float lse_i = q_ok ? p.lse[static_cast<int64_t>(b) * p.n + qi] : 0.f;
    const float d_i = q_ok ? g * p.e[static_cast<int64_t>(b) * p.n + qi] : 0.f;

    f32x16 dq;
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[e] = 0.f;

    f32x4 kreg, vreg[VPT];
    auto load_tile = [&](int j0) {
        if (tid < KF4) {
            const int row = tid / (DQ / 4), c4 = tid % (DQ / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j0 + row < p.n) v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(j0 + row) * p.lq + DQ + c4 * 4);
            kreg = v;
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (CT / 4), c4 = f % (CT / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j0 + row < p.n) v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(j0 + row) * p.lq + 2 * DQ + c4 * 4);
            vreg[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        if (tid < KF4) {
            const int row = tid / (DQ / 4), c4 = tid % (DQ / 4);
            *reinterpret_cast<f32x4*>(&sK[buf][row * KLD + c4 * 4]) = kreg;
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (CT / 4), c4 = f % (CT / 4);
            *reinterpret_cast<f32x4*>(&sV[buf][row * VLD + c4 * 4]) = vreg[i];
        }
    };

    const int ktiles = (p.n + 31) / 32;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ktiles; ++t) {
        const int buf = t & 1, j0 = t * 32;
        const bool more = t + 1 < ktiles;
        if (more) load_tile(j0 + 32);
        if (wave_active) {
            const float* k = sK[buf];
            const float* v = sV[buf];
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < DQ / 8; ++kk) {
                const f32x4 kfr = *reinterpret_cast<const f32x4*>(k + r * KLD + kk * 8 + h * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kfr[e], qf[kk][e], s, 0, 0, 0);
            }
#pragma unroll
            for (int kk = 0; kk < CT / 8; ++kk) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(v + r * VLD + kk * 8 + h * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], dof[kk][e], dp, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool ok = q_ok && (j0 + rowmap(e, h) < p.n);
                const float pe = ok ? expf(s[e] - lse_i) : 0.f;
                dp[e] = pe * (dp[e] - d_i);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float a = r < DQ ? k[rowmap(e, h) * KLD + r] : 0.f;
                dq = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dp[e], dq, 0, 0, 0);
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }
    if (!wave_active || !q_ok) return;
    float* drow = p.dqkv + (static_cast<int64_t>(b) * p.n + qi) * p.lq;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int d = gq * 8 + h * 4;
        if (d >= DQ) continue;
        f32x4 v = {dq[gq * 4], dq[gq * 4 + 1], dq[gq * 4 + 2], dq[gq * 4 + 3]};
        *reinterpret_cast<f32x4*>(drow + d) = v;
    }
}


// ---- split-operand variants of the fast paths -----------------------------------------------------
// Same mathematics and wave ownership as the fast kernels above, on the 16-bit matrix instructions. Tiles that
// stream past are split on their way to LDS; the wave's fixed operand is split once into registers; P and dS are
// split in registers and, being in the accumulator layout (lane = column, register = row), serve as B operands as
// they stand. The operands that the fp32 kernels fetched with one 4-byte LDS read per MFMA (dO^T, Q^T, K^T) come from
// the transposing read ds_read_b64_tr_b16, in the key/query order the score registers hold.
//  * three bf16 planes (F16 = false): every fp32 operand is the exact sum of three bf16 numbers (hi + mid + lo) and a
//    product is evaluated as its six cross terms of weight >= 2^-16 on v_mfma_f32_32x32x16_bf16 with fp32
//    accumulation (dropped terms below 2^-23 of the product);
//  * fp16 pair (F16 = true; conv_common.h): hi and the remainder scaled by 2^11, three products per fragment pair on
//    v_mfma_f32_32x32x16_f16 -- hi*hi into a main accumulator, the two cross products into a second one that is
//    scaled by 2^-11 and added when the sum is complete. The forward pass of a training call runs in the same
//    arithmetic (attention.hip), so that the probabilities recomputed here from its log-sum-exp are the ones it used.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
using ddpm::conv::kLoInv;
using ddpm::conv::mfma16;

__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
    unsigned u[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        u[e] = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(u[e] & 0xffff0000u);
        m[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(m[e] & 0xffff0000u);
        l[e] = __float_as_uint(r2);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = __builtin_amdgcn_perm(u[2 * e + 1], u[2 * e], 0x07060302u);
        mid[e] = __builtin_amdgcn_perm(m[2 * e + 1], m[2 * e], 0x07060302u);
        lo[e] = __builtin_amdgcn_perm(l[2 * e + 1], l[2 * e], 0x07060302u);
    }
}

__device__ __forceinline__ void split4_store(const f32x4 v, unsigned char* dst, int plane_stride) {
    unsigned u[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u[e] = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(u[e] & 0xffff0000u);
        m[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(m[e] & 0xffff0000u);
        l[e] = __float_as_uint(r2);
    }
    u32x2 hi, mid, lo;
    hi[0] = __builtin_amdgcn_perm(u[1], u[0], 0x07060302u);
    hi[1] = __builtin_amdgcn_perm(u[3], u[2], 0x07060302u);
    mid[0] = __builtin_amdgcn_perm(m[1], m[0], 0x07060302u);
    mid[1] = __builtin_amdgcn_perm(m[3], m[2], 0x07060302u);
    lo[0] = __builtin_amdgcn_perm(l[1], l[0], 0x07060302u);
    lo[1] = __builtin_amdgcn_perm(l[3], l[2], 0x07060302u);
    *reinterpret_cast<u32x2*>(dst) = hi;
    *reinterpret_cast<u32x2*>(dst + plane_stride) = mid;
    *reinterpret_cast<u32x2*>(dst + 2 * plane_stride) = lo;
}

// The same for the fp16 pair: two planes; `amax` is the thread's range watch (conv_common.h).
template <class W>
__device__ __forceinline__ void split8_f16(const float (&v)[8], u32x4& hi, u32x4& lo, W& amax) {
    unsigned hh[4], ll[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const ddpm::conv::f32x2 x = {v[2 * e], v[2 * e + 1]};
        ddpm::conv::watch2(amax, x);
        ddpm::conv::split2_pair(x, hh[e], ll[e]);
    }
    hi = u32x4{hh[0], hh[1], hh[2], hh[3]};
    lo = u32x4{ll[0], ll[1], ll[2], ll[3]};
}

template <class W>
__device__ __forceinline__ void split4_store_f16(const f32x4 v, unsigned char* dst, int plane_stride, W& amax) {
    unsigned h0, l0, h1, l1;
    ddpm::conv::watch2(amax, ddpm::conv::f32x2{v[0], v[1]});
    ddpm::conv::watch2(amax, ddpm::conv::f32x2{v[2], v[3]});
    ddpm::conv::split2_pair(ddpm::conv::f32x2{v[0], v[1]}, h0, l0);
    ddpm::conv::split2_pair(ddpm::conv::f32x2{v[2], v[3]}, h1, l1);
    *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(dst + plane_stride) = u32x2{l0, l1};
}

// 8 floats at p (two float4; zeros when !ok), times `scale` -> the operand's planes (pl[2] unused by the fp16 pair)
template <bool F16, class W>
__device__ __forceinline__ void load_split8(const float* p, bool ok, u32x4 (&pl)[F16 ? 2 : 3], float scale, W& amax) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (ok) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = a[j] * scale;
            v[4 + j] = b[j] * scale;
        }
    }
    if constexpr (F16) split8_f16(v, pl[0], pl[1], amax);
    else split8(v, pl[0], pl[1], pl[2]);
}

template <bool F16, class W>
__device__ __forceinline__ void split8_any(const float (&v)[8], u32x4 (&pl)[F16 ? 2 : 3], W& amax) {
    if constexpr (F16) split8_f16(v, pl[0], pl[1], amax);
    else split8(v, pl[0], pl[1], pl[2]);
}

template <bool F16, class W>
__device__ __forceinline__ void split4_store_any(const f32x4 v, unsigned char* dst, int plane_stride, W& amax) {
    if constexpr (F16) split4_store_f16(v, dst, plane_stride, amax);
    else split4_store(v, dst, plane_stride);
}

__device__ __forceinline__ u32x4 tr_pair(const unsigned char* a0, const unsigned char* a1) {
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
    return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
}

// The products of one fragment pair, in issue order: plane of the A operand, plane of the B operand, and (fp16 pair)
// whether the product is a cross term, which goes to the cross accumulator.
template <bool F16>
struct Products {
    static constexpr int N = F16 ? 3 : 6;
    __host__ __device__ static constexpr int pa(int k) {
        constexpr int t[6] = {F16 ? 1 : 2, 0, F16 ? 0 : 1, 1, 0, 0};
        return t[k];
    }
    __host__ __device__ static constexpr int pb(int k) {
        constexpr int t[6] = {0, F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0};
        return t[k];
    }
    __host__ __device__ static constexpr bool cross(int k) { return F16 && k < 2; }
};

// 256-byte rows of 16-bit elements read both row-wise (ds_read_b128) and transposed (ds_read_b64_tr_b16):
// 16-byte chunk ch of row `row` lives at this offset; both kinds of read are then bank-conflict-free.
__device__ __forceinline__ int swz(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// Key-side gradients of 32 keys per wave. DK: dK (needs dP, hence V of all CT = C channels in
// registers; the dO tile is only read row-wise and sits in padded rows). DV: dV for the CT channels
// starting at bid.y * CT (CT <= 128; the dO tile is read transposed and sits in swizzled 256-byte
// rows). Both at once when the registers allow (C <= 128); wider layers run the two halves as separate
// launches, each recomputing the scores (depth-DQ products, cheap beside the C-deep ones).
// SW: this workgroup watches the range of the dO and Q tiles that stream past (every key block of a sample stages the
// same ones: the first does the checking for all); its own K, V and dS it always watches.
template <int DQ, int CT, bool DK, bool DV, bool F16, int SW>   // SW: 0 = the streamed tiles are not watched, 1 = watched, 2 = watched by `amax` itself
__device__ __forceinline__ void attn_bwd_dkv_x3_body(const BwdParams& p, const dim3 bid) {
    if constexpr (F16) ddpm::conv::fp16_saturate();
    ddpm::conv::RangeWatch amax;                     // F16: largest magnitude this thread has turned into an fp16 pair
    std::conditional_t<SW == 1, ddpm::conv::RangeWatch, ddpm::conv::NoWatch> swatch_own;   // the same for the streamed tiles
    auto& swatch = [&]() -> auto& {
        if constexpr (SW == 2) return amax;
        else return swatch_own;
    }();
    static_assert(!DV || CT <= 128, "transposed dO reads assume 256-byte rows");
    using PR = Products<F16>;
    constexpr int NPL = F16 ? 2 : 3;
    constexpr int NCT = CT / 32, KCV = CT / 16;
    constexpr int KD = DQ < 16 ? 16 : DQ, KQ = KD / 16;   // padded depth, 16-deep chunks of it
    constexpr int QROW = 2 * KD, QPLANE = 32 * QROW;      // bytes
    constexpr int DROW = DV ? 256 : CT * 2 + 16, DPLANE = 32 * DROW;
    constexpr int DOPT = TQ * CT / 4 / 256;               // dO float4 per thread
    constexpr int QF4 = TQ * DQ / 4;                      // Q float4 per tile (<= 256)
    extern __shared__ __attribute__((aligned(256))) unsigned char dyn_lds[];
    unsigned char* sDO = dyn_lds;                          // [2][NPL * DPLANE]
    unsigned char* sQ = sDO + 2 * NPL * DPLANE;            // [2][NPL * QPLANE]
    float* sL = reinterpret_cast<float*>(sQ + 2 * NPL * QPLANE);   // [2][TQ]
    float* sD = sL + 2 * TQ;                               // [2][TQ]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int b = bid.z;
    const int c0 = bid.y * CT;
    const int j0 = (bid.x * 4 + wave) * 32;
    const bool wave_active = j0 < p.n;
    const int kj = j0 + r;
    const bool k_ok = kj < p.n;
    const float g = p.gamma[0];
    const float* base = p.qkv + static_cast<int64_t>(b) * p.n * p.lq;
    const float* dyb = p.dy + static_cast<int64_t>(b) * p.n * p.ldd + c0;
    const float* lseb = p.lse + static_cast<int64_t>(b) * p.n;
    const float* eb = p.e + static_cast<int64_t>(b) * p.n;

    if (DQ < 16) {   // the upper half of the padded depth of the Q tile stays zero
        for (int i = tid; i < 2 * NPL * QPLANE / 4; i += 256) reinterpret_cast<unsigned*>(sQ)[i] = 0u;
    }

    // fixed operands of this wave's 32 keys: lane (key, h) holds depth / channels 16*chunk + 8h .. + 7
    u32x4 kf[KQ][NPL], vf[DK ? KCV : 1][NPL];
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq)
        load_split8<F16>(base + static_cast<int64_t>(kj) * p.lq + DQ + 16 * kq + 8 * h, k_ok && 16 * kq + 8 * h < DQ, kf[kq], 1.0f, amax);
    if (DK) {
#pragma unroll
        for (int kc = 0; kc < KCV; ++kc)
            load_split8<F16>(base + static_cast<int64_t>(kj) * p.lq + 2 * DQ + 16 * kc + 8 * h, k_ok, vf[kc], 1.0f, amax);
    }

    // An MFMA that accumulates into the registers the previous one wrote waits for it (a chain of dependent
    // 32x32x16 MFMAs runs at half the rate of two interleaved chains), so every product below rotates over two or
    // three accumulators. bf16 planes: dK keeps one per 16-query chunk and adds them at the end, the scores and dP keep
    // two partial sums per tile, dV takes its channel blocks in pairs. fp16 pair: main and cross accumulators alternate
    // by themselves; where two chunks run side by side each has its own cross accumulator.
    f32x16 dv[DV ? NCT : 1], dvx[(DV && F16) ? NCT : 1], dk2[2];   // fp16 pair: dk2[0] main, dk2[1] cross products
#pragma unroll
    for (int t = 0; t < (DV ? NCT : 1); ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            dv[t][e] = 0.f;
            if constexpr (DV && F16) dvx[t][e] = 0.f;
        }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        dk2[0][e] = 0.f;
        dk2[1][e] = 0.f;
    }

    auto dofs = [](int row, int ch) { return DV ? swz(row, ch) : row * DROW + 16 * ch; };

    // The tiles that stream past come through range-checked descriptors over the sample's rows: rows past the last
    // query read as zero without a branch (with branches around its loads the compiler cannot count the outstanding
    // ones and waits for all of them at the first use)
    const __amdgpu_buffer_rsrc_t rs_dy = ddpm::buffer_over(dyb, (static_cast<long long>(p.n) * p.ldd - c0) * 4ll);
    const __amdgpu_buffer_rsrc_t rs_q = ddpm::buffer_over(base, static_cast<long long>(p.n) * p.lq * 4ll);
    const __amdgpu_buffer_rsrc_t rs_l = ddpm::buffer_over(lseb, p.n * 4ll);
    const __amdgpu_buffer_rsrc_t rs_e = ddpm::buffer_over(eb, p.n * 4ll);
    f32x4 doreg[DOPT], qreg;
    float lreg = 0.f, dreg = 0.f;
    auto load_tile = [&](int i0) {
#pragma unroll
        for (int i = 0; i < DOPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (CT / 4), c4 = f % (CT / 4);
            doreg[i] = ddpm::buffer_load4(rs_dy, static_cast<unsigned>((i0 + row) * p.ldd + c4 * 4) * 4u);
        }
        {
            const int row = tid / (DQ / 4), c4 = tid % (DQ / 4);
            qreg = ddpm::buffer_load4(rs_q, tid < QF4 ? static_cast<unsigned>((i0 + row) * p.lq + c4 * 4) * 4u : ddpm::kOob);
        }
        const unsigned lo = tid < TQ ? static_cast<unsigned>(i0 + tid) * 4u : ddpm::kOob;
        lreg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_l, lo, 0, 0));
        dreg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_e, lo, 0, 0));
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < DOPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (CT / 4), c4 = f % (CT / 4);
            split4_store_any<F16>(doreg[i] * g, sDO + buf * NPL * DPLANE + dofs(row, c4 >> 1) + 8 * (c4 & 1), DPLANE, swatch);
        }
        if (tid < QF4) {
            const int row = tid / (DQ / 4), c4 = tid % (DQ / 4);
            split4_store_any<F16>(qreg, sQ + buf * NPL * QPLANE + row * QROW + c4 * 8, QPLANE, swatch);
        }
        if (tid < TQ) {
            sL[buf * TQ + tid] = lreg;
            sD[buf * TQ + tid] = g * dreg;
        }
    };

    // lane-constant parts of the LDS addresses
    const int sw_row = DV ? (((r & 3) << 2) | ((r >> 2) & 3)) : 0;        // swizzle of row r (row reads)
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;   // transposed reads: row quartet, column quad, column group
    // transposed read of a 32-row tile: rows 16*ck + 4h + tq (+8 for the second half); row & 3 = tq, (row >> 2) & 3 = h (+2)
    const int tsw0 = (tq << 2) | (h & 3), tsw1 = (tq << 2) | ((h + 2) & 3);
    // Q tile: QROW-byte rows of KD depth values; with KD = 16 the second column group (depth rows >= 16) is never stored
    const int qtr = (4 * h + tq) * QROW + tp * 8 + (KD > 16 ? tg * 32 : 0);

    ddpm::conv::RangeWatch amax_ds;   // largest |dS| split by this lane; counts only if the lane's key exists
    const int qtiles = (p.n + TQ - 1) / TQ;
    load_tile(0);
    __syncthreads();
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < qtiles; ++t) {
        const int buf = t & 1, i0 = t * TQ;
        const bool more = t + 1 < qtiles;
        if (more) load_tile(i0 + TQ);
        if (wave_active) {
            const unsigned char* q = sQ + buf * NPL * QPLANE;
            const unsigned char* dO = sDO + buf * NPL * DPLANE;
            // bf16 planes: s + s1 and dp + dp1 are two partial sums each; fp16 pair: s1 and dp1 hold the cross
            // products (scaled and added below), dp2 the cross products of the second chunk of a pair
            f32x16 s, s1, dp, dp1, dp2;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] = 0.f;
                s1[e] = 0.f;
                dp[e] = 0.f;
                dp1[e] = 0.f;
                if constexpr (F16) dp2[e] = 0.f;
            }
            auto mma = [&](f32x16& acc, const u32x4& a, const u32x4& bb) { acc = mfma16<F16>(a, bb, acc); };
            // S = Q K^T: rows = queries (this tile), columns = keys (lanes)
            u32x4 aq[KQ][NPL];
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) aq[kq][pl] = *reinterpret_cast<const u32x4*>(q + pl * QPLANE + r * QROW + kq * 32 + h * 16);
            if constexpr (F16) {
                // (the pair's scores are three products per depth chunk: they go first, so that their cross accumulator
                // is free again before the accumulators of dP are taken)
#pragma unroll
                for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
                    for (int k = 0; k < PR::N; ++k) mma(PR::cross(k) ? s1 : s, aq[kq][PR::pa(k)], kf[kq][PR::pb(k)]);
                s = s + s1 * kLoInv;
            }
            if (DK) {
                // dP = dO V^T, two 16-channel chunks at a time; (bf16 planes) the MFMAs of the scores go between the first of them
                static_assert(!DK || (KCV % 2 == 0 && KQ <= KCV / 2), "chunk pairs; the scores fit between them");
#pragma unroll
                for (int kc = 0; kc < KCV; kc += 2) {
                    u32x4 a[2][NPL];
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl)
                            a[c2][pl] = *reinterpret_cast<const u32x4*>(dO + pl * DPLANE + r * DROW + 16 * ((2 * (kc + c2) + h) ^ sw_row));
#pragma unroll
                    for (int k = 0; k < PR::N; ++k) {
                        const bool x = PR::cross(k);
                        if (!F16 && kc / 2 < KQ) mma(s, aq[kc / 2 < KQ ? kc / 2 : 0][PR::pa(k)], kf[kc / 2 < KQ ? kc / 2 : 0][PR::pb(k)]);
                        if constexpr (F16) {
                            mma(x ? dp1 : dp, a[0][PR::pa(k)], vf[kc][PR::pb(k)]);
                            mma(x ? dp2 : dp, a[1][PR::pa(k)], vf[kc + 1][PR::pb(k)]);
                        } else {
                            mma(dp, a[0][PR::pa(k)], vf[kc][PR::pb(k)]);
                            mma(dp1, a[1][PR::pa(k)], vf[kc + 1][PR::pb(k)]);
                        }
                    }
                }
                if constexpr (F16) dp = dp + (dp1 + dp2) * kLoInv;
                else dp += dp1;
            } else if constexpr (!F16) {
#pragma unroll
                for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
                    for (int k = 0; k < PR::N; ++k) mma((k & 1) ? s1 : s, aq[kq][PR::pa(k)], kf[kq][PR::pb(k)]);
                s += s1;
            }
            // P and dS, split into planes per 16-query chunk (register e <-> query rowmap(e, h))
            u32x4 pp[2][NPL], ds[2][NPL];
#pragma unroll
            for (int ck = 0; ck < 2; ++ck) {
                float pv[8], dsv[8];
#pragma unroll
                for (int g4 = 0; g4 < 2; ++g4) {
                    const int il0 = 16 * ck + 8 * g4 + 4 * h;   // queries il0 .. il0 + 3 = registers 8ck + 4g4 .. + 3
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(&sL[buf * TQ + il0]);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(&sD[buf * TQ + il0]);
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int e = 8 * ck + 4 * g4 + e4;
                        // No masks here. A query past the end has zero rows of Q and dO and zero lse and D (range-checked
                        // loads): P = 1 and dS = 0 there, and its column of dO^T / Q^T is zero. A key past the end only
                        // ever affects ITS OWN output column (lane = key is the B operand's column), which is not stored.
                        const float pe = __builtin_amdgcn_exp2f((s[e] - l4[e4]) * 1.4426950408889634f);
                        pv[4 * g4 + e4] = pe;
                        dsv[4 * g4 + e4] = pe * (dp[e] - d4[e4]);
                    }
                }
                // (range watch: probabilities are at most 1 on real keys and need none; a key past the end holds
                // exp(-lse), possibly infinite, in a column that is never stored, and must not raise the flag either)
                ddpm::conv::RangeWatch unwatched;
                if (DV) split8_any<F16>(pv, pp[ck], unwatched);
                if (DK) split8_any<F16>(dsv, ds[ck], amax_ds);
            }
            if (DV) {
                // dV^T += dO^T P : A = dO^T through transposed reads (rows = channels of block ct, k = queries)
                static_assert(!DV || NCT % 2 == 0, "channel blocks are taken in pairs");
#pragma unroll
                for (int ck = 0; ck < 2; ++ck) {
#pragma unroll
                    for (int ct = 0; ct < NCT; ct += 2) {
                        u32x4 a[2][NPL];
                        const int row0 = 16 * ck + 4 * h + tq;
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) {
                            const int ch0 = 4 * (ct + c2) + 2 * tg + (tp >> 1);
#pragma unroll
                            for (int pl = 0; pl < NPL; ++pl) {
                                const unsigned char* pbase = dO + pl * DPLANE + 8 * (tp & 1);
                                a[c2][pl] = tr_pair(pbase + 256 * row0 + 16 * (ch0 ^ tsw0), pbase + 256 * (row0 + 8) + 16 * (ch0 ^ tsw1));
                            }
                        }
#pragma unroll
                        for (int k = 0; k < PR::N; ++k) {
                            if (PR::cross(k)) {
                                mma(dvx[F16 ? ct : 0], a[0][PR::pa(k)], pp[ck][PR::pb(k)]);
                                mma(dvx[F16 ? ct + 1 : 0], a[1][PR::pa(k)], pp[ck][PR::pb(k)]);
                            } else {
                                mma(dv[ct], a[0][PR::pa(k)], pp[ck][PR::pb(k)]);
                                mma(dv[ct + 1], a[1][PR::pa(k)], pp[ck][PR::pb(k)]);
                            }
                        }
                    }
                }
            }
            if (DK) {   // dK^T += Q^T dS, the two 16-query chunks side by side
                u32x4 a[2][NPL];
#pragma unroll
                for (int ck = 0; ck < 2; ++ck)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                        const unsigned char* pbase = q + pl * QPLANE + 16 * ck * QROW + qtr;
                        a[ck][pl] = tr_pair(pbase, pbase + 8 * QROW);
                    }
                if constexpr (F16) {
                    // six products, main and cross accumulator as far as possible in turn
                    mma(dk2[1], a[0][1], ds[0][0]);
                    mma(dk2[0], a[0][0], ds[0][0]);
                    mma(dk2[1], a[0][0], ds[0][1]);
                    mma(dk2[0], a[1][0], ds[1][0]);
                    mma(dk2[1], a[1][1], ds[1][0]);
                    mma(dk2[1], a[1][0], ds[1][1]);
                } else {
#pragma unroll
                    for (int k = 0; k < PR::N; ++k) {
                        mma(dk2[0], a[0][PR::pa(k)], ds[0][PR::pb(k)]);
                        mma(dk2[1], a[1][PR::pa(k)], ds[1][PR::pb(k)]);
                    }
                }
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    if constexpr (F16) ddpm::conv::raise_range_flag(p.range_flag, ddpm::conv::merged(k_ok ? ddpm::conv::merged(amax, amax_ds) : amax, swatch));
    if (!wave_active || !k_ok) return;
    float* drow = p.dqkv + (static_cast<int64_t>(b) * p.n + kj) * p.lq;
    if (DV) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            if constexpr (F16) dv[ct] = dv[ct] + dvx[ct] * kLoInv;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int ch = c0 + ct * 32 + gq * 8 + h * 4;
                f32x4 v = {dv[ct][gq * 4], dv[ct][gq * 4 + 1], dv[ct][gq * 4 + 2], dv[ct][gq * 4 + 3]};
                *reinterpret_cast<f32x4*>(drow + 2 * DQ + ch) = v;
            }
        }
    }
    if (DK) {
        const f32x16 dk = F16 ? dk2[0] + dk2[1] * kLoInv : dk2[0] + dk2[1];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int d = gq * 8 + h * 4;
            if (d >= DQ) continue;
            f32x4 v = {dk[gq * 4], dk[gq * 4 + 1], dk[gq * 4 + 2], dk[gq * 4 + 3]};
            *reinterpret_cast<f32x4*>(drow + DQ + d) = v;
        }
    }
}

template <int DQ, int CT, bool DK, bool DV, int WPS = 1, bool F16 = false>
__global__ __launch_bounds__(256, WPS) void attn_bwd_dkv_x3_kernel(const BwdParams p) {
    const dim3 bid = ddpm::xcd_block();   // the block to work on: neighbours in launch order share an XCD's L2
    // (the dK kernel of 128 channels holds V of all channels in registers and has none to spare for a second copy of
    // its code: there every block keeps watching what it stages)
    constexpr bool kTwoBodies = F16 && !(DK && CT > 64);
    if constexpr (!kTwoBodies) {
        attn_bwd_dkv_x3_body<DQ, CT, DK, DV, F16, F16 ? 2 : 0>(p, bid);
    } else {
        if (bid.x == 0) attn_bwd_dkv_x3_body<DQ, CT, DK, DV, F16, 2>(p, bid);
        else attn_bwd_dkv_x3_body<DQ, CT, DK, DV, F16, 0>(p, bid);
    }
}

template <int DQ, int CT, bool DK, bool DV, bool F16 = false>
constexpr int dkv_x3_lds() {
    constexpr int KD = DQ < 16 ? 16 : DQ, NPL = F16 ? 2 : 3;
    return 2 * NPL * 32 * (DV ? 256 : CT * 2 + 16) + 2 * NPL * 32 * 2 * KD + 4 * TQ * 4;
}

// Query-side gradient of 32 queries per wave (dO of all CT = C channels in registers).
// SW: this workgroup watches the range of the K and V tiles that stream past (as above: the first query block of a sample
// does it for all); its own Q, dO and dS it always watches.
template <int DQ, int CT, bool F16, bool SW>
__device__ __forceinline__ void attn_bwd_dq_x3_body(const BwdParams& p, const dim3 bid) {
    if constexpr (F16) ddpm::conv::fp16_saturate();
    ddpm::conv::RangeWatch amax;                     // F16: largest magnitude this thread has turned into an fp16 pair
    ddpm::conv::NoWatch unwatched_stream;
    auto& swatch = [&]() -> auto& {   // the streamed tiles: watched by `amax` too, or not at all
        if constexpr (SW) return amax;
        else return unwatched_stream;
    }();
    using PR = Products<F16>;
    constexpr int NPL = F16 ? 2 : 3;
    constexpr int KCV = CT / 16;
    constexpr int KD = DQ < 16 ? 16 : DQ, KQ = KD / 16;
    constexpr int KROW = 2 * KD, KPLANE = 32 * KROW;
    constexpr int VROW = CT * 2 + 16;          // V plane row + pad: row reads conflict-free
    constexpr int VPLANE = 32 * VROW;
    constexpr int VPT = 32 * CT / 4 / 256;
    constexpr int KF4 = 32 * DQ / 4;
    extern __shared__ __attribute__((aligned(256))) unsigned char dyn_lds[];
    unsigned char* sV = dyn_lds;                    // [2][NPL * VPLANE]
    unsigned char* sK = sV + 2 * NPL * VPLANE;      // [2][NPL * KPLANE]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int b = bid.z;
    const int i0 = (bid.x * 4 + wave) * 32;
    const bool wave_active = i0 < p.n;
    const int qi = i0 + r;
    const bool q_ok = qi < p.n;
    const float g = p.gamma[0];
    const float* base = p.qkv + static_cast<int64_t>(b) * p.n * p.lq;

    if (DQ < 16) {
        for (int i = tid; i < 2 * NPL * KPLANE / 4; i += 256) reinterpret_cast<unsigned*>(sK)[i] = 0u;
    }

    // fixed operands of this wave's 32 queries: lane (query, h) holds depth / channels 16*chunk + 8h .. + 7
    u32x4 qf[KQ][NPL], dof[KCV][NPL];
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq)
        load_split8<F16>(base + static_cast<int64_t>(qi) * p.lq + 16 * kq + 8 * h, q_ok && 16 * kq + 8 * h < DQ, qf[kq], 1.0f, amax);
#pragma unroll
    for (int kc = 0; kc < KCV; ++kc)
        load_split8<F16>(p.dy + (static_cast<int64_t>(b) * p.n + qi) * p.ldd + 16 * kc + 8 * h, q_ok, dof[kc], g, amax);
    const float lse_i = q_ok ? p.lse[static_cast<int64_t>(b) * p.n + qi] : 0.f;
    const float d_i = q_ok ? g * p.e[static_cast<int64_t>(b) * p.n + qi] : 0.f;

    // (accumulators taken in turn: see the key-side kernel; fp16 pair: dq2[0] main, dq2[1] cross products)
    f32x16 dq2[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        dq2[0][e] = 0.f;
        dq2[1][e] = 0.f;
    }

    // K and V rows through a range-checked descriptor over the sample's [n][lq] block: no branch around the loads
    const __amdgpu_buffer_rsrc_t rs_kv = ddpm::buffer_over(base, static_cast<long long>(p.n) * p.lq * 4ll);
    f32x4 kreg, vreg[VPT];
    auto load_tile = [&](int j0) {
        {
            const int row = tid / (DQ / 4), c4 = tid % (DQ / 4);
            kreg = ddpm::buffer_load4(rs_kv, tid < KF4 ? static_cast<unsigned>((j0 + row) * p.lq + DQ + c4 * 4) * 4u : ddpm::kOob);
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (CT / 4), c4 = f % (CT / 4);
            vreg[i] = ddpm::buffer_load4(rs_kv, static_cast<unsigned>((j0 + row) * p.lq + 2 * DQ + c4 * 4) * 4u);
        }
    };
    auto store_tile = [&](int buf) {
        if (tid < KF4) {
            const int row = tid / (DQ / 4), c4 = tid % (DQ / 4);
            split4_store_any<F16>(kreg, sK + buf * NPL * KPLANE + row * KROW + c4 * 8, KPLANE, swatch);
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (CT / 4), c4 = f % (CT / 4);
            split4_store_any<F16>(vreg[i], sV + buf * NPL * VPLANE + row * VROW + c4 * 8, VPLANE, swatch);
        }
    };

    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    // K tile transposed read: rows = keys 4h + tq (+8, +16 per chunk), depth columns 16 * tg + 4 * tp ..
    const int ktr = (4 * h + tq) * KROW + tp * 8 + (KD > 16 ? tg * 32 : 0);

    const int ktiles = (p.n + 31) / 32;
    load_tile(0);
    __syncthreads();
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ktiles; ++t) {
        const int buf = t & 1, j0 = t * 32;
        const bool more = t + 1 < ktiles;
        if (more) load_tile(j0 + 32);
        if (wave_active) {
            const unsigned char* k = sK + buf * NPL * KPLANE;
            const unsigned char* v = sV + buf * NPL * VPLANE;
            // fp16 pair: sx, dp1 and dp2 hold cross products (dp2: those of the second chunk of a pair)
            f32x16 s, sx, dp, dp1, dp2;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] = 0.f;
                dp[e] = 0.f;
                dp1[e] = 0.f;
                if constexpr (F16) {
                    sx[e] = 0.f;
                    dp2[e] = 0.f;
                }
            }
            auto mma = [&](f32x16& acc, const u32x4& a, const u32x4& bb) { acc = mfma16<F16>(a, bb, acc); };
            // S^T = K Q^T: rows = keys (this tile), columns = queries (lanes)
            u32x4 ak[KQ][NPL];
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) ak[kq][pl] = *reinterpret_cast<const u32x4*>(k + pl * KPLANE + r * KROW + kq * 32 + h * 16);
            if constexpr (F16) {   // (the pair's scores first: see the key-side kernel)
#pragma unroll
                for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
                    for (int kk = 0; kk < PR::N; ++kk) mma(PR::cross(kk) ? sx : s, ak[kq][PR::pa(kk)], qf[kq][PR::pb(kk)]);
                s = s + sx * kLoInv;
            }
            // dP^T = V dO^T, two 16-channel chunks at a time; (bf16 planes) the MFMAs of the scores go between the first of them
            static_assert(KCV % 2 == 0 && KQ <= KCV / 2, "chunk pairs; the scores fit between them");
#pragma unroll
            for (int kc = 0; kc < KCV; kc += 2) {
                u32x4 a[2][NPL];
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) a[c2][pl] = *reinterpret_cast<const u32x4*>(v + pl * VPLANE + r * VROW + (kc + c2) * 32 + h * 16);
#pragma unroll
                for (int kk = 0; kk < PR::N; ++kk) {
                    const bool x = PR::cross(kk);
                    if (!F16 && kc / 2 < KQ) mma(s, ak[kc / 2 < KQ ? kc / 2 : 0][PR::pa(kk)], qf[kc / 2 < KQ ? kc / 2 : 0][PR::pb(kk)]);
                    if constexpr (F16) {
                        mma(x ? dp1 : dp, a[0][PR::pa(kk)], dof[kc][PR::pb(kk)]);
                        mma(x ? dp2 : dp, a[1][PR::pa(kk)], dof[kc + 1][PR::pb(kk)]);
                    } else {
                        mma(dp, a[0][PR::pa(kk)], dof[kc][PR::pb(kk)]);
                        mma(dp1, a[1][PR::pa(kk)], dof[kc + 1][PR::pb(kk)]);
                    }
                }
            }
            if constexpr (F16) dp = dp + (dp1 + dp2) * kLoInv;
            else dp += dp1;
            u32x4 ds[2][NPL];
#pragma unroll
            for (int ck = 0; ck < 2; ++ck) {
                // Only keys past the end need a mask (their rows of K^T meet dS in the product below; a probability that
                // overflowed there would turn 0 * inf into NaN), and only the last tile has any: one uniform branch. A
                // query past the end has zero Q, dO, lse and D, hence dS = 0.
                float dsv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) dsv[j] = __builtin_amdgcn_exp2f((s[8 * ck + j] - lse_i) * 1.4426950408889634f);
                if (j0 + 32 > p.n) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (j0 + rowmap(8 * ck + j, h) >= p.n) dsv[j] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) dsv[j] *= dp[8 * ck + j] - d_i;
                split8_any<F16>(dsv, ds[ck], amax);
            }
            {   // dQ^T += K^T dS^T, the two 16-key chunks side by side
                u32x4 a[2][NPL];
#pragma unroll
                for (int ck = 0; ck < 2; ++ck)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                        const unsigned char* pbase = k + pl * KPLANE + 16 * ck * KROW + ktr;
                        a[ck][pl] = tr_pair(pbase, pbase + 8 * KROW);
                    }
                if constexpr (F16) {
                    mma(dq2[1], a[0][1], ds[0][0]);
                    mma(dq2[0], a[0][0], ds[0][0]);
                    mma(dq2[1], a[0][0], ds[0][1]);
                    mma(dq2[0], a[1][0], ds[1][0]);
                    mma(dq2[1], a[1][1], ds[1][0]);
                    mma(dq2[1], a[1][0], ds[1][1]);
                } else {
#pragma unroll
                    for (int kk = 0; kk < PR::N; ++kk) {
                        mma(dq2[0], a[0][PR::pa(kk)], ds[0][PR::pb(kk)]);
                        mma(dq2[1], a[1][PR::pa(kk)], ds[1][PR::pb(kk)]);
                    }
                }
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }
    if constexpr (F16) ddpm::conv::raise_range_flag(p.range_flag, ddpm::conv::merged(amax, swatch));
    if (!wave_active || !q_ok) return;
    float* drow = p.dqkv + (static_cast<int64_t>(b) * p.n + qi) * p.lq;
    const f32x16 dq = F16 ? dq2[0] + dq2[1] * kLoInv : dq2[0] + dq2[1];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int d = gq * 8 + h * 4;
        if (d >= DQ) continue;
        f32x4 v = {dq[gq * 4], dq[gq * 4 + 1], dq[gq * 4 + 2], dq[gq * 4 + 3]};
        *reinterpret_cast<f32x4*>(drow + d) = v;
    }
}

template <int DQ, int CT, bool F16 = false>
__global__ __launch_bounds__(256, CT <= 128 ? 2 : 1) void attn_bwd_dq_x3_kernel(const BwdParams p) {
    const dim3 bid = ddpm::xcd_block();   // the block to work on: neighbours in launch order share an XCD's L2
    if (F16 && bid.x == 0) attn_bwd_dq_x3_body<DQ, CT, F16, F16>(p, bid);
    else attn_bwd_dq_x3_body<DQ, CT, F16, false>(p, bid);
}

template <int DQ, int CT, bool F16 = false>
constexpr int dq_x3_lds() {
    constexpr int KD = DQ < 16 ? 16 : DQ, NPL = F16 ? 2 : 3;
    return 2 * NPL * 32 * (CT * 2 + 16) + 2 * NPL * 32 * 2 * KD;
}

template <typename K>
bool allow_lds(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}

// DDPM_ATTN_TRAIN_F16=0 (read once) keeps the attention of training calls (forward with a log-sum-exp, and this
// backward) on three bf16 planes when the process computes with fp16 pairs: a diagnostic switch for A/B measurements.
// attention.hip reads the same variable: the two sides always agree.
inline bool pair_in_training() {
    static const bool on = [] {
        const char* e = getenv("DDPM_ATTN_TRAIN_F16");
        return !(e && strcmp(e, "0") == 0);
    }();
    return on;
}

// split-operand backward for the layers whose operands fit the register file: C <= 128 in three launches
// (dK, dV, dQ), C = 256 likewise (dV per 128 channels). The fp16 pair takes the layers whose forward takes it:
// at most 16 query channels (C = 64 and 128).
template <int DQ, int C, bool F16>
int launch_bwd_x3_math(const BwdParams& p, int batch, hipStream_t s) {
    const int bn = (p.n + 127) / 128;
    if constexpr (C <= 128) {
        // two lighter kernels that fit two waves per SIMD beat a fused dK + dV kernel at one wave per SIMD
        constexpr int lds_k = dkv_x3_lds<DQ, C, true, false, F16>(), lds_v = dkv_x3_lds<DQ, C, false, true, F16>();
        if (!allow_lds(attn_bwd_dkv_x3_kernel<DQ, C, true, false, 2, F16>, lds_k) || !allow_lds(attn_bwd_dkv_x3_kernel<DQ, C, false, true, 2, F16>, lds_v))
            return ddpm::fail_invalid("LDS size for attention backward (dk, dv)");
        attn_bwd_dkv_x3_kernel<DQ, C, true, false, 2, F16><<<dim3(bn, 1, batch), 256, lds_k, s>>>(p);
        attn_bwd_dkv_x3_kernel<DQ, C, false, true, 2, F16><<<dim3(bn, 1, batch), 256, lds_v, s>>>(p);
    } else {
        constexpr int CV = C > 128 ? 128 : C;
        constexpr int lds_k = dkv_x3_lds<DQ, C, true, false, F16>(), lds_v = dkv_x3_lds<DQ, CV, false, true, F16>();
        if (!allow_lds(attn_bwd_dkv_x3_kernel<DQ, C, true, false, 1, F16>, lds_k) || !allow_lds(attn_bwd_dkv_x3_kernel<DQ, CV, false, true, 2, F16>, lds_v))
            return ddpm::fail_invalid("LDS size for attention backward (dk, dv)");
        attn_bwd_dkv_x3_kernel<DQ, C, true, false, 1, F16><<<dim3(bn, 1, batch), 256, lds_k, s>>>(p);
        attn_bwd_dkv_x3_kernel<DQ, CV, false, true, 2, F16><<<dim3(bn, C / CV, batch), 256, lds_v, s>>>(p);
    }
    constexpr int lds_q = dq_x3_lds<DQ, C, F16>();
    if (!allow_lds(attn_bwd_dq_x3_kernel<DQ, C, F16>, lds_q)) return ddpm::fail_invalid("LDS size for attention backward (dq)");
    attn_bwd_dq_x3_kernel<DQ, C, F16><<<dim3(bn, 1, batch), 256, lds_q, s>>>(p);
    return ddpm::check_launch("attn_bwd_x3");
}

template <int DQ, int C>
int launch_bwd_x3(const BwdParams& p, int batch, hipStream_t s) {
    if constexpr (DQ <= 16) {
        if (ddpm::f16_math() && pair_in_training()) {
            if (!p.range_flag) return DDPM_E_LAUNCH;
            return launch_bwd_x3_math<DQ, C, true>(p, batch, s);
        }
    }
    return launch_bwd_x3_math<DQ, C, false>(p, batch, s);
}

template <int DQ, int CT>
int launch_bwd_fast(const BwdParams& p, int batch, hipStream_t s) {
    const dim3 grid((p.n + 127) / 128, 1, batch);
    if (ddpm::split_math()) return launch_bwd_x3<DQ, CT>(p, batch, s);
#ifndef DDPM_DKV_WPS
#define DDPM_DKV_WPS 1
#endif
    // 128 channels: V rows (64 registers) + dV accumulators (64) + scores do not fit 256 registers
    // without spilling into the tile loop, so that variant is sized for one wave per SIMD.
    if (CT > 64) attn_bwd_dkv_fast_kernel<DQ, CT, DDPM_DKV_WPS><<<grid, 256, 0, s>>>(p);
    else attn_bwd_dkv_fast_kernel<DQ, CT, 2><<<grid, 256, 0, s>>>(p);
    attn_bwd_dq_fast_kernel<DQ, CT><<<grid, 256, 0, s>>>(p);
    return ddpm::check_launch("attn_bwd_fast");
}

template <int DQ>
int launch_bwd(const BwdParams& p, int batch, hipStream_t s) {
    const int blocks_n = (p.n + 127) / 128;
    const size_t lds_kv = sizeof(float) * (TQ * (DQ + 4) + TQ * (p.c + 4) + 2 * TQ);
    const size_t lds_q = sizeof(float) * (32 * (DQ + 4) + 32 * (p.c + 4));
    if (p.c > 64) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<DQ, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(lds_kv)) != hipSuccess)
            return ddpm::fail_invalid("LDS size for attention backward (dkv)");
        attn_bwd_dkv_kernel<DQ, 4><<<dim3(blocks_n, (p.c + 127) / 128, batch), 256, lds_kv, s>>>(p);
    } else if (p.c > 32) {
        attn_bwd_dkv_kernel<DQ, 2><<<dim3(blocks_n, 1, batch), 256, lds_kv, s>>>(p);
    } else {
        attn_bwd_dkv_kernel<DQ, 1><<<dim3(blocks_n, 1, batch), 256, lds_kv, s>>>(p);
    }
    if (int rc = ddpm::check_launch("attn_bwd_dkv_kernel")) return rc;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<DQ>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds_q)) != hipSuccess)
        return ddpm::fail_invalid("LDS size for attention backward (dq)");
    attn_bwd_dq_kernel<DQ><<<dim3(blocks_n, 1, batch), 256, lds_q, s>>>(p);
    return ddpm::check_launch("attn_bwd_dq_kernel");
}

}  // namespace

extern "C" int ddpm_attention_bwd(const float* qkv, const float* dy, int ldd, const float* attn_raw, const float* lse, const float* gamma,
                                  double* workspace, float* e_scratch, float* dqkv, float* dgamma, int batch, int n, int c, int dq,
                                  void* stream) {
    DDPM_REQUIRE(qkv && dy && attn_raw && lse && gamma && workspace && e_scratch && dqkv && dgamma);
    DDPM_REQUIRE(batch > 0 && n > 0 && c > 0 && (c & 3) == 0 && c <= 1024 && (ldd & 3) == 0 && ldd >= c);
    DDPM_REQUIRE(dq == 8 || dq == 16 || dq == 32 || dq == 64 || dq == 128);
    DDPM_REQUIRE(ddpm::aligned16(qkv) && ddpm::aligned16(dy) && ddpm::aligned16(attn_raw) && ddpm::aligned16(dqkv));
    hipStream_t s = ddpm::as_stream(stream);
    const int64_t rows = static_cast<int64_t>(batch) * n;
    int pre_blocks = static_cast<int>((rows + 3) / 4 > 1024 ? 1024 : (rows + 3) / 4);
    attn_bwd_pre_kernel<<<pre_blocks, 256, 0, s>>>(dy, ldd, attn_raw, c, rows, e_scratch, workspace);
    attn_bwd_dgamma_kernel<<<1, 256, 0, s>>>(workspace, pre_blocks, dgamma);
    if (int rc = ddpm::check_launch("attn_bwd_pre")) return rc;
    if (ddpm::small_attention() && ddpm::attention_small_bwd_fits(n, c, dq))
        return ddpm::attention_small_bwd(qkv, dy, ldd, lse, e_scratch, gamma, dqkv, batch, n, c, dq, s);
    BwdParams p;
    p.qkv = qkv;
    p.dy = dy;
    p.o = attn_raw;
    p.lse = lse;
    p.e = e_scratch;
    p.gamma = gamma;
    p.dqkv = dqkv;
    p.n = n;
    p.c = c;
    p.dqp = dq;
    p.lq = 2 * dq + c;
    p.ldd = ldd;
    p.range_flag = ddpm::f16_math() ? ddpm::range_flag() : nullptr;
    // the sample index is the grid's z axis (at most 65535): a larger batch runs as consecutive launches over its parts
    constexpr int kMaxZ = 65535;
    for (int b0 = 0; b0 < batch; b0 += kMaxZ) {
        const int nb = batch - b0 < kMaxZ ? batch - b0 : kMaxZ;
        const int64_t r0 = static_cast<int64_t>(b0) * n;
        BwdParams q = p;
        q.qkv = p.qkv + r0 * p.lq;
        q.dy = p.dy + r0 * ldd;
        q.o = p.o + r0 * c;
        q.lse = p.lse + r0;
        q.e = p.e + r0;
        q.dqkv = p.dqkv + r0 * p.lq;
        int rc;
        if (c == 64 && dq == 8) rc = launch_bwd_fast<8, 64>(q, nb, s);
        else if (c == 128 && dq == 16) rc = launch_bwd_fast<16, 128>(q, nb, s);
        else if (c == 256 && dq == 32 && ddpm::split_math()) rc = launch_bwd_x3<32, 256>(q, nb, s);
        else switch (dq) {
            case 8: rc = launch_bwd<8>(q, nb, s); break;
            case 16: rc = launch_bwd<16>(q, nb, s); break;
            case 32: rc = launch_bwd<32>(q, nb, s); break;
            case 64: rc = launch_bwd<64>(q, nb, s); break;
            default: rc = launch_bwd<128>(q, nb, s); break;
        }
        if (rc) return rc;
    }
    return DDPM_OK;
}

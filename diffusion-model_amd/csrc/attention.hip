// Single-head, unscaled self-attention of the UNet (d_qk = C/8 zero-padded to a multiple of 8,
// d_v = C), flash-style on the fp32 matrix cores: the N x N score matrix lives only in registers.
//
// One wave owns 32 queries and a slice of up to 128 value channels; the 4 waves of a workgroup
// share each 32-key K/V tile through LDS (double-buffered, register-prefetched).
//
//   S^T = K Q^T   : A = K tile (row = key),   B = Q fragment (col = query)  -> lane = query,
//                   register = key. Row max / row sum are therefore lane-local plus ONE exchange
//                   with lane^32.
//   O^T = V^T P^T : A = V^T (row = channel),  B = P^T. MFMA number `reg` of a key tile uses the
//                   score register `reg` as its B operand unchanged (the k index of that MFMA is
//                   exactly the key that register holds in each lane half), so the probabilities
//                   never leave the register file. The output tile again has lane = query, so the
//                   online-softmax rescale is a per-lane multiply.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KT = 32;  // keys per tile

__device__ __forceinline__ int rowmap(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

struct AttnParams {
    const float* qkv;   // [B][N][lq]: q | k | v
    const float* x;     // [B][N][.] stride ldx
    const float* gamma;
    const float* temb;  // optional [B][C], added after the gate
    float* out;         // stride ldo
    float* attn_raw;    // optional [B][N][C]
    float* lse;         // optional [B][N]
    int n, c, dqp, lq, ldx, ldo;
};

template <int DQ, int NCT>
__global__ __launch_bounds__(256) void attention_fwd_kernel(const AttnParams p) {
    constexpr int CS = 32 * NCT;      // value channels per workgroup
    constexpr int KLD = DQ + 4;       // padded K row (floats): 16-byte row reads stay conflict-free
    constexpr int VLD = CS;
    constexpr int KF4 = KT * DQ / 4;  // float4 per K tile
    constexpr int VF4 = KT * CS / 4;
    constexpr int KPT = (KF4 + 255) / 256, VPT = VF4 / 256;
    static_assert(VF4 % 256 == 0, "V tile must split evenly over 256 threads");

    __shared__ __attribute__((aligned(16))) float sK[2][KT * KLD];
    __shared__ __attribute__((aligned(16))) float sV[2][KT * VLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const int cbase = blockIdx.y * CS;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const bool wave_active = q0 < p.n;          // wave-uniform
    const int qi = q0 + r;                      // this lane's query
    const float* base = p.qkv + static_cast<int64_t>(b) * p.n * p.lq;

    // Q fragment: element e of chunk kk is Q[qi][8kk + 4h + e].
    f32x4 qf[DQ / 8];
#pragma unroll
    for (int kk = 0; kk < DQ / 8; ++kk) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (qi < p.n) v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(qi) * p.lq + kk * 8 + h * 4);
        qf[kk] = v;
    }

    f32x16 o[NCT];
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    f32x4 kreg[KPT], vreg[VPT];
    auto load_tile = [&](int j0) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (DQ / 4), c4 = f % (DQ / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (f < KF4 && j0 + row < p.n)
                v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(j0 + row) * p.lq + p.dqp + c4 * 4);
            kreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (CS / 4), c4 = f % (CS / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j0 + row < p.n && cbase + c4 * 4 < p.c)
                v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(j0 + row) * p.lq + 2 * p.dqp + cbase + c4 * 4);
            vreg[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (DQ / 4), c4 = f % (DQ / 4);
            if (f < KF4) *reinterpret_cast<f32x4*>(&sK[buf][row * KLD + c4 * 4]) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (CS / 4), c4 = f % (CS / 4);
            *reinterpret_cast<f32x4*>(&sV[buf][row * VLD + c4 * 4]) = vreg[i];
        }
    };

    const int tiles = (p.n + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < tiles; ++t) {
        const int buf = t & 1, j0 = t * KT;
        const bool more = t + 1 < tiles;
        if (more) load_tile(j0 + KT);
        if (wave_active) {
            // scores: lane = query, register = key rowmap(reg, h)
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < DQ / 8; ++kk) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(&sK[buf][r * KLD + kk * 8 + h * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[kk][e], s, 0, 0, 0);
            }
            float mt = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (j0 + rowmap(e, h) >= p.n) s[e] = -INFINITY;
                mt = fmaxf(mt, s[e]);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = fmaxf(m_run, mt);
            const float alpha = expf(m_run - m_new);   // 0 on the first tile (m_run = -inf)
            float ps = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] = expf(s[e] - m_new);
                ps += s[e];
            }
            ps += __shfl_xor(ps, 32);
            l_run = l_run * alpha + ps;
            m_run = m_new;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
                for (int e = 0; e < 16; ++e) o[ct][e] *= alpha;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float va = sV[buf][rowmap(e, h) * VLD + ct * 32 + r];
                    o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, s[e], o[ct], 0, 0, 0);
                }
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    if (!wave_active || qi >= p.n) return;
    const float inv_l = 1.0f / l_run;
    const float g = p.gamma[0];
    const int64_t pix = static_cast<int64_t>(b) * p.n + qi;
    if (p.lse && h == 0 && blockIdx.y == 0) p.lse[pix] = m_run + logf(l_run);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int c = cbase + ct * 32 + gq * 8 + h * 4;   // channels c..c+3 = registers 4gq..4gq+3
            if (c >= p.c) continue;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + pix * p.ldx + c);
            f32x4 te = {0.f, 0.f, 0.f, 0.f};
            if (p.temb) te = *reinterpret_cast<const f32x4*>(p.temb + static_cast<int64_t>(b) * p.c + c);
            f32x4 a, y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = o[ct][gq * 4 + e] * inv_l;
                y[e] = (g * a[e] + xv[e]) + te[e];
            }
            if (p.attn_raw) *reinterpret_cast<f32x4*>(p.attn_raw + pix * p.c + c) = a;
            *reinterpret_cast<f32x4*>(p.out + pix * p.ldo + c) = y;
        }
    }
}

template <int DQ>
int launch_dq(const AttnParams& p, int batch, hipStream_t s) {
    const int q_blocks = (p.n + 127) / 128;
    if (p.c > 64) {
        const dim3 g(q_blocks, (p.c + 127) / 128, batch);
        attention_fwd_kernel<DQ, 4><<<g, 256, 0, s>>>(p);
    } else if (p.c > 32) {
        const dim3 g(q_blocks, 1, batch);
        attention_fwd_kernel<DQ, 2><<<g, 256, 0, s>>>(p);
    } else {
        // 32 channels: the V tile (32 x 32 floats) is one float4 per thread
        const dim3 g(q_blocks, 1, batch);
        attention_fwd_kernel<DQ, 1><<<g, 256, 0, s>>>(p);
    }
    return ddpm::check_launch("attention_fwd_kernel");
}

}  // namespace

extern "C" int ddpm_attention_fwd(const float* qkv, const float* x, int ldx, const float* gamma, float* out, int ldo,
                                  float* attn_raw, float* lse, const float* temb, int batch, int n, int c, int dq,
                                  void* stream) {
    DDPM_REQUIRE(qkv && x && gamma && out && batch > 0 && batch <= 65535 && n > 0 && c > 0);
    DDPM_REQUIRE((c & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0 && ldx >= c && ldo >= c);
    DDPM_REQUIRE(dq == 8 || dq == 16 || dq == 32 || dq == 64);
    DDPM_REQUIRE(ddpm::aligned16(qkv) && ddpm::aligned16(x) && ddpm::aligned16(out));
    DDPM_REQUIRE(!attn_raw || ddpm::aligned16(attn_raw));
    AttnParams p;
    p.qkv = qkv;
    p.x = x;
    p.gamma = gamma;
    p.temb = temb;
    p.out = out;
    p.attn_raw = attn_raw;
    p.lse = lse;
    p.n = n;
    p.c = c;
    p.dqp = dq;
    p.lq = 2 * dq + c;
    p.ldx = ldx;
    p.ldo = ldo;
    hipStream_t s = ddpm::as_stream(stream);
    switch (dq) {
        case 8: return launch_dq<8>(p, batch, s);
        case 16: return launch_dq<16>(p, batch, s);
        case 32: return launch_dq<32>(p, batch, s);
        default: return launch_dq<64>(p, batch, s);
    }
}

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
#include "attention_small.h"
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
    const dim3 bid = ddpm::xcd_block();   // the block to work on: neighbours in launch order share an XCD's L2
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
    const int b = bid.z;
    const int cbase = bid.y * CS;
    const int q0 = (bid.x * 4 + wave) * 32;
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
    if (p.lse && h == 0 && bid.y == 0) p.lse[pix] = m_run + logf(l_run);
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


// ---- split-bf16 variant (long sequences) ------------------------------------------------------
// Both products run on v_mfma_f32_32x32x16_bf16 with every fp32 operand written as the exact sum of
// three bf16 numbers and a product evaluated as its six cross terms of weight >= 2^-16 (fp32
// accumulation; the dropped terms are below 2^-23 of the product): 2.67x the matrix throughput of
// the fp32 MFMA at fp32 accuracy.
//
// A wave owns 64 queries (two 32-query blocks) and 64 value channels; the 4 waves of a workgroup
// share each 32-key tile. K and V are split on their way from global memory to LDS (three bf16
// planes). S^T = K Q^T takes K row-wise (ds_read_b128) and the pre-split Q from registers. The
// probabilities are split in registers; their accumulator layout (lane = query, register = key)
// is already the B-operand layout of O^T = V^T P^T once the 16 keys of an MFMA are taken in the
// order the registers hold them, and V^T comes out of the [key][channel] LDS image in exactly that
// key order through the transposing read ds_read_b64_tr_b16 (each lane names its own key row).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 8 floats -> three planes of 8 bf16 (hi | mid | lo), element j in the j-th half-word
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

// QB = 32-query blocks per wave, NCT = 32-channel blocks per workgroup. Per 32 x 32 tile of scores a wave spends
// about 1050 cycles of VALU on the softmax and the split of the probabilities, 192 * KC cycles of MFMA on the scores
// and 384 * NCT cycles of MFMA on the value product: with 64 channels (NCT = 2) the two are level, and every further
// 64-channel slice handled by ANOTHER workgroup repeats the scores and the softmax. Tensors of 128 channels and more
// therefore run (QB, NCT) = (1, 4): 32 queries x 128 channels per wave, which makes the matrix pipe the longer side
// and halves the repeated work; 64-channel tensors keep (2, 2).
template <int DQ, int QB, int NCT>
__global__ __launch_bounds__(256, 2) void attention_fwd_x3_kernel(const AttnParams p) {
    const dim3 bid = ddpm::xcd_block();   // the block to work on: neighbours in launch order share an XCD's L2
    constexpr int CS = 32 * NCT;
    constexpr int KD = DQ < 16 ? 16 : DQ;       // K/Q depth padded to the MFMA's 16
    constexpr int KC = KD / 16;                 // depth chunks
    constexpr int KPL = KD * 2;                 // bytes of one plane in a K row
    constexpr int KROW = 3 * KPL + 16;          // K row: hi | mid | lo + pad (row reads conflict-free)
    constexpr int VROW = CS * 2 + 64;           // V row of one plane + pad (transposed reads conflict-free)
    constexpr int VPLANE = KT * VROW;
    constexpr int KF4 = KT * DQ / 4;            // float4 per K tile
    constexpr int VF4 = KT * CS / 4;            // float4 per V tile
    constexpr int KPT = (KF4 + 255) / 256, VPT = VF4 / 256;
    static_assert(VF4 % 256 == 0, "V tile must split evenly over 256 threads");

    __shared__ __attribute__((aligned(16))) unsigned char sK[2][KT * KROW];
    __shared__ __attribute__((aligned(16))) unsigned char sV[2][3 * VPLANE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int b = bid.z;
    const int cbase = bid.y * CS;
    const int q0 = (bid.x * 4 + wave) * (32 * QB);
    const bool wave_active = q0 < p.n;          // wave-uniform
    const float* base = p.qkv + static_cast<int64_t>(b) * p.n * p.lq;

    if (DQ < 16) {   // the upper half of the padded depth stays zero
        for (int i = tid; i < 2 * KT * KROW / 4; i += 256) reinterpret_cast<unsigned*>(&sK[0][0])[i] = 0u;
    }

    // Q fragments (B operand of S^T = K Q^T): lane (query, h) holds depth 16kc + 8h .. + 7, three planes
    u32x4 qf[QB][KC][3];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + 32 * qb + r;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            const int d0 = 16 * kc + 8 * h;
            if (qi < p.n && d0 < DQ) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(qi) * p.lq + d0);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(qi) * p.lq + d0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = x0[j];
                    v[4 + j] = x1[j];
                }
            }
            split8(v, qf[qb][kc][0], qf[qb][kc][1], qf[qb][kc][2]);
        }
    }

    f32x16 o[QB][NCT];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int t = 0; t < NCT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[qb][t][e] = 0.f;
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -INFINITY;
        l_run[qb] = 0.f;
    }

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
            if (f < KF4) split4_store(kreg[i], &sK[buf][row * KROW + c4 * 8], KPL);
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (CS / 4), c4 = f % (CS / 4);
            split4_store(vreg[i], &sV[buf][row * VROW + c4 * 8], VPLANE);
        }
    };

    // transposing V read: lane 4q + p of a 16-lane group names key row q, channels 4p .. 4p+3 of the
    // group's 16 channels; the lane then holds ITS channel (lane & 15 of the group) for the 4 key rows.
    // Key rows of a lane half: 4h + q (+ 8 for the second read, + 16 per 16-key chunk): the order in
    // which the score registers hold the keys.
    const int vlane = (4 * h + ((lane & 15) >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    const int tiles = (p.n + KT - 1) / KT;
    load_tile(0);
    __syncthreads();   // the zero fill of sK, if any
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < tiles; ++t) {
        const int buf = t & 1, j0 = t * KT;
        const bool more = t + 1 < tiles;
        if (more) load_tile(j0 + KT);
        if (wave_active) {
            // ---- scores: lane = query, register = key rowmap(reg, h)
            f32x16 s[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[qb][e] = 0.f;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                bf16x8 kf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) kf[pl] = *reinterpret_cast<const bf16x8*>(&sK[buf][r * KROW + pl * KPL + kc * 32 + h * 16]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    auto term = [&](int pa, int pb) {
                        s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[pa], __builtin_bit_cast(bf16x8, qf[qb][kc][pb]), s[qb], 0, 0, 0);
                    };
                    term(2, 0);
                    term(0, 2);
                    term(1, 1);
                    term(1, 0);
                    term(0, 1);
                    term(0, 0);
                }
            }
            // ---- online softmax, probabilities split into three bf16 planes per 16-key chunk
            u32x4 pp[QB][2][3];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (j0 + KT > p.n) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (j0 + rowmap(e, h) >= p.n) s[qb][e] = -INFINITY;
                }
                float mt = s[qb][0];
#pragma unroll
                for (int e = 1; e < 16; ++e) mt = fmaxf(mt, s[qb][e]);
                mt = fmaxf(mt, __shfl_xor(mt, 32));
                const float m_new = fmaxf(m_run[qb], mt);
                const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * 1.4426950408889634f);   // 0 on the first tile
                float ps = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    s[qb][e] = __builtin_amdgcn_exp2f((s[qb][e] - m_new) * 1.4426950408889634f);
                    ps += s[qb][e];
                }
                ps += __shfl_xor(ps, 32);
                l_run[qb] = l_run[qb] * alpha + ps;
                m_run[qb] = m_new;
                if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {   // the running maximum moved for some query
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int e = 0; e < 16; ++e) o[qb][ct][e] *= alpha;
                }
#pragma unroll
                for (int ck = 0; ck < 2; ++ck) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = s[qb][8 * ck + j];
                    split8(v, pp[qb][ck][0], pp[qb][ck][1], pp[qb][ck][2]);
                }
            }
            // ---- O^T += V^T P^T
#pragma unroll
            for (int ck = 0; ck < 2; ++ck) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    bf16x8 vf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const unsigned char* src = &sV[buf][pl * VPLANE + (16 * ck) * VROW + ct * 64 + vlane];
                        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(src));
                        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(src + 8 * VROW));
                        const s16x8 both = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
                        vf[pl] = __builtin_bit_cast(bf16x8, both);
                    }
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        auto term = [&](int pa, int pb) {
                            o[qb][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[pa], __builtin_bit_cast(bf16x8, pp[qb][ck][pb]), o[qb][ct], 0, 0, 0);
                        };
                        term(2, 0);
                        term(0, 2);
                        term(1, 1);
                        term(1, 0);
                        term(0, 1);
                        term(0, 0);
                    }
                }
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    if (!wave_active) return;
    const float g = p.gamma[0];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + 32 * qb + r;
        if (qi >= p.n) continue;
        const float inv_l = 1.0f / l_run[qb];
        const int64_t pix = static_cast<int64_t>(b) * p.n + qi;
        if (p.lse && h == 0 && bid.y == 0) p.lse[pix] = m_run[qb] + logf(l_run[qb]);
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
                    a[e] = o[qb][ct][gq * 4 + e] * inv_l;
                    y[e] = (g * a[e] + xv[e]) + te[e];
                }
                if (p.attn_raw) *reinterpret_cast<f32x4*>(p.attn_raw + pix * p.c + c) = a;
                *reinterpret_cast<f32x4*>(p.out + pix * p.ldo + c) = y;
            }
        }
    }
}

// DDPM_ATTN_WIDE=0 (read once) keeps tensors of 128 channels and more on the 64-channel configuration: a diagnostic
// switch for A/B measurements.
inline bool wide_attention() {
    static const bool on = [] {
        const char* e = getenv("DDPM_ATTN_WIDE");
        return !(e && strcmp(e, "0") == 0);
    }();
    return on;
}

template <int DQ>
int launch_dq(const AttnParams& p, int batch, hipStream_t s) {
    if constexpr (DQ <= 64) {   // 128 query channels (C = 1024): only the bottleneck has them, its maps are short
      if (ddpm::split_math() && p.n >= 128 && p.c >= 64) {
        if (p.c >= 128 && wide_attention()) {
            const dim3 g((p.n + 127) / 128, (p.c + 127) / 128, batch);
            attention_fwd_x3_kernel<DQ, 1, 4><<<g, 256, 0, s>>>(p);
        } else {
            const dim3 g((p.n + 255) / 256, (p.c + 63) / 64, batch);
            attention_fwd_x3_kernel<DQ, 2, 2><<<g, 256, 0, s>>>(p);
        }
        return ddpm::check_launch("attention_fwd_x3_kernel");
      }
    }
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
    DDPM_REQUIRE(dq == 8 || dq == 16 || dq == 32 || dq == 64 || dq == 128);
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
    if (ddpm::small_attention() && ddpm::attention_small_fwd_fits(n, c, dq))
        return ddpm::attention_small_fwd(qkv, x, ldx, gamma, temb, out, ldo, attn_raw, lse, batch, n, c, dq, s);
    switch (dq) {
        case 8: return launch_dq<8>(p, batch, s);
        case 16: return launch_dq<16>(p, batch, s);
        case 32: return launch_dq<32>(p, batch, s);
        case 64: return launch_dq<64>(p, batch, s);
        default: return launch_dq<128>(p, batch, s);
    }
}

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
#include <type_traits>

#include "conv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KT = 32;  // keys per tile
// In INFERENCE calls (no log-sum-exp wanted) the matrix kernels' running row maximum follows the scores only when a tile
// exceeds it by more than this (natural-log units: probabilities stay below e^5.5 = 245, far inside fp32 and the fp16
// pair). A softmax is the same for every shift, so the result is unchanged up to rounding, while the output
// accumulators are rescaled on the first tiles only instead of whenever any row of the wave meets a larger score
// (5-8 % of these calls). Training calls keep the exact maximum, and with it the bits the backward pass was tuned on.
constexpr float kMaxSlack = 5.5f;

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
    unsigned* range_flag;   // fp16-pair kernels: raised when q, k or v lies beyond fp16's range
    // PRE kernels (inference, fp16 pair): the projection has split k and v already. qkv then holds q | k only (lq = 2 dqp),
    // every k element as one (hi | lo << 16) word; v is here, as pair planes in operand order: 16 bytes (the eight keys
    // 16ck + 4h + i and 16ck + 8 + 4h + i, i < 4, that half h of a matrix operand takes from 16-key chunk ck) at
    // [row / 16 = chunk][h][plane][channel], rows counted over the whole batch (n is a multiple of 32).
    const unsigned char* vpl;
};

// four k elements as (hi | lo << 16) words -> their two plane words each, into a K row of the tile
__device__ __forceinline__ void unzip4_store(const f32x4 v, unsigned char* dst, int plane_stride) {
    const unsigned w0 = __float_as_uint(v[0]), w1 = __float_as_uint(v[1]), w2 = __float_as_uint(v[2]), w3 = __float_as_uint(v[3]);
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2_t*>(dst) = u32x2_t{__builtin_amdgcn_perm(w1, w0, 0x05040100u), __builtin_amdgcn_perm(w3, w2, 0x05040100u)};
    *reinterpret_cast<u32x2_t*>(dst + plane_stride) = u32x2_t{__builtin_amdgcn_perm(w1, w0, 0x07060302u), __builtin_amdgcn_perm(w3, w2, 0x07060302u)};
}

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
    // K and V rows come through a range-checked descriptor over the sample's [n][lq] block: rows past the last key
    // and channels past the last one read as zero without a branch
    const __amdgpu_buffer_rsrc_t rs_kv = ddpm::buffer_over(base, static_cast<long long>(p.n) * p.lq * 4ll);
    auto load_tile = [&](int j0) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (DQ / 4), c4 = f % (DQ / 4);
            kreg[i] = ddpm::buffer_load4(rs_kv, f < KF4 ? static_cast<unsigned>((j0 + row) * p.lq + p.dqp + c4 * 4) * 4u : ddpm::kOob);
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (CS / 4), c4 = f % (CS / 4);
            vreg[i] = ddpm::buffer_load4(rs_kv, cbase + c4 * 4 < p.c ? static_cast<unsigned>((j0 + row) * p.lq + 2 * p.dqp + cbase + c4 * 4) * 4u : ddpm::kOob);
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

// Two floats -> one 32-bit word of each bf16 plane (element 0 in the low half-word). x = hi + mid + lo exactly: two
// truncations to the upper 16 bits, each followed by an exact subtraction. (Scalar subtractions on purpose: the packed
// v_pk_add_f32 form measured 2 % slower in the kernels below.)
__device__ __forceinline__ void split2(const float v0, const float v1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned u0 = __float_as_uint(v0), u1 = __float_as_uint(v1);
    const float r0 = v0 - __uint_as_float(u0 & 0xffff0000u), r1 = v1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned m0 = __float_as_uint(r0), m1 = __float_as_uint(r1);
    const float q0 = r0 - __uint_as_float(m0 & 0xffff0000u), q1 = r1 - __uint_as_float(m1 & 0xffff0000u);
    hi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    mid = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    lo = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}

// 8 floats -> three planes of 8 bf16 (hi | mid | lo), element j in the j-th half-word
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned a, b, c;
        split2(v[2 * e], v[2 * e + 1], a, b, c);
        hi[e] = a;
        mid[e] = b;
        lo[e] = c;
    }
}

__device__ __forceinline__ void split4_store(const f32x4 v, unsigned char* dst, int plane_stride) {
    u32x2 hi, mid, lo;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        unsigned a, b, c;
        split2(v[2 * e], v[2 * e + 1], a, b, c);
        hi[e] = a;
        mid[e] = b;
        lo[e] = c;
    }
    *reinterpret_cast<u32x2*>(dst) = hi;
    *reinterpret_cast<u32x2*>(dst + plane_stride) = mid;
    *reinterpret_cast<u32x2*>(dst + 2 * plane_stride) = lo;
}

// The same three helpers for the fp16 pair (hi, and the remainder scaled by 2^11; conv_common.h): two planes.
__device__ __forceinline__ void split8_f16(const float (&v)[8], u32x4& hi, u32x4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ddpm::conv::split2_pair(ddpm::conv::f32x2{v[2 * e], v[2 * e + 1]}, h[e], l[e]);
    hi = u32x4{h[0], h[1], h[2], h[3]};
    lo = u32x4{l[0], l[1], l[2], l[3]};
}

// (the same with the range watch of conv_common.h: for q, k and v, which come from memory; probabilities need none)
template <class W>
__device__ __forceinline__ void split8_f16(const float (&v)[8], u32x4& hi, u32x4& lo, W& amax) {
#pragma unroll
    for (int e = 0; e < 4; ++e) ddpm::conv::watch2(amax, ddpm::conv::f32x2{v[2 * e], v[2 * e + 1]});
    split8_f16(v, hi, lo);
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

using ddpm::conv::kLoInv;
using ddpm::conv::mfma16;

// With F16 (the default arithmetic, DDPM_CONV_MATH) the kernels below take every operand as an fp16 pair instead:
// two planes, three MFMAs per fragment pair, the hi*hi products in one accumulator and the two cross products in a
// second one that is scaled by 2^-11 and added when the sum is complete (the scores of a tile before their softmax,
// the output after the last key tile; the online-softmax factor multiplies both output accumulators).
//
// QB = 32-query blocks per wave, NCT = 32-channel blocks per workgroup. Per 32 x 32 tile of scores a wave spends
// about 1050 cycles of VALU on the softmax and the split of the probabilities, 192 * KC cycles of MFMA on the scores
// and 384 * NCT cycles of MFMA on the value product: with 64 channels (NCT = 2) the two are level, and every further
// 64-channel slice handled by ANOTHER workgroup repeats the scores and the softmax. Tensors of 128 channels and more
// therefore run (QB, NCT) = (1, 4): 32 queries x 128 channels per wave, which makes the matrix pipe the longer side
// and halves the repeated work; 64-channel tensors keep (2, 2).
#ifndef DDPM_ATTN_ABLATE
#define DDPM_ATTN_ABLATE 0   // diagnostic builds only (wrong results): bit 0 no softmax arithmetic, bit 1 no value-product MFMAs, bit 2 no staging in the loop, bit 3 no transposed V reads in the loop; pipelined kernel: bit 4 no softmax slices, bit 5 no staging, bit 6 V fragments of the first pair only, bit 7 no barrier in the loop
#endif
// KVW: this workgroup watches the range of the K and V tiles it stages (the first query block of a sample does it for
// all of them: they stage the same tiles); each watches its own queries. The tile buffers are the caller's, so that
// the two instantiations inside one kernel share them.
// PRE: k and v arrive split (AttnParams::vpl): the staging is a copy and the V fragments are plain 16-byte reads.
template <int DQ, int QB, int NCT, bool F16, bool KVW, bool PRE = false>
__device__ __forceinline__ void attention_fwd_x3_body(const AttnParams& p, const dim3 bid, unsigned char* sK_mem, unsigned char* sV_mem) {
    static_assert(!PRE || (F16 && !KVW), "pre-split operands are fp16 pairs, checked where they were split");
    if constexpr (F16) ddpm::conv::fp16_saturate();
    ddpm::conv::RangeWatch amax;                           // F16: largest magnitude of q this thread has turned into an fp16 pair
    ddpm::conv::NoWatch unwatched_kv;
    auto& kvw = [&]() -> auto& {   // k and v: watched by `amax` too, or not at all
        if constexpr (KVW) return amax;
        else return unwatched_kv;
    }();
    constexpr int NPL = F16 ? 2 : 3;            // operand planes
    constexpr int CS = 32 * NCT;
    constexpr int KD = DQ < 16 ? 16 : DQ;       // K/Q depth padded to the MFMA's 16
    constexpr int KC = KD / 16;                 // depth chunks
    constexpr int KPL = KD * 2;                 // bytes of one plane in a K row
    constexpr int KROW = NPL * KPL + 16;        // K row: the planes + pad (row reads conflict-free)
    constexpr int VROW = CS * 2 + 64;           // V row of one plane + pad (transposed reads conflict-free)
    constexpr int VPLANE = KT * VROW;
    constexpr int KF4 = KT * DQ / 4;            // float4 per K tile
    constexpr int VF4 = KT * CS / 4;            // float4 per V tile (PRE: 16-byte pieces [chunk][h][plane][channel], as many)
    constexpr int KPT = (KF4 + 255) / 256, VPT = VF4 / 256;
    static_assert(VF4 % 256 == 0, "V tile must split evenly over 256 threads");

    unsigned char (*sK)[KT * KROW] = reinterpret_cast<unsigned char (*)[KT * KROW]>(sK_mem);        // [2]
    unsigned char (*sV)[NPL * VPLANE] = reinterpret_cast<unsigned char (*)[NPL * VPLANE]>(sV_mem);  // [2] (PRE uses 16 * VF4 bytes of each)

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
    u32x4 qf[QB][KC][NPL];
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
            if constexpr (F16) split8_f16(v, qf[qb][kc][0], qf[qb][kc][1], amax);
            else split8(v, qf[qb][kc][0], qf[qb][kc][1], qf[qb][kc][NPL - 1]);
        }
    }

    f32x16 o[QB][NCT], ox[F16 ? QB : 1][F16 ? NCT : 1];   // ox: the cross products of the fp16 pair
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int t = 0; t < NCT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                o[qb][t][e] = 0.f;
                if constexpr (F16) ox[qb][t][e] = 0.f;
            }
    float m_run[QB], l_run[QB];
    const float slack = p.lse ? 0.f : kMaxSlack;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -INFINITY;
        l_run[qb] = 0.f;
    }

    f32x4 kreg[KPT], vreg[VPT];
    // K and V rows come through a range-checked descriptor over the sample's [n][lq] block: rows past the last key
    // and channels past the last one read as zero without a branch
    const __amdgpu_buffer_rsrc_t rs_kv = ddpm::buffer_over(base, static_cast<long long>(p.n) * p.lq * 4ll);
    // PRE: the sample's v planes (n * c * 4 bytes, tile t at t * 128 * c)
    const __amdgpu_buffer_rsrc_t rs_v = ddpm::buffer_over(
        PRE ? reinterpret_cast<const float*>(p.vpl + static_cast<int64_t>(b) * p.n * p.c * 4) : base, PRE ? static_cast<long long>(p.n) * p.c * 4ll : 0ll);
    auto load_tile = [&](int j0) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (DQ / 4), c4 = f % (DQ / 4);
            kreg[i] = ddpm::buffer_load4(rs_kv, f < KF4 ? static_cast<unsigned>((j0 + row) * p.lq + p.dqp + c4 * 4) * 4u : ddpm::kOob);
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = tid + 256 * i;
            if constexpr (PRE) {
                const int g = f / CS, ch = f % CS;   // g = (chunk of the tile, h, plane)
                vreg[i] = ddpm::buffer_load4(rs_v, cbase + ch < p.c ? static_cast<unsigned>(((j0 >> 4) * 4 + g) * p.c + cbase + ch) * 16u : ddpm::kOob);
            } else {
                const int row = f / (CS / 4), c4 = f % (CS / 4);
                vreg[i] = ddpm::buffer_load4(rs_kv, cbase + c4 * 4 < p.c ? static_cast<unsigned>((j0 + row) * p.lq + 2 * p.dqp + cbase + c4 * 4) * 4u : ddpm::kOob);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (DQ / 4), c4 = f % (DQ / 4);
            if (f < KF4) {
                if constexpr (PRE) unzip4_store(kreg[i], &sK[buf][row * KROW + c4 * 8], KPL);
                else if constexpr (F16) split4_store_f16(kreg[i], &sK[buf][row * KROW + c4 * 8], KPL, kvw);
                else split4_store(kreg[i], &sK[buf][row * KROW + c4 * 8], KPL);
            }
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = tid + 256 * i;
            const int row = f / (CS / 4), c4 = f % (CS / 4);
            if constexpr (PRE) *reinterpret_cast<f32x4*>(&sV[buf][f * 16]) = vreg[i];
            else if constexpr (F16) split4_store_f16(vreg[i], &sV[buf][row * VROW + c4 * 8], VPLANE, kvw);
            else split4_store(vreg[i], &sV[buf][row * VROW + c4 * 8], VPLANE);
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
        if (more && !(DDPM_ATTN_ABLATE & 4)) load_tile(j0 + KT);
        if (wave_active) {
            // ---- scores: lane = query, register = key rowmap(reg, h)
            f32x16 s[QB], sx[F16 ? QB : 1];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    s[qb][e] = 0.f;
                    if constexpr (F16) sx[qb][e] = 0.f;
                }
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                u32x4 kf[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) kf[pl] = *reinterpret_cast<const u32x4*>(&sK[buf][r * KROW + pl * KPL + kc * 32 + h * 16]);
                auto term = [&](f32x16 (&acc)[QB], int pa, int pb) {
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) acc[qb] = mfma16<F16>(kf[pa], qf[qb][kc][pb], acc[qb]);
                };
                if constexpr (F16) {
                    term(sx, 1, 0);
                    term(sx, 0, 1);
                    term(s, 0, 0);
                } else {
                    term(s, 2, 0);
                    term(s, 0, 2);
                    term(s, 1, 1);
                    term(s, 1, 0);
                    term(s, 0, 1);
                    term(s, 0, 0);
                }
            }
            if constexpr (F16) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) s[qb] = s[qb] + sx[qb] * kLoInv;
            }
            // ---- online softmax, probabilities split into their planes per 16-key chunk
            u32x4 pp[QB][2][NPL];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (DDPM_ATTN_ABLATE & 1) {
#pragma unroll
                    for (int ck = 0; ck < 2; ++ck)
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                            for (int e = 0; e < 4; ++e) pp[qb][ck][pl][e] = __float_as_uint(s[qb][(8 * ck + 4 * (pl & 1) + e) & 15]);
                    continue;
                }
                if (j0 + KT > p.n) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (j0 + rowmap(e, h) >= p.n) s[qb][e] = -INFINITY;
                }
                float mt = s[qb][0];
#pragma unroll
                for (int e = 1; e < 16; ++e) mt = fmaxf(mt, s[qb][e]);
                mt = fmaxf(mt, __shfl_xor(mt, 32));
                const float m_new = mt > m_run[qb] + slack ? mt : m_run[qb];       // (first tile: m_run = -inf, so m_new = mt)
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
                        for (int e = 0; e < 16; ++e) {
                            o[qb][ct][e] *= alpha;
                            if constexpr (F16) ox[qb][ct][e] *= alpha;
                        }
                }
#pragma unroll
                for (int ck = 0; ck < 2; ++ck) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = s[qb][8 * ck + j];
                    if constexpr (F16) split8_f16(v, pp[qb][ck][0], pp[qb][ck][1]);
                    else split8(v, pp[qb][ck][0], pp[qb][ck][1], pp[qb][ck][NPL - 1]);
                }
            }
            // ---- O^T += V^T P^T, two channel blocks at a time with their accumulators taken in turn (an MFMA that
            // accumulates into the registers the previous one wrote waits for it)
#pragma unroll
            for (int ck = 0; ck < 2; ++ck) {
#pragma unroll
                for (int ct0 = 0; ct0 < NCT; ct0 += 2) {
                    u32x4 vf[2][NPL];
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl) {
                            if ((DDPM_ATTN_ABLATE & 8) && (ck | ct0 | c2)) {
                                vf[c2][pl] = qf[0][0][pl];
                                continue;
                            }
                            if constexpr (PRE) {
                                vf[c2][pl] = *reinterpret_cast<const u32x4*>(&sV[buf][((((ck * 2 + h) * 2 + pl) * CS) + (ct0 + c2) * 32 + r) * 16]);
                                continue;
                            }
                            const unsigned char* src = &sV[buf][pl * VPLANE + (16 * ck) * VROW + (ct0 + c2) * 64 + vlane];
                            const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(src));
                            const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(src + 8 * VROW));
                            vf[c2][pl] = __builtin_bit_cast(u32x4, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
                        }
                    if constexpr (F16) {
                        // lo*hi and hi*lo into the cross accumulator, hi*hi into the main one
                        const int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};
#pragma unroll
                        for (int k = 0; k < 3; ++k)
#pragma unroll
                            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                                for (int qb = 0; qb < QB; ++qb) {
                                    if (k < 2) ox[qb][ct0 + c2] = mfma16<true>(vf[c2][pa[k]], pp[qb][ck][pb[k]], ox[qb][ct0 + c2]);
                                    else o[qb][ct0 + c2] = mfma16<true>(vf[c2][pa[k]], pp[qb][ck][pb[k]], o[qb][ct0 + c2]);
                                }
                    } else {
                        const int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            if ((DDPM_ATTN_ABLATE & 2) && k < 5) continue;
#pragma unroll
                            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                                for (int qb = 0; qb < QB; ++qb)
                                    o[qb][ct0 + c2] = mfma16<false>(vf[c2][pa[k]], pp[qb][ck][pb[k]], o[qb][ct0 + c2]);
                        }
                    }
                }
            }
        }
        if (more && !(DDPM_ATTN_ABLATE & 4)) store_tile(buf ^ 1);
        __syncthreads();
    }

    if constexpr (F16) ddpm::conv::raise_range_flag(p.range_flag, ddpm::conv::merged(amax, kvw));
    if (!wave_active) return;
    if constexpr (F16) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) o[qb][ct] = o[qb][ct] + ox[qb][ct] * kLoInv;
    }
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

template <int DQ, int QB, int NCT, bool F16, bool PRE = false>
__global__ __launch_bounds__(256, 2) void attention_fwd_x3_kernel(const AttnParams p) {
    constexpr int NPL = F16 ? 2 : 3, KD = DQ < 16 ? 16 : DQ;
    constexpr int KROW = NPL * KD * 2 + 16, VPLANE = KT * (32 * NCT * 2 + 64);   // as in the body
    __shared__ __attribute__((aligned(16))) unsigned char sK[2 * KT * KROW];
    __shared__ __attribute__((aligned(16))) unsigned char sV[2 * NPL * VPLANE];
    const dim3 bid = ddpm::xcd_block();   // the block to work on: neighbours in launch order share an XCD's L2
    if constexpr (PRE) attention_fwd_x3_body<DQ, QB, NCT, F16, false, true>(p, bid, sK, sV);
    else if (F16 && bid.x == 0) attention_fwd_x3_body<DQ, QB, NCT, F16, true>(p, bid, sK, sV);
    else attention_fwd_x3_body<DQ, QB, NCT, F16, false>(p, bid, sK, sV);
}

// Software-pipelined version of the kernel above (one 32-query block per wave). A wave of the kernel above goes
// through its tile in phases -- scores (MFMA), softmax and splitting (VALU), value product (MFMA), staging of the
// next tile (VALU) -- and the two waves that share a SIMD fall into step with each other, so that the matrix pipe
// idles through every VALU phase. Here iteration t issues the value-product MFMAs of tile t one by one and puts a
// slice of the VALU work of tile t + 1 behind each of them: its softmax (the scores of tile t + 1 are the first
// MFMAs of the iteration, from a K tile staged one iteration further ahead, hence three K buffers), the split of its
// probabilities, and the splitting + LDS stores of the tiles being staged. A VALU instruction issues while the MFMA
// in front of it executes, so the wave keeps both pipes busy by itself. Same operations on the same values in the
// same order per query as the kernel above: the results are bit-identical.
// KVW: this workgroup watches the range of the K and V tiles it stages. Every query block of a sample stages the same
// K and V, so one of them (the first) does the checking for all; each checks its own queries.
// PRE: k and v arrive split (AttnParams::vpl): the staging is a copy and the V fragments are plain 16-byte reads.
template <int DQ, int NCT, int NW, bool F16, bool KVW, bool PRE = false>
__device__ __forceinline__ void attention_fwd_x3p_body(const AttnParams& p, const dim3 bid) {
    static_assert(!PRE || (F16 && !KVW), "pre-split operands are fp16 pairs, checked where they were split");
    if constexpr (F16) ddpm::conv::fp16_saturate();
    ddpm::conv::RangeWatch amax;                   // F16: largest magnitude of q this thread has turned into an fp16 pair
    ddpm::conv::NoWatch unwatched_kv;
    auto& kvw = [&]() -> auto& {   // k and v: watched by `amax` too, or not at all
        if constexpr (KVW) return amax;
        else return unwatched_kv;
    }();
    constexpr int NPL = F16 ? 2 : 3;    // operand planes
    constexpr int NPR = F16 ? 3 : 6;    // MFMAs per fragment pair
    constexpr int NT = 64 * NW;
    constexpr int CS = 32 * NCT;
    constexpr int KD = DQ < 16 ? 16 : DQ, KC = KD / 16, KPL = KD * 2, KROW = NPL * KPL + 16;
    constexpr int VROW = CS * 2 + 64, VPLANE = KT * VROW;
    constexpr int KF4 = KT * DQ / 4, VF4 = KT * CS / 4;
    constexpr int KPT = (KF4 + NT - 1) / NT, VPT = VF4 / NT;
    static_assert(VF4 % NT == 0, "V tile must split evenly over the workgroup");
    constexpr int KBUF = KT * KROW, VBUF = PRE ? 16 * VF4 : NPL * VPLANE;   // PRE: 16-byte pieces [chunk][h][plane][channel]
    constexpr float kLog2e = 1.4426950408889634f;

    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    unsigned char* sK = psm;                // [3][KBUF]
    unsigned char* sV = psm + 3 * KBUF;     // [2][VBUF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int b = bid.z;
    const int cbase = bid.y * CS;
    const int q0 = (bid.x * NW + wave) * 32;
    const bool wave_active = q0 < p.n;          // wave-uniform
    const float* base = p.qkv + static_cast<int64_t>(b) * p.n * p.lq;

    if (DQ < 16) {   // the upper half of the padded depth stays zero
        for (int i = tid; i < 3 * KBUF / 4; i += NT) reinterpret_cast<unsigned*>(sK)[i] = 0u;
    }

    u32x4 qf[KC][NPL];
    {
        const int qi = q0 + r;
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
            if constexpr (F16) split8_f16(v, qf[kc][0], qf[kc][1], amax);
            else split8(v, qf[kc][0], qf[kc][1], qf[kc][NPL - 1]);
        }
    }

    f32x16 o[NCT], ox[F16 ? NCT : 1];   // ox: the cross products of the fp16 pair
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            o[t][e] = 0.f;
            if constexpr (F16) ox[t][e] = 0.f;
        }
    float m_run = -INFINITY, l_run = 0.f;
    const float slack = p.lse ? 0.f : kMaxSlack;

    f32x4 kreg[KPT], vreg[VPT];
    // K and V rows come through a range-checked descriptor over the sample's [n][lq] block: rows past the last key
    // and channels past the last one read as zero without a branch
    const __amdgpu_buffer_rsrc_t rs_kv = ddpm::buffer_over(base, static_cast<long long>(p.n) * p.lq * 4ll);
    auto load_k = [&](int j0) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int f = tid + NT * i;
            const int row = f / (DQ / 4), c4 = f % (DQ / 4);
            kreg[i] = ddpm::buffer_load4(rs_kv, f < KF4 ? static_cast<unsigned>((j0 + row) * p.lq + p.dqp + c4 * 4) * 4u : ddpm::kOob);
        }
    };
    // PRE: the sample's v planes (n * c * 4 bytes, tile t at t * 128 * c)
    const __amdgpu_buffer_rsrc_t rs_v = ddpm::buffer_over(
        PRE ? reinterpret_cast<const float*>(p.vpl + static_cast<int64_t>(b) * p.n * p.c * 4) : base, PRE ? static_cast<long long>(p.n) * p.c * 4ll : 0ll);
    auto load_v = [&](int j0) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = tid + NT * i;
            if constexpr (PRE) {
                const int g = f / CS, ch = f % CS;   // g = (chunk of the tile, h, plane)
                vreg[i] = ddpm::buffer_load4(rs_v, cbase + ch < p.c ? static_cast<unsigned>(((j0 >> 4) * 4 + g) * p.c + cbase + ch) * 16u : ddpm::kOob);
            } else {
                const int row = f / (CS / 4), c4 = f % (CS / 4);
                vreg[i] = ddpm::buffer_load4(rs_kv, cbase + c4 * 4 < p.c ? static_cast<unsigned>((j0 + row) * p.lq + 2 * p.dqp + cbase + c4 * 4) * 4u : ddpm::kOob);
            }
        }
    };
    auto store_k1 = [&](int i, unsigned char* dst) {
        const int f = tid + NT * i;
        const int row = f / (DQ / 4), c4 = f % (DQ / 4);
        if (f < KF4) {
            if constexpr (PRE) unzip4_store(kreg[i], dst + row * KROW + c4 * 8, KPL);
            else if constexpr (F16) split4_store_f16(kreg[i], dst + row * KROW + c4 * 8, KPL, kvw);
            else split4_store(kreg[i], dst + row * KROW + c4 * 8, KPL);
        }
    };
    auto store_v1 = [&](int i, unsigned char* dst) {
        const int f = tid + NT * i;
        const int row = f / (CS / 4), c4 = f % (CS / 4);
        if constexpr (PRE) *reinterpret_cast<f32x4*>(dst + f * 16) = vreg[i];
        else if constexpr (F16) split4_store_f16(vreg[i], dst + row * VROW + c4 * 8, VPLANE, kvw);
        else split4_store(vreg[i], dst + row * VROW + c4 * 8, VPLANE);
    };

    const int vlane = (4 * h + ((lane & 15) >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int tiles = (p.n + KT - 1) / KT;

    // The score MFMAs of one depth chunk, in issue order: (K plane, Q plane, into the cross accumulator?)
    constexpr int spa[6] = {F16 ? 1 : 2, 0, F16 ? 0 : 1, 1, 0, 0}, spb[6] = {0, F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0};
    // scores of one tile: lane = query, register = key rowmap(reg, h)
    auto scores = [&](const unsigned char* kb, f32x16& s) {
        f32x16 sx;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = sx[e] = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            u32x4 kf[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) kf[pl] = *reinterpret_cast<const u32x4*>(kb + r * KROW + pl * KPL + kc * 32 + h * 16);
#pragma unroll
            for (int k = 0; k < NPR; ++k) {
                if (F16 && k < 2) sx = mfma16<F16>(kf[spa[k]], qf[kc][spb[k]], sx);
                else s = mfma16<F16>(kf[spa[k]], qf[kc][spb[k]], s);
            }
        }
        if constexpr (F16) s = s + sx * kLoInv;
    };

    // The softmax of one tile in 29 slices (the arithmetic of the kernel above, cut up): s holds scores on entry and
    // probabilities after slice 19; pn receives their planes; alpha the factor that the accumulators owe.
    constexpr int SM = 29;
    float mt = 0.f, m_new = 0.f, alpha = 1.f, ps = 0.f;
    auto softmax_slice = [&](auto ic, f32x16& s, u32x4 (&pn)[2][NPL], int j0) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i == 0) {
            if (j0 + KT > p.n) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (j0 + rowmap(e, h) >= p.n) s[e] = -INFINITY;
            }
            mt = s[0];
#pragma unroll
            for (int e = 1; e < 6; ++e) mt = fmaxf(mt, s[e]);
        } else if constexpr (i == 1) {
#pragma unroll
            for (int e = 6; e < 11; ++e) mt = fmaxf(mt, s[e]);
        } else if constexpr (i == 2) {
#pragma unroll
            for (int e = 11; e < 16; ++e) mt = fmaxf(mt, s[e]);
            mt = fmaxf(mt, __shfl_xor(mt, 32));
        } else if constexpr (i == 3) {
            m_new = mt > m_run + slack ? mt : m_run;                 // (first tile: m_run = -inf, so m_new = mt)
            alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);   // 0 on the first tile
            ps = 0.f;
        } else if constexpr (i < 20) {
            constexpr int e = i - 4;
            s[e] = __builtin_amdgcn_exp2f((s[e] - m_new) * kLog2e);
            ps += s[e];
        } else if constexpr (i == 20) {
            ps += __shfl_xor(ps, 32);
            l_run = l_run * alpha + ps;
            m_run = m_new;
        } else {
            // elements 2j, 2j + 1 -> one 32-bit word of each plane
            constexpr int j = i - 21, ck = j >> 2, w = j & 3;
            if constexpr (F16) {
                unsigned a, b;
                ddpm::conv::split2_pair(ddpm::conv::f32x2{s[2 * j], s[2 * j + 1]}, a, b);
                pn[ck][0][w] = a;
                pn[ck][1][w] = b;
            } else {
                unsigned a, b, c;
                split2(s[2 * j], s[2 * j + 1], a, b, c);
                pn[ck][0][w] = a;
                pn[ck][1][w] = b;
                pn[ck][NPL - 1][w] = c;
            }
        }
    };

    auto read_vf = [&](u32x4 (&vf)[NPL], const unsigned char* vb, int blk) {
        const int ck = blk / NCT, ct = blk % NCT;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            if constexpr (PRE) {
                vf[pl] = *reinterpret_cast<const u32x4*>(vb + ((((ck * 2 + h) * 2 + pl) * CS) + ct * 32 + r) * 16);
                continue;
            }
            const unsigned char* src = vb + pl * VPLANE + (16 * ck) * VROW + ct * 64 + vlane;
            const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(src));
            const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(src + 8 * VROW));
            vf[pl] = __builtin_bit_cast(u32x4, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };

    // ---- prologue: K(0), V(0), K(1) staged; scores and softmax of tile 0
    load_k(0);
    load_v(0);
    __syncthreads();   // the zero fill of sK, if any
#pragma unroll
    for (int i = 0; i < KPT; ++i) store_k1(i, sK);
#pragma unroll
    for (int i = 0; i < VPT; ++i) store_v1(i, sV);
    if (tiles > 1) {
        load_k(KT);
#pragma unroll
        for (int i = 0; i < KPT; ++i) store_k1(i, sK + KBUF);
    }
    __syncthreads();

    u32x4 pp[2][NPL];
    if (wave_active) {
        f32x16 s0;
        scores(sK, s0);
        ddpm::static_for<SM>([&](auto ic) { softmax_slice(ic, s0, pp, 0); });
        // alpha = exp2(-inf) = 0 and the accumulators are zero: nothing owed
    }

    // The value product takes its (16-key chunk, 32-channel block) pieces two at a time and alternates between their
    // accumulators, and the scores of the next tile are slipped in between the first of them: an MFMA that accumulates
    // into the registers the previous one wrote waits for it (measured: a chain of dependent v_mfma_f32_32x32x16_bf16
    // runs at half the rate of two interleaved chains).
    constexpr int NB = 2 * NCT;         // pieces
    constexpr int M = NPR * NB;         // their MFMAs
    constexpr int NS = NPR * KC;        // MFMAs of one tile of scores
    constexpr int LEAD = NS + 2;        // value-product MFMAs in front of the first softmax slice (the scores must have left the pipe)
    static_assert(NCT % 2 == 0 && LEAD < M, "pairs of channel blocks; the scores fit in front of the softmax");
    auto body = [&](auto next_tag, int t) {
        constexpr bool NEXT = decltype(next_tag)::value;
        const int j0n = (t + 1) * KT;
        const bool more2 = t + 2 < tiles;
        const unsigned char* vb = sV + (t & 1) * VBUF;
        unsigned char* vnext = sV + ((t + 1) & 1) * VBUF;
        unsigned char* knext2 = sK + ((t + 2) % 3) * KBUF;
        if (NEXT && !(DDPM_ATTN_ABLATE & 32)) {
            load_v(j0n);
            if (more2) load_k(j0n + KT);
        }
        constexpr int P = NEXT ? SM + VPT + KPT : 0;     // VALU slices of this iteration
        if (wave_active) {
            f32x16 sn, snx;
            u32x4 pn[2][NPL];
            u32x4 vf[2][2][NPL], kf[KC][NPL];
            read_vf(vf[0][0], vb, 0);
            read_vf(vf[0][1], vb, 1);
            if (NEXT) {
                const unsigned char* kb = sK + ((t + 1) % 3) * KBUF;
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) kf[kc][pl] = *reinterpret_cast<const u32x4*>(kb + r * KROW + pl * KPL + kc * 32 + h * 16);
#pragma unroll
                for (int e = 0; e < 16; ++e) sn[e] = snx[e] = 0.f;
            }
            ddpm::static_for<M>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int pr = i / (2 * NPR), k = (i % (2 * NPR)) >> 1, c2 = i & 1;
                constexpr int blk = 2 * pr + c2, ck = blk / NCT, ct = blk % NCT;
                if constexpr (i % (2 * NPR) == 0 && 2 * pr + 2 < NB) {
                    if constexpr (DDPM_ATTN_ABLATE & 64) {
#pragma unroll
                        for (int q = 0; q < 2 * NPL; ++q) vf[(pr + 1) & 1][q / NPL][q % NPL] = vf[pr & 1][q / NPL][q % NPL];
                    } else {
                        read_vf(vf[(pr + 1) & 1][0], vb, 2 * pr + 2);
                        read_vf(vf[(pr + 1) & 1][1], vb, 2 * pr + 3);
                    }
                }
                if constexpr (F16 && k < 2) ox[ct] = mfma16<F16>(vf[pr & 1][c2][spa[k]], pp[ck][spb[k]], ox[ct]);
                else o[ct] = mfma16<F16>(vf[pr & 1][c2][spa[k]], pp[ck][spb[k]], o[ct]);
                if constexpr (NEXT) {
                    if constexpr (i < NS) {
                        if constexpr (F16 && i % NPR < 2) snx = mfma16<F16>(kf[i / NPR][spa[i % NPR]], qf[i / NPR][spb[i % NPR]], snx);
                        else sn = mfma16<F16>(kf[i / NPR][spa[i % NPR]], qf[i / NPR][spb[i % NPR]], sn);
                    }
                    constexpr int lo = i < LEAD ? 0 : (i - LEAD) * P / (M - LEAD);
                    constexpr int hi = i < LEAD ? 0 : (i + 1 == M ? P : (i + 1 - LEAD) * P / (M - LEAD));
                    ddpm::static_for<hi - lo>([&](auto dc) {
                        constexpr int sl = lo + decltype(dc)::value;
                        if constexpr (sl < SM) {
                            if constexpr (F16 && sl == 0) sn = sn + snx * kLoInv;   // the tile's scores are complete
                            if constexpr (!(DDPM_ATTN_ABLATE & 16)) softmax_slice(std::integral_constant<int, sl>{}, sn, pn, j0n);
                            else if constexpr (sl == 0) {
#pragma unroll
                                for (int q = 0; q < 8 * NPL; ++q) pn[q / (4 * NPL)][(q / 4) % NPL][q & 3] = __float_as_uint(sn[q & 15]);
                            }
                        } else if constexpr (DDPM_ATTN_ABLATE & 32) {
                        } else if constexpr (sl < SM + VPT) store_v1(sl - SM, vnext);
                        else if (more2) store_k1(sl - SM - VPT, knext2);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            if (NEXT) {
                if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {   // the running maximum moved for some query
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            o[ct][e] *= alpha;
                            if constexpr (F16) ox[ct][e] *= alpha;
                        }
                }
#pragma unroll
                for (int ck = 0; ck < 2; ++ck)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) pp[ck][pl] = pn[ck][pl];
            }
        } else if (NEXT) {
            // (the registers pass through an empty asm so that this copy of the splitting code is not merged with the
            // slices above and hoisted in front of both, where it would wait for the loads at the top of the iteration)
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                asm volatile("" : "+v"(vreg[i]));
                store_v1(i, vnext);
            }
            if (more2) {
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    asm volatile("" : "+v"(kreg[i]));
                    store_k1(i, knext2);
                }
            }
        }
    };
    for (int t = 0; t + 1 < tiles; ++t) {
        body(std::true_type{}, t);
        if (!(DDPM_ATTN_ABLATE & 128)) __syncthreads();
    }
    body(std::false_type{}, tiles - 1);

    if constexpr (F16) ddpm::conv::raise_range_flag(p.range_flag, ddpm::conv::merged(amax, kvw));
    if (!wave_active) return;
    if constexpr (F16) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) o[ct] = o[ct] + ox[ct] * kLoInv;
    }
    const float g = p.gamma[0];
    const int qi = q0 + r;
    if (qi >= p.n) return;
    const float inv_l = 1.0f / l_run;
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

template <int DQ, int NCT, int NW, bool F16, bool PRE = false>
__global__ __launch_bounds__(64 * NW, 2) void attention_fwd_x3p_kernel(const AttnParams p) {
    const dim3 bid = ddpm::xcd_block();   // the block to work on: neighbours in launch order share an XCD's L2
    if constexpr (PRE) attention_fwd_x3p_body<DQ, NCT, NW, F16, false, true>(p, bid);
    else if (F16 && bid.x == 0) attention_fwd_x3p_body<DQ, NCT, NW, F16, true>(p, bid);
    else attention_fwd_x3p_body<DQ, NCT, NW, F16, false>(p, bid);
}

template <int DQ, int NCT, int NW, bool F16, bool PRE = false>
int launch_x3p_math(const AttnParams& p, int batch, hipStream_t s) {
    constexpr int KD = DQ < 16 ? 16 : DQ, NPL = F16 ? 2 : 3;
    constexpr int lds = 3 * KT * (NPL * KD * 2 + 16) + 2 * (PRE ? 16 * KT * 32 * NCT / 4 : NPL * KT * (32 * NCT * 2 + 64));
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_fwd_x3p_kernel<DQ, NCT, NW, F16, PRE>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) return ddpm::fail_invalid("LDS size for the attention forward");
    const dim3 g((p.n + 32 * NW - 1) / (32 * NW), (p.c + 32 * NCT - 1) / (32 * NCT), batch);
    attention_fwd_x3p_kernel<DQ, NCT, NW, F16, PRE><<<g, 64 * NW, lds, s>>>(p);
    return ddpm::check_launch("attention_fwd_x3p_kernel");
}

// Which calls take the fp16 pair (when it is the process's arithmetic): those with at most 16 query channels. The pair
// needs a second set of output accumulators, and with 32 query channels (two depth chunks of Q and K fragments besides)
// that no longer fits the register file at two waves per SIMD; that shape (256 channels, maps of 16 x 16 and smaller) is
// short anyway. Training calls (a log-sum-exp is requested) take it too: the backward kernels of these shapes recompute
// the probabilities from that log-sum-exp in the same arithmetic (attention_bwd.hip), so that scores are rounded the same
// way on both sides. DDPM_ATTN_TRAIN_F16=0 (read once, by both files) keeps training calls on three bf16 planes:
// a diagnostic switch for A/B measurements.
template <int DQ>
bool fp16_pair_call(const AttnParams& p) {
    static const bool train_too = [] {
        const char* e = getenv("DDPM_ATTN_TRAIN_F16");
        return !(e && strcmp(e, "0") == 0);
    }();
    return DQ <= 16 && ddpm::f16_math() && (p.lse == nullptr || train_too);
}

template <int DQ, int NCT, int NW>
int launch_x3p(const AttnParams& p, int batch, hipStream_t s) {
    if constexpr (DQ <= 16) {
        if (fp16_pair_call<DQ>(p)) return launch_x3p_math<DQ, NCT, NW, true>(p, batch, s);
    }
    return launch_x3p_math<DQ, NCT, NW, false>(p, batch, s);
}

// DDPM_ATTN_PIPE=0 (read once) selects the phase-by-phase kernel instead of the software-pipelined one: a diagnostic
// switch for A/B measurements.
inline bool pipelined_attention() {
    static const bool on = [] {
        const char* e = getenv("DDPM_ATTN_PIPE");
        return !(e && strcmp(e, "0") == 0);
    }();
    return on;
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

// Shortest sequence that takes the split-bf16 kernels (DDPM_ATTN_X3_MIN_N overrides it: a diagnostic switch).
inline int x3_min_n() {
    static const int n = [] {
        const char* e = getenv("DDPM_ATTN_X3_MIN_N");
        return e ? atoi(e) : 128;
    }();
    return n;
}

template <int DQ>
int launch_dq(const AttnParams& p, int batch, hipStream_t s) {
    if constexpr (DQ <= 64) {   // 128 query channels (C = 1024): only the bottleneck has them, its maps are short
      if (ddpm::split_math() && p.n >= x3_min_n() && p.c >= 64) {
        // 128 channels and more: the software-pipelined kernel (2-6 % faster there; 64-channel tensors are faster with two
        // query blocks per wave, and with 64 query channels the scores alone are half of a tile's MFMAs)
        if constexpr (DQ <= 32) {
            if (p.c >= 128 && pipelined_attention()) return launch_x3p<DQ, 4, 4>(p, batch, s);
        }
        const bool f16 = fp16_pair_call<DQ>(p);
        if (p.c >= 128 && wide_attention()) {
            const dim3 g((p.n + 127) / 128, (p.c + 127) / 128, batch);
            if constexpr (DQ <= 16) {
                if (f16) {
                    attention_fwd_x3_kernel<DQ, 1, 4, true><<<g, 256, 0, s>>>(p);
                    return ddpm::check_launch("attention_fwd_x3_kernel");
                }
            }
            attention_fwd_x3_kernel<DQ, 1, 4, false><<<g, 256, 0, s>>>(p);
        } else {
            const dim3 g((p.n + 255) / 256, (p.c + 63) / 64, batch);
            if constexpr (DQ <= 16) {
                if (f16) {
                    attention_fwd_x3_kernel<DQ, 2, 2, true><<<g, 256, 0, s>>>(p);
                    return ddpm::check_launch("attention_fwd_x3_kernel");
                }
            }
            attention_fwd_x3_kernel<DQ, 2, 2, false><<<g, 256, 0, s>>>(p);
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

namespace ddpm {

// The two shapes with pair kernels behind them: 64 channels (two query blocks per wave) and 128 (software-pipelined),
// when those are the kernels ddpm_attention_fwd itself would run.
bool attention_fwd_pairs_fits(int batch, int n, int c, int dq) {
    if (!f16_math() || batch > 65535 || (n & 31) != 0 || n < x3_min_n()) return false;
    if (small_attention() && attention_small_fwd_fits(n, c, dq)) return false;
    return (c == 64 && dq == 8) || (c == 128 && dq == 16 && pipelined_attention());
}

int attention_fwd_pairs(const float* qk, const unsigned char* vpl, const float* x, int ldx, const float* gamma, const float* temb,
                        float* out, int ldo, int batch, int n, int c, int dq, hipStream_t s) {
    if (!attention_fwd_pairs_fits(batch, n, c, dq)) return fail_invalid("attention_fwd_pairs: not a shape of the pair kernels");
    AttnParams p;
    p.qkv = qk;
    p.vpl = vpl;
    p.x = x;
    p.gamma = gamma;
    p.temb = temb;
    p.out = out;
    p.attn_raw = nullptr;
    p.lse = nullptr;
    p.n = n;
    p.c = c;
    p.dqp = dq;
    p.lq = 2 * dq;
    p.ldx = ldx;
    p.ldo = ldo;
    p.range_flag = range_flag();
    if (!p.range_flag) return DDPM_E_LAUNCH;
    if (c == 128) return launch_x3p_math<16, 4, 4, true, true>(p, batch, s);
    const dim3 g((n + 255) / 256, 1, batch);
    attention_fwd_x3_kernel<8, 2, 2, true, true><<<g, 256, 0, s>>>(p);
    return check_launch("attention_fwd_x3_kernel (pairs)");
}

}  // namespace ddpm

extern "C" int ddpm_attention_fwd(const float* qkv, const float* x, int ldx, const float* gamma, float* out, int ldo,
                                  float* attn_raw, float* lse, const float* temb, int batch, int n, int c, int dq,
                                  void* stream) {
    DDPM_REQUIRE(qkv && x && gamma && out && batch > 0 && n > 0 && c > 0);
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
    p.vpl = nullptr;
    p.range_flag = ddpm::f16_math() ? ddpm::range_flag() : nullptr;
    if (ddpm::f16_math() && !p.range_flag) return DDPM_E_LAUNCH;
    hipStream_t s = ddpm::as_stream(stream);
    if (ddpm::small_attention() && ddpm::attention_small_fwd_fits(n, c, dq))
        return ddpm::attention_small_fwd(qkv, x, ldx, gamma, temb, out, ldo, attn_raw, lse, batch, n, c, dq, s);
    // the sample index is the grid's z axis (at most 65535): a larger batch runs as consecutive launches over its parts
    constexpr int kMaxZ = 65535;
    for (int b0 = 0; b0 < batch; b0 += kMaxZ) {
        const int nb = batch - b0 < kMaxZ ? batch - b0 : kMaxZ;
        const int64_t r0 = static_cast<int64_t>(b0) * n;
        AttnParams q = p;
        q.qkv = p.qkv + r0 * p.lq;
        q.x = p.x + r0 * ldx;
        q.out = p.out + r0 * ldo;
        if (p.attn_raw) q.attn_raw = p.attn_raw + r0 * c;
        if (p.lse) q.lse = p.lse + r0;
        if (p.temb) q.temb = p.temb + static_cast<int64_t>(b0) * c;
        int rc;
        switch (dq) {
            case 8: rc = launch_dq<8>(q, nb, s); break;
            case 16: rc = launch_dq<16>(q, nb, s); break;
            case 32: rc = launch_dq<32>(q, nb, s); break;
            case 64: rc = launch_dq<64>(q, nb, s); break;
            default: rc = launch_dq<128>(q, nb, s); break;
        }
        if (rc) return rc;
    }
    return DDPM_OK;
}

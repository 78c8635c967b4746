// 3x3 convolution (padding 1) as implicit GEMM on split-bf16 MFMA with the activation tile resident
// in LDS across the nine taps.
//
// A workgroup of 4 waves (2 x 2; two workgroups per CU, so that one's prologue, flush and epilogue
// overlap the other's matrix work) owns a run of 32-pixel row groups of one 64-column tile and walks
// it in tiles of up to 8 groups (256 consecutive pixels of the flattened (b, y, x) axis). For every
// 16-channel chunk it stages the tile's pixels plus a halo ONCE -- fp32 from global memory, split into
// three bf16 planes in registers, 112-byte LDS rows -- and the nine taps then read their operand
// fragments from that one tile at a fixed row offset per tap.
//
// The LDS tile holds consecutive pixels of the flattened axis, W + 1 before and after the tile's own.
// Tap (dy, dx) of the pixel in LDS row R is row R + dy*W + dx; a lane whose neighbour in that
// direction does not exist (image border, other sample) reads from a small region of zeros instead,
// at the same address modulo 256 bytes -- the banks its own row would have used -- so the 16-byte
// fragment reads (consecutive lanes on consecutive 112-byte rows) stay conflict-free. Which
// neighbours exist is four wave-level lane masks per row group, computed once per tile.
//
// Weights never touch LDS: the packed buffer is fragment-major (one 1 KiB block per (tap, chunk,
// 32-column block, plane)), so each wave loads exactly its own operand registers with coalesced
// 16-byte loads, one step ahead of the MFMAs. A wave computes (4 row groups) x (one 32-column block)
// = 4 accumulators, 24 MFMAs per step; the next chunk's activations are staged in passes of 64
// rows, one pass per tap, into the other LDS buffer: one barrier per chunk.
// Accuracy scheme as in conv_gemm.hip: six cross products, MFMA partial sums folded into a master
// accumulator (round-to-nearest) after every chunk.
//
// With F16 the operands are fp16 pairs instead (conv_common.h): two planes per row (80-byte LDS rows), three
// MFMAs per fragment pair -- the hi*hi products in one accumulator, the two cross products in a second one that is
// scaled by 2^-11 and added after the last chunk -- and no fold.
//
// With STATS the epilogue also sums the values it stores and their squares per output channel (train-mode BatchNorm
// statistics of the convolution's output without another pass over it): the 16 values of a register block in fp32
// (round-to-nearest, so the errors of the thousands of blocks of a channel average out), those sums in float64 per lane
// across the workgroup's whole run, then across the two lane halves and the two waves that share a column block, and one
// (sum, sum of squares) pair per workgroup and channel goes to p.stat_partial in the layout the BatchNorm finishing
// kernel reads. The order of every addition is fixed: the result is deterministic.
#include <type_traits>

#include "conv_common.h"

namespace ddpm {
namespace conv {
namespace {

#ifndef DDPM_HALO_ABLATE
#define DDPM_HALO_ABLATE 0   // diagnostic builds only (wrong results): bit 0 no staging in the chunk loop, bit 1 no weight loads in the loop, bit 2 no LDS fragment reads in the loop, bit 3 no master fold, bit 4 no epilogue, bit 5 fragment reads at tap 0 only (real data, 1/9 of the reads), bit 6 no border select on the fragment addresses
#endif
constexpr int kThreads = 256;
constexpr int kPassRows = kThreads / 4;    // LDS rows staged per pass
constexpr int NPASS = 6;                   // staging passes per chunk (<= 384 LDS rows)
constexpr int kMaxLds = 160 * 1024 - 4096;
constexpr int WM = 2, WN = 2;              // wave grid
constexpr int BN = WN * 32;

// WATCH: this workgroup watches the range of the activations it stages (F16). The column tiles of one run of rows stage
// the same activations, so the first of them does the checking for all.
template <bool DUAL, bool F16, bool STATS, bool WATCH>
__device__ __forceinline__ void conv3x3_halo_body(const GemmParams& p, int id, unsigned char* hsm, int* rowb) {
    if constexpr (F16) fp16_saturate();
    std::conditional_t<WATCH, RangeWatch, NoWatch> amax;   // F16: largest magnitude this thread has turned into an fp16 pair
    constexpr int XR = F16 ? XROW_F16 : XROW;   // LDS bytes per tile row
    constexpr int NPL = F16 ? 2 : 3;            // operand planes

    const int W = p.in_w, H = p.in_h, HW = H * W;
    const int HBUF = p.hrows * XR;
    const int ZOFF = (2 * HBUF + 255) & ~255;   // 384 bytes of zeros behind the two buffers, 256-byte aligned

    const int nwg = gridDim.x;
    const int n_tile = id % p.n_tiles;
    const int m_block = id / p.n_tiles;
    const int n0 = n_tile * BN;
    const int g_begin = m_block * p.gpb;
    const int g_end = min(p.groups_total, g_begin + p.gpb);
    if (g_begin >= g_end) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int srow = tid >> 2, lq = tid & 3;   // staging: LDS row (+ 64 per pass), 4-channel quad

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
    // byte offset of this lane's fragment piece in a (tap, chunk) slab, or OOB past the last column block
    const int nb = (n0 >> 5) + wn;
    const unsigned wcol = nb < p.nbp ? static_cast<unsigned>(nb) * 3072u + lane * 16u : OOB;
    const unsigned chunk_stride_w = static_cast<unsigned>(p.nbp) * 3072u;
    const unsigned tap_stride_w = chunk_stride_w * static_cast<unsigned>(p.kc16);

    if (tid < 96) reinterpret_cast<unsigned*>(hsm + ZOFF)[tid] = 0u;

    // Tiles of this run: as even as possible, at most p.tile_groups groups each. The second half of the
    // grid (the workgroups that share CUs with the first half) starts with a half-size tile, so that the
    // two workgroups of a CU are out of phase: one's prologue, fold and epilogue then overlap the
    // other's MFMA steps instead of coinciding with them.
    const int cnt = g_end - g_begin;
    int first = 0;
    if (blockIdx.x >= (nwg >> 1) && cnt > p.tile_groups) first = max(2, (min(cnt, p.tile_groups) / 2) & ~1);
    const int rest = cnt - first;
    const int ntile = (rest + p.tile_groups - 1) / p.tile_groups;
    const int tq = rest / ntile, trem = rest - tq * ntile;

    double stat_sum = 0.0, stat_sq = 0.0;   // STATS: this lane's column, its rows of every tile of the run
    int g0 = g_begin;
    for (int t = first ? -1 : 0; t < ntile; ++t) {
        const int kg = t < 0 ? first : tq + (t < trem ? 1 : 0);
        const int m0 = g0 * 32;
        g0 += kg;
        const int hr = kg * 32 + 2 * W + 2;   // LDS rows in use: the tile's pixels and W + 1 on either side
        const int hp0 = m0 - W - 1;           // pixel behind LDS row 0 (may be negative)

        __syncthreads();   // the previous tile's epilogue reads rowb; its last chunk reads LDS
        {
            const int m = m0 + tid;
            rowb[tid] = (tid < kg * 32 && m < p.m_total) ? m / HW : -1;
        }

        // This lane's row in each of its (up to 4) row groups: LDS address, and which neighbours exist.
        // The four conditions are wave-level lane masks (SGPR pairs) that the tap selects combine.
        int abase[4];
        bool up[4], down[4], left[4], right[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (WM * i + wm) * 32 + r;
            const int m = m0 + row;
            const int yy = m / W;
            const int x = m - yy * W;
            const int y = yy % H;
            up[i] = y > 0;
            down[i] = y < H - 1;
            left[i] = x > 0;
            right[i] = x < W - 1;
            abase[i] = (row + W + 1) * XR + h * 16;
        }

        const int pix0 = __builtin_amdgcn_readfirstlane(max(hp0, 0));   // first pixel the tile can touch: base of the descriptors
        const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(p.src1 + static_cast<long long>(pix0) * p.ld1, (p.src1_pixels - pix0) * p.ld1 * 4ll);
        const __amdgpu_buffer_rsrc_t rs2 = make_rsrc(p.src2 + static_cast<long long>(pix0) * p.ld2, (p.src2_pixels - pix0) * p.ld2 * 4ll);
        const int shift0 = hp0 - pix0;   // <= 0: LDS row L holds descriptor pixel L + shift0

        // (with two sources a chunk of 16 channels may straddle their boundary: both loads are issued,
        // one of them out of range, and added at store time so that no wait follows the loads)
        auto load_rel = [&](int rel, int kc, f32x4& second) -> f32x4 {
            const bool ok = rel >= 0;
            const int c = kc * XK + lq * 4;
            if (DUAL) {
                const unsigned o1 = (ok && c < p.c1) ? static_cast<unsigned>(rel) * p.ld1 * 4u + c * 4u : OOB;
                const unsigned o2 = (ok && c >= p.c1 && c < p.ctot) ? static_cast<unsigned>(rel) * p.ld2 * 4u + (c - p.c1) * 4u : OOB;
                second = buf_load(rs2, o2);
                return buf_load(rs1, o1);
            }
            return buf_load(rs1, (ok && c < p.ctot) ? static_cast<unsigned>(rel) * p.ld1 * 4u + c * 4u : OOB);
        };
        // staging pass k: LDS row srow + 64*k; rows past the tile and pixels before the tensor load zeros
        // (pixels past its end are cut off by the descriptor)
        auto load_pass = [&](int k, int kc, f32x4& second) -> f32x4 {
            const int L = srow + kPassRows * k;
            return load_rel(L < hr ? L + shift0 : -1, kc, second);
        };
        auto store_pass = [&](const f32x4 v, int k, unsigned char* buf) {
            const int L = srow + kPassRows * k;
            if (L < p.hrows) {
                if constexpr (F16) split_store_f16(v, buf + L * XR + lq * 8, amax);
                else split_store(v, buf + L * XR + lq * 8);
            }
        };

        // prologue: chunk 0 into buffer 0
        {
            constexpr int PG = DUAL ? 3 : NPASS;   // passes in flight together (register budget)
#pragma unroll
            for (int k0 = 0; k0 < NPASS; k0 += PG) {
                f32x4 pre[PG], pre2[PG];
#pragma unroll
                for (int k = 0; k < PG; ++k) pre[k] = load_pass(k0 + k, 0, pre2[k]);
#pragma unroll
                for (int k = 0; k < PG; ++k) store_pass(DUAL ? pre[k] + pre2[k] : pre[k], k0 + k, hsm);
            }
        }

        const int my_slots = (kg - wm + WM - 1) / WM;
        f32x16 master[4];

        auto k_loop = [&](auto slots_tag) {
            constexpr int SL = decltype(slots_tag)::value;
            constexpr int SLA = SL > 0 ? SL : 1;
            f32x16 acc[SLA];
#pragma unroll
            for (int i = 0; i < SLA; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    acc[i][e] = 0.f;
                    master[i][e] = 0.f;
                }
            // Weight fragments alternate between two register sets by tap parity; after tap 8 the first
            // tap of the next chunk is parked in set 1 and moved to set 0 at the top of the chunk.
            u32x4 bf[2][NPL];
            auto load_b = [&](u32x4 (&dst)[NPL], int tap, int kc) {
                const unsigned base = static_cast<unsigned>(tap) * tap_stride_w + static_cast<unsigned>(kc) * chunk_stride_w;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) dst[pl] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wcol, base + pl * 1024u, 0);
            };
            // Activation fragments: plane 2 of the NEXT tap is requested as soon as this tap's only
            // term with plane 2 has issued, plane 1 after its last term, plane 0 at the top of the step
            // (it is needed last): the reads of a step are covered by its own MFMAs.
            u32x4 a0[SLA], a1[SLA], a2[F16 ? 1 : SLA];
            // Fragment of tap (dy, dx) for row group i: the row at the tap's offset, or -- for a lane whose
            // neighbour in that direction does not exist -- the zero region at the same offset modulo 256,
            // i.e. on the banks the row itself would have used: the read stays conflict-free.
            auto read_plane = [&](u32x4 (&dst)[SLA], int curoff, int tap, int pl) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                const int off = curoff + (dy * W + dx) * XR;
#pragma unroll
                for (int i = 0; i < SLA; ++i) {
                    const int nat = abase[i] + off;
                    bool ok = true;
                    if (dy < 0) ok = ok && up[i];
                    if (dy > 0) ok = ok && down[i];
                    if (dx < 0) ok = ok && left[i];
                    if (dx > 0) ok = ok && right[i];
                    const int addr = ((DDPM_HALO_ABLATE & 64) || (dy == 0 && dx == 0)) ? nat : (ok ? nat : (ZOFF | (nat & 255)));
                    dst[i] = *reinterpret_cast<const u32x4*>(hsm + addr + pl * 32);
                }
            };
            auto mma = [&](auto& c, const u32x4 (&a)[SLA], const u32x4& b) {
#pragma unroll
                for (int i = 0; i < SLA; ++i) c[i] = mfma16<F16>(a[i], b, c[i]);
            };
            if (SL > 0) load_b(bf[1], 0, 0);
            __syncthreads();
            // The two workgroups of a CU are out of phase: while this one multiplies, the other is mostly in its prologue or
            // epilogue (loads, GELU, stores). Raised priority for the matrix loop keeps its issue slots from waiting behind
            // that vector work: 1 % per launch on the large layers, 0.7 % of a sampling step.
            __builtin_amdgcn_s_setprio(2);

            for (int kc = 0; kc < p.kc16; ++kc) {
                const int curoff = (kc & 1) * HBUF;
                unsigned char* nxt = hsm + ((kc & 1) ^ 1) * HBUF;
                const bool has_next = kc + 1 < p.kc16;
                f32x4 areg = {0.f, 0.f, 0.f, 0.f}, areg2 = {0.f, 0.f, 0.f, 0.f};
                if (SL > 0) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) bf[0][pl] = bf[1][pl];
                    if (!(DDPM_HALO_ABLATE & 4)) {
                        if constexpr (!F16) read_plane(a2, curoff, 0, 2);
                        read_plane(a1, curoff, 0, 1);
                    }
                }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    if (SL > 0 && !(DDPM_HALO_ABLATE & 4) && (!(DDPM_HALO_ABLATE & 32) || tap == 0)) read_plane(a0, curoff, tap, 0);
                    // weights of the next step (after tap 8: the first tap of the next chunk)
                    if (SL > 0 && !(DDPM_HALO_ABLATE & 2)) {
                        if (tap < 8) load_b(bf[(tap + 1) & 1], tap + 1, kc);
                        else if (has_next) load_b(bf[1], 0, kc + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // the weight loads stay here, a full step ahead of their use
                    if (has_next && !(DDPM_HALO_ABLATE & 1)) {
                        if (tap >= 1 && tap <= NPASS) store_pass(DUAL ? areg + areg2 : areg, tap - 1, nxt);
                        if (tap < NPASS) areg = load_pass(tap, kc + 1, areg2);
                    }
                    if (SL > 0) {
                        const u32x4 (&b)[NPL] = bf[tap & 1];
                        if constexpr (F16) {
                            mma(master, a1, b[0]);   // lo * hi
                            if (tap < 8 && !(DDPM_HALO_ABLATE & (4 | 32))) read_plane(a1, curoff, tap + 1, 1);
                            mma(master, a0, b[1]);   // hi * lo
                            mma(acc, a0, b[0]);      // hi * hi
                        } else {
                            mma(acc, a2, b[0]);
                            if (tap < 8 && !(DDPM_HALO_ABLATE & (4 | 32))) read_plane(a2, curoff, tap + 1, 2);
                            mma(acc, a1, b[1]);
                            mma(acc, a1, b[0]);
                            if (tap < 8 && !(DDPM_HALO_ABLATE & (4 | 32))) read_plane(a1, curoff, tap + 1, 1);
                            mma(acc, a0, b[2]);
                            mma(acc, a0, b[1]);
                            mma(acc, a0, b[0]);
                        }
                    }
                }
                if (!F16 && SL > 0 && (!(DDPM_HALO_ABLATE & 8) || !has_next)) {
#pragma unroll
                    for (int i = 0; i < SLA; ++i) {
                        master[i] += acc[i];
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
                    }
                }
                __syncthreads();
            }
            __builtin_amdgcn_s_setprio(0);
            if (F16 && SL > 0) {
#pragma unroll
                for (int i = 0; i < SLA; ++i) master[i] = acc[i] + master[i] * kLoInv;
            }
        };
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        using T2 = std::integral_constant<int, 2>;
        using T3 = std::integral_constant<int, 3>;
        using T4 = std::integral_constant<int, 4>;
        if (my_slots == 4) k_loop(T4{});
        else if (my_slots == 3) k_loop(T3{});
        else if (my_slots == 2) k_loop(T2{});
        else if (my_slots == 1) k_loop(T1{});
        else k_loop(T0{});

        // ---- epilogue -----------------------------------------------------------------------
        // Stores and residual loads go through buffer descriptors that begin at the tile's first pixel
        // and end with the tensor: rows past the last pixel fall outside and are dropped (or read as
        // zero) by the hardware's range check, so no element carries a compare, a select or a branch.
        int n = n0 + wn * 32 + r;
        int hq = 4 * h;
        asm volatile("" : "+v"(n), "+v"(hq));   // keeps the 64 per-element offsets below from being hoisted out of the tile loop
        if (n < p.n_total && (!(DDPM_HALO_ABLATE & 16) || p.act == 77)) {
            const float sc = p.scale ? p.scale[n] : 1.0f;
            const float sh = p.shift ? p.shift[n] : 0.0f;
            const long long rows_left = p.m_total - m0;
            const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out + static_cast<int64_t>(m0) * p.ldo, rows_left * p.ldo * 4ll);
            const __amdgpu_buffer_rsrc_t rs_r =
                make_rsrc(p.residual ? p.residual + static_cast<int64_t>(m0) * p.ldr : p.out, p.residual ? rows_left * p.ldr * 4ll : 0ll);
            const unsigned vo = static_cast<unsigned>((wm * 32 + hq) * p.ldo + n) * 4u;
            const unsigned vr = static_cast<unsigned>((wm * 32 + hq) * p.ldr + n) * 4u;
            const unsigned ldo4 = static_cast<unsigned>(p.ldo) * 4u, ldr4 = static_cast<unsigned>(p.ldr) * 4u;
            // Every mode decision (residual, activation, time embedding) is taken once per 16 elements and each
            // phase runs over all 16: the conditions are uniform, but a branch per element costs more than the
            // arithmetic it guards, and a load per element would be waited for 64 times instead of four.
            const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(p.temb ? p.temb : p.out, p.temb ? static_cast<long long>(p.batch) * p.cout * 4ll : 0ll);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i >= my_slots) continue;
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = master[i][e] * sc + sh;
                if (p.residual) {
                    float rv[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const unsigned rel = static_cast<unsigned>(i * WM * 32 + 8 * (e >> 2) + (e & 3));   // + wm * 32 + hq
                        rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, vr + rel * ldr4, 0, 0));
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] += rv[e];
                }
                if (p.act == DDPM_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                } else if (p.act == DDPM_ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = apply_act_fast(v[e], DDPM_ACT_GELU);
                }
                if (p.temb) {
                    // sample of every row from the tile's table; rows past the last pixel (-1) read out of range: zero
                    float tvv[16];
#pragma unroll
                    for (int eq = 0; eq < 4; ++eq) {
                        const int4 bs = *reinterpret_cast<const int4*>(&rowb[i * WM * 32 + 8 * eq + wm * 32 + hq]);
                        const int bb[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
                        for (int ei = 0; ei < 4; ++ei)
                            tvv[eq * 4 + ei] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                rs_t, bb[ei] >= 0 ? static_cast<unsigned>(bb[ei] * p.cout + n) * 4u : OOB, 0, 0));
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] += tvv[e];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned rel = static_cast<unsigned>(i * WM * 32 + 8 * (e >> 2) + (e & 3));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), rs_o, vo + rel * ldo4, 0, DDPM_STORE_AUX);
                }
                if constexpr (STATS) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (i * WM + wm) * 32 + 8 * (e >> 2) + (e & 3) + hq;
                        const float x = m0 + row < p.m_total ? v[e] : 0.f;   // rows past the last pixel hold the bias alone
                        s1 += x;
                        s2 += x * x;
                    }
                    stat_sum += static_cast<double>(s1);
                    stat_sq += static_cast<double>(s2);
                }
            }
        }
    }
    if constexpr (F16 && WATCH) raise_range_flag(p.range_flag, amax);
    if constexpr (STATS) {
        // the two lane halves hold different rows of the same column; so do the two waves with the same wn
        stat_sum += __shfl_xor(stat_sum, 32);
        stat_sq += __shfl_xor(stat_sq, 32);
        __syncthreads();                       // the last tile's fragment reads are done: the tile buffers are free
        double* red = reinterpret_cast<double*>(hsm);
        if (wm == 1 && h == 0) {
            red[(wn * 32 + r) * 2] = stat_sum;
            red[(wn * 32 + r) * 2 + 1] = stat_sq;
        }
        __syncthreads();
        const int n = n0 + wn * 32 + r;
        if (wm == 0 && h == 0 && n < p.n_total) {
            double* dst = p.stat_partial + static_cast<int64_t>(m_block) * 2 * p.n_total + n;
            dst[0] = stat_sum + red[(wn * 32 + r) * 2];
            dst[p.n_total] = stat_sq + red[(wn * 32 + r) * 2 + 1];
        }
    }
}

template <bool DUAL, bool F16, bool STATS>
__global__ __launch_bounds__(kThreads, 2) void conv3x3_halo_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    __shared__ __attribute__((aligned(16))) int rowb[256];   // sample index of every tile row, -1 = no such pixel
    int id = blockIdx.x;
    {   // workgroups that share rows (the column tiles of one run) take consecutive numbers on ONE XCD
        const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = id & 7;
        id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
    }
    if (F16 && id % p.n_tiles == 0) conv3x3_halo_body<DUAL, F16, STATS, true>(p, id, hsm, rowb);
    else conv3x3_halo_body<DUAL, F16, STATS, false>(p, id, hsm, rowb);
}

template <bool DUAL, bool F16, bool STATS = false>
int launch_one(const GemmParams& p, int grid, int lds, hipStream_t s) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<DUAL, F16, STATS>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (attr != hipSuccess) {
        set_error("conv3x3_halo_kernel: cannot reserve %d bytes of LDS: %s", kMaxLds, hipGetErrorString(attr));
        return DDPM_E_LAUNCH;
    }
    conv3x3_halo_kernel<DUAL, F16, STATS><<<dim3(grid), dim3(kThreads), lds, s>>>(p);
    return check_launch("conv3x3_halo_kernel");
}

// LDS rows of a tile of g groups: its pixels and W + 1 pixels of halo on either side.
int tile_rows(int g, int w) { return g * 32 + 2 * w + 2; }

}  // namespace

int launch_conv3x3_halo(GemmParams& p, hipStream_t s) {
    // shapes left to the generic kernel: very few columns, or images too wide for the LDS tile
    if (p.n_total <= 32) return 1;
    if (p.m_total != p.batch * p.in_h * p.in_w) return 1;
    if (p.stat_partial && p.c2 > 0) return 1;   // statistics in the epilogue: single-source layers only
    // two workgroups per CU: 2 buffers of hrows rows each and the zero region within half of the LDS
    const bool f16 = f16_math();
    if (f16 && !p.range_flag) return DDPM_E_LAUNCH;
    const int xr = f16 ? XROW_F16 : XROW;
    const int max_rows = min(NPASS * kPassRows, (160 * 1024 / 2 - 1024 - 256 - 384 - 512) / (2 * xr));
    int tg = 0;
    for (int g = 8; g >= 2; g -= 2)
        if (tile_rows(g, p.in_w) <= max_rows) {
            tg = g;
            break;
        }
    if (tg == 0) return 1;
    p.tile_groups = tg;
    p.hrows = tile_rows(tg, p.in_w);
    const int lds = ((2 * p.hrows * xr + 255) & ~255) + 384;
    p.groups_total = (p.m_total + 31) / 32;
    p.kc16 = (p.ctot + XK - 1) / XK;
    p.nbp = (p.n_total + 31) / 32;
    const long long wb = 3072ll * 9 * p.nbp * p.kc16;
    if (wb >= 0x7fffffffll) return fail_invalid("weight tensor of 2 GiB or more");
    p.w_bytes = static_cast<unsigned>(wb);
    // one resident round of 512 workgroups, each covering 64 columns
    p.n_tiles = (p.n_total + BN - 1) / BN;
    int per_tile = 512 / p.n_tiles;
    if (per_tile < 1) per_tile = 1;
    p.gpb = (p.groups_total + per_tile - 1) / per_tile;
    const int64_t grid = static_cast<int64_t>((p.groups_total + p.gpb - 1) / p.gpb) * p.n_tiles;
    if (grid <= 0 || grid > 0x7fffffffll) return fail_invalid("conv grid size");
    p.stat_blocks = static_cast<int>(grid) / p.n_tiles;
    if (p.stat_partial) return f16 ? launch_one<false, true, true>(p, static_cast<int>(grid), lds, s) : launch_one<false, false, true>(p, static_cast<int>(grid), lds, s);
    if (f16) return p.c2 > 0 ? launch_one<true, true>(p, static_cast<int>(grid), lds, s) : launch_one<false, true>(p, static_cast<int>(grid), lds, s);
    return p.c2 > 0 ? launch_one<true, false>(p, static_cast<int>(grid), lds, s) : launch_one<false, false>(p, static_cast<int>(grid), lds, s);
}

}  // namespace conv
}  // namespace ddpm

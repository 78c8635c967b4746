/*
 * ddpm_mi355x.h -- C ABI of libddpm_mi355x.so: the DDPM hot path of vemunderoe/diffusion-model
 * as hand-written gfx950 (MI355X) kernels.
 *
 * The reference has no FFI; its boundary for this path is the Python module surface
 * (unet.UNet, diffusion.DiffusionModel, beta_scheduler.BetaScheduler,
 * embeddings.SinusoidalTimeEmbedding). The drop-in modules shipped next to this library keep
 * that surface and bind the entry points below with ctypes. Every entry point cites the
 * reference code it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *  - All device pointers are BORROWED (owned by the caller, e.g. torch tensors). The library
 *    never allocates, frees or synchronises; every kernel is enqueued on the `stream` argument
 *    (a hipStream_t passed as void*), so HIP-graph capture and multi-stream overlap work.
 *  - Activations are fp32 "NHWC": element (b, y, x, c) of a tensor with C channels lives at
 *    ((b*H + y)*W + x)*ld + c, where the pixel stride `ld` >= C lets a tensor be a channel
 *    slice of a wider buffer (the skip/upsample concatenation is never materialised twice).
 *    This is the memory image of a torch channels_last tensor of logical shape [B,C,H,W].
 *  - Return value: 0 on success, a negative DDPM_E_* code on failure; ddpm_last_error() gives
 *    a thread-local message. Nothing throws across the boundary. Shapes are validated on the
 *    host BEFORE any launch.
 *  - One device per call: the device that is current on the calling thread.
 */
#ifndef DDPM_MI355X_H
#define DDPM_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDPM_OK 0
#define DDPM_E_INVALID (-1)  /* bad argument / unsupported shape */
#define DDPM_E_LAUNCH (-2)   /* HIP reported an error at launch   */

#define DDPM_ACT_NONE 0
#define DDPM_ACT_RELU 1 /* first-iteration/unet.py:18,21 */
#define DDPM_ACT_GELU 2 /* exact erf GELU, second-iteration/unet.py:19,26,35 */

const char* ddpm_last_error(void);
int ddpm_abi_version(void);

/* ---------------------------------------------------------------------------------------
 * Host-side tables (no GPU work).
 * ------------------------------------------------------------------------------------- */

/* beta schedule, float64 math rounded once to fp32. kind: 0 linear, 1 quadratic, 2 cosine.
 * Replaces BetaScheduler.linear/quadratic/cosine, first-iteration/beta_scheduler.py:12-26.
 * betas: host array [timesteps]. */
int ddpm_schedule_betas(int kind, int timesteps, double beta_start, double beta_end, float* betas);

/* alphas = 1 - betas (fp32); alpha_bar = running product accumulated in float64 and rounded to
 * fp32 per element (what torch.cumprod does on CPU). Replaces DiffusionModel.__init__,
 * first-iteration/diffusion.py:9-10. Host arrays [timesteps]. */
int ddpm_schedule_tables(int timesteps, const float* betas, float* alphas, float* alpha_bar);

/* c[0] = 1/sqrt(alpha_t), c[1] = (1-alpha_t)/sqrt(1-alpha_bar_t), c[2] = sqrt(beta_t); every
 * operation a correctly rounded fp32 op. Replaces the 0-dim tensor arithmetic of
 * first-iteration/diffusion.py:26-28,36-37. */
int ddpm_step_coefficients(float beta_t, float alpha_t, float alpha_bar_t, float c[3]);

/* ---------------------------------------------------------------------------------------
 * Elementwise process kernels (HBM-bound).
 * ------------------------------------------------------------------------------------- */

/* x_out = c1*(x - c2*eps) + c3*z, five separately rounded fp32 ops (no FMA contraction), n
 * elements. z == NULL means the t == 0 step (no noise term; the reference adds sqrt(beta)*0).
 * eps_uncond != NULL applies classifier-free guidance first:
 * eps = lerp(eps_uncond, eps, cfg_w) with torch.lerp's two-branch formula.
 * Replaces first-iteration/diffusion.py:30-37 and third-iteration/diffusion.py:29-31,42-44.
 * x_out may alias x. 16 B per element algorithmic traffic (12 B when z == NULL). */
int ddpm_reverse_step(const float* x, const float* eps, const float* eps_uncond, float cfg_w,
                      const float* z, float c1, float c2, float c3, float* x_out, int64_t n,
                      void* stream);

/* z[i] = standard-normal element i of the counter-based stream (seed, call): Philox4x32-10 with
 * counter (i / 4, call) and key seed, two Box-Muller pairs per block. The same stream that
 * ddpm_final_conv_reverse_step draws in place with noise_mode 2 (bit for bit). Stands in for the
 * per-step torch.randn_like(x_t), first-iteration/diffusion.py:31 (the reference never seeds it). */
int ddpm_philox_normal(float* z, int64_t n, uint64_t seed, uint64_t call, void* stream);

/* The last layer of the UNet and the reverse step as ONE kernel, for the sampling loop: eps-hat =
 * final 1x1 conv of the NHWC features (c channels, stride ld; w [cout][c], bias [cout]) is formed per
 * pixel exactly as ddpm_conv1x1_final_fwd forms it (same bits) and consumed on the spot by the
 * reverse-step expression of ddpm_reverse_step (same five roundings): x_out = c1*(x_t - c2*eps) +
 * c3*z, all NCHW [batch][cout][h][w]. eps-hat makes no round trip through memory: 12 B per element
 * beside the feature read (x_t, z, x_out), 8 B with in-kernel noise.
 *   noise_mode 0: no noise term (the t == 0 step); 1: z is read from `z`; 2: z is drawn in place,
 *   element i of the stream (seed, call) of ddpm_philox_normal.
 *   eps_cond != NULL: this pass is the unconditional one of a classifier-free-guidance step and
 *   eps = lerp(eps_this, eps_cond, cfg_w) first (third-iteration/diffusion.py:29-31).
 *   eps_out != NULL additionally stores this pass's eps-hat (NCHW).
 * x_out may alias x_t. Replaces first-iteration/unet.py:115 + first-iteration/diffusion.py:30-37. */
int ddpm_final_conv_reverse_step(const float* feat, int ld, const float* w, const float* bias,
                                 const float* x_t, const float* eps_cond, float cfg_w, int noise_mode,
                                 const float* z, uint64_t seed, uint64_t call, float c1, float c2, float c3,
                                 float* x_out, float* eps_out, int batch, int h, int w_, int c, int cout,
                                 void* stream);

/* Hand-off of finished samples to the callers: out = clamp((x + 1) / 2, 0, 1), the two lines every
 * caller applies to sample()'s last stage (first-iteration/generate_samples.py:36-37,87-88), with
 * torch's roundings and NaN propagation. out may alias x. */
int ddpm_postprocess_images(const float* x, float* out, int64_t n, void* stream);

/* The image grid the callers save (make_grid(images, nrow=..., normalize=True),
 * first-iteration/generate_samples.py:71,93): images [n][c][h][w] -> grid [c == 1 ? 3 : c][gh][gw],
 * xmaps = min(nrow, n) images per row, ymaps = ceil(n / xmaps) rows, gh = (h + padding) * ymaps +
 * padding, gw likewise; image k sits at row k / xmaps, column k % xmaps behind `padding` pixels of
 * pad_value; single-channel images are repeated over three channels; normalize != 0 maps v to
 * (clamp(v, lo, hi) - lo) / max(hi - lo, 1e-5) (lo, hi = the tensor's min and max, the caller's).
 * torchvision's n == 1 shortcut (the image itself, no grid) is the caller's. */
int ddpm_make_grid(const float* images, int n, int c, int h, int w, int nrow, int padding, float pad_value,
                   int normalize, float lo, float hi, float* grid, void* stream);

/* out[b,:] = sqrt(abar[t[b]])*x0[b,:] + sqrt(1-abar[t[b]])*noise[b,:]; t: device int64 [batch],
 * alpha_bar: device fp32 [timesteps]; per_sample = C*H*W elements.
 * Replaces DiffusionModel.add_noise, first-iteration/diffusion.py:12-15. */
int ddpm_add_noise(const float* x0, const float* noise, const int64_t* t, const float* alpha_bar,
                   int timesteps, float* out, int batch, int64_t per_sample, void* stream);

/* loss = mean((pred-target)^2) over n elements -> loss_out[0] (device). partial: device scratch
 * of ddpm_mse_partial_count() doubles. Replaces nn.MSELoss() forward, first-iteration/main.py:48,79. */
int ddpm_mse_partial_count(void);
int ddpm_mse_fwd(const float* pred, const float* target, int64_t n, double* partial, float* loss_out,
                 void* stream);
/* grad[i] = gout[0] * 2*(pred[i]-target[i])/n   (gout: device scalar = d loss). */
int ddpm_mse_bwd(const float* pred, const float* target, const float* gout, int64_t n, float* grad,
                 void* stream);

/* Power-of-two scaling of the backward pass (loss.backward(), first-iteration/main.py:82-83: the reference has no
 * counterpart because its fp32 kernels need none). With the fp16-pair arithmetic the MFMA operands of the
 * data-gradient convolutions must sit inside fp16's exponent range and loss gradients (2*diff/numel) are far below
 * it, so the backward runs on s*g with s the power of two that brings max|g| into [1, 2) and multiplies every
 * parameter gradient by 1/s at the end; both multiplications are exact in fp32.
 * state: 3 device floats: [0] max|g|, [1] s, [2] 1/s. ddpm_grad_amax writes [0] (a data-parallel job all-reduces it
 * with MAX so that every rank uses the same s); ddpm_grad_scale writes out = s*g and [1], [2];
 * ddpm_scale_tensors multiplies ntensors tensors in place by the device scalar factor[0] (host arrays of device
 * pointers and element counts, consumed before the call returns). */
int ddpm_grad_amax(const float* g, int64_t n, float* state, void* stream);
int ddpm_grad_scale(const float* g, int64_t n, float* state, float* out, void* stream);
int ddpm_scale_tensors(int ntensors, float* const* tensors, const int64_t* numel, const float* factor, void* stream);

/* Arithmetic of the MFMA-fed convolutions and attention kernels: 0 fp32 MFMA, 1 three bf16 planes per operand and
 * six cross products, 2 fp16 pair and three cross products (the default). The process starts in the mode that
 * DDPM_CONV_MATH names in the environment ("fp32", "bf16x3", "f16x2"); ddpm_set_conv_math changes it for every later
 * call of the process (not thread-local). The packed-weight format follows the mode: weights must be used in the mode
 * they were packed in. No counterpart in the reference (torch picks its own kernels; on a CPU it computes in IEEE fp32,
 * first-iteration/unet.py:16-21 run without autocast). */
int ddpm_conv_math(void);
int ddpm_set_conv_math(int mode);

/* Range watch of the fp16-pair arithmetic. An fp16 pair represents magnitudes up to 65504; an operand beyond that is
 * clipped on its way into the matrix instructions (fp32, and mode 1, have no such limit). Every kernel that forms pairs
 * -- the convolutions, weight gradients, attention forward and the weight-packing kernels in mode 2 -- raises a flag in
 * device memory (one word per GPU, built into the library) when one of its operands was beyond that range, so that a
 * caller can repeat the work in mode 1 instead of keeping a clipped result. ddpm_range_flag_read enqueues on `stream`
 * a copy of the current GPU's flag into host_dst[0] (pinned host memory, or the copy is synchronous; may be NULL) and,
 * when clear != 0, resets the flag behind it; the value is valid once the stream has reached that point. The host
 * mirror does this at the stage hand-offs of sample() and once per training step. No counterpart in the reference. */
int ddpm_range_flag_read(uint32_t* host_dst, int clear, void* stream);

/* Adam update of ntensors tensors (host arrays of device pointers and element counts; the arrays are
 * consumed before the call returns): exp_avg = lerp(exp_avg, grad, 1-beta1); exp_avg_sq = beta2*exp_avg_sq
 * + (1-beta2)*grad^2; param -= lr/(1-beta1^step) * exp_avg / (sqrt(exp_avg_sq)/sqrt(1-beta2^step) + eps);
 * weight_decay > 0 adds weight_decay*param to grad first (L2, as torch.optim.Adam). step >= 1 is the
 * count AFTER this update. Replaces optimizer.step() of optim.Adam(model.parameters(), lr),
 * first-iteration/main.py:51,84; third-iteration/main.py:61. */
int ddpm_adam_step(int ntensors, float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const int64_t* numel, double lr, double beta1, double beta2,
                   double eps, double weight_decay, int64_t step, void* stream);

/* ---------------------------------------------------------------------------------------
 * Time / label embedding: sinusoidal embedding followed by the four time Linears (and, when
 * labels are given, the label Embedding + four label Linears added on top).
 * Replaces SinusoidalTimeEmbedding.forward (first-iteration/embeddings.py:11-20), the time_fc
 * calls of first-iteration/unet.py:78-82 and third-iteration/unet.py:135-145.
 * ------------------------------------------------------------------------------------- */
typedef struct ddpm_embed_args {
    int32_t batch;          /* B */
    int32_t emb_dim;        /* E (even, >= 4) */
    int32_t n_labels;       /* 0 = unconditional; 1 = one label broadcast over the batch; else B */
    int32_t t_is_float;     /* timesteps are fp32 instead of int64 */
    int32_t n_classes;      /* rows of label_table (required when n_labels > 0). A label outside
                               [0, n_classes) is never dereferenced: that sample's four output rows
                               are filled with NaN (nn.Embedding raises there; a kernel cannot) */
    int32_t uniform;        /* != 0: the caller guarantees that all B timesteps are equal (and n_labels <= 1), as in
                               every step of the sampling loop (first-iteration/diffusion.py:22-24): row 0 is computed
                               once and copied to the other rows (same bits as computing each row) */
    const void* t;          /* device [B] int64 (or fp32) */
    const int64_t* y;       /* device [n_labels] or NULL */
    const float* label_table; /* device [num_classes][E] or NULL */
    int32_t ch[4];          /* output widths C1..C4 */
    const float* w[4];      /* time Linear weights [Ck][E] */
    const float* b[4];      /* time Linear biases  [Ck]    */
    const float* lw[4];     /* label Linear weights (or NULL) */
    const float* lb[4];
    float* out[4];          /* device [B][Ck] */
    float* emb_out;         /* optional device [B][E]: the sinusoidal embedding itself (or NULL) */
} ddpm_embed_args;
int ddpm_time_embed(const ddpm_embed_args* a, void* stream);

/* The sinusoidal embedding alone: out[b] = [sin(t_b*f_k) | cos(t_b*f_k)], f_k = exp(-k*ln(1e4)/(E/2-1)),
 * written as [batch][2*(E/2)]. Replaces SinusoidalTimeEmbedding.forward, first-iteration/embeddings.py:11-20. */
int ddpm_sinusoidal_embedding(const void* t, int t_is_float, int batch, int emb_dim, float* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Convolutions as implicit GEMM on the matrix cores, LDS-staged tiles.
 * rows = pixels, columns = output channels, K = taps * input channels.
 * Arithmetic: each fp32 operand is split exactly into three bf16 terms and a product is the sum of
 * the six cross terms of weight >= 2^-16 on v_mfma_f32_32x32x16_bf16, accumulated in fp32 (error per
 * product below 2^-23, i.e. fp32 grade; see conv_gemm.hip).
 * ------------------------------------------------------------------------------------- */
typedef struct ddpm_conv_desc {
    int32_t batch, in_h, in_w;       /* input geometry */
    int32_t c1, ld1;                 /* channels read from src1 and its pixel stride (floats) */
    int32_t c2, ld2;                 /* channels read from src2 (0 = single source): the channel
                                        concatenation [src1 | src2] of UpBlock, never materialised */
    int32_t cout;                    /* output channels */
    int32_t ldo;                     /* pixel stride of the output */
    int32_t act;                     /* DDPM_ACT_* applied after scale/shift (+residual) */
    int32_t ldr;                     /* pixel stride of the residual tensor */
    /* transposed conv only: output geometry and placement of the un-padded 2H x 2W result in it
     * (negative offsets crop, positive offsets leave a zero border the caller has cleared). */
    int32_t out_h, out_w, off_y, off_x;
} ddpm_conv_desc;

/* Epilogue applied to every output element v (column n = output channel, pixel p, sample b):
 *   v = scale ? v*scale[n] + shift[n] : (shift ? v + shift[n] : v)   -- bias or folded BatchNorm
 *   v = residual ? v + residual[p*ldr + n] : v
 *   v = act(v)
 *   v = temb ? v + temb[b*cout + n] : v
 */
typedef struct ddpm_epilogue {
    const float* scale;
    const float* shift;
    const float* residual;
    const float* temb;
} ddpm_epilogue;

/* 3x3, padding 1, stride 1. w_packed from ddpm_pack_conv3x3 (cin = c1+c2).
 * Replaces nn.Conv2d(k=3,p=1) (+ eval BatchNorm2d + ReLU/GELU + residual + time-embedding add):
 * first-iteration/unet.py:16-21, second-iteration/unet.py:16-21,26,32-36.
 * Requires c1 % 4 == c2 % 4 == ld % 4 == 0. */
int ddpm_conv3x3_fwd(const ddpm_conv_desc* d, const float* src1, const float* src2,
                     const float* w_packed, const ddpm_epilogue* ep, float* out, void* stream);

/* ddpm_conv3x3_fwd that also leaves the partial sums of train-mode BatchNorm statistics over the values it writes
 * (what ddpm_bn_partials would compute from `out`): the 3x3 kernel's epilogue sums them while they are in registers
 * (16 values at a time in fp32, those sums in float64), which saves the statistics' own pass over the tensor; shapes
 * that kernel does not take (at most 32 output channels, two sources, very wide images) run the convolution and then
 * ddpm_bn_partials. Finish with ddpm_bn_stats_from_partials(partial, *nblocks, batch*in_h*in_w, cout, ...). Replaces
 * nn.Conv2d + the reduction of nn.BatchNorm2d in training, first-iteration/unet.py:16-17,
 * second-iteration/unet.py:16-17,24-25,30-33. */
int ddpm_conv3x3_fwd_stats(const ddpm_conv_desc* d, const float* src1, const float* src2, const float* w_packed,
                           const ddpm_epilogue* ep, float* out, double* partial, int* nblocks, void* stream);

/* 1x1. w_packed from ddpm_pack_conv1x1 (rows = cout, cin = c1+c2). Replaces the query/key/value projections
 * (second-iteration/unet.py:45-47,52-57) as ONE GEMM with cout = 2*(C/8) + C. */
int ddpm_conv1x1_fwd(const ddpm_conv_desc* d, const float* src1, const float* src2,
                     const float* w_packed, const ddpm_epilogue* ep, float* out, void* stream);

/* ConvTranspose2d(k=2, s=2) followed by F.pad/crop to (out_h, out_w): output pixel
 * (2y+a+off_y, 2x+b+off_x), dropped when outside. w_packed from ddpm_pack_convT2x2;
 * shift = bias[co] (length cout). Only input pixels whose outputs survive the
 * crop are computed. Replaces first-iteration/unet.py:41,45-48. */
int ddpm_convT2x2_fwd(const ddpm_conv_desc* d, const float* src, const float* w_packed,
                      const ddpm_epilogue* ep, float* out, void* stream);

/* Weight repacking (device -> device) from torch's parameter layouts into the library's packed
 * layout, an opaque buffer of ddpm_packed_weight_floats(taps, rows, cin) floats for a contraction of
 * `taps` taps, `rows` GEMM columns (output channels) and `cin` input channels:
 *   conv3x3: [cout][cin][3][3] -> taps 9, rows cout, depth cin; flip=1 packs the data-gradient conv
 *            instead (a 3x3 conv of dY): taps 9, rows cin, depth cout, W'(tap, ci, co) = w[co][ci][8-tap].
 *   convT:   [cin][cout][2][2] -> taps 1, rows 4*cout (row (a*2+b)*cout + co), depth cin.
 *   conv1x1: [rows][cin] (transpose=0) or its transpose [cin][rows] (transpose=1, the data-gradient
 *            GEMM of a 1x1 conv) -> taps 1.
 * Contents: every weight as an exact sum of three bf16 numbers (hi + mid + lo), laid out
 * [tap][cin/16][rows/32][plane][lane = 32*h + row%32][8 channels 8h..8h+7] -- one (tap, 16-channel
 * chunk, 32-row block, plane) is the 1 KiB MFMA operand fragment of a wave -- with rows zero-padded to
 * a multiple of 32 and channels to a multiple of 16; with DDPM_CONV_MATH=fp32 in the environment
 * (diagnostic: plain fp32 MFMA kernels) fp32 [tap][row][cin]. */
int64_t ddpm_packed_weight_floats(int taps, int rows, int cin);
int ddpm_pack_conv3x3(const float* w, int cout, int cin, int flip, float* out, void* stream);

/* The same for n weight tensors in one or a few launches (host arrays of n entries each): a training step repacks every
 * convolution weight after the optimizer step (first-iteration/main.py:84), both orientations. */
int ddpm_pack_conv3x3_batch(int n, const float* const* w, const int* cout, const int* cin, const int* flip, float* const* out,
                            void* stream);
int ddpm_pack_convT2x2(const float* w, int cin, int cout, float* out, void* stream);
int ddpm_pack_conv1x1(const float* w, int rows, int cin, int transpose, float* out, void* stream);

/* The attention block's three 1x1 projections as ONE GEMM (second-iteration/unet.py:45-47): packs
 * the fused [2*dqp + C][C] weight (query | key | value rows, query/key zero-padded from dq = C/8 to
 * dqp rows) for ddpm_conv1x1_fwd into w_packed (ddpm_packed_weight_floats(1, 2*dqp + C, C) floats),
 * its transpose for the data-gradient GEMM into wt_packed (ddpm_packed_weight_floats(1, C, 2*dqp + C)
 * floats) and the fused bias [2*dqp + C]. Weights in torch layout [rows][C] (the 1x1 kernel dims dropped). */
int ddpm_pack_qkv(const float* qw, const float* kw, const float* vw, const float* qb, const float* kb,
                  const float* vb, int c, int dq, int dqp, float* w_packed, float* wt_packed, float* bias,
                  void* stream);

/* First layer: 3x3 pad 1 on an NCHW input with few channels (1 or 3), direct (non-MFMA),
 * writes NHWC with the same epilogue. w: torch layout [cout][cin][3][3].
 * Replaces the first Conv2d of enc1, first-iteration/unet.py:61. */
int ddpm_conv3x3_first_fwd(const float* x_nchw, const float* w, const ddpm_epilogue* ep, int act,
                           float* out, int batch, int cin, int h, int w_, int cout, int ldo,
                           void* stream);

/* Last layer: 1x1 conv NHWC (c channels, stride ld) -> NCHW [B][cout][H][W] + bias.
 * w: [cout][c]. Replaces final_conv, first-iteration/unet.py:71,115. */
int ddpm_conv1x1_final_fwd(const float* x, int ld, const float* w, const float* bias, float* out_nchw,
                           int batch, int h, int w_, int c, int cout, void* stream);

/* ---------------------------------------------------------------------------------------
 * Pooling + skip hand-off: skip = x + temb (full resolution), pooled = maxpool2x2(x) WITHOUT the
 * embedding. Replaces DownBlock.pool and the `skip + time_emb` adds,
 * first-iteration/unet.py:31,35,89-94. temb may be NULL (plain copy), pooled may be NULL
 * (add only), argmax (uint8 per pooled element, 0..3 = dy*2+dx) may be NULL.
 * ------------------------------------------------------------------------------------- */
int ddpm_pool_skip_fwd(const float* x, int ldx, const float* temb, float* skip, int lds, float* pooled,
                       int ldp, uint8_t* argmax, int batch, int h, int w, int c, void* stream);

/* y = x + temb[b][c] (broadcast over pixels); y may alias x. first-iteration/unet.py:100,105,109,113. */
int ddpm_add_temb(const float* x, int ldx, const float* temb, float* y, int ldy, int batch, int hw,
                  int c, void* stream);

/* ---------------------------------------------------------------------------------------
 * Self-attention, single head, unscaled, d_qk = C/8, d_v = C, flash-style (the N x N matrix is
 * never materialised): out = gamma * softmax(Q K^T) V + x.
 * qkv: [B][N][2*dq + C] (q | k | v per pixel, from ddpm_conv1x1_fwd), x/out: [B][N][C].
 * gamma: device scalar. Optional outputs for the backward pass: attn_raw [B][N][C] (the attention
 * output before the gate) and lse [B][N] (log-sum-exp of every score row). temb (optional,
 * [B][C]) is added after the gate: the `x + time_emb` that follows every attention block outside
 * the encoder. Replaces SelfAttention.forward, second-iteration/unet.py:50-61 (+ :142,147,151,155).
 * ------------------------------------------------------------------------------------- */
int ddpm_attention_fwd(const float* qkv, const float* x, int ldx, const float* gamma, float* out,
                       int ldo, float* attn_raw, float* lse, const float* temb, int batch, int n, int c,
                       int dq, void* stream);

/* The whole attention block of an INFERENCE call in one entry: the fused q|k|v projection (weights and bias from
 * ddpm_pack_qkv) followed by the attention above, out = gamma * softmax(Q K^T) V + x (+ temb). Nothing is kept for a
 * backward pass, so the layout of the projection's output is the library's own affair: for the shapes whose attention
 * kernels take fp16 pairs (C = 64 or 128, N a multiple of 32, at least 16384 rows, default arithmetic) the projection's
 * epilogue writes k and v already split, in the order the attention kernels' matrix operands take them, instead of
 * every query block of every sample splitting them again on its way to LDS; the same pairs, products and order of
 * operations, so the same bits as ddpm_conv1x1_fwd + ddpm_attention_fwd, which it runs for every other shape.
 * x/out: [B][N][.] with strides ldx/ldo; workspace: B*N*(2*dq + C) floats, 16-byte aligned, contents unspecified
 * afterwards. Replaces SelfAttention.forward under torch.no_grad(), second-iteration/unet.py:45-61. */
int ddpm_self_attention_infer(const float* x, int ldx, const float* wqkv_packed, const float* bias_qkv,
                              const float* gamma, const float* temb, float* out, int ldo, float* workspace,
                              int batch, int n, int c, int dq, void* stream);

/* ---------------------------------------------------------------------------------------
 * Train-mode BatchNorm2d (+ activation / residual / time-embedding add), forward and backward.
 * Replaces nn.BatchNorm2d in training mode together with the ReLU/GELU, the residual add of
 * ResidualBlock and the `+ time_emb` that follow it: first-iteration/unet.py:17-21,
 * second-iteration/unet.py:17-26,33-36. Every reduction over (batch, pixels) accumulates in
 * float64 in a fixed order. `workspace`: ddpm_reduce_workspace_doubles(c) doubles of scratch.
 * ------------------------------------------------------------------------------------- */
int64_t ddpm_reduce_workspace_doubles(int c);

/* mean[c], invstd[c] = 1/sqrt(biased var + eps) over `rows` pixels; when running_mean/var are
 * given they are updated in place: r = (1-momentum)*r + momentum*stat (unbiased variance). */
int ddpm_bn_stats(const float* x, int ldx, int64_t rows, int c, float eps, float momentum,
                  double* workspace, float* mean, float* invstd, float* running_mean,
                  float* running_var, void* stream);

/* The two halves of ddpm_bn_stats as separate calls: ddpm_bn_partials leaves per-block partial sums
 * ([block][sum | sum of squares][c] doubles, *nblocks blocks, at most 1024: `partial` holds
 * ddpm_reduce_workspace_doubles(c) doubles) and ddpm_bn_stats_from_partials finishes them in float64 in block order
 * (mean, invstd, running statistics, as ddpm_bn_stats). ddpm_conv3x3_fwd_stats (below, with the convolutions) is the
 * producer they exist for: nn.Conv2d followed by nn.BatchNorm2d in train mode, second-iteration/unet.py:16-21. */
int ddpm_bn_partials(const float* x, int ldx, int64_t rows, int c, double* partial, int* nblocks, void* stream);
int ddpm_bn_stats_from_partials(const double* partial, int nblocks, int64_t rows, int c, float eps, float momentum,
                                float* mean, float* invstd, float* running_mean, float* running_var, void* stream);

/* ddpm_bn_stats without a workspace argument and without any allocation. No counterpart in the
 * reference: it exists for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc). The per-block partial sums go to
 * a buffer built into the library's device image (8 MiB per GPU, held until the process ends),
 * which limits c to ddpm_bn_stats_noalloc_max_channels() = 512; a larger c returns DDPM_E_INVALID.
 * Results are bit-identical to ddpm_bn_stats. All calls on one GPU share that buffer: calls that
 * overlap on the same GPU (different streams or threads) race; calls ordered on one stream do not.
 * The first call on a device looks the buffer's address up (hipGetSymbolAddress) and should be made
 * outside a stream capture. The buffer is the CURRENT device's: a `stream` of another device returns
 * DDPM_E_INVALID (make the stream's device current first). */
int ddpm_bn_stats_noalloc_max_channels(void);
int ddpm_bn_stats_noalloc(const float* x, int ldx, int64_t rows, int c, float eps, float momentum,
                          float* mean, float* invstd, float* running_mean, float* running_var,
                          void* stream);

/* out = act((x-mean)*invstd*w + b [+ residual]) [+ temb[b][c]] */
int ddpm_bn_apply_fwd(const float* x, int ldx, const float* mean, const float* invstd, const float* w,
                      const float* b, const float* residual, int ldr, const float* temb, int act,
                      float* out, int ldo, int batch, int hw, int c, void* stream);

/* Backward of the block above given dout = d loss / d out (the embedding add passes gradients
 * through unchanged). Produces dw, db (written, or added to when accumulate != 0), dx, and
 * optionally dresidual (= dz) and dconv_bias (= column sums of dx: the gradient of the bias of the
 * convolution feeding the BatchNorm). sums: scratch of 2*c floats. */
int ddpm_bn_bwd(const float* x, int ldx, const float* dout, int ldd, const float* residual, int ldr,
                const float* mean, const float* invstd, const float* w, const float* b, int act,
                int64_t rows, int c, double* workspace, float* sums, float* dw, float* db,
                int accumulate, float* dx, int lddx, float* dresidual, int lddr, float* dconv_bias,
                void* stream);

/* The same BatchNorm in two halves, for data-parallel training with GLOBAL batch statistics (the
 * reference's semantics at the global batch: one process normalises over all samples,
 * second-iteration/unet.py:17,21,33). Between the halves the caller all-reduces (sum) the 2*c doubles
 * across ranks -- with torch.distributed over RCCL; the library itself holds no communicator.
 *   forward : ddpm_bn_partial_sums -> sums = [sum x | sum x^2] of this rank's rows; all-reduce;
 *             ddpm_bn_finalize_stats(sums, total_rows = rows of all ranks) -> mean, invstd, running
 *             statistics; then ddpm_bn_apply_fwd as usual.
 *   backward: ddpm_bn_bwd_sums -> sums = [sum dz | sum dz*xhat] of this rank's rows (their float values
 *             are this rank's db | dw); all-reduce; ddpm_bn_bwd_apply with the reduced sums (as floats)
 *             and total_rows -> dx (+ dresidual, dconv_bias = column sums of this rank's dx).
 * With one rank the two halves compute exactly what ddpm_bn_stats / ddpm_bn_bwd compute. */
int ddpm_bn_partial_sums(const float* x, int ldx, int64_t rows, int c, double* workspace, double* sums,
                         void* stream);
/* This rank's raw float64 sums (sum x | sum x^2, 2*c doubles) from partial sums that a producer left behind
 * (ddpm_conv3x3_fwd_stats, ddpm_bn_partials): the second half of ddpm_bn_partial_sums, for data-parallel training
 * with global-batch statistics when the convolution's epilogue has already summed its output. Same layer as above:
 * second-iteration/unet.py:17 at the global batch. */
int ddpm_bn_sums_from_partials(const double* partial, int nblocks, int c, double* sums, void* stream);
int ddpm_bn_finalize_stats(const double* sums, int64_t total_rows, int c, float eps, float momentum,
                           float* mean, float* invstd, float* running_mean, float* running_var, void* stream);
int ddpm_bn_bwd_sums(const float* x, int ldx, const float* dout, int ldd, const float* residual, int ldr,
                     const float* mean, const float* invstd, const float* w, const float* b, int act,
                     int64_t rows, int c, double* workspace, double* sums, void* stream);
int ddpm_bn_bwd_apply(const float* x, int ldx, const float* dout, int ldd, const float* residual, int ldr,
                      const float* mean, const float* invstd, const float* w, const float* b,
                      const float* sums, int64_t total_rows, int act, int64_t rows, int c, double* workspace,
                      float* dx, int lddx, float* dresidual, int lddr, float* dconv_bias, void* stream);

/* out[g][c] (+)= sum of x over the rows of group g (groups of rows_per_group consecutive pixels):
 * groups = batch gives the gradient of a broadcast time-embedding add, groups = 1 a bias gradient. */
int ddpm_colsum(const float* x, int ld, int groups, int64_t rows_per_group, int c, double* workspace,
                int accumulate, float* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Backward pass of the convolutions (the gradient of nn.Conv2d / nn.ConvTranspose2d that
 * loss.backward() computes in first-iteration/main.py:83).
 *  - data gradient of a 3x3 conv = ddpm_conv3x3_fwd on dY with weights packed by
 *    ddpm_pack_conv3x3(flip = 1); of a 1x1 conv = ddpm_conv1x1_fwd with the transposed weight.
 *  - weight gradients are split-K MFMA GEMMs over the pixel axis (fp32 atomics into `packed`,
 *    scratch of the same size as the weight, then permuted into torch's parameter layout).
 * ------------------------------------------------------------------------------------- */
int ddpm_conv3x3_wgrad(const float* dy, int ldy, int cout, const float* x, int ldx, int cin, int batch,
                       int h, int w, float* packed, float* grad_w /* [cout][cin][3][3] */, void* stream);
int ddpm_conv1x1_wgrad(const float* dy, int ldy, int cout, const float* x, int ldx, int cin, int64_t rows,
                       float* grad_w /* [cout][cin] */, void* stream);

/* Transposed conv: dy is the gradient of the padded/cropped output image (out_h x out_w, stride
 * ldy); x the layer input (h x w). ddpm_convT2x2_dgrad: the in_h/in_w fields give the input geometry, the
 * out_h/out_w/off_y/off_x fields the output placement, c1/ld1 describe dy, cout = channels of the gradient produced; w_packed from
 * ddpm_pack_convT2x2_dgrad ([4][cin][cout]). */
int ddpm_pack_convT2x2_dgrad(const float* w, int cin, int cout, float* out, void* stream);
int ddpm_convT2x2_dgrad(const ddpm_conv_desc* d, const float* dy, const float* w_packed,
                        const ddpm_epilogue* ep, float* out, void* stream);
int ddpm_convT2x2_wgrad(const float* dy, int ldy, int cout, int out_h, int out_w, int off_y, int off_x,
                        const float* x, int ldx, int cin, int batch, int h, int w, float* packed,
                        float* grad_w /* [cin][cout][2][2] */, void* stream);

/* First conv (NCHW image in): weight gradient [cout][cin][3][3]; no data gradient is needed. */
int ddpm_conv3x3_first_wgrad(const float* dy, int ldy, const float* x_nchw, double* workspace, float* dw,
                             int batch, int cin, int h, int w_, int cout, void* stream);

/* Final 1x1 conv: dout is NCHW; produces dx (NHWC), dw [cout][c], db [cout]. */
int ddpm_conv1x1_final_bwd(const float* dout_nchw, const float* x, int ldx, const float* w,
                           double* workspace, float* dx, int lddx, float* dw, float* db, int batch, int h,
                           int w_, int c, int cout, void* stream);

/* dx = dskip + (dpooled routed to the arg-max position of each 2x2 window): backward of
 * ddpm_pool_skip_fwd (MaxPool2d + the skip branch), first-iteration/unet.py:33-36. */
int ddpm_pool_skip_bwd(const float* dskip, int ldds, const float* dpooled, int ldp, const uint8_t* argmax,
                       float* dx, int lddx, int batch, int h, int w, int c, void* stream);

/* out[c] = sum of x over every sample and the pixel window [y0,y0+win_h) x [x0,x0+win_w): bias
 * gradient of a transposed conv whose output was padded. */
int ddpm_colsum_window(const float* x, int ld, int batch, int img_h, int img_w, int y0, int x0, int win_h,
                       int win_w, int c, double* workspace, float* out, void* stream);

/* Backward of ddpm_time_embed w.r.t. the Linear / Embedding parameters. */
typedef struct ddpm_embed_bwd_args {
    int32_t batch, emb_dim, n_labels;
    int32_t n_classes;         /* rows of label_table / dlabel_table; samples whose label is outside
                                  [0, n_classes) contribute nothing to the label-path gradients */
    const float* emb;          /* [B][E] saved sinusoidal embedding */
    const int64_t* y;
    const float* label_table;
    int32_t ch[4];
    const float* dtemb[4];     /* [B][Ck] gradients of the four embedding vectors */
    const float* lw[4];        /* label Linear weights (forward values) or NULL */
    float* dw[4];              /* out: time Linear weight grads [Ck][E] */
    float* db[4];              /* out: time Linear bias grads */
    float* dlw[4];             /* out: label Linear grads (or NULL) */
    float* dlb[4];
    float* dlabel_table;       /* out: [classes][E], zeroed by the caller (rows are added to) */
} ddpm_embed_bwd_args;
int ddpm_time_embed_bwd(const ddpm_embed_bwd_args* a, void* stream);

/* Backward of ddpm_attention_fwd given dy = d loss / d out. Produces dqkv [B][N][2*dq+C] (the
 * gradient of the fused q|k|v projection output) and dgamma; the pass-through term dx += dy is
 * the caller's. attn_raw/lse: saved by the forward; e_scratch: [B][N] floats;
 * workspace: >= 1024 doubles. Replaces autograd through second-iteration/unet.py:50-61. */
int ddpm_attention_bwd(const float* qkv, const float* dy, int ldd, const float* attn_raw, const float* lse,
                       const float* gamma, double* workspace, float* e_scratch, float* dqkv, float* dgamma,
                       int batch, int n, int c, int dq, void* stream);

/* ---------------------------------------------------------------------------------------
 * Scoring of generated images with the small MNIST classifier (the "+classifier.py path" of the
 * sampling benchmark). The classifier itself -- conv3x3(1->32)+ReLU+pool, conv3x3(32->64)+ReLU+pool,
 * fc 4096->128 + ReLU, fc 128->10 (third-iteration/classifier.py:9-24) -- runs on
 * ddpm_conv3x3_first_fwd, ddpm_pool_skip_fwd, ddpm_conv3x3_fwd and ddpm_conv1x1_fwd (a Linear is a
 * 1x1 convolution over a one-pixel image); the entry points below are the reductions over its
 * [n][k] logits that the metric scripts perform (k <= 16 classes).
 * ------------------------------------------------------------------------------------- */

/* probs = softmax(logits, dim=1), fp32. Replaces F.softmax(classifier(batch), dim=1),
 * third-iteration/metrics2.py:77. */
int ddpm_softmax_rows(const float* logits, int64_t n, int k, float* probs, void* stream);

/* Inception-Score terms: for split s of n // splits consecutive rows, py = mean of the rows,
 * scores[s] = exp(mean_i sum_j p_ij * (log p_ij - log py_j)); rows beyond splits * (n // splits) are
 * ignored as in the reference. Replaces the loop of third-iteration/metrics2.py:83-87; the mean and
 * standard deviation over the splits (:89) are the caller's. */
int ddpm_inception_split_scores(const float* probs, int64_t n, int k, int splits, float* scores, void* stream);

/* mean[k] and unbiased covariance cov[k][k] (float64, as numpy.cov) of the rows of x [n][k]: the
 * statistics the Frechet distance is computed from. Replaces np.mean(act, axis=0), np.cov(act,
 * rowvar=False), third-iteration/metrics2.py:107-108 (the 10x10 matrix square root of :112 stays on
 * the host). workspace: ddpm_mean_cov_workspace_doubles(k) doubles. */
int64_t ddpm_mean_cov_workspace_doubles(int k);
int ddpm_mean_cov(const float* x, int64_t n, int k, double* workspace, double* mean, double* cov, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DDPM_MI355X_H */

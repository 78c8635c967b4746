// One-workgroup-per-sample attention for feature maps of at most 64 pixels (attention_small.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace ddpm {

bool attention_small_fwd_fits(int n, int c, int dq);
bool attention_small_bwd_fits(int n, int c, int dq);

// operands as in ddpm_attention_fwd; dq is the padded q/k width
int attention_small_fwd(const float* qkv, const float* x, int ldx, const float* gamma, const float* temb, float* out, int ldo,
                        float* attn_raw, float* lse, int batch, int n, int c, int dq, hipStream_t s);

// operands as in ddpm_attention_bwd after its row-sum pass: e[b][i] = dY[i] . O[i]
int attention_small_bwd(const float* qkv, const float* dy, int ldd, const float* lse, const float* e, const float* gamma, float* dqkv,
                        int batch, int n, int c, int dq, hipStream_t s);

// Attention forward of an inference call whose k and v the projection has already split into fp16 pairs (attention.hip;
// layouts at AttnParams::vpl): qk = [rows][2 dq] (q as fp32, k as (hi | lo << 16) words), vpl = the v planes. The same
// kernels, products and order of operations as ddpm_attention_fwd runs for these shapes: the same bits.
bool attention_fwd_pairs_fits(int batch, int n, int c, int dq);
int attention_fwd_pairs(const float* qk, const unsigned char* vpl, const float* x, int ldx, const float* gamma, const float* temb,
                        float* out, int ldo, int batch, int n, int c, int dq, hipStream_t s);

}  // namespace ddpm

// Token-sequence building blocks shared by the transformer-shaped stages (Whisper encoder, MuseTalk UNet
// attention): LayerNorm, row softmax, device-side packing of a GEMM "B" operand, dynamic GEMM plans.
//
// A token sequence of length T with C channels is an ActBuf{C, H=1, W=T, halo=1}: token t is pixel
// (0, t), so every linear layer is a 1x1 convolution on the implicit-GEMM kernel and shares its
// epilogues (bias, GELU, residual).
#pragma once
#include "mf_conv.h"

// y = (x - mean) / sqrt(var + eps) * gamma + beta over the C channels of every token (fp32 math)
int mf_layernorm(const ActView& x, const ActView& y, const float* gamma, const float* beta, float eps, int batch,
                 hipStream_t s);

// p[t][j] = softmax_j(scale * s[t][j]) for j < n_keys; columns n_keys..p.C-1 are written as zero
int mf_softmax_rows(const ActView& scores, const ActView& probs, int n_keys, float scale, int batch, hipStream_t s);

// Packs B[n][k] = src[n*stride_n + k*stride_k] (bf16 hi/lo planes) into the implicit-GEMM weight layout
// [K/64][Npad][64] of `plan` (zero padded), so activations can be the "weight" operand of a GEMM.
int mf_pack_b(ConvPlan* plan, const bf16_t* src_hi, const bf16_t* src_lo, int64_t stride_n, int64_t stride_k, int N,
              int K, hipStream_t s);

// A ConvPlan shell for out[t][n] = sum_k in[t][k] * B[n][k] whose B is filled on the device by mf_pack_b.
int mf_gemm_plan_create(ConvPlan* p, int K, int N, int T, int precision);

// (hi + lo) planes of the interior of a view -> fp32 [batch][T][C] row-major
int mf_rows_to_f32(const ActView& x, float* dst, int batch, hipStream_t s);

inline int64_t mf_interior(const ActBuf& b) { return ((int64_t)b.halo * b.Wp() + b.halo) * b.C; }

// Token-sequence building blocks shared by the transformer-shaped stages (Whisper encoder, MuseTalk UNet
// attention): LayerNorm, row softmax, device-side packing of a GEMM "B" operand, dynamic GEMM plans.
//
// A token sequence of length T with C channels is an ActBuf{C, H=1, W=T, halo=1}: token t is pixel
// (0, t), so every linear layer is a 1x1 convolution on the implicit-GEMM kernel and shares its
// epilogues (bias, GELU, residual).
#pragma once
#include "mf_conv.h"

// y = (x - mean) / sqrt(var + eps) * gamma + beta over the C channels of every token (fp32 math)
// tokens > 0: only the first `tokens` tokens of every batch item (a sequence prefix; the buffers keep their geometry)
// act = 3: GELU (erf form) applied to the normalised value (wav2vec2's conv -> LayerNorm -> GELU feature layers)
int mf_layernorm(const ActView& x, const ActView& y, const float* gamma, const float* beta, float eps, int batch,
                 hipStream_t s, int tokens = 0, int act = 0);

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
// Whisper's `encoder_embeddings` gather (audio2feature.py:103-110): the first `tokens` tokens of every batch item into
// dst[b][t][layer][c] of an fp32 [batch][tokens][n_layers][C] tensor
int mf_rows_to_f32_layered(const ActView& x, float* dst, int batch, int tokens, int layer, int n_layers, hipStream_t s);

// fp32 [batch][T][C] row-major (+ optional addend [T][C], e.g. a positional encoding) -> planes of a view
int mf_rows_from_f32(const float* src, const float* addend, const ActView& y, int batch, hipStream_t s);

// GroupNorm over (H*W x C/groups) per (batch, group), optional SiLU, on any view (fp64 sums, fp32 apply): two launches.
// `stats` is a device scratch of batch*groups*2 doubles owned by the caller and must be ZERO on entry (mf_zero_f64;
// a kernel, not hipMemsetAsync -- memset nodes corrupted captured graphs on ROCm 7.2).
int mf_zero_f64(double* p, int n, hipStream_t s);
// have_stats: `stats` already holds the sums (the producing conv's epilogue added them, ConvPlan::out_stats): no statistics pass
int mf_groupnorm_affine(const ActView& x, const float* gamma, const float* beta, int groups, float eps, double* stats, float* scale, float* shift,
                        int batch, hipStream_t s, bool have_stats = false);
int mf_groupnorm(const ActView& x, const ActView& y, const float* gamma, const float* beta, int groups, float eps,
                 bool silu, double* stats, int batch, hipStream_t s, bool have_stats = false);

// GroupNorm [+ SiLU] -> Conv2d 3x3 s1 p1 with <= 16 output channels as ONE pass over x (mf_conv_tail.hip): the `conv_norm_out` -> `conv_act` -> `conv_out`
// tail of the VAE decoder and of the UNet.  bf16x3 only; cin a multiple of 32.  `stats` as for mf_groupnorm.
struct TailConv { int cin = 0, cout = 0; bf16_t* w = nullptr; float* bias = nullptr; };
bool mf_tail_conv_supported(int cin, int cout, int precision);
int mf_tail_conv_create(TailConv* p, const float* weight, const float* bias, int cin, int cout);
void mf_tail_conv_destroy(TailConv* p);
int mf_gn_conv3_tail(const TailConv& p, const ActView& x, const float* gamma, const float* beta, int groups, float eps, bool silu, double* stats,
                     bool have_stats, const ActView& out, int batch, hipStream_t s);

// GEGLU (diffusers): y[t][c] = x[t][c] * gelu(x[t][C + c]) for c < C = x.C / 2
int mf_geglu(const ActView& x, const ActView& y, int batch, hipStream_t s);

// Packs `groups` = batch*heads B operands at once (see mf_pack_b): group z = (b, h) reads
// src[b*sb + h*sh + n*stride_n + k*stride_k] into plan weights + z * (KT*Npad*64).
int mf_pack_b_grouped(ConvPlan* plan, const bf16_t* src_hi, const bf16_t* src_lo, int64_t sb, int64_t sh, int64_t stride_n,
                      int64_t stride_k, int N, int K, int groups, int heads, hipStream_t s);
// a mf_gemm_plan_create shell with room for `groups` packed operands
int mf_gemm_plan_create_grouped(ConvPlan* p, int K, int N, int T, int groups, int precision);

// image = (x / 2 + 0.5).clamp(0, 1); (image * 255).round() -> uint8 [B][H][W][3] with the channel order reversed
// (RGB -> BGR): the tail of VAE.decode_latents, musetalk/models/vae.py:104-107
int mf_vae_post_u8(const ActView& x, uint8_t* dst, int batch, hipStream_t s);

// Fused softmax(q k^T / sqrt(dh)) v (mf_attn.hip) on contiguous (halo 0) token buffers; head dims 40 / 64 / 80 / 160.
bool mf_attention_supported(int dh);
// tq / tk > 0: only the first tq queries attend to only the first tk keys of every batch item (sequence prefixes)
int mf_attention(const ActView& q, const ActView& k, const ActView& v, const ActView& out, int heads, int batch, int precision,
                 hipStream_t s, int tq = 0, int tk = 0);

inline int64_t mf_interior(const ActBuf& b) { return ((int64_t)b.halo * b.Wp() + b.halo) * b.C; }

// Layout-conversion and head kernels around the MFMA convolutions (all HBM-bound, one pass).
#pragma once
#include "mf_conv.h"

// fp32 NCHW [B,C,H,W] -> interior of the padded NHWC (hi, lo) planes of `dst` (channels >= C
// of the destination view are written as zero up to dst.buf->C when C is not a multiple of 8).
int mf_nchw_to_act(const float* src, int C, const ActBuf& dst, int batch, hipStream_t s);

// interior of a channel slice of padded NHWC planes -> fp32 NCHW [B,C,H,W]
int mf_act_to_nchw(const ActView& src, float* dst, int batch, hipStream_t s);

// lipreal.py:115-122 fused: uint8 [B,96,96,3] BGR crops -> 8-channel padded NHWC
// [masked b,g,r (rows >= H/2 zero), full b,g,r, 0, 0] / 255
int mf_faces_u8_to_act(const uint8_t* faces, const ActBuf& dst, int batch, hipStream_t s);

// output_block.1 (plain 1x1 Conv2d 32->3) + Sigmoid (wav2lip.py:84-85) on the NHWC activation
// of output_block.0.  hwc255 == 0: fp32 NCHW [B,3,H,W] in [0,1];  hwc255 == 1: fp32 [B,H,W,3]*255
// (the `pred.cpu().numpy().transpose(0,2,3,1) * 255.` of lipreal.py:126).
// w: device fp32 [3][cin], b: device fp32 [3].
int mf_head_1x1_sigmoid(const ActView& src, const float* w, const float* b, float* dst, int hwc255,
                        int batch, hipStream_t s);

// VAE.preprocess_img (musetalk/models/vae.py:52-82) for an in-memory crop: uint8 [B,H,W,3] BGR -> RGB, / 255., optional half mask (rows >= H/2
// zeroed BEFORE the normalisation, vae.py:75-76), Normalize(mean .5, std .5) -> the padded NHWC planes of `dst` (channels 3.. zero).
int mf_vae_image_u8_to_act(const uint8_t* img, const ActBuf& dst, int half_mask, int batch, hipStream_t s);

// MF_PREC_F16Q input planes from fp32 NCHW (test seam): hi plane = f16(x); lo plane = per pixel and 32-channel block 64 bytes
// [q6(x - f16(x)) : 24 B of e2m3 codes, E8M0 scale byte, 7 B pad | q6(f16(x)) likewise] (OCP-MX: the block maximum scaled into [4, 8))
int mf_nchw_to_act_q(const float* src, int C, const ActBuf& dst, int batch, hipStream_t s);

// The producer side of MF_PREC_F16Q inside a network: v = x * scale[b][c] + shift[b][c] (a finalised GroupNorm, mf_groupnorm_affine), optional SiLU, written
// straight into the f16 + FP6-block planes of `dst` (same geometry as x) -- the GroupNorm-apply pass of a resnet whose convolution reads the new format.
// post: per-channel multiplier applied BEHIND the activation (device, x.C floats, powers of two), or null -- the channel equalisation of the MX blocks (mf_musetalk.hip gn_conv)
// gn_stats != null (maps of a multiple of 64 pixels): scale / shift are not read -- the kernel forms the affine itself from the GroupNorm's (sum, sum of squares) per
// (sample, group), gamma and beta (the values mf_groupnorm_affine would have written, bit for bit): no k_gn_affine launch in front
int mf_affine_silu_to_act_q(const ActView& x, const float* scale, const float* shift, int silu, const ActBuf& dst, int batch, hipStream_t s, const float* post = nullptr,
                            const double* gn_stats = nullptr, const float* gn_gamma = nullptr, const float* gn_beta = nullptr, int gn_groups = 0, float gn_eps = 0.f);

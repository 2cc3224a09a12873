// Implicit-GEMM convolution on MFMA for gfx950: plan/launch interface.
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   activations : padded NHWC  [B][H+2*halo][W+2*halo][C], bf16; the halo ring is zero and is
//                 never written, so the kernel's gather needs no bounds predicates.  In
//                 MF_PREC_BF16X3 every tensor is a (hi, lo) pair of such planes.
//   weights     : BatchNorm-folded, packed per phase as [K/32][Npad][32] bf16 (hi, lo), where the
//                 K axis enumerates 8-channel groups tap-major: g = tap*(Cin/8) + c/8.
//   goff        : int32 per K group = element offset of that (tap, channel-group) relative to the
//                 output pixel's input anchor; staged in LDS by every workgroup.
#pragma once
#include "mf_common.h"
#include <vector>

struct ActBuf {
    int C = 0, H = 0, W = 0, halo = 0;
    bf16_t* hi = nullptr;
    bf16_t* lo = nullptr;
    int Hp() const { return H + 2 * halo; }
    int Wp() const { return W + 2 * halo; }
    int64_t per_batch() const { return (int64_t)Hp() * Wp() * C; }
};

// channel slice [coff, coff+C) of a buffer (concat-free skip connections, wav2lip.py:108)
struct ActView {
    const ActBuf* buf = nullptr;
    int coff = 0;
    int C = 0;
};

#define MF_MAX_PHASE 9

struct ConvPhase {
    int goff_begin;   // first entry of this phase in the goff table
    int ngroups;      // K groups incl. padding = KT*4
    int KT;           // number of 32-deep K tiles
    int64_t w_off;    // element offset of this phase's packed weights
    int64_t y_off;    // element offset of this phase's first output pixel (relative to y base)
};

struct ConvArgs {
    const bf16_t* x_hi; const bf16_t* x_lo;
    const bf16_t* w_hi; const bf16_t* w_lo;
    const float* bias;
    const bf16_t* r_hi; const bf16_t* r_lo;
    bf16_t* y_hi; bf16_t* y_lo;
    const int* goff;
    int M, N, Npad;
    int HqWq, Wq;
    int64_t xb; int xi, xj;   // input element strides per (batch, quotient row, quotient col)
    int64_t yb; int yi, yj;   // output strides
    int64_t rb; int ri, rj;   // residual strides
    int act;                  // 0 none, 1 relu, 2 sigmoid
    int tiles_m, tiles_n;
    int goff_total;
    ConvPhase ph[MF_MAX_PHASE];
};

struct ConvPlan {
    mf_conv2d_desc d{};
    int precision = 0;
    int cin_pad = 0;      // cin rounded up to 8
    int out_h = 0, out_w = 0;
    int Hq = 0, Wq = 0;   // quotient grid (== output grid for Conv2d, input grid for stride-2 ConvT)
    int nphase = 1;
    int Npad = 0;
    // device
    bf16_t* w_hi = nullptr;
    bf16_t* w_lo = nullptr;
    float* bias = nullptr;
    int* goff = nullptr;
    // host-side phase description, independent of the buffers the layer is later bound to
    struct Tap { int dy, dx; };               // input displacement in pixels relative to anchor
    std::vector<std::vector<Tap>> phase_taps; // per phase
    std::vector<int> phase_oy, phase_ox;      // output pixel offset of the phase
    int out_step = 1;                         // output pixel stride of the quotient grid
    int in_step_h = 1, in_step_w = 1;         // input pixel stride of the quotient grid
    int in_halo_need = 0;
    ConvPhase ph[MF_MAX_PHASE]{};
    int goff_total = 0;
    // binding-dependent (built by mf_conv_bind)
    int bound_in_ld = -1, bound_in_wp = -1;
};

// Folds BN, packs weights, uploads.  Returns mf_status.
int mf_conv_plan_create(ConvPlan* p, const mf_conv2d_desc& d, const float* weight, const float* bias,
                        const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                        const float* bn_var, int precision);
void mf_conv_plan_destroy(ConvPlan* p);

// Builds the goff table for the input buffer geometry the plan will read (pixel stride = buffer C,
// row stride = Wp*C).  Must be called once before launch; rebinding to another geometry is allowed.
int mf_conv_bind(ConvPlan* p, const ActBuf& in);

// Enqueues the layer.  res may have buf == nullptr.
int mf_conv_launch(const ConvPlan* p, const ActView& in, const ActView& out, const ActView& res,
                   int batch, hipStream_t stream);

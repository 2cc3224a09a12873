// Implicit-GEMM convolution on MFMA for gfx950: plan/launch interface.
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   activations : padded NHWC  [B][H+2*halo][W+2*halo][C], bf16; the halo ring is zero and is
//                 never written, so the kernel's gather needs no bounds predicates.  In
//                 MF_PREC_BF16X3 every tensor is a (hi, lo) pair of such planes.
//   weights     : BatchNorm-folded, packed per phase as [K/BK][Npad][BK] bf16 (hi, lo), where the
//                 K axis enumerates 8-channel groups tap-major: g = tap*(Cin/8) + c/8.
//   goff        : int32 per K group = element offset of that (tap, channel-group) relative to the
//                 output pixel's input anchor; staged in LDS by every workgroup.
#pragma once
#include "mf_common.h"
#include <map>
#include <vector>
#include <utility>

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
    int ngroups;      // K groups incl. padding = KT*(BK/8)
    int KT;           // number of BK-deep K tiles
    int64_t w_off;    // element offset of this phase's packed weights
    int64_t y_off;    // element offset of this phase's first output pixel (relative to y base)
    int64_t ws_off;   // same, in the unpadded fp32 split-K workspace
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
    // m / HqWq and rem / Wq by multiplication (mf_fastdiv below): the kernels split a pixel index into (image, row, column) per fragment row in the prologue and again
    // in the epilogue -- as hardware-less 32-bit divisions that was ~30 vector instructions each, 16 - 24 of them per lane and workgroup
    uint32_t dv_hw_mul, dv_hw_shr, dv_w_mul, dv_w_shr;
    uint32_t dv_t_mul, dv_t_shr, dv_s_mul, dv_s_shr;     // tile index / (tiles_m or tiles_n, whichever runs fastest) and K tiles / split count: set by the launcher
    int64_t xb; int xi, xj;   // input element strides per (batch, quotient row, quotient col)
    int64_t yb; int yi, yj;   // output strides
    int64_t rb; int ri, rj;   // residual strides
    int act;                  // 0 none, 1 relu, 2 sigmoid, 3 gelu(erf), 4 silu
    int res_after_act;        // residual added after the activation (x = act(conv) + r)
    int tiles_m, tiles_n;
    unsigned long long* dbg;  // MF_DEBUG=times: 4 s_memtime stamps per workgroup, or null
    int m_fastest;            // XCD tile order: pixel tiles fastest (weight-heavy layers), see k_conv_igemm
    int ld;                   // operand path of the 4-wave tiles (k_conv_igemm's LD): -1 = the library default, 0 / 1 / 2 = measured choice (mf_conv_tune)
    int wide_store;           // output view starts on an 8-channel group and N % 8 == 0: 16-byte epilogue stores (lane pairs exchange halves)
    int goff_total;
    // grouped launch (attention: one GEMM per (batch, head) on blockIdx.z): element offsets per group
    int zgroups, zheads;
    int64_t zx_b, zx_h, zw, zy_b, zy_h;
    float* ws;                // split-K fp32 partials [split][B][Ho][Wo][N], or null
    int64_t ws_split, wsb; int wsi, wsj;
    // GroupNorm statistics of the OUTPUT for the layer's consumer: (sum, sum of squares) per (sample, group) added to gn_out[2 * (b * groups + g)]
    // by the epilogue (4-wave tiles whose pixel tile lies in one sample) or by the split-K combine; null = off
    double* gn_out; int gn_out_cpg, gn_out_groups;
    // LayerNorm folded into the GEMMs either side of it (round 5; 1 x 1 layers over token sequences).  PRODUCER (ln_out != null): the epilogue adds every
    // output row's (sum, sum of squares) over this tile's channels to ln_out[2 * m], one fp64 atomic per (wave, row, moment).  CONSUMER (ln_in != null): the
    // layer multiplies the RAW tensor by gamma-scaled weights and its epilogue finishes the normalisation,
    //     y[m][n] = rstd_m * (acc[m][n] - mean_m * ln_cs[n]) + bias'[n],   ln_cs[n] = sum_k (gamma_k W[n][k]) as the kernel multiplies it,  bias' = bias + W beta
    // -- the k_layernorm pass between the two (read + write of the tensor, one launch per LayerNorm: 48 per UNet step) disappears.
    double* ln_out; const double* ln_in; const float* ln_cs; float ln_inv_c, ln_eps;
    ConvPhase ph[MF_MAX_PHASE];
};

// ---- 3x3 stride-1 halo-tile kernel (mf_conv_halo.hip) ---------------------------------------------
struct HaloArgs {
    const bf16_t* x_hi; const bf16_t* x_lo;     // input view base (channel offset applied)
    const bf16_t* w_hi; const bf16_t* w_lo;     // packed [slice][tap][Npad][CK]
    const float* bias;
    const bf16_t* r_hi; const bf16_t* r_lo;
    bf16_t* y_hi; bf16_t* y_lo;                 // interior origin of the output view
    int batch, H, W, N, Npad, n_slices;
    int in_halo, in_hp, in_wp, x_ld;            // input buffer geometry (padded rows/cols, pixel stride)
    int64_t xb;
    int64_t yb; int yi, yj;
    int64_t rb; int ri, rj;
    int act;
    int res_from_halo;                          // residual == the layer input: taken from the LDS halo image
    int patches_x, patches_per_img, n_patches, tiles_n;   // filled by mf_halo_launch
    // LDS-weights kernel only: channel slices split over blockIdx.y, fp32 partial tiles [split][B][H][W][N] combined by k_splitk_epilogue
    float* ws; int64_t ws_split; int nsplit;
    unsigned long long* dbg;                    // MF_DEBUG=times: 4 s_memtime stamps per workgroup (entry, loop start, loop end, exit), or null
    int q;                                      // operands in the f16 + FP6-residual format (MF_PREC_F16Q): x_lo / w_lo hold [q6 block | q6 block] rows
    double* gn_out; int gn_out_cpg, gn_out_groups;   // f16 + FP6 kernel: (sum, sum of squares) of the OUTPUT per (sample, group) added here for the consumer GroupNorm; null = off
    int wide_store;                             // f16 + FP6 tiles: the output view starts on an 8-channel group, C % 8 == 0, N % 32 == 0: 16-byte epilogue stores
};
struct HaloTile { int ph, bn, wgm, wgn; };

// thin-input convolution (mf_conv_thin.hip): cin <= 16, cout <= 32, k7 s1 / k3 s1 / k3 s2 -- the input patch in LDS once, every tap read out of it
struct ThinArgs {
    const bf16_t* x_hi; const bf16_t* x_lo;     // input view base (channel offset applied)
    const bf16_t* w;                            // lane-ordered A fragments [fragment][step][plane][64][8] (mf_thin_pack)
    const float* bias;
    bf16_t* y_hi; bf16_t* y_lo;                 // interior origin of the output view
    int batch, H, W, N;                         // OUTPUT map, output channels
    int pad, in_halo, in_hp, in_wp, x_ld;       // input buffer geometry
    int64_t xb, yb; int yi, yj;
    int act;
};
bool mf_thin_supported(int k, int stride, int cin, int cout);
int mf_thin_nks(int k, int cin);
void mf_thin_pack(const float* w, const float* scale, int cout, int cin, int k, bool x3, std::vector<bf16_t>& dst);
int mf_thin_launch(const ThinArgs& a, int k, int stride, int cin, int cout, bool x3, hipStream_t s);
HaloTile mf_halo_pick_tile(int H, int W, int N, int batch, int cin = 0);
int mf_halo_launch(const HaloArgs& a, const HaloTile& t, bool x3, hipStream_t s);
// second generation (mf_conv_halo2.hip): weights shared through an LDS ring; pick_tile returns ph == 0 to decline
HaloTile mf_halo_w_pick_tile(int H, int W, int N, int batch, int cin = 0);
int mf_halo_w_launch(const HaloArgs& a, const HaloTile& t, bool x3, hipStream_t s, int phase = -1);
struct ConvTile { int bm, bn, wgm, wgn, nsplit; };
struct ConvTuned { ConvTile tile; int ld; };   // a measured launch configuration (mf_conv_tune)

struct ConvPlan {
    mf_conv2d_desc d{};
    int precision = 0;
    int cin_pad = 0;      // cin rounded up to 8
    int out_h = 0, out_w = 0;
    int Hq = 0, Wq = 0;   // quotient grid (== output grid for Conv2d, input grid for stride-2 ConvT)
    int nphase = 1;
    int Npad = 0;
    int BK = 64;          // contraction depth of one LDS tile: 64 (bf16) or 32 (bf16x3)
    // device
    bf16_t* w_hi = nullptr;
    bf16_t* w_lo = nullptr;
    float* bias = nullptr;
    int* goff = nullptr;
    double* out_stats = nullptr; int out_stats_groups = 0;   // GroupNorm (sum, sum of squares) of the output, left in out_stats by every launch (set by the network builder): from the
                                                             // epilogue where the kernel can (f16 + FP6 tiles, 4-wave implicit-GEMM tiles, split-K combine), else by a k_gn_stats pass behind the conv
    // LayerNorm folding (ConvArgs::ln_*).  Consumer: set ln_gamma / ln_beta (host, cin floats, read by mf_conv_plan_create only) BEFORE the plan is created -- the
    // weights are scaled by gamma, the bias takes W beta, ln_cs (device, Npad floats) the column sums -- and ln_in / ln_eps before the first launch.  Producer: ln_out.
    const float* ln_gamma = nullptr; const float* ln_beta = nullptr;
    float* ln_cs = nullptr; const double* ln_in = nullptr; float ln_eps = 1e-5f;
    double* ln_out = nullptr;
    bool q_small_maps = false;   // set BEFORE mf_conv_plan_create (MF_PREC_F16Q): take the f16 + FP6 halo tile on maps from 16 x 16 and up to 2048 input channels as well
                                 // (the UNet's 640-channel 16 x 16 layers at >= 40 frames per step; the caller keeps a bf16x3 plan for smaller steps)
    bool q = false;       // MF_PREC_F16Q: w_hi = f16 [slice][tap][Npad][32], w_lo = [slice][tap][Npad][q6(wh) 32 B | q6(wl) 32 B] (24 B codes + E8M0 byte + pad)
    bool halo = false;    // 3x3 s1 p1 on a >= 16x16 map: LDS halo-tile kernel, weights packed [slice][tap][Npad][CK]
    bool thin = false;    // (with halo: the kernel addresses the input itself) cin <= 16 on a large map: k_conv_thin, weights packed by mf_thin_pack into w_hi
    bf16_t* up_hi = nullptr;  // nearest-2x-upsample + 3x3 layers that qualify for the fat halo tiles: [phase][slice][4 taps][Npad][CK], pre-summed taps
    bf16_t* up_lo = nullptr;
    ConvPlan* alt = nullptr;  // wide halo layers (> 256 channels) only run on the LDS-weights kernel's fat tiles, which need a few hundred
                              // workgroups: this implicit-GEMM twin (own weight pack) takes the launches whose batch is too small
    int n_slices = 0;
    int groups_cap = 1;   // grouped GEMM shells: packed operands the weight buffers hold
    hipEvent_t prof_mid = nullptr;   // measurement only: recorded between the MFMA kernel and its split-K combine
    float* ws = nullptr;  // split-K workspace, grown on the first (eager) launch that needs it
    int64_t ws_cap = 0;
    // Outgrown workspaces / counters.  A hipGraph captured at one batch size keeps the pointer it was captured with; the split count comes
    // from a batch-dependent cost model, so a SMALLER batch can need a LARGER workspace later.  Outgrown buffers are therefore never freed
    // while the plan lives (a replayed graph may still write them): they are parked here until mf_conv_plan_destroy.
    std::vector<void*> retired;
    // host-side phase description, independent of the buffers the layer is later bound to
    struct Tap {                              // input displacement in pixels relative to anchor
        int dy, dx;
        std::vector<std::pair<int, int>> src;  // kernel taps (ky, kx) summed into this tap; empty = derived
    };
    std::vector<std::vector<Tap>> phase_taps; // per phase
    std::vector<int> phase_oy, phase_ox;      // output pixel offset of the phase
    int out_step = 1;                         // output pixel stride of the quotient grid
    int in_step_h = 1, in_step_w = 1;         // input pixel stride of the quotient grid
    int in_halo_need = 0;
    ConvPhase ph[MF_MAX_PHASE]{};
    int goff_total = 0;
    // binding-dependent (built by mf_conv_bind)
    int bound_in_ld = -1, bound_in_wp = -1;
    std::map<int, ConvTuned> tuned;           // batch -> configuration measured on this device (mf_conv_tune); empty: the cost model decides
};

// Folds BN, packs weights, uploads.  Returns mf_status.
int mf_conv_plan_create(ConvPlan* p, const mf_conv2d_desc& d, const float* weight, const float* bias,
                        const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                        const float* bn_var, int precision);
void mf_conv_plan_destroy(ConvPlan* p);

// Builds the goff table for the input buffer geometry the plan will read (pixel stride = buffer C,
// row stride = Wp*C).  Must be called once before launch; rebinding to another geometry is allowed.
int mf_conv_bind(ConvPlan* p, const ActBuf& in);

// rocprofv3-style name of the kernel mf_conv_launch will use at this batch size
void mf_conv_kernel_name(const ConvPlan* p, int batch, char* buf, int cap);
// Workgroup tile the launch will use for this batch size (kernel = k_conv_igemm<bm,bn,wgm,wgn,x3>).
ConvTile mf_conv_pick_tile(const ConvPlan* p, int batch);
// Measures the implicit-GEMM launch configurations (tile x split-K x operand path) of this layer at this batch ON the bound buffers and keeps the
// fastest in p->tuned (layers that run on the halo kernels are left alone).  Eager only: call between two uncaptured forwards.  The cost model
// behind mf_conv_pick_tile was fitted to one kernel generation; on the UNet's small GEMMs its pick is 0-15 % off per layer in either direction.
int mf_conv_tune(ConvPlan* p, const ActView& in, const ActView& out, const ActView& res, int batch, hipStream_t stream);
// The table lookup alone (no launch, no timing): applies the configuration recorded for this layer's signature at this batch -- from MF_TUNE_CACHE, from the
// table shipped beside the library (tune/gfx950.txt), or from an mf_conv_tune earlier in this process.  Returns 1 if the signature was found.  This is all a
// forward ever does; measuring is explicit (mf_unet_tune / mf_vae_tune / mf_wav2lip_tune / mf_net_tune).
int mf_conv_tune_lookup(ConvPlan* p, const ActView& in, int batch);
// MF_AUTOTUNE=1 (development only): measure on the first forward at a batch size, as rounds 1-2 did; default off
bool mf_autotune_enabled();
// Algorithmic FLOPs of the layer (2 x MACs of the convolution itself; BN/ReLU/residual excluded).
double mf_conv_flops(const ConvPlan* p, int batch);

// Grouped GEMM on raw pointers (attention): for group z = (b, h), b = z / heads:
//   out[z][m][n] = sum_k x[b*zx_b + h*zx_h + m*x_row + k] * B_z[n][k],  B_z = plan weights + z*zw
// written at y + b*zy_b + h*zy_h + m*y_row + n.  The plan is a mf_gemm_plan_create shell whose packed weights
// hold `groups` consecutive operands.  No split-K, no residual.
struct GroupedGemm {
    const bf16_t* x_hi; const bf16_t* x_lo; int64_t zx_b, zx_h; int x_row;
    bf16_t* y_hi; bf16_t* y_lo; int64_t zy_b, zy_h; int y_row;
    int M, groups, heads;
};
int mf_gemm_grouped_launch(ConvPlan* p, const GroupedGemm& g, hipStream_t stream);

// channel-slice split the f16 + FP6 kernel runs a layer with at this batch (1 = none): maps with fewer 16 x 16 x 128-channel tiles than CUs
int mf_q_split_count(const ConvPlan* p, int batch);
// GroupNorm statistics pass alone (mf_nn.hip): (sum, sum of squares) per (sample, group) of view x ADDED to stats[2 * (b * groups + g)]
int mf_groupnorm_stats(const ActView& x, int groups, double* stats, int batch, hipStream_t s);
// Enqueues the layer.  res may have buf == nullptr.
// tokens > 0 (single-row sequences only, H == 1): compute only the first `tokens` output positions of every batch item -- a sequence
// prefix.  The buffers keep their geometry (base pointers, batch strides); the GEMM simply has M = batch * tokens rows.
int mf_conv_launch(ConvPlan* p, const ActView& in, const ActView& out, const ActView& res,
                   int batch, hipStream_t stream, int tokens = 0);

#include "mf_fastdiv.h"

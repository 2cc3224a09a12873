/*
 * merefusion.h -- C ABI of libmerefusion_hip.so, the MI355X (gfx950) frame generator that sits
 * underneath mere-fusion's lipreal.py / musereal.py render loops.
 *
 * The reference has no native FFI on this path (its hot loop is torch modules called from
 * Python); each entry point below names the Python interface it replaces (paths relative to the
 * reference checkout).  INTEGRATION.md shows the ctypes stub a maintainer adds on the reference
 * side.  All functions return 0 on success or a negative mf_status; they never throw across
 * the boundary.  mf_last_error() returns a thread-local message for the last failure.
 *
 * Pointers marked "device" are HIP device pointers borrowed for the duration of the call
 * (caller owns them, lipreal.py:121-126); "host" pointers are read during the call only.
 * A handle owns its packed weights, its activation workspace and its captured hipGraphs.
 * Handles are not thread-safe: one handle per (session, stream), as the reference runs one
 * single-threaded inference process per session (lipreal.py:85, app.py:549).
 */
#ifndef MEREFUSION_H
#define MEREFUSION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mf_status {
    MF_OK = 0,
    MF_ERR_INVALID = -1,   /* bad argument / shape / missing tensor */
    MF_ERR_HIP = -2,       /* a HIP runtime call failed */
    MF_ERR_NODEVICE = -3,  /* no gfx950 device visible */
    MF_ERR_NOMEM = -4
} mf_status;

/* Arithmetic mode of the MFMA convolution kernels.
 * MF_PREC_BF16   : bf16 activations/weights, fp32 accumulate (BASELINE config 2).
 * MF_PREC_BF16X3 : activations and weights carried as bf16 (hi, lo) pairs, three bf16 MFMAs per
 *                  product (hi*hi + lo*hi + hi*lo), fp32 accumulate: ~2^-17 relative operand
 *                  error; meets the north-star parity bound (L-inf <= 1e-3 vs the fp32 CPU
 *                  generator) that plain bf16 cannot. */
typedef enum mf_precision { MF_PREC_BF16 = 0, MF_PREC_BF16X3 = 1, MF_PREC_F16Q = 2 } mf_precision;
/* MF_PREC_F16Q (mf_conv2d_* test seam; inside the VAE decoder it is chosen by the schedule; wide 3x3 stride-1 layers, optionally behind a nearest 2x upsample): operands as f16 + FP6 (e2m3, OCP-MX block scales) residuals --
 * w.x = wh.xh (one f16 MFMA) + q6(wh).xl + wl.q6(xh) (one block-scaled 16x16x128 MFMA at 4x the rate); outputs stay bf16 (hi, lo).  DESIGN.md. */

/* One named fp32 host tensor of a checkpoint (a state_dict item). */
typedef struct mf_tensor {
    const char* name;   /* e.g. "face_encoder_blocks.0.0.conv_block.0.weight" */
    const float* data;  /* host, contiguous, fp32 */
    int ndim;
    int64_t shape[4];
} mf_tensor;

/* ---- library ---------------------------------------------------------------------------- */
/* Selects the HIP device (replaces `device = 'cuda'`, lipreal.py:29). */
int mf_init(int device);
const char* mf_last_error(void);
/* ABI version, bumped on any signature change. */
int mf_abi_version(void);

/* ---- Wav2Lip generator (H2) -------------------------------------------------------------- */
typedef struct mf_wav2lip mf_wav2lip;

/* Replaces `Wav2Lip()` + `load_state_dict` + `.to(device).eval()` (lipreal.py:43-53,
 * wav2lip/models/wav2lip.py:12-85).  `weights` are the state-dict tensors (101 weight, 101 bias,
 * 50 x running_mean / running_var; num_batches_tracked is ignored); a leading "module." in a name
 * is stripped as lipreal.py:48-49 does.  BatchNorm (eval, eps 1e-5, conv.py:10) is folded in. */
int mf_wav2lip_create(const mf_tensor* weights, int n_weights, int precision, mf_wav2lip** out);

/* Replaces `pred = model(mel_batch, img_batch)` (lipreal.py:124-125 -> wav2lip.py:87-125).
 * mel  : device fp32 [B,1,80,16]    face : device fp32 [B,6,96,96] (NCHW, values in [0,1])
 * out  : device fp32 [B,3,96,96] in [0,1], BGR.  Enqueued on `stream` (a hipStream_t; NULL =
 * default stream); returns without synchronising. */
int mf_wav2lip_forward(mf_wav2lip* h, const float* mel, const float* face, float* out, int batch,
                       void* stream);

/* Fuses the per-batch glue around the generator (lipreal.py:115-126): faces arrive as uint8
 * [B,96,96,3] BGR crops, are masked (rows >= 48 zeroed in the first 3 channels), concatenated
 * and scaled by 1/255 on the fly; frames leave as fp32 [B,96,96,3] = pred*255 (HWC, what
 * lipreal.py:126 hands to process_frames).  mel: device fp32 [B,1,80,16]. */
int mf_wav2lip_forward_u8(mf_wav2lip* h, const float* mel, const uint8_t* faces_u8,
                          float* frames_hwc, int batch, void* stream);

/* Copies the NCHW fp32 activation named `tap` of the LAST forward into `dst` (device), for
 * parity tests.  tap: "audio_embedding", "face_encoder_blocks.N", "face_decoder_blocks.N". */
int mf_wav2lip_read_tap(mf_wav2lip* h, const char* tap, float* dst, int batch, void* stream);

/* Measurement seam for bench.py (roofline): the generator's KERNEL launches in execution order at a
 * batch size (a split-K layer is two launches: the MFMA kernel and its combine pass).
 * launch_info: name = the reference module path ("face_decoder_blocks.4.1"), kernel = the HIP kernel
 * name as rocprofv3 prints it (template arguments without spaces), flops = algorithmic FLOPs of that
 * launch (2 x conv MACs; 0 for layout / combine kernels).
 * profile: runs `iters` forwards WITHOUT the hipGraph on `stream`, a hipEvent before every launch,
 * and returns the mean milliseconds of each launch. */
int mf_wav2lip_num_launches(const mf_wav2lip* h, int batch);
int mf_wav2lip_launch_info(const mf_wav2lip* h, int index, int batch, char* name, int name_cap,
                           char* kernel, int kernel_cap, double* flops);
int mf_wav2lip_profile(mf_wav2lip* h, const float* mel, const float* face, float* out, int batch,
                       int iters, float* ms_per_launch, void* stream);

/* Launch configurations.  A forward never measures: at the first forward of a batch size every implicit-GEMM layer looks its signature up in the
 * tuning table -- the file MF_TUNE_CACHE names, else `tune/gfx950.txt` beside the library (measured on an MI355X for the BASELINE.json shapes) --
 * and a signature that is not there runs the cost model's pick.  mf_*_tune is the explicit warm-up a server calls at start-up for every batch size its
 * loop can emit: it times each layer's tile x split-K x operand-path candidates on the buffers the last forward at `batch` filled (one forward at
 * that batch must have run), records the winners in the table (appended to MF_TUNE_CACHE when set) and drops the hipGraph captured with the old
 * ones.  Seconds per full-size network; results differ from the un-tuned ones by fp32 summation order only.  (ABI version 4) */
int mf_wav2lip_tune(mf_wav2lip* h, int batch, void* stream);

void mf_wav2lip_destroy(mf_wav2lip* h);

/* ---- single fused convolution layer (building block, also the per-geometry test seam) ---- */
typedef struct mf_conv2d mf_conv2d;

typedef struct mf_conv2d_desc {
    int cin, cout;
    int kh, kw;
    int stride_h, stride_w;
    int pad_h, pad_w;
    int transposed;       /* 0 = Conv2d (conv.py:5), 1 = ConvTranspose2d (conv.py:33) */
    int output_padding;   /* transposed only */
    int residual;         /* 1: add the layer input before the activation (conv.py:17-18); 2: after it */
    int act;              /* 0 none, 1 ReLU, 2 sigmoid, 3 GELU (erf), 4 SiLU, 5 GEGLU: y = x[:, :cout/2] * gelu(x[:, cout/2:])
                           * (diffusers GEGLU of the UNet feed-forward), cout % 32 == 0, the output has cout/2 channels */
    int in_h, in_w;       /* spatial size of the input this layer is built for */
    int upsample;         /* 1: nearest-neighbour 2x upsampling in front of a 3x3 s1 p1 conv (diffusers Upsample2D) */
    int pad_hi;           /* extra zero rows / columns at the BOTTOM / RIGHT only (diffusers Downsample2D of the VAE encoder: F.pad(x, (0, 1, 0, 1)) +
                           * conv k3 s2 p0); 0 for every other layer.  (ABI version 2) */
} mf_conv2d_desc;

/* weight: host fp32, [cout,cin,kh,kw] (Conv2d) or [cin,cout,kh,kw] (ConvTranspose2d);
 * bias: host fp32 [cout] or NULL; bn_*: host fp32 [cout] or all NULL (no BatchNorm). */
int mf_conv2d_create(const mf_conv2d_desc* desc, const float* weight, const float* bias,
                     const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                     const float* bn_var, int precision, mf_conv2d** out);
/* x: device fp32 NCHW [B,cin,in_h,in_w]; y: device fp32 NCHW [B,cout,out_h,out_w]. */
int mf_conv2d_forward(mf_conv2d* h, const float* x, float* y, int batch, void* stream);
int mf_conv2d_out_shape(const mf_conv2d* h, int* out_h, int* out_w);
/* The same layer as the producer of a GroupNorm(groups) (diffusers ResnetBlock2D: conv -> norm): besides y, the launch leaves the (sum, sum of
 * squares) of the stored output per (sample, group) in stats -- device fp64 [B][groups][2] -- from the kernel's epilogue or split-K combine
 * where the chosen configuration can, else from a statistics pass behind it (test seam of what the UNet / VAE schedules do between layers). */
int mf_conv2d_forward_stats(mf_conv2d* h, const float* x, float* y, int groups, double* stats, int batch, void* stream);
/* Measurement seam: mean milliseconds of the convolution launch alone (layout passes excluded) over
 * `iters` repeats on the buffers of the last mf_conv2d_forward, bracketed by hipEvents on `stream`. */
int mf_conv2d_time(mf_conv2d* h, int batch, int iters, float* ms, void* stream);
void mf_conv2d_destroy(mf_conv2d* h);

/* ---- Wav2Lip mel-spectrogram (H1) -------------------------------------------------------- */
/* Replaces `audio.melspectrogram(inputs)` (lipasr.py:23 -> wav2lip/audio.py:45-51 with the
 * constants of wav2lip/hparams.py:33-73): pre-emphasis 0.97, centred STFT n_fft=800 hop=200
 * periodic Hann, 80 Slaney mel bands 55-7600 Hz, 20*log10(max(1e-5,.)) - 20, clip-normalise
 * to [-4,4].  wav: device fp32 [n]; out: device fp32 [80, T], T = 1 + n/200.
 * pad_mode: 0 = zeros (librosa >= 0.10 default "constant"), 1 = reflect (librosa < 0.10). */
int mf_melspec(const float* wav, int n, float* out, int pad_mode, void* stream);
int mf_melspec_frames(int n);

/* ---- fused multi-head attention (test seam of the kernel both transformer stages run on) ---- */
/* out = softmax(q k^T / sqrt(head_dim)) v per (batch, head): the `Attention` of the diffusers
 * BasicTransformerBlock that musetalk/models/unet.py:36-47 instantiates (attn1: tk == tq; attn2: tk = audio
 * tokens) and `MultiHeadAttention.qkv_attention` of musetalk/whisper/whisper/model.py:62-93.
 * q, out: device fp32 [batch][tq][heads*head_dim]; k, v: device fp32 [batch][tk][heads*head_dim];
 * head_dim in {40, 64, 80, 160}.  Synchronises the stream before returning. */
int mf_attention_forward(const float* q, const float* k, const float* v, float* out, int batch, int tq, int tk,
                         int heads, int head_dim, int precision, void* stream);

/* ---- MuseTalk Whisper audio features (H3) -------------------------------------------------- */
typedef struct mf_whisper mf_whisper;

/* Replaces `load_model(path)` of musetalk/whisper/whisper/__init__.py:108-116 for the ENCODER half
 * (`Whisper.encoder`, model.py:131-171): weights = the encoder state-dict tensors ("conv1.weight",
 * "blocks.N.attn.query.weight", ...; an "encoder." prefix is accepted); n_head comes from the checkpoint's
 * dims (6 for tiny); n_state / n_layer are inferred from the tensors, n_ctx is 1500 (30 s). */
int mf_whisper_create(const mf_tensor* weights, int n_weights, int n_head, int precision, mf_whisper** out);
int mf_whisper_dims(const mf_whisper* h, int* n_layer, int* n_ctx, int* n_state);

/* Replaces `log_mel_spectrogram(audio)` (whisper/audio.py:92-125): wav device fp32 [n], 160 <= n <= 480000
 * -> out device fp32 [80, n/160]. */
int mf_whisper_log_mel(mf_whisper* h, const float* wav, int n, float* out, void* stream);

/* One segment of `transcribe` (transcribe.py:103-126): log-mel, pad to 3000 frames,
 * `encoder(segment, include_embeddings=True)`.  emb: device fp32 [n_layer+1][1500][n_state]
 * (= the reference's embeddings[0], model.py:158-168).  n <= 480000 samples (one 30 s segment). */
int mf_whisper_encode_audio(mf_whisper* h, const float* wav, int n, float* emb, void* stream);

/* The streaming form of the same stage: what `MuseASR.run_step` needs from `audio2feat` for the
 * sliding windows of one or several sessions (museasr.py:25-26 -> audio2feature.py:99-112 -> transcribe.py:103-126).
 * wav: device fp32 [n_windows][n], every window the same length (the render loop's (2B + l + r) * 320 samples);
 * feat: device fp32 [n_windows][n/320][n_layer+1][n_state] -- `concatenated_array` of audio2feature.py:110 per window.
 * ctx_tokens = 0 keeps the reference's 1500-token context; the result equals mf_whisper_encode_audio's first n/320 tokens
 * (the last block evaluates only the consumed queries -- its keys / values still span the whole context -- and all
 * windows share every launch).  ctx_tokens > 0 shortens the context to that many tokens: an approximation, because the
 * encoder's attention is global and unmasked; bench.py reports its error next to its time.
 * mf_whisper_set_batch sizes the workspace for up to max_windows windows per call (default 1). */
int mf_whisper_set_batch(mf_whisper* h, int max_windows);
int mf_whisper_encode_windows(mf_whisper* h, const float* wav, int n, int n_windows, int ctx_tokens, float* feat, void* stream);
void mf_whisper_destroy(mf_whisper* h);

/* ---- MuseTalk UNet (H4) and VAE decode (H5) --------------------------------------------------- */
/* The reference builds both from diffusers with config files it does not ship (musetalk/models/unet.py:34-37,
 * vae.py:24), so the architecture is a parameter here.  Field meaning = the diffusers config keys. */
typedef struct mf_unet_config {
    int in_channels, out_channels;      /* 8, 4 */
    int n_blocks;                        /* len(block_out_channels), <= 4 */
    int block_out_channels[4];          /* 320, 640, 1280, 1280 */
    int layers_per_block;               /* 2 */
    int cross_attention_dim;            /* 384 */
    int attention_heads;                /* config key "attention_head_dim": 8 heads */
    int norm_num_groups;                /* 32 */
    int down_attn[4], up_attn[4];       /* CrossAttn{Down,Up}Block2D vs {Down,Up}Block2D */
    int sample_size;                    /* latent H = W: 32 */
    int ctx_len;                        /* audio tokens per frame: 50 */
} mf_unet_config;

typedef struct mf_vae_config {
    int latent_channels, out_channels;  /* 4, 3 */
    int n_blocks;
    int block_out_channels[4];          /* 128, 256, 512, 512 */
    int layers_per_block;               /* 2 */
    int norm_num_groups;                /* 32 */
    int sample_size;                    /* latent H = W: 32 */
    float scaling_factor;               /* 0.18215 */
} mf_vae_config;

typedef struct mf_unet mf_unet;
typedef struct mf_vae mf_vae;

/* Replaces `UNet(unet_config, model_path)` (musetalk/models/unet.py:29-44): weights = the
 * UNet2DConditionModel state dict (diffusers key names).  The timestep is fixed to 0 as musereal.py:59 does;
 * its embedding is folded into the resnet biases at create time. */
int mf_unet_create(const mf_unet_config* cfg, const mf_tensor* weights, int n_weights, int precision, int max_batch,
                   mf_unet** out);
/* Replaces `pe(audio)` + `unet.model(latent_batch, timesteps, encoder_hidden_states=audio).sample`
 * (musereal.py:102-107).  latents: device fp32 [B,in_channels,S,S]; audio: device fp32 [B,ctx_len,cross_dim];
 * add_pe != 0 adds the PositionalEncoding of unet.py:12-27 to `audio` first; out: device fp32 [B,out_channels,S,S]. */
int mf_unet_forward(mf_unet* h, const float* latents, const float* audio, int add_pe, float* out, int batch, void* stream);
/* Measurement seam (bench.py roofline): the schedule's ops in execution order on the buffers of the last forward.
 * op_info: name = the diffusers module path, kernel = HIP kernel(s) it launches, flops_per_frame = 2 x MACs.
 * profile: every op alone between two hipEvents on `stream`, hipGraph off; mean milliseconds per op. */
int mf_unet_num_ops(const mf_unet* h);
int mf_unet_op_info(const mf_unet* h, int i, char* name, int ncap, char* kernel, int kcap, double* flops_per_frame);
int mf_unet_profile(mf_unet* h, int batch, int iters, float* ms_per_op, void* stream);
int mf_unet_tune(mf_unet* h, int batch, void* stream);                       /* explicit launch-configuration warm-up: see mf_wav2lip_tune */
void mf_unet_destroy(mf_unet* h);

/* Replaces `AutoencoderKL.from_pretrained` for the DECODER half (musetalk/models/vae.py:24,96-108). */
int mf_vae_create(const mf_vae_config* cfg, const mf_tensor* weights, int n_weights, int precision, int max_batch,
                  mf_vae** out);
/* `VAE.decode_latents` (vae.py:96-108): latents device fp32 [B,4,S,S] -> frames device uint8 [B,8S,8S,3] BGR.
 * image_f32 (optional, may be NULL): the decoder output before post-processing, device fp32 [B,3,8S,8S]. */
int mf_vae_decode_latents(mf_vae* h, const float* latents, uint8_t* frames, float* image_f32, int batch, void* stream);
int mf_vae_num_ops(const mf_vae* h);
int mf_vae_op_info(const mf_vae* h, int i, char* name, int ncap, char* kernel, int kcap, double* flops_per_frame);
int mf_vae_profile(mf_vae* h, int batch, int iters, float* ms_per_op, void* stream);
int mf_vae_tune(mf_vae* h, int batch, void* stream);                         /* explicit launch-configuration warm-up: see mf_wav2lip_tune */
void mf_vae_destroy(mf_vae* h);

/* ---- ER-NeRF inference kernels (H6): the functions of the reference's four torch extensions ------------------
 * Same argument order and in-place output conventions as the pybind entry points the Python wrappers call, so
 * `ernerf/raymarching/raymarching.py`, `gridencoder/grid.py`, `shencoder/sphere_harmonics.py`, `freqencoder/freq.py`
 * run unchanged on top of them.  All pointers are device fp32 / int32 / uint8 unless
 * noted; all calls are asynchronous on `stream`. */

/* `_backend.near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars)` (raymarching.py:44 ->
 * kernel_near_far_from_aabb, raymarching.cu:92-145).  rays_o/d [N,3], aabb [6]; a miss writes FLT_MAX to both. */
int mf_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t n_rays, float min_near,
                          float* nears, float* fars, void* stream);
/* `_backend.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
 * density_bitfield, near, far, xyzs, dirs, deltas, noises)` (raymarching.py:393 -> kernel_march_rays,
 * raymarching.cu:828-929).  xyzs/dirs [n_alive*n_step,3] and deltas [n_alive*n_step,2] must be zero-filled by the
 * caller (raymarching.py:383-385): dt == 0 marks "no sample". */
int mf_march_rays(uint32_t n_alive, uint32_t n_step, const int* rays_alive, const float* rays_t, const float* rays_o,
                  const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t cascades, uint32_t grid_size,
                  const uint8_t* density_bitfield, const float* nears, const float* fars, float* xyzs, float* dirs,
                  float* deltas, const float* noises, void* stream);
/* `_backend.composite_rays_triplane(...)` (raymarching.py:666 -> kernel_composite_rays_triplane,
 * raymarching.cu:2142-2249): updates rays_alive (-1 = terminated), rays_t and the [N]-sized accumulators in place. */
int mf_composite_rays_triplane(uint32_t n_alive, uint32_t n_step, float T_thresh, int* rays_alive, float* rays_t,
                               const float* sigmas, const float* rgbs, const float* deltas, const float* ambs_aud,
                               const float* ambs_eye, const float* uncertainties, float* weights_sum, float* depth,
                               float* image, float* amb_aud_sum, float* amb_eye_sum, float* uncertainty_sum, void* stream);
/* `_backend.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype,
 * align_corners)` (grid.py:49 -> kernel_grid, gridencoder.cu:76-165), inference only (no dy_dx).
 * inputs [B,D] in [0,1]; embeddings [offsets[L], C]; offsets_host: HOST int32 [L+1]; S = log2(per_level_scale);
 * outputs [L,B,C] as the extension writes it, or -- out_blc != 0 -- [B, L*C], the layout grid.py:52 permutes to.
 * D in {2,3}, C in {1,2,4,8}, L <= 32. */
int mf_grid_encode_forward(const float* inputs, const float* embeddings, const int* offsets_host, float* outputs, uint32_t B,
                           uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                           int out_blc, void* stream);
/* `_backend.sh_encode_forward(inputs, outputs, B, input_dim, degree, dy_dx)` (sphere_harmonics.py:32 -> kernel_sh,
 * shencoder.cu:28-110), inference only; inputs [B,3], outputs [B,degree^2], degree 1..4. */
int mf_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t degree, void* stream);
/* `_backend.freq_encode_forward(inputs, B, input_dim, degree, output_dim, outputs)` (freq.py:29 -> kernel_freq,
 * freqencoder.cu:30-58); outputs [B, D + 2*D*degree]. */
int mf_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t degree, uint32_t C, float* outputs, void* stream);

/* Tail of `run_cuda` (renderer.py:275-280) and the uint8 conversion of nerfreal.py:111, in place:
 * image[N,3] = clamp(image + (1 - weights_sum) * bg, 0, 1); depth[N] = clamp(depth - near, 0) / (far - near);
 * frame_u8 (optional, may be NULL) = uint8(image * 255), truncating like ndarray.astype.  bg_color: device fp32 [N,3]
 * (bg_per_ray != 0), [3], or NULL for the scalar bg_const (`bg_color = 1`, renderer.py:309-310). */
int mf_nerf_finish(float* image, float* depth, const float* weights_sum, const float* nears, const float* fars, const float* bg_color,
                   int bg_per_ray, float bg_const, uint32_t n_rays, uint8_t* frame_u8, void* stream);

/* ---- ER-NeRF radiance field (H6) ---------------------------------------------------------- */
typedef struct mf_nerf_field mf_nerf_field;
typedef struct mf_nerf_field_config {
    float bound;                 /* opt.bound (app.py:603): positions live in [-bound, bound] */
    int num_levels;              /* 12 (network.py:122) */
    int level_dim;               /* 1 */
    int base_resolution;         /* 64 */
    float log2_per_level_scale;  /* S of grid.py:33 */
    int offsets[33];             /* GridEncoder.offsets (grid.py:108-123), num_levels + 1 entries */
    int audio_dim;               /* 32 */
    int geo_feat_dim;            /* 64 */
    int hidden_dim;              /* 64 */
    int individual_dim;          /* opt.ind_dim, 4 */
    int exp_eye;                 /* opt.exp_eye */
} mf_nerf_field_config;
/* Replaces the inference half of `NeRFNetwork` (network.py:93-160): weights = the state-dict tensors
 * "encoder_{xy,yz,xz}.embeddings", "{sigma_net,color_net,aud_ch_att_net,eye_att_net}.net.N.weight" (fp32, host). */
int mf_nerf_field_create(const mf_nerf_field_config* cfg, const mf_tensor* weights, int n_weights, int precision,
                         int max_samples, mf_nerf_field** out);
/* Replaces `self.forward(xyzs, dirs, enc_a, ind_code, eye)` (renderer.py:260 -> network.py:249-277, density :280-308).
 * xyzs, dirs: device fp32 [M,3]; enc_a: device fp32 [32]; ind_code: device fp32 [individual_dim] or NULL; eye: the
 * scalar of opt.exp_eye.  Outputs (device fp32): sigmas [M], rgbs [M,3], amb_aud [M] (= ||aud_ch_att||), amb_eye [M],
 * uncertainty [M] (ln 2 in test mode, may be NULL). */
int mf_nerf_field_forward(mf_nerf_field* h, const float* xyzs, const float* dirs, const float* enc_a, const float* ind_code,
                          float eye, int n_samples, float* sigmas, float* rgbs, float* amb_aud, float* amb_eye,
                          float* uncertainty, void* stream);
void mf_nerf_field_destroy(mf_nerf_field* h);

/* a24, `Trainer.test_gui_with_data` utils.py:1208-1216 + nerfreal.py:111: the [h,w] render resized to the GUI's [H,W].
 * image [h,w,3] -> out_image [H,W,3] as F.interpolate(mode='bilinear') (align_corners False, no antialias); depth [h,w] ->
 * out_depth [H,W] as mode='nearest'; frame_u8 [H,W,3] = uint8(out_image * 255) truncating.  Any of the three outputs may
 * be NULL (depth may be NULL when out_depth is). */
int mf_nerf_resize_frame(const float* image, const float* depth, int h, int w, int H, int W, float* out_image, float* out_depth,
                         uint8_t* frame_u8, void* stream);

/* ---- ER-NeRF head frame without host round trips ----------------------------------------------- */
typedef struct mf_nerf_head mf_nerf_head;
/* Scratch for up to max_rays rays over `field` (which must outlive the head and use the fused field kernel). */
int mf_nerf_head_create(mf_nerf_field* field, int max_rays, mf_nerf_head** out);
/* The inference branch of `NeRFRenderer.run_cuda` (renderer.py:231-291) for one frame, enqueued in one go: near/far,
 * up to max_steps rounds of (round control -> march_rays -> field -> composite_rays_triplane -> compaction) whose counters
 * (n_alive, n_step = max(min(N // n_alive, 8), 1), step) live on the device, then the background mix / depth
 * normalisation of :275-280.  No host synchronisation: the call only enqueues (capturable in a hipGraph).
 * Only the rounds the frames before needed (every round until a first frame has reported) go out as launches; ONE
 * further launch stands for the rest of the max_steps rounds and runs them itself if the loop has not ended by then
 * (mf_nerf_head_plan_rounds / _set_rounds below; MF_NERF_TAIL_AFTER=<k>|off overrides).
 * rays_o, rays_d [N,3]; density_bitfield [cascades * grid_size^3 / 8]; enc_a [32]; ind_code [individual_dim] or NULL;
 * bg as in mf_nerf_finish.  Outputs: image [N,3], depth [N], weights_sum [N] (optional), frame_u8 [N,3] (optional).
 * Survivors of a round keep no particular order (rays are independent), unlike the reference's boolean mask.
 * bg_per_ray < 0 leaves image / depth unfinished (no background mix, depth not normalised): the caller finishes with
 * mf_nerf_head_finish, e.g. after joining a stream on which the torso (mf_nerf_torso_forward) produced bg_color. */
int mf_nerf_head_render(mf_nerf_head* h, const float* rays_o, const float* rays_d, int n_rays, const uint8_t* density_bitfield, int cascades,
                        int grid_size, float min_near, float dt_gamma, int max_steps, float T_thresh, float density_scale, const float* enc_a,
                        const float* ind_code, float eye, const float* bg_color, int bg_per_ray, float bg_const, float* image, float* depth,
                        float* weights_sum, uint8_t* frame_u8, void* stream);
/* renderer.py:275-280 + nerfreal.py:111 for the frame a mf_nerf_head_render(bg_per_ray < 0) left unfinished; weights_sum may be
 * NULL when none was passed to the render call. */
int mf_nerf_head_finish(mf_nerf_head* h, int n_rays, const float* bg_color, int bg_per_ray, float bg_const, float* image, float* depth,
                        const float* weights_sum, uint8_t* frame_u8, void* stream);
/* The eye feature `e` of `NeRFNetwork.forward` (network.py:249, `data['eye']`: a [1, 1] device tensor in the reference's loader) read from DEVICE memory by every
 * later mf_nerf_head_render on `h` instead of its by-value `eye` argument: no host copy of a value that lives on the device, and a captured graph follows a changing
 * value.  NULL: back to the by-value argument.  The float must stay valid until the renders that use it have finished. */
int mf_nerf_head_set_eye(mf_nerf_head* h, const float* eye_dev);
/* Rounds of a frame enqueued as (march, field, composite) launches before the tail launch.  _plan_rounds returns what the next mf_nerf_head_render would
 * choose from the round counts earlier frames posted (device -> pinned host word, read without a sync); _set_rounds(r >= 0) pins the count -- a caller that
 * captures the enqueue in a hipGraph pins what it keyed the graph on -- and _set_rounds(-1) returns to following the feedback.  _last_rounds: the round count
 * the most recently FINISHED frame posted (0 before the first), and the tail's error flag. */
int mf_nerf_head_plan_rounds(mf_nerf_head* h, int max_steps, int* rounds);
int mf_nerf_head_set_rounds(mf_nerf_head* h, int rounds);
int mf_nerf_head_last_rounds(mf_nerf_head* h, int* rounds, int* error_flag);
/* Diagnostics: the first n_ints words of the loop's control block (synchronous copy; layout in csrc/mf_nerf_march.h). */
int mf_nerf_head_ctl_snapshot(mf_nerf_head* h, int* out, int n_ints);
/* The per-ray sums `run_cuda` also returns at inference (`results['ambient_aud' | 'ambient_eye' | 'uncertainty']`, renderer.py:286-288) of the LAST frame
 * rendered through `h`: device-to-device copies enqueued on `stream` behind that frame (any of the three may be NULL). */
int mf_nerf_head_sums(mf_nerf_head* h, int n_rays, float* ambient_aud, float* ambient_eye, float* uncertainty, void* stream);
void mf_nerf_head_destroy(mf_nerf_head* h);

/* ---- ER-NeRF torso branch --------------------------------------------------------------------- */
typedef struct mf_nerf_torso mf_nerf_torso;
typedef struct mf_nerf_torso_config {
    float torso_shrink;          /* opt.torso_shrink, 0.8 (app.py:595) */
    int num_levels;              /* 16: tiled grid of network.py:158 */
    int level_dim;               /* 2 */
    int base_resolution;         /* 16 */
    float log2_per_level_scale;
    int offsets[33];             /* GridEncoder.offsets of torso_encoder */
    int individual_dim;          /* opt.ind_dim_torso, 8 */
    int grid_size;               /* 128: density_grid_torso is grid_size^2 (renderer.py:121) */
} mf_nerf_torso_config;
/* weights: "torso_deform_net.net.N.weight", "torso_net.net.N.weight", "torso_encoder.embeddings", "density_grid_torso". */
int mf_nerf_torso_create(const mf_nerf_torso_config* cfg, const mf_tensor* weights, int n_weights, int precision, int max_pixels,
                         mf_nerf_torso** out);
/* Replaces `run_torso` + `forward_torso` (renderer.py:294-352, network.py:166-201) for one frame.
 * bg_coords: device fp32 [N,2] in [-1,1]; frame_consts_host: HOST fp32 [42 + individual_dim] = FreqEncoder(6,3) of the
 * wrapped anchors (network.py:175-178) followed by the torso individual code; bg_color as in mf_nerf_finish;
 * density_thresh = min(opt.density_thresh_torso, mean_density_torso) (renderer.py:325).
 * bg_out: device fp32 [N,3] = torso_color * torso_alpha + bg * (1 - torso_alpha); torso_alpha [N] and deform [N,2] optional. */
int mf_nerf_torso_forward(mf_nerf_torso* h, const float* bg_coords, const float* frame_consts_host, const float* bg_color,
                          int bg_per_ray, float bg_const, float density_thresh, int n_pixels, float* bg_out, float* torso_alpha,
                          float* deform, void* stream);
void mf_nerf_torso_destroy(mf_nerf_torso* h);

/* ---- ER-NeRF audio features ------------------------------------------------------------------- */
typedef struct mf_audio_encoder mf_audio_encoder;
/* weights: "audio_net.*" and (use_att) "audio_att_net.*" of the NeRFNetwork state dict (network.py:9-66), fp32 host. */
int mf_audio_encoder_create(const mf_tensor* weights, int n_weights, int use_att, mf_audio_encoder** out);
/* Replaces `NeRFNetwork.encode_audio(a)` (network.py:222-237): auds device fp32 [n_windows, audio_in_dim, 16]
 * (8 windows with the attention net, 1 without) -> enc_a device fp32 [32]. */
int mf_audio_encoder_forward(mf_audio_encoder* h, const float* auds, int n_windows, float* enc_a, void* stream);
/* The same followed by the lip-smoothing step of `NeRFRenderer.run_cuda` (ernerf/nerf_triplane/renderer.py:190-194):
 * enc_a = 0.35 * prev_enc_a + (1 - 0.35) * encode_audio(a), in torch's fp32 evaluation order (bit-identical to the three torch launches it replaces).
 * prev_enc_a: device fp32 [32], the previous frame's smoothed features (may alias enc_a); null = no smoothing (first frame). */
int mf_audio_encoder_forward_smooth(mf_audio_encoder* h, const float* auds, int n_windows, const float* prev_enc_a, float* enc_a, void* stream);
void mf_audio_encoder_destroy(mf_audio_encoder* h);

/* ---- sd-vae encoder: avatar preparation -------------------------------------------------------- */
typedef struct mf_vae_encoder mf_vae_encoder;
/* Replaces `self.vae.encode(image).latent_dist` of musetalk/models/vae.py:84-94 (diffusers AutoencoderKL.encode: Encoder +
 * quant_conv; used once per avatar frame by mere_musetalk.py:175-188, 303-304).  cfg as for mf_vae_create (sample_size = the
 * LATENT grid, 32; the image is sample_size * 2^(n_blocks-1) = 256); weights: the `encoder.*` and `quant_conv.*` tensors. */
int mf_vae_encoder_create(const mf_vae_config* cfg, const mf_tensor* weights, int n_weights, int precision, int max_batch,
                          mf_vae_encoder** out);
/* Exactly one of image / image_u8_bgr: image = device fp32 [B,3,256,256], already normalised (the tensor vae.py:84 receives);
 * image_u8_bgr = device uint8 [B,256,256,3] BGR crops, preprocessed on the device as vae.py:52-82 does (RGB, / 255.,
 * half_mask: rows >= 128 zeroed, Normalize(.5, .5)).  moments: device fp32 [B, 2 * latent_channels, 32, 32] = (mean | logvar) of
 * `latent_dist`; `sample()` = mean + exp(0.5 * clamp(logvar, -30, 20)) * noise and the scaling factor stay with the caller. */
int mf_vae_encode(mf_vae_encoder* h, const float* image, const uint8_t* image_u8_bgr, int half_mask, float* moments, int batch,
                  void* stream);
int mf_vae_encoder_image_size(const mf_vae_encoder* h);
void mf_vae_encoder_destroy(mf_vae_encoder* h);

/* ---- per-batch glue of the MuseTalk loop ------------------------------------------------ */
/* musereal.py:92-97: `latent_batch = torch.cat([input_latent_list_cycle[__mirror_index(length, index + i)] ...])`.
 * pool: device fp32 [n_pool_rows][row_elems] (every session's cached latents, 8*32*32 = 8192 floats per row);
 * rows: HOST int[n], the pool row of each batch entry (mirror index + the session's offset); out: device fp32
 * [n][row_elems].  row_elems must be a multiple of 4. */
int mf_gather_rows_f32(const float* pool, int n_pool_rows, int64_t row_elems, const int* rows, int n, float* out, void* stream);

/* Audio2Feature.feature2chunks / get_sliced_feature (musetalk/whisper/audio2feature.py:16-45, called at museasr.py:27)
 * on the device: chunk i = feature rows [left_rows[i], left_rows[i] + rows_per_chunk), each clamped to [0, T-1]
 * (`min(length - 1, max(0, idx))`), concatenated.  feat: device fp32 [T][row_elems] (row_elems = 5 * 384 for
 * Whisper-tiny); left_rows: HOST int[n_chunks] = int(vid_idx * 50 / fps) - 2 * audio_feat_length[0]; out: device fp32
 * [n_chunks][rows_per_chunk][row_elems] == [n_chunks][50][384]. */
int mf_whisper_feature_chunks(const float* feat, int T, int row_elems, const int* left_rows, int rows_per_chunk, int n_chunks,
                              float* out, void* stream);

/* ---- paste-back of the generated face into the cached full frame --------------------------- */
/* One output frame.  Wav2Lip (lipreal.py:207-214): `combine_frame = deepcopy(frame_list_cycle[idx]);
 * res = cv2.resize(res_frame.astype(np.uint8), (x2 - x1, y2 - y1)); combine_frame[y1:y2, x1:x2] = res` -> mask == NULL.
 * MuseTalk (musereal.py:238-247 + musetalk/utils/blending.py:103-125): the same resize, then
 * `get_image_blending(ori_frame, res, bbox, mask, mask_crop_box)`: inside the crop box the frame becomes
 * cv2.blendLinear(crop with the face pasted in, crop, gray(mask) / 255, 1 - gray(mask) / 255). */
typedef struct mf_paste_job {
    int frame_index;          /* cached full frame this output starts from (frame_list_cycle[idx]) */
    int x1, y1, x2, y2;       /* face bbox, x / y order as blending.py:105 (lipreal.py's coords are (y1, y2, x1, x2): reorder) */
    int cx1, cy1, cx2, cy2;   /* MuseTalk: mask crop box (x_s, y_s, x_e, y_e), blending.py:106; ignored when mask == NULL */
    const uint8_t* mask;      /* device uint8 [cy2-cy1][cx2-cx1][3] BGR (mask_list_cycle[idx]), or NULL: rectangle copy */
} mf_paste_job;
/* res: device, n_jobs generated frames [res_h][res_w][3], uint8, or fp32 with res_is_f32 (Wav2Lip's pred * 255, truncated as
 * `astype(np.uint8)` does); frames: device uint8 [n_frames][H][W][3] BGR; jobs: HOST array; out: device uint8
 * [n_jobs][H][W][3].  Bit-exact with OpenCV's 8-bit INTER_LINEAR resize / BGR2GRAY / blendLinear (csrc/mf_blend.hip). */
int mf_paste_frames(const void* res, int res_is_f32, int res_h, int res_w, const uint8_t* frames, int n_frames, int H, int W,
                    const mf_paste_job* jobs, int n_jobs, uint8_t* out, void* stream);
/* cv2.resize(src, (dw, dh)) for uint8 [sh][sw][3] with the default INTER_LINEAR (lipreal.py:211, musereal.py:241). */
int mf_resize_linear_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, void* stream);

/* ---- ER-NeRF audio front-end: wav2vec2 / HuBERT CTC network ----------------------------------- */
/* Replaces `self.processor(frame, ...)` + `self.model(inputs.input_values)` of NerfASR.__frame_to_text (nerfasr.py:128-143):
 * transformers' Wav2Vec2FeatureExtractor normalisation, then Wav2Vec2ForCTC (`.logits`) or HubertModel (`.last_hidden_state`, out_hidden = 1)
 * -- feature extractor (conv -> LayerNorm -> GELU per layer, feat_extract_norm = "layer"), feature projection, weight-normed grouped
 * positional convolution, pre-LN ("do_stable_layer_norm") or post-LN transformer layers, final LayerNorm, lm_head.  Field names follow
 * transformers' Wav2Vec2Config; weights = the model's state dict ("wav2vec2." / "hubert." prefixes accepted, the positional conv as
 * `weight`, `weight_g` / `weight_v` or `parametrizations.weight.original0 / 1`).  (ABI version 3) */
typedef struct mf_wav2vec2 mf_wav2vec2;
typedef struct {
    int hidden, n_layer, n_head, ffn, vocab;            /* hidden_size, num_hidden_layers, num_attention_heads, intermediate_size, vocab_size */
    int n_conv, conv_dim[8], conv_kernel[8], conv_stride[8], conv_bias;
    int feat_norm_layer;                                /* 1: feat_extract_norm == "layer" (required) */
    int stable_ln;                                      /* do_stable_layer_norm */
    int pos_k, pos_groups;                              /* num_conv_pos_embeddings, num_conv_pos_embedding_groups */
    float layer_norm_eps;
    int do_normalize;                                   /* the processor's zero-mean / unit-variance step */
    int out_hidden;                                     /* 0: logits [T][vocab]; 1: the final hidden state [T][hidden] */
} mf_wav2vec2_config;
/* One handle serves windows of exactly n_samples samples (NerfASR feeds (l + m + r) * 320 every step), up to max_windows per call. */
int mf_wav2vec2_create(const mf_wav2vec2_config* cfg, const mf_tensor* weights, int n_weights, int n_samples, int max_windows, int precision,
                       mf_wav2vec2** out);
/* frames per window (20 ms each) and the width of an output row */
int mf_wav2vec2_frames(const mf_wav2vec2* h, int* n_frames, int* width);
/* wav: device fp32 [n_windows][n_samples] (raw samples: the normalisation happens here); out: device fp32 [n_windows][n_frames][width]. */
int mf_wav2vec2_forward(mf_wav2vec2* h, const float* wav, int n_samples, int n_windows, float* out, void* stream);
void mf_wav2vec2_destroy(mf_wav2vec2* h);

/* ---- avatar preparation: static CNN graphs --------------------------------------------------------- */
/* The S3FD face detector (`s3fd.forward`, face_detection/detection/sfd/net_s3fd.py:72-129; called through `detect` / `batch_detect`,
 * sfd/detect.py:19-92, from genavatar.py:61-99 and musetalk/utils/preprocessing.py:63,104) and the BiSeNet face parser (`BiSeNet.forward`,
 * musetalk/utils/face_parsing/model.py:245-262 over resnet.py:60-95; called at face_parsing/__init__.py:51) are plain Python module trees
 * in the reference.  The drop-in modules (mere-fusion_amd/avatar/) walk the same trees and emit one op per module through this builder; all
 * arithmetic runs in the library.  Buffers are padded NHWC (hi, lo) planes; `coff` arguments select channel slices (torch.cat = two
 * producers writing slices of one buffer).  Build: mf_net_create, mf_net_buffer (returns a buffer id >= 0), ops in execution order.
 * Run: mf_net_set_input -> mf_net_run (captured into a hipGraph per batch size) -> mf_net_get_output*.  (ABI version 3) */
typedef struct mf_net mf_net;
int mf_net_create(int max_batch, int precision, mf_net** out);
int mf_net_buffer(mf_net* h, int C, int H, int W, int halo);
/* nn.Conv2d (+ eval-mode BatchNorm2d folded when bn_* are given, + residual, + d->act) on the MFMA kernels; in_h / in_w of `d` are taken from
 * the input buffer; bias may be NULL; res_buf < 0: no residual. */
int mf_net_conv(mf_net* h, const mf_conv2d_desc* d, const float* weight, const float* bias, const float* bn_gamma, const float* bn_beta,
                const float* bn_mean, const float* bn_var, int in_buf, int in_coff, int out_buf, int out_coff, int res_buf, int res_coff,
                const char* name);
int mf_net_maxpool(mf_net* h, int in_buf, int out_buf, int k, int stride, int pad);                       /* F.max_pool2d / nn.MaxPool2d */
int mf_net_l2norm(mf_net* h, int in_buf, int out_buf, const float* weight, int C, float eps);            /* L2Norm, net_s3fd.py:6-19 */
int mf_net_global_avgpool(mf_net* h, int in_buf, int in_coff, int C, int out_buf);                       /* F.avg_pool2d(x, x.size()[2:]) -> 1x1 map */
/* out = x * s[b][c] + t + v[b][c]: s, v are 1x1 maps (channel attention `torch.mul(feat, atten)`, the nearest-upsampled global feature of
 * model.py:103-107), t a map of x's size (`feat_atten + feat`, `feat16_arm + feat32_up`); any of s_buf / t_buf / v_buf may be < 0 */
int mf_net_scale_add(mf_net* h, int x_buf, int x_coff, int C, int s_buf, int t_buf, int t_coff, int v_buf, int out_buf, int out_coff);
int mf_net_upsample_nearest(mf_net* h, int in_buf, int out_buf);                                         /* F.interpolate(x, size, mode='nearest') */
int mf_net_num_ops(const mf_net* h);
double mf_net_flops_per_item(const mf_net* h);                                                           /* 2 x MACs of the convolutions, one batch item */
int mf_net_set_input(mf_net* h, int buf, const float* nchw, int C, int batch, void* stream);             /* device fp32 [batch][C][H][W] */
int mf_net_run(mf_net* h, int batch, void* stream);
int mf_net_tune(mf_net* h, int batch, void* stream);                                                     /* explicit launch-configuration warm-up: see mf_wav2lip_tune */
int mf_net_get_output(mf_net* h, int buf, int coff, int C, float* nchw, int batch, void* stream);        /* device fp32 [batch][C][H][W] */
/* F.interpolate(x, (H, W), mode='bilinear', align_corners=True) of a channel slice -> device fp32 [batch][C][H][W] (model.py:257-259) */
int mf_net_get_output_bilinear(mf_net* h, int buf, int coff, int C, float* nchw, int H, int W, int batch, void* stream);
/* "max-out background label" of net_s3fd.py:123-126: device fp32 [batch][4][hw] -> [batch][2][hw] = (max(c0, c1, c2), c3) */
int mf_s3fd_maxout_bg(const float* cls4, float* cls2, int batch, int hw, void* stream);
void mf_net_destroy(mf_net* h);

/* ---- measurement seam --------------------------------------------------------------------------------------------- */
/* TFLOP/s of the convolution's own arithmetic that a kernel issuing NOTHING but matrix instructions sustains on this device (random operand bits, 8
 * waves per CU) for the instruction mix one product costs: mix 0 = bf16x3 as shipped (3 bf16 MFMAs), 1 = f16 + two FP8 block-scaled correction
 * terms, 2 = f16 + two FP6 ones.  bench.py reports it beside the nominal peak.  (ABI version 3)
 * mix 3 / 4 = mix 2 / 0 with operand VALUES drawn the way the layers' are (activations silu(z), z ~ N(0, 1); He-initialised weights) and split by the packers'
 * own rules: the ceiling is data dependent (the chip clocks to its power budget), random bits are the pessimistic end. */
int mf_probe_mfma_ceiling(int mix, float* tflops_algorithmic);

/* ---- frame transport ----------------------------------------------------------------------- */
/* Host-side plumbing of the shared-memory frame ring that replaces the pickled `res_frame_queue` items of
 * lipreal.py:136,161 / musereal.py:116,153 (mere-fusion_amd/transport.py keeps the (res_frame, idx, audio_frames) tuple
 * contract).  mf_host_register page-locks a host range (the ring's shared-memory block) once, so that
 * mf_copy_d2h_async of a batch of frames is one asynchronous DMA into the slot the consumer reads;
 * mf_stream_synchronize is the fence before the slot is published. */
int mf_host_register(void* host, size_t bytes);
int mf_host_unregister(void* host);
int mf_copy_d2h_async(const void* dev, void* host, size_t bytes, void* stream);
/* `rows` pieces of width_bytes each: dense on the device (dev_pitch apart), one ring slot apart on the host (host_pitch): a batch of
 * frames into consecutive slots as one DMA. */
int mf_copy_d2h_2d_async(const void* dev, size_t dev_pitch, void* host, size_t host_pitch, size_t width_bytes, size_t rows, void* stream);
int mf_stream_synchronize(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MEREFUSION_H */

// ER-NeRF radiance field on gfx950: tri-plane hash-grid features -> audio / eye attention -> sigma MLP -> colour MLP.
//
// Replaces `NeRFNetwork.forward` + `density` + `encode_x` (reference: ernerf/nerf_triplane/network.py:204-219, 249-308) for the
// inference call `self.forward(xyzs, dirs, enc_a, ind_code, eye)` of the render loop (renderer.py:260): per sample 36 grid
// features, SH(dir), and ten bias-free Linear layers (23 184 MACs).
//
// The reference runs ~10 tiny cuBLAS GEMMs plus cat / repeat / elementwise passes per loop iteration.  Here the samples are a
// token sequence ([M/1024] "images" of 1 x 1024 tokens) and every Linear is a 1x1 convolution on the MFMA implicit-GEMM kernel
// (mf_conv.hip) with its ReLU / sigmoid epilogue; the torch.cat calls disappear because producers write channel slices of the
// consumer's input buffer:
//     SI [72] = [ enc_x 36 | enc_a * aud_ch_att 32 | e * eye_att 1 | 0 0 0 ]        <- sigma_net input   (network.py:287-296)
//     CI [96] = [ h0 | geo_feat 64 | 0 x7 | SH 16 | ind_code 4 | 0 x4 ]             <- colour_net input  (network.py:262-269)
// sigma_net's last layer writes its 65 outputs straight into CI[0..64]; colour_net's weight columns are permuted to that layout
// (and zero on h0), so geo_feat is never copied.
#include "mf_nn.h"
#include "mf_nerf_march.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

extern "C" int mf_grid_encode_forward(const float* inputs, const float* embeddings, const int* offsets_host, float* outputs, uint32_t B,
                                      uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                      int out_blc, void* stream);

// mf_nerf_fused.hip: the whole field as one kernel (default); the per-layer GEMM path below stays for A/B (MF_NERF_FIELD=gemm)
int mf_nerf_fused_pack(const float* const w[9], int n_ind, bool has_eye, bool x3, bf16_t** dev_out);
int mf_nerf_fused_launch(const bf16_t* packed, bool x3, const float* const emb[3], const int* offsets, float log2_pls, int base_res, float bound,
                         const float* xyzs, const float* dirs, const float* enc_a, const float* ind, int n_ind, float eye, int has_eye, int M,
                         float* sigmas, float* rgbs, float* amb_aud, float* amb_eye, float* unc, hipStream_t s, const int* M_dev = nullptr,
                         float sigma_scale = 1.f, const float* eye_dev = nullptr, const float* deltas = nullptr);
int mf_nerf_tail_launch(const bf16_t* packed, bool x3, const float* const emb[3], const int* offsets, float log2_pls, int base_res, float bound, const float* enc_a,
                        const float* ind, int n_ind, float eye, int has_eye, float sigma_scale, const float* eye_dev, float* sigmas, float* rgbs, float* amb_aud,
                        float* amb_eye, float* unc, int* ctl, int N, int max_steps, int rounds_launched, float T_thresh, float dt_gamma, uint32_t cascades,
                        uint32_t grid_size, int* alive0, int* alive1, float* rays_t, const float* rays_o, const float* rays_d, const float* fars,
                        const uint8_t* bitfield, float* xyzs, float* dirs, float* deltas, float* wsum, float* depth, float* image, float* aasum, float* aesum,
                        float* unsum, hipStream_t s);

// mf_nerf_torso.hip: the whole torso branch as one fp32 kernel (default); the GEMM chain below stays for A/B (MF_TORSO=gemm)
int mf_nerf_torso_fused_weight_count();
int mf_nerf_torso_fused_launch(const float* w, const float* bias_d, const float* bias_t, const float* emb, const int* offsets_host, float log2_pls,
                               int base_res, const float* density, int G, const float* bg_coords, float shrink, float thresh, const float* bg,
                               int bg_per_ray, float bg_const, int N, float* out, float* alpha_out, float* deform, hipStream_t s);

namespace {

constexpr int TW = 1024;          // tokens per "image" of the token buffers
constexpr int ENC = 36, AUD = 32, GEO = 64, SHD = 16;
constexpr int SI_C = 72, CI_C = 96, CI_SH = 72, CI_IND = 88;

__device__ __forceinline__ uint32_t qf2bf(float f) {
    return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)f);      // round to nearest even in hardware (v_cvt_pk_bf16_f32)
}
__device__ __forceinline__ float qbf2f(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ void put(bf16_t* hi, bf16_t* lo, int64_t o, float v) {
    const uint32_t h = qf2bf(v);
    hi[o] = (bf16_t)h;
    if (lo) lo[o] = (bf16_t)qf2bf(v - qbf2f(h));
}
__device__ __forceinline__ float get(const bf16_t* hi, const bf16_t* lo, int64_t o) {
    float v = qbf2f(hi[o]);
    if (lo) v += qbf2f(lo[o]);
    return v;
}

// per sample: plane coordinates mapped to [0, 1] (grid.py:144 + network.py:204-208), SH basis of the view direction
// (shencoder.cu:50-68, degree 4) and the individual code into the colour-net input
__global__ __launch_bounds__(256) void k_nf_prep(const float* __restrict__ xyzs, const float* __restrict__ dirs, const float* __restrict__ ind,
                                                 int n_ind, float bound, int M, float* coords, bf16_t* ci_hi, bf16_t* ci_lo) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float inv = 1.f / (2.f * bound);
    const float x = (xyzs[3 * m] + bound) * inv, y = (xyzs[3 * m + 1] + bound) * inv, z = (xyzs[3 * m + 2] + bound) * inv;
    float* cxy = coords + (size_t)m * 2;
    float* cyz = coords + (size_t)M * 2 + (size_t)m * 2;
    float* cxz = coords + (size_t)M * 4 + (size_t)m * 2;
    cxy[0] = x; cxy[1] = y;
    cyz[0] = y; cyz[1] = z;
    cxz[0] = x; cxz[1] = z;
    const float dx = dirs[3 * m], dy = dirs[3 * m + 1], dz = dirs[3 * m + 2];
    const float xy = dx * dy, xz = dx * dz, yz = dy * dz, x2 = dx * dx, y2 = dy * dy, z2 = dz * dz;
    float sh[SHD];
    sh[0] = 0.28209479177387814f;
    sh[1] = -0.48860251190291987f * dy; sh[2] = 0.48860251190291987f * dz; sh[3] = -0.48860251190291987f * dx;
    sh[4] = 1.0925484305920792f * xy; sh[5] = -1.0925484305920792f * yz; sh[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    sh[7] = -1.0925484305920792f * xz; sh[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    sh[9] = 0.59004358992664352f * dy * (-3.0f * x2 + y2); sh[10] = 2.8906114426405538f * xy * dz;
    sh[11] = 0.45704579946446572f * dy * (1.0f - 5.0f * z2); sh[12] = 0.3731763325901154f * dz * (5.0f * z2 - 3.0f);
    sh[13] = 0.45704579946446572f * dx * (1.0f - 5.0f * z2); sh[14] = 1.4453057213202769f * dz * (x2 - y2);
    sh[15] = 0.59004358992664352f * dx * (-x2 + 3.0f * y2);
    // 16 SH values + 8 individual-code slots per sample as 16-byte stores (CI_SH and CI_IND are multiples of 8 channels)
    const int64_t o = (int64_t)m * CI_C;
    float v[SHD + 8];
#pragma unroll
    for (int i = 0; i < SHD; ++i) v[i] = sh[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[SHD + i] = i < n_ind ? ind[i] : 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        uint32_t hb[8], lb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { hb[e] = qf2bf(v[q * 8 + e]); lb[e] = qf2bf(v[q * 8 + e] - qbf2f(hb[e])); }
        const int64_t oo = o + (q < 2 ? CI_SH + q * 8 : CI_IND);
        *reinterpret_cast<uint4*>(ci_hi + oo) = make_uint4(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16));
        if (ci_lo) *reinterpret_cast<uint4*>(ci_lo + oo) = make_uint4(lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16), lb[4] | (lb[5] << 16), lb[6] | (lb[7] << 16));
    }
}

// fp32 grid features [3][M][12] -> SI[m][0..35]
__global__ __launch_bounds__(256) void k_nf_pack(const float* __restrict__ enc, int M, bf16_t* si_hi, bf16_t* si_lo) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * ENC) return;
    const int m = (int)(idx / ENC), c = (int)(idx - (int64_t)m * ENC);
    const int plane = c / 12, l = c - plane * 12;
    put(si_hi, si_lo, (int64_t)m * SI_C + c, enc[((size_t)plane * M + m) * 12 + l]);
}

// enc_w = enc_a * aud_ch_att, e * eye_att into the sigma-net input; the two ambient outputs (network.py:284-306).
// One lane per (sample, audio channel): 32 lanes read / write one sample's 64 contiguous bytes, the norm is a 32-lane butterfly.
__global__ __launch_bounds__(256) void k_nf_mix(const bf16_t* __restrict__ a_hi, const bf16_t* __restrict__ a_lo, const bf16_t* __restrict__ e_hi,
                                                const bf16_t* __restrict__ e_lo, const float* __restrict__ enc_a, float eye, int has_eye, int M,
                                                bf16_t* si_hi, bf16_t* si_lo, float* amb_aud, float* amb_eye) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int m = (int)(idx >> 5), c = (int)(idx & 31);
    const bool live = m < M;
    const float a = live ? get(a_hi, a_lo, (int64_t)m * AUD + c) : 0.f;
    float n2 = a * a;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) n2 += __shfl_xor(n2, o, 32);
    if (!live) return;
    const int64_t so = (int64_t)m * SI_C + ENC;
    put(si_hi, si_lo, so + c, enc_a[c] * a);
    if (c < 4) {
        const float ea = has_eye ? get(e_hi, e_lo, (int64_t)m * 8) : 0.f;
        put(si_hi, si_lo, so + AUD + c, (c == 0 && has_eye) ? eye * ea : 0.f);
        if (c == 0) { amb_aud[m] = sqrtf(n2); amb_eye[m] = ea; }
    }
}

// sigma = exp(h0) (network.py:300), colour = sigmoid(.) * 1.002 - 0.001 (network.py:272; the sigmoid ran in the GEMM epilogue),
// uncertainty = log(1 + exp(0)) in test mode (network.py:240-246, 275)
__global__ __launch_bounds__(256) void k_nf_out(const bf16_t* __restrict__ ci_hi, const bf16_t* __restrict__ ci_lo, const bf16_t* __restrict__ c_hi,
                                                const bf16_t* __restrict__ c_lo, int M, float* sigmas, float* rgbs, float* unc) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    sigmas[m] = expf(get(ci_hi, ci_lo, (int64_t)m * CI_C));
#pragma unroll
    for (int k = 0; k < 3; ++k) rgbs[3 * m + k] = get(c_hi, c_lo, (int64_t)m * 8 + k) * (1.f + 2.f * 0.001f) - 0.001f;
    if (unc) unc[m] = 0.69314718055994530942f;
}

}  // namespace

// shared by the field and the torso net: token buffers of 1 x TW "images", bias-free Linears as bound 1x1-conv plans
struct TokenNet {
    int precision = MF_PREC_BF16X3;
    int cap_batches = 0;
    std::vector<std::unique_ptr<ActBuf>> bufs;
    std::vector<std::unique_ptr<ConvPlan>> plans;
    std::vector<void*> dev;

    ~TokenNet() {
        for (auto& p : plans) mf_conv_plan_destroy(p.get());
        for (auto& b : bufs) { if (b->hi) (void)hipFree(b->hi); if (b->lo) (void)hipFree(b->lo); }
        for (void* d : dev) (void)hipFree(d);
    }
    ActBuf* seq(int C) {
        bufs.emplace_back(new ActBuf());
        ActBuf* b = bufs.back().get();
        b->C = C; b->H = 1; b->W = TW; b->halo = 0;
        return b;
    }
    int alloc() {
        for (auto& b : bufs) {
            const size_t bytes = ((size_t)cap_batches * b->per_batch() + 64) * sizeof(bf16_t);
            MF_HIP(hipMalloc(&b->hi, bytes)); MF_HIP(hipMemset(b->hi, 0, bytes));
            if (precision == MF_PREC_BF16X3) { MF_HIP(hipMalloc(&b->lo, bytes)); MF_HIP(hipMemset(b->lo, 0, bytes)); }
        }
        return MF_OK;
    }
    // Linear(cin -> cout, bias=False) (network.py:79) as a 1x1 convolution; `w` is [cout][cin_buf] with the input-channel order of
    // the buffer view it reads
    int linear(ConvPlan** out, const std::vector<float>& w, int cin, int cout, int act, ActBuf* in) {
        plans.emplace_back(new ConvPlan());
        ConvPlan* p = plans.back().get();
        mf_conv2d_desc d{};
        d.cin = cin; d.cout = cout; d.kh = d.kw = 1; d.stride_h = d.stride_w = 1; d.act = act; d.in_h = 1; d.in_w = TW;
        int rc = mf_conv_plan_create(p, d, w.data(), nullptr, nullptr, nullptr, nullptr, nullptr, precision);
        if (rc) return rc;
        if ((rc = mf_conv_bind(p, *in))) return rc;
        *out = p;
        return MF_OK;
    }
};

struct mf_nerf_field : TokenNet {
    mf_nerf_field_config cfg{};
    ActBuf *SI = nullptr, *CI = nullptr, *T1 = nullptr, *AU = nullptr, *T2 = nullptr, *EY = nullptr, *S1 = nullptr, *S2 = nullptr, *C1 = nullptr, *RGB = nullptr;
    ConvPlan *a1, *a2, *e1, *e2, *s1, *s2, *s3, *c1, *c2;
    float* emb[3] = {nullptr, nullptr, nullptr};
    float *coords = nullptr, *enc = nullptr, *d_enc_a = nullptr, *d_ind = nullptr;
    bf16_t* fused_w = nullptr;      // weight fragments of k_nerf_field_fused, or null (MF_NERF_FIELD=gemm)
};

static const mf_tensor* nf_find(const std::map<std::string, const mf_tensor*>& sd, const std::string& k, int64_t r, int64_t c) {
    auto it = sd.find(k);
    if (it == sd.end()) { mf_set_error("nerf_field_create: tensor '%s' missing", k.c_str()); return nullptr; }
    const mf_tensor* t = it->second;
    if (t->ndim != 2 || t->shape[0] != r || t->shape[1] != c) {
        mf_set_error("nerf_field_create: '%s' has shape [%lld, %lld], expected [%lld, %lld]", k.c_str(), (long long)t->shape[0],
                     (long long)(t->ndim > 1 ? t->shape[1] : 0), (long long)r, (long long)c);
        return nullptr;
    }
    return t;
}

extern "C" int mf_nerf_field_create(const mf_nerf_field_config* cfg, const mf_tensor* weights, int n_weights, int precision,
                                    int max_samples, mf_nerf_field** out) {
    MF_REQUIRE(cfg && weights && out && n_weights > 0 && max_samples > 0, "nerf_field_create: bad argument");
    MF_REQUIRE(precision == MF_PREC_BF16 || precision == MF_PREC_BF16X3, "nerf_field_create: unknown precision %d", precision);
    MF_REQUIRE(cfg->num_levels == 12 && cfg->level_dim == 1 && cfg->audio_dim == AUD && cfg->geo_feat_dim == GEO && cfg->hidden_dim == 64,
               "nerf_field_create: built for the reference's fixed field (12 levels x 1, audio 32, hidden 64, geo 64: network.py:122-143)");
    MF_REQUIRE(cfg->individual_dim >= 0 && cfg->individual_dim <= 8 && cfg->bound > 0, "nerf_field_create: individual_dim / bound");
    *out = nullptr;
    std::map<std::string, const mf_tensor*> sd;
    for (int i = 0; i < n_weights; ++i) {
        MF_REQUIRE(weights[i].name && weights[i].data, "nerf_field_create: tensor %d has no name/data", i);
        sd[weights[i].name] = &weights[i];
    }
    std::unique_ptr<mf_nerf_field> h(new mf_nerf_field());
    h->precision = precision;
    h->cfg = *cfg;
    h->cap_batches = (max_samples + TW - 1) / TW;
    const int n_emb = cfg->offsets[cfg->num_levels];
    const int eye = cfg->exp_eye ? 1 : 0, nind = cfg->individual_dim;
    const int sig_in = ENC + AUD + eye, col_in = SHD + GEO + nind;

    h->SI = h->seq(SI_C); h->CI = h->seq(CI_C); h->T1 = h->seq(64); h->AU = h->seq(AUD); h->T2 = h->seq(16); h->EY = h->seq(8);
    h->S1 = h->seq(64); h->S2 = h->seq(64); h->C1 = h->seq(64); h->RGB = h->seq(8);
    int rc = h->alloc();
    if (rc) return rc;
    const size_t cap = (size_t)h->cap_batches * TW;
    MF_HIP(hipMalloc(&h->coords, cap * 6 * sizeof(float))); h->dev.push_back(h->coords);
    MF_HIP(hipMalloc(&h->enc, cap * ENC * sizeof(float))); h->dev.push_back(h->enc);
    MF_HIP(hipMalloc(&h->d_enc_a, AUD * sizeof(float))); h->dev.push_back(h->d_enc_a);
    MF_HIP(hipMalloc(&h->d_ind, 8 * sizeof(float))); h->dev.push_back(h->d_ind);
    const char* planes[3] = {"encoder_xy.embeddings", "encoder_yz.embeddings", "encoder_xz.embeddings"};
    for (int p = 0; p < 3; ++p) {
        const mf_tensor* t = nf_find(sd, planes[p], n_emb, 1);
        if (!t) return MF_ERR_INVALID;
        MF_HIP(hipMalloc(&h->emb[p], (size_t)n_emb * sizeof(float))); h->dev.push_back(h->emb[p]);
        MF_HIP(hipMemcpy(h->emb[p], t->data, (size_t)n_emb * sizeof(float), hipMemcpyHostToDevice));
    }

    // Linear weights, re-laid onto the channel order of the buffers they read (zero columns for padding / unused channels)
    auto relay = [&](const char* name, int cout, int cin_ref, int cin_buf, const std::vector<int>& col_of, std::vector<float>& w) -> int {
        const mf_tensor* t = nf_find(sd, name, cout, cin_ref);
        if (!t) return MF_ERR_INVALID;
        const float* src = (const float*)t->data;
        w.assign((size_t)cout * cin_buf, 0.f);
        for (int o = 0; o < cout; ++o)
            for (int i = 0; i < cin_ref; ++i) w[(size_t)o * cin_buf + col_of[i]] = src[(size_t)o * cin_ref + i];
        return MF_OK;
    };
    auto iota = [](int n, int start = 0) { std::vector<int> v(n); for (int i = 0; i < n; ++i) v[i] = start + i; return v; };
    std::vector<float> w;
    if ((rc = relay("aud_ch_att_net.net.0.weight", 64, ENC, 40, iota(ENC), w)) || (rc = h->linear(&h->a1, w, 40, 64, 1, h->SI))) return rc;
    if ((rc = relay("aud_ch_att_net.net.1.weight", AUD, 64, 64, iota(64), w)) || (rc = h->linear(&h->a2, w, 64, AUD, 0, h->T1))) return rc;
    if (eye) {
        if ((rc = relay("eye_att_net.net.0.weight", 16, ENC, 40, iota(ENC), w)) || (rc = h->linear(&h->e1, w, 40, 16, 1, h->SI))) return rc;
        if ((rc = relay("eye_att_net.net.1.weight", 1, 16, 16, iota(16), w)) || (rc = h->linear(&h->e2, w, 16, 1, 2, h->T2))) return rc;
    }
    if ((rc = relay("sigma_net.net.0.weight", 64, sig_in, SI_C, iota(sig_in), w)) || (rc = h->linear(&h->s1, w, SI_C, 64, 1, h->SI))) return rc;
    if ((rc = relay("sigma_net.net.1.weight", 64, 64, 64, iota(64), w)) || (rc = h->linear(&h->s2, w, 64, 64, 1, h->S1))) return rc;
    if ((rc = relay("sigma_net.net.2.weight", 1 + GEO, 64, 64, iota(64), w)) || (rc = h->linear(&h->s3, w, 64, 1 + GEO, 0, h->S2))) return rc;
    {
        // colour-net input order in the reference: [SH 16 | geo 64 | ind] (network.py:265); in CI: geo at 1..64, SH at 72.., ind at 88..
        std::vector<int> col(col_in);
        for (int i = 0; i < SHD; ++i) col[i] = CI_SH + i;
        for (int i = 0; i < GEO; ++i) col[SHD + i] = 1 + i;
        for (int i = 0; i < nind; ++i) col[SHD + GEO + i] = CI_IND + i;
        if ((rc = relay("color_net.net.0.weight", 64, col_in, CI_C, col, w)) || (rc = h->linear(&h->c1, w, CI_C, 64, 1, h->CI))) return rc;
    }
    if ((rc = relay("color_net.net.1.weight", 3, 64, 64, iota(64), w)) || (rc = h->linear(&h->c2, w, 64, 3, 2, h->C1))) return rc;
    {
        const char* e = getenv("MF_NERF_FIELD");
        if (!(e && !strcmp(e, "gemm"))) {
            const char* names[9] = {"aud_ch_att_net.net.0.weight", "aud_ch_att_net.net.1.weight", "eye_att_net.net.0.weight", "eye_att_net.net.1.weight",
                                    "sigma_net.net.0.weight", "sigma_net.net.1.weight", "sigma_net.net.2.weight", "color_net.net.0.weight",
                                    "color_net.net.1.weight"};
            const float* w9[9];
            for (int i = 0; i < 9; ++i) {
                auto it = sd.find(names[i]);
                w9[i] = it == sd.end() ? nullptr : (const float*)it->second->data;   // shapes were checked by the relay() calls above
                MF_REQUIRE(w9[i] || ((i == 2 || i == 3) && !eye), "nerf_field_create: tensor '%s' missing", names[i]);
            }
            if ((rc = mf_nerf_fused_pack(w9, nind, eye != 0, precision == MF_PREC_BF16X3, &h->fused_w))) return rc;
            h->dev.push_back(h->fused_w);
        }
    }
    MF_HIP(hipDeviceSynchronize());
    *out = h.release();
    return MF_OK;
}

extern "C" int mf_nerf_field_forward(mf_nerf_field* h, const float* xyzs, const float* dirs, const float* enc_a, const float* ind_code,
                                     float eye, int n_samples, float* sigmas, float* rgbs, float* amb_aud, float* amb_eye,
                                     float* uncertainty, void* stream) {
    MF_REQUIRE(h && xyzs && dirs && enc_a && sigmas && rgbs && amb_aud && amb_eye, "nerf_field_forward: null argument");
    MF_REQUIRE(n_samples >= 0 && n_samples <= h->cap_batches * TW, "nerf_field_forward: %d samples exceed the capacity %d", n_samples,
               h->cap_batches * TW);
    MF_REQUIRE(h->cfg.individual_dim == 0 || ind_code, "nerf_field_forward: the field was built with an individual code");
    if (n_samples == 0) return MF_OK;
    hipStream_t s = (hipStream_t)stream;
    const int M = n_samples, nb = (M + TW - 1) / TW, gb = (M + 255) / 256;
    const bool x3 = h->precision == MF_PREC_BF16X3;
    const mf_nerf_field_config& c = h->cfg;
    if (h->fused_w)
        return mf_nerf_fused_launch(h->fused_w, x3, h->emb, c.offsets, c.log2_per_level_scale, c.base_resolution, c.bound, xyzs, dirs, enc_a, ind_code,
                                    c.individual_dim, eye, c.exp_eye, M, sigmas, rgbs, amb_aud, amb_eye, uncertainty, s);
    MF_HIP(hipMemcpyAsync(h->d_enc_a, enc_a, AUD * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (c.individual_dim) MF_HIP(hipMemcpyAsync(h->d_ind, ind_code, c.individual_dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_nf_prep, dim3(gb), dim3(256), 0, s, xyzs, dirs, h->d_ind, c.individual_dim, c.bound, M, h->coords, h->CI->hi, h->CI->lo);
    MF_HIP(hipGetLastError());
    int rc;
    for (int p = 0; p < 3; ++p)
        if ((rc = mf_grid_encode_forward(h->coords + (size_t)p * M * 2, h->emb[p], c.offsets, h->enc + (size_t)p * M * 12, M, 2, 1, 12,
                                         c.log2_per_level_scale, c.base_resolution, 0, 0, 1, stream)))
            return rc;
    hipLaunchKernelGGL(k_nf_pack, dim3((unsigned)(((int64_t)M * ENC + 255) / 256)), dim3(256), 0, s, h->enc, M, h->SI->hi, h->SI->lo);
    MF_HIP(hipGetLastError());
    auto V = [](ActBuf* b, int coff, int C) { return ActView{b, coff, C}; };
    if ((rc = mf_conv_launch(h->a1, V(h->SI, 0, 40), V(h->T1, 0, 64), ActView{}, nb, s))) return rc;
    if ((rc = mf_conv_launch(h->a2, V(h->T1, 0, 64), V(h->AU, 0, AUD), ActView{}, nb, s))) return rc;
    if (c.exp_eye) {
        if ((rc = mf_conv_launch(h->e1, V(h->SI, 0, 40), V(h->T2, 0, 16), ActView{}, nb, s))) return rc;
        if ((rc = mf_conv_launch(h->e2, V(h->T2, 0, 16), V(h->EY, 0, 1), ActView{}, nb, s))) return rc;
    }
    hipLaunchKernelGGL(k_nf_mix, dim3((unsigned)(((int64_t)M * 32 + 255) / 256)), dim3(256), 0, s, h->AU->hi, h->AU->lo, h->EY->hi, h->EY->lo, h->d_enc_a, eye, c.exp_eye, M, h->SI->hi,
                       h->SI->lo, amb_aud, amb_eye);
    MF_HIP(hipGetLastError());
    if ((rc = mf_conv_launch(h->s1, V(h->SI, 0, SI_C), V(h->S1, 0, 64), ActView{}, nb, s))) return rc;
    if ((rc = mf_conv_launch(h->s2, V(h->S1, 0, 64), V(h->S2, 0, 64), ActView{}, nb, s))) return rc;
    if ((rc = mf_conv_launch(h->s3, V(h->S2, 0, 64), V(h->CI, 0, 1 + GEO), ActView{}, nb, s))) return rc;
    if ((rc = mf_conv_launch(h->c1, V(h->CI, 0, CI_C), V(h->C1, 0, 64), ActView{}, nb, s))) return rc;
    if ((rc = mf_conv_launch(h->c2, V(h->C1, 0, 64), V(h->RGB, 0, 3), ActView{}, nb, s))) return rc;
    hipLaunchKernelGGL(k_nf_out, dim3(gb), dim3(256), 0, s, h->CI->hi, x3 ? h->CI->lo : nullptr, h->RGB->hi, x3 ? h->RGB->lo : nullptr, M, sigmas, rgbs,
                       uncertainty);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

extern "C" void mf_nerf_field_destroy(mf_nerf_field* h) { delete h; }

// =========================================================================================================================
// Torso branch: `run_torso` (renderer.py:294-352) + `forward_torso` (network.py:166-201).
//
// The reference gathers the pixels whose torso occupancy exceeds a threshold (boolean mask -> host sync), runs the deform and
// colour MLPs on them and scatters back.  Here every pixel goes through the (cheap: 8.6 kMAC) nets and the mask is applied in the
// final mix, so there is no compaction and no sync.  The per-frame constant part of the MLP inputs -- the frequency-encoded
// wrapped anchors (42) and the individual code (8) -- never becomes a channel: it is folded into a per-frame BIAS of the first
// layer of each MLP (W[:, const columns] . const), written into the plans' bias arrays before the launch.
//     TX [40] = [ freq(x, 8) 34 | 0 x6 ]                         <- deform-net input (+ bias)     (network.py:177-185)
//     TH [72] = [ tiled-grid(x + dx) 32 | freq(x, 8) 34 | 0 x6 ] <- torso-net input (+ bias)     (network.py:189-196)
// =========================================================================================================================
namespace {

constexpr int TX_C = 40, TH_C = 72, FQ = 34, TG = 32, NCONST = 50;

// x = bg_coords * shrink; FreqEncoder(2, 8) (freqencoder.cu:30-58, with the accurate sinf) into TX[0..33] and TH[32..65]
__global__ __launch_bounds__(256) void k_torso_prep(const float* __restrict__ bg_coords, float shrink, int N, float* xs, bf16_t* tx_hi, bf16_t* tx_lo,
                                                    bf16_t* th_hi, bf16_t* th_lo) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n = (int)(idx / TX_C), c = (int)(idx - (int64_t)n * TX_C);
    if (n >= N) return;
    float v = 0.f;
    if (c < FQ) {
        if (c < 2) {
            v = bg_coords[2 * n + c] * shrink;
            xs[2 * n + c] = v;
        } else {
            const int col = c / 2 - 1, d = c & 1, freq = col >> 1;
            v = sinf(scalbnf(bg_coords[2 * n + d] * shrink, freq) + (float)(col & 1) * 1.57079632679489661923f);
        }
        put(th_hi, th_lo, (int64_t)n * TH_C + TG + c, v);
    } else {
        put(th_hi, th_lo, (int64_t)n * TH_C + TG + c, 0.f);      // TH[66..71]
    }
    put(tx_hi, tx_lo, (int64_t)n * TX_C + c, v);
}

// x2 = clamp(x + dx, -1, 1) (network.py:187), mapped to [0, 1] for the tiled grid (grid.py:144, bound 1)
__global__ __launch_bounds__(256) void k_torso_warp(const float* __restrict__ xs, const bf16_t* __restrict__ d_hi, const bf16_t* __restrict__ d_lo, int N,
                                                    float* coords01, float* deform) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float dx = get(d_hi, d_lo, (int64_t)n * 8 + k);
        const float x2 = fminf(fmaxf(xs[2 * n + k] + dx, -1.f), 1.f);
        coords01[2 * n + k] = (x2 + 1.f) * 0.5f;
        if (deform) deform[2 * n + k] = dx;
    }
}

__global__ __launch_bounds__(256) void k_torso_pack(const float* __restrict__ feat, int N, bf16_t* th_hi, bf16_t* th_lo) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)N * TG) return;
    const int n = (int)(idx / TG), c = (int)(idx - (int64_t)n * TG);
    put(th_hi, th_lo, (int64_t)n * TH_C + c, feat[idx]);
}

// occupancy = grid_sample(density_grid_torso, bg_coords, align_corners=True) (renderer.py:326), mask = occupancy > thresh;
// alpha / colour = sigmoid(.) * 1.002 - 0.001 (network.py:198-199, the sigmoid ran in the GEMM epilogue);
// bg = colour * alpha + bg * (1 - alpha) (renderer.py:343)
__global__ __launch_bounds__(256) void k_torso_mix(const float* __restrict__ bg_coords, const float* __restrict__ density, int G, float thresh,
                                                   const bf16_t* __restrict__ o_hi, const bf16_t* __restrict__ o_lo, const float* __restrict__ bg,
                                                   int bg_per_ray, float bg_const, int N, float* out, float* alpha_out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float gx = (bg_coords[2 * n] + 1.f) * 0.5f * (float)(G - 1), gy = (bg_coords[2 * n + 1] + 1.f) * 0.5f * (float)(G - 1);
    const float fx = floorf(gx), fy = floorf(gy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx = gx - fx, wy = gy - fy;
    auto at = [&](int y, int x) { return (x >= 0 && x < G && y >= 0 && y < G) ? density[y * G + x] : 0.f; };   // zeros padding
    const float occ = at(y0, x0) * (1 - wx) * (1 - wy) + at(y0, x0 + 1) * wx * (1 - wy) + at(y0 + 1, x0) * (1 - wx) * wy + at(y0 + 1, x0 + 1) * wx * wy;
    const bool m = occ > thresh;
    const float a = m ? get(o_hi, o_lo, (int64_t)n * 8) * 1.002f - 0.001f : 0.f;
    if (alpha_out) alpha_out[n] = a;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float col = m ? get(o_hi, o_lo, (int64_t)n * 8 + 1 + k) * 1.002f - 0.001f : 0.f;
        const float b = bg ? (bg_per_ray ? bg[3 * n + k] : bg[k]) : bg_const;
        out[3 * n + k] = col * a + b * (1.f - a);
    }
}

}  // namespace

struct mf_nerf_torso : TokenNet {
    mf_nerf_torso_config cfg{};
    ActBuf *TX = nullptr, *TH = nullptr, *D1 = nullptr, *D2 = nullptr, *DX = nullptr, *U1 = nullptr, *U2 = nullptr, *OUT = nullptr;
    ConvPlan *d1, *d2, *d3, *t1, *t2, *t3;
    float *emb = nullptr, *density = nullptr, *xs = nullptr, *coords01 = nullptr, *feat = nullptr;
    std::vector<float> wd_const, wt_const;     // [32][50]: the constant-input columns of the two first layers
    float* wf = nullptr;                       // fp32 [out][in] tables of the six layers for the fused kernel (mf_nerf_torso.hip)
    bool fused = true;
};

extern "C" int mf_nerf_torso_create(const mf_nerf_torso_config* cfg, const mf_tensor* weights, int n_weights, int precision, int max_pixels,
                                    mf_nerf_torso** out) {
    MF_REQUIRE(cfg && weights && out && n_weights > 0 && max_pixels > 0, "nerf_torso_create: bad argument");
    MF_REQUIRE(precision == MF_PREC_BF16 || precision == MF_PREC_BF16X3, "nerf_torso_create: unknown precision %d", precision);
    MF_REQUIRE(cfg->num_levels == 16 && cfg->level_dim == 2 && cfg->individual_dim >= 0 && cfg->individual_dim <= 8 && cfg->grid_size > 1,
               "nerf_torso_create: built for the reference's torso nets (tiled grid 16 x 2, ind_dim_torso <= 8: network.py:155-159)");
    *out = nullptr;
    std::map<std::string, const mf_tensor*> sd;
    for (int i = 0; i < n_weights; ++i) {
        MF_REQUIRE(weights[i].name && weights[i].data, "nerf_torso_create: tensor %d has no name/data", i);
        sd[weights[i].name] = &weights[i];
    }
    std::unique_ptr<mf_nerf_torso> h(new mf_nerf_torso());
    h->precision = precision;
    h->cfg = *cfg;
    h->cap_batches = (max_pixels + TW - 1) / TW;
    const int nind = cfg->individual_dim, ncst = 42 + nind;                 // anchors freq(6, 3) = 42 (network.py:151) + code
    const int din = FQ + ncst, tin = TG + FQ + ncst;
    h->TX = h->seq(TX_C); h->TH = h->seq(TH_C); h->D1 = h->seq(32); h->D2 = h->seq(32); h->DX = h->seq(8); h->U1 = h->seq(32); h->U2 = h->seq(32);
    h->OUT = h->seq(8);
    int rc = h->alloc();
    if (rc) return rc;
    const size_t cap = (size_t)h->cap_batches * TW;
    auto dmalloc = [&](float** p, size_t n) -> int { MF_HIP(hipMalloc(p, n * sizeof(float))); h->dev.push_back(*p); return MF_OK; };
    if ((rc = dmalloc(&h->xs, cap * 2)) || (rc = dmalloc(&h->coords01, cap * 2)) || (rc = dmalloc(&h->feat, cap * TG))) return rc;
    const int n_emb = cfg->offsets[cfg->num_levels];
    const mf_tensor* te = nf_find(sd, "torso_encoder.embeddings", n_emb, 2);
    if (!te) return MF_ERR_INVALID;
    if ((rc = dmalloc(&h->emb, (size_t)n_emb * 2))) return rc;
    MF_HIP(hipMemcpy(h->emb, te->data, (size_t)n_emb * 2 * sizeof(float), hipMemcpyHostToDevice));
    {
        auto it = sd.find("density_grid_torso");
        MF_REQUIRE(it != sd.end(), "nerf_torso_create: tensor 'density_grid_torso' missing");
        int64_t n = 1;
        for (int d = 0; d < it->second->ndim; ++d) n *= it->second->shape[d];
        MF_REQUIRE(n == (int64_t)cfg->grid_size * cfg->grid_size, "nerf_torso_create: density_grid_torso has %lld entries, expected %d^2", (long long)n, cfg->grid_size);
        if ((rc = dmalloc(&h->density, (size_t)n))) return rc;
        MF_HIP(hipMemcpy(h->density, it->second->data, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    }
    auto first_layer = [&](const char* name, int cin_ref, int var_cols, int cin_buf, const std::vector<int>& col_of, std::vector<float>& w,
                           std::vector<float>& wconst) -> int {
        const mf_tensor* t = nf_find(sd, name, 32, cin_ref);
        if (!t) return MF_ERR_INVALID;
        const float* src = (const float*)t->data;
        w.assign((size_t)32 * cin_buf, 0.f);
        wconst.assign((size_t)32 * NCONST, 0.f);
        for (int o = 0; o < 32; ++o) {
            for (int i = 0; i < var_cols; ++i) w[(size_t)o * cin_buf + col_of[i]] = src[(size_t)o * cin_ref + i];
            for (int i = var_cols; i < cin_ref; ++i) wconst[(size_t)o * NCONST + (i - var_cols)] = src[(size_t)o * cin_ref + i];
        }
        return MF_OK;
    };
    auto plain = [&](const char* name, int cout, int cin, std::vector<float>& w) -> int {
        const mf_tensor* t = nf_find(sd, name, cout, cin);
        if (!t) return MF_ERR_INVALID;
        w.assign((const float*)t->data, (const float*)t->data + (size_t)cout * cin);
        return MF_OK;
    };
    std::vector<float> w;
    std::vector<int> col(FQ + TG);
    for (int i = 0; i < FQ; ++i) col[i] = i;
    // deform net input order: [freq(x) 34 | anchors 42 | code] (network.py:180-183)
    if ((rc = first_layer("torso_deform_net.net.0.weight", din, FQ, TX_C, col, w, h->wd_const)) || (rc = h->linear(&h->d1, w, TX_C, 32, 1, h->TX))) return rc;
    if ((rc = plain("torso_deform_net.net.1.weight", 32, 32, w)) || (rc = h->linear(&h->d2, w, 32, 32, 1, h->D1))) return rc;
    if ((rc = plain("torso_deform_net.net.2.weight", 2, 32, w)) || (rc = h->linear(&h->d3, w, 32, 2, 0, h->D2))) return rc;
    // torso net input order: [grid 32 | freq(x) 34 | anchors 42 | code] (network.py:189-194) -- the same layout as TH
    for (int i = 0; i < TG + FQ; ++i) col[i] = i;
    if ((rc = first_layer("torso_net.net.0.weight", tin, TG + FQ, TH_C, col, w, h->wt_const)) || (rc = h->linear(&h->t1, w, TH_C, 32, 1, h->TH))) return rc;
    if ((rc = plain("torso_net.net.1.weight", 32, 32, w)) || (rc = h->linear(&h->t2, w, 32, 32, 1, h->U1))) return rc;
    if ((rc = plain("torso_net.net.2.weight", 4, 32, w)) || (rc = h->linear(&h->t3, w, 32, 4, 2, h->U2))) return rc;
    {
        // the fused kernel's table: the variable-input columns of the two first layers ([freq 34], [grid 32 | freq 34]: the leading columns of
        // the reference's weight rows, network.py:180-183 / 189-194) and the four other matrices as they are
        std::vector<float> tab;
        auto take = [&](const char* name, int cout, int cin_ref, int cols) -> int {
            const mf_tensor* t = nf_find(sd, name, cout, cin_ref);
            if (!t) return MF_ERR_INVALID;
            const float* src = (const float*)t->data;
            for (int o = 0; o < cout; ++o) tab.insert(tab.end(), src + (size_t)o * cin_ref, src + (size_t)o * cin_ref + cols);
            return MF_OK;
        };
        if ((rc = take("torso_deform_net.net.0.weight", 32, din, FQ)) || (rc = take("torso_deform_net.net.1.weight", 32, 32, 32)) ||
            (rc = take("torso_deform_net.net.2.weight", 2, 32, 32)) || (rc = take("torso_net.net.0.weight", 32, tin, TG + FQ)) ||
            (rc = take("torso_net.net.1.weight", 32, 32, 32)) || (rc = take("torso_net.net.2.weight", 4, 32, 32)))
            return rc;
        MF_REQUIRE((int)tab.size() == mf_nerf_torso_fused_weight_count(), "nerf_torso_create: fused table has %zu entries", tab.size());
        if ((rc = dmalloc(&h->wf, tab.size()))) return rc;
        MF_HIP(hipMemcpy(h->wf, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
        const char* e = std::getenv("MF_TORSO");
        h->fused = !(e && std::strcmp(e, "gemm") == 0);
    }
    MF_HIP(hipDeviceSynchronize());
    *out = h.release();
    return MF_OK;
}

namespace {
struct FrameBias { float v[64]; };
__global__ void k_torso_bias(FrameBias fb, float* bd, float* bt) {
    if (threadIdx.x < 32) bd[threadIdx.x] = fb.v[threadIdx.x];
    else bt[threadIdx.x - 32] = fb.v[threadIdx.x];
}
}  // namespace

extern "C" int mf_nerf_torso_forward(mf_nerf_torso* h, const float* bg_coords, const float* frame_consts_host, const float* bg_color,
                                     int bg_per_ray, float bg_const, float density_thresh, int n_pixels, float* bg_out, float* torso_alpha,
                                     float* deform, void* stream) {
    MF_REQUIRE(h && bg_coords && frame_consts_host && bg_out, "nerf_torso_forward: null argument");
    MF_REQUIRE(n_pixels >= 0 && n_pixels <= h->cap_batches * TW, "nerf_torso_forward: %d pixels exceed the capacity %d", n_pixels, h->cap_batches * TW);
    if (n_pixels == 0) return MF_OK;
    hipStream_t s = (hipStream_t)stream;
    const int N = n_pixels, nb = (N + TW - 1) / TW, gb = (N + 255) / 256;
    const int ncst = 42 + h->cfg.individual_dim;
    // per-frame biases of the two first layers: W[:, constant columns] . [freq(wrapped anchors) | individual code]
    FrameBias fb;
    float *bd = fb.v, *bt = fb.v + 32;
    for (int o = 0; o < 32; ++o) {
        double a = 0, b = 0;
        for (int i = 0; i < ncst; ++i) {
            a += (double)h->wd_const[(size_t)o * NCONST + i] * frame_consts_host[i];
            b += (double)h->wt_const[(size_t)o * NCONST + i] * frame_consts_host[i];
        }
        bd[o] = (float)a; bt[o] = (float)b;
    }
    if (h->fused)
        return mf_nerf_torso_fused_launch(h->wf, bd, bt, h->emb, h->cfg.offsets, h->cfg.log2_per_level_scale, h->cfg.base_resolution, h->density,
                                          h->cfg.grid_size, bg_coords, h->cfg.torso_shrink, density_thresh, bg_color, bg_per_ray, bg_const, N, bg_out,
                                          torso_alpha, deform, s);
    hipLaunchKernelGGL(k_torso_bias, dim3(1), dim3(64), 0, s, fb, h->d1->bias, h->t1->bias);   // by value: no staging copy, no host sync
    hipLaunchKernelGGL(k_torso_prep, dim3((unsigned)(((int64_t)N * TX_C + 255) / 256)), dim3(256), 0, s, bg_coords, h->cfg.torso_shrink, N, h->xs,
                       h->TX->hi, h->TX->lo, h->TH->hi, h->TH->lo);
    MF_HIP(hipGetLastError());
    auto V = [](ActBuf* b, int C) { return ActView{b, 0, C}; };
    int rc;
    if ((rc = mf_conv_launch(h->d1, V(h->TX, TX_C), V(h->D1, 32), ActView{}, nb, s))) return rc;
    if ((rc = mf_conv_launch(h->d2, V(h->D1, 32), V(h->D2, 32), ActView{}, nb, s))) return rc;
    if ((rc = mf_conv_launch(h->d3, V(h->D2, 32), V(h->DX, 2), ActView{}, nb, s))) return rc;
    hipLaunchKernelGGL(k_torso_warp, dim3(gb), dim3(256), 0, s, h->xs, h->DX->hi, h->DX->lo, N, h->coords01, deform);
    MF_HIP(hipGetLastError());
    if ((rc = mf_grid_encode_forward(h->coords01, h->emb, h->cfg.offsets, h->feat, N, 2, 2, 16, h->cfg.log2_per_level_scale, h->cfg.base_resolution, 1, 0,
                                     1, stream)))
        return rc;
    hipLaunchKernelGGL(k_torso_pack, dim3((unsigned)(((int64_t)N * TG + 255) / 256)), dim3(256), 0, s, h->feat, N, h->TH->hi, h->TH->lo);
    MF_HIP(hipGetLastError());
    if ((rc = mf_conv_launch(h->t1, V(h->TH, TH_C), V(h->U1, 32), ActView{}, nb, s))) return rc;
    if ((rc = mf_conv_launch(h->t2, V(h->U1, 32), V(h->U2, 32), ActView{}, nb, s))) return rc;
    if ((rc = mf_conv_launch(h->t3, V(h->U2, 32), V(h->OUT, 4), ActView{}, nb, s))) return rc;
    hipLaunchKernelGGL(k_torso_mix, dim3(gb), dim3(256), 0, s, bg_coords, h->density, h->cfg.grid_size, density_thresh, h->OUT->hi, h->OUT->lo, bg_color,
                       bg_per_ray, bg_const, N, bg_out, torso_alpha);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

extern "C" void mf_nerf_torso_destroy(mf_nerf_torso* h) { delete h; }

// =========================================================================================================================
// Head render loop without host round trips: `run_cuda`'s inference branch (renderer.py:231-291) with the round control
// (n_alive, n_step, step) kept in HBM.  The reference compacts `rays_alive` with a boolean mask and reads its length back on the
// host every round; here k_loop_ctl / k_loop_compact (mf_nerf.hip) do both on the device, every launch has a fixed grid and
// exits when its round has nothing to do.  A frame needs ~5 of the max_steps = 16 rounds the reference allows, so only as many rounds as the frames before
// needed (+ 1) are enqueued as launches and ONE tail launch (k_loop_tail, mf_nerf_fused.hip) stands for the rest: it finds the loop ended, or runs the
// remaining rounds itself -- no round is ever dropped, and nothing waits for the host.  The whole frame is capturable as one hipGraph.
// =========================================================================================================================
extern "C" int mf_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t n_rays, float min_near, float* nears,
                                     float* fars, void* stream);
extern "C" int mf_nerf_finish(float* image, float* depth, const float* weights_sum, const float* nears, const float* fars, const float* bg_color,
                              int bg_per_ray, float bg_const, uint32_t n_rays, uint8_t* frame_u8, void* stream);
int mf_nerf_loop_init(int* ctl, int N, int max_steps, int* alive, float* rays_t, float* nears, float* weights_sum, float* depth, float* image,
                      float* amb_aud_sum, float* amb_eye_sum, float* unc_sum, hipStream_t s, const float* rays_o, const float* rays_d, const float* aabb, float min_near,
                      float* fars);
int mf_nerf_loop_round(int* ctl, int N, int max_steps, const int* alive_in, int* alive_out, float* rays_t, const float* rays_o, const float* rays_d,
                       float bound, float dt_gamma, uint32_t cascades, uint32_t grid_size, const uint8_t* bitfield, const float* fars,
                       float* xyzs, float* dirs, float* deltas, int phase, float T_thresh, const float* sigmas, const float* rgbs, const float* amb_aud,
                       const float* amb_eye, const float* unc, float* weights_sum, float* depth, float* image, float* amb_aud_sum, float* amb_eye_sum,
                       float* unc_sum, hipStream_t s);

struct mf_nerf_head {
    mf_nerf_field* field = nullptr;
    int cap = 0;
    std::vector<void*> dev;
    float *aabb, *nears, *fars, *rays_t, *xyzs, *dirs, *deltas, *sig, *rgb, *aa, *ae, *un, *wsum, *aasum, *aesum, *unsum;
    int *alive[2], *ctl;
    const float* eye_dev = nullptr;      // mf_nerf_head_set_eye: the eye feature stays on the device
    // round-count feedback: two words of pinned host memory the device posts to when a frame's loop ends ([0] rounds the frame ran, [1] the tail's error flag);
    // read without a sync, so it tells about a frame that finished some calls ago
    volatile int* fb = nullptr;
    int ctl_rounds = 0;                  // the control block holds tail tickets for this many rounds
    int hist[4] = {0, 0, 0, 0}, hist_n = 0;
    int fixed_rounds = -1;               // mf_nerf_head_set_rounds: >= 0 pins the launched-round count (a captured graph is keyed on it), -1 = follow the feedback
    ~mf_nerf_head() {
        for (void* d : dev) (void)hipFree(d);
        if (fb) (void)hipHostFree((void*)fb);
    }
};

constexpr int HEAD_MAX_ROUNDS = 1024;     // max_steps accepted by mf_nerf_head_render (one round per step at least: renderer.py:270)

// How many rounds of the next frame are enqueued as (march, field, composite) launches; the tail launch covers the rest.  The most rounds any of the last four
// observed frames ran; every round (no tail) until a first frame has reported.  MF_NERF_TAIL_AFTER=<k> fixes it (tests, A/B: 0 = the tail runs the whole
// loop, "off" = launches only, as before round 6).
static int head_plan(mf_nerf_head* h, int max_steps) {
    const char* e = getenv("MF_NERF_TAIL_AFTER");
    if (e && *e) {
        if (!strcmp(e, "off")) return max_steps;
        const int k = atoi(e);
        return k < 0 ? max_steps : (k < max_steps ? k : max_steps);
    }
    const int seen = h->fb ? h->fb[0] : 0;
    if (seen > 0) { h->hist[h->hist_n & 3] = seen; h->hist_n++; }
    if (h->hist_n == 0) return max_steps;
    int m = 0;
    for (int i = 0; i < 4; ++i) m = h->hist[i] > m ? h->hist[i] : m;
    // no margin: a frame that needs more rounds than its predecessors runs them in the tail launch (and the next frames follow); a margin of one round was three
    // launches per frame that found nothing to do (2 043 -> 2 068 frames/s)
    return m < max_steps ? m : max_steps;
}

extern "C" int mf_nerf_head_create(mf_nerf_field* field, int max_rays, mf_nerf_head** out) {
    MF_REQUIRE(field && out && max_rays > 0, "nerf_head_create: bad argument");
    MF_REQUIRE(field->fused_w, "nerf_head_create: needs the fused field kernel (unset MF_NERF_FIELD=gemm)");
    MF_REQUIRE(max_rays <= field->cap_batches * TW, "nerf_head_create: %d rays exceed the field's sample capacity %d", max_rays, field->cap_batches * TW);
    *out = nullptr;
    std::unique_ptr<mf_nerf_head> h(new mf_nerf_head());
    h->field = field; h->cap = max_rays;
    auto fm = [&](float** p, size_t n) -> int { MF_HIP(hipMalloc(p, n * sizeof(float))); h->dev.push_back(*p); return MF_OK; };
    auto im = [&](int** p, size_t n) -> int { MF_HIP(hipMalloc(p, n * sizeof(int))); h->dev.push_back(*p); return MF_OK; };
    const size_t N = (size_t)max_rays;
    int rc;
    if ((rc = fm(&h->aabb, 6)) || (rc = fm(&h->nears, N)) || (rc = fm(&h->fars, N)) || (rc = fm(&h->rays_t, N)) || (rc = fm(&h->xyzs, 3 * N)) ||
        (rc = fm(&h->dirs, 3 * N)) || (rc = fm(&h->deltas, 2 * N)) || (rc = fm(&h->sig, N)) || (rc = fm(&h->rgb, 3 * N)) || (rc = fm(&h->aa, N)) ||
        (rc = fm(&h->ae, N)) || (rc = fm(&h->un, N)) || (rc = fm(&h->wsum, N)) || (rc = fm(&h->aasum, N)) || (rc = fm(&h->aesum, N)) ||
        (rc = fm(&h->unsum, N)) || (rc = im(&h->alive[0], N)) || (rc = im(&h->alive[1], N)) ||
        (rc = im(&h->ctl, (size_t)LOOP_CTL_TAIL + (size_t)LOOP_TAIL_ARRAYS * (HEAD_MAX_ROUNDS + 1))))
        return rc;
    h->ctl_rounds = HEAD_MAX_ROUNDS;
    MF_HIP(hipMemset(h->ctl, 0, ((size_t)LOOP_CTL_TAIL + (size_t)LOOP_TAIL_ARRAYS * (HEAD_MAX_ROUNDS + 1)) * sizeof(int)));
    {
        int* fb = nullptr;
        MF_HIP(hipHostMalloc((void**)&fb, 4 * sizeof(int), hipHostMallocMapped));
        fb[0] = fb[1] = fb[2] = fb[3] = 0;
        h->fb = fb;
        void* fb_dev = nullptr;
        MF_HIP(hipHostGetDevicePointer(&fb_dev, fb, 0));
        MF_HIP(hipMemcpy(h->ctl + LOOP_CTL_FB, &fb_dev, sizeof(fb_dev), hipMemcpyHostToDevice));
    }
    const float b = field->cfg.bound;
    const float aabb[6] = {-b, -b / 2, -b, b, b / 2, b};                                    // aabb_infer, renderer.py:86-89
    MF_HIP(hipMemcpy(h->aabb, aabb, sizeof(aabb), hipMemcpyHostToDevice));
    *out = h.release();
    return MF_OK;
}

extern "C" int mf_nerf_head_render(mf_nerf_head* h, const float* rays_o, const float* rays_d, int n_rays, const uint8_t* density_bitfield, int cascades,
                                   int grid_size, float min_near, float dt_gamma, int max_steps, float T_thresh, float density_scale, const float* enc_a,
                                   const float* ind_code, float eye, const float* bg_color, int bg_per_ray, float bg_const, float* image, float* depth,
                                   float* weights_sum, uint8_t* frame_u8, void* stream) {
    MF_REQUIRE(h && rays_o && rays_d && density_bitfield && enc_a && image && depth, "nerf_head_render: null argument");
    MF_REQUIRE(n_rays > 0 && n_rays <= h->cap && max_steps > 0 && max_steps <= HEAD_MAX_ROUNDS, "nerf_head_render: n_rays=%d (capacity %d) max_steps=%d", n_rays, h->cap, max_steps);
    hipStream_t s = (hipStream_t)stream;
    mf_nerf_field* f = h->field;
    const mf_nerf_field_config& c = f->cfg;
    MF_REQUIRE(c.individual_dim == 0 || ind_code, "nerf_head_render: the field was built with an individual code");
    const int N = n_rays;
    float* ws = weights_sum ? weights_sum : h->wsum;
    int rc;
    // near / far (raymarching.cu:92-145) inside the loop's init launch
    if ((rc = mf_nerf_loop_init(h->ctl, N, max_steps, h->alive[0], h->rays_t, h->nears, ws, depth, image, h->aasum, h->aesum, h->unsum, s, rays_o, rays_d, h->aabb, min_near,
                                h->fars)))
        return rc;
    const bool x3 = f->precision == MF_PREC_BF16X3;
    // at least one sample per alive ray and round, so max_steps rounds always suffice (step += n_step >= 1, renderer.py:270)
    MF_REQUIRE(!h->fb || h->fb[1] == 0, "nerf_head_render: the tail kernel of an earlier frame gave up waiting for a round (control block error flag)");
    const int rounds = h->fixed_rounds >= 0 ? (h->fixed_rounds < max_steps ? h->fixed_rounds : max_steps) : head_plan(h, max_steps);
    const char* skip_env = getenv("MF_NERF_SKIP_EMPTY");                       // "0": evaluate every slot as the reference does (A/B)
    const bool skip_empty = !(skip_env && skip_env[0] == '0');
    for (int it = 0; it < rounds; ++it) {
        int* a_in = h->alive[it & 1];
        int* a_out = h->alive[(it + 1) & 1];
        if ((rc = mf_nerf_loop_round(h->ctl, N, max_steps, a_in, a_out, h->rays_t, rays_o, rays_d, c.bound, dt_gamma, cascades, grid_size, density_bitfield,
                                     h->fars, h->xyzs, h->dirs, h->deltas, 0, T_thresh, nullptr, nullptr, nullptr, nullptr, nullptr, ws, depth, image,
                                     h->aasum, h->aesum, h->unsum, s)))
            return rc;
        if ((rc = mf_nerf_fused_launch(f->fused_w, x3, f->emb, c.offsets, c.log2_per_level_scale, c.base_resolution, c.bound, h->xyzs, h->dirs, enc_a, ind_code,
                                       c.individual_dim, eye, c.exp_eye, N, h->sig, h->rgb, h->aa, h->ae, h->un, s, h->ctl + 3, density_scale, h->eye_dev,
                                       skip_empty ? h->deltas : nullptr)))
            return rc;
        if ((rc = mf_nerf_loop_round(h->ctl, N, max_steps, a_in, a_out, h->rays_t, rays_o, rays_d, c.bound, dt_gamma, cascades, grid_size, density_bitfield,
                                     h->fars, h->xyzs, h->dirs, h->deltas, 1, T_thresh, h->sig, h->rgb, h->aa, h->ae, h->un, ws, depth, image,
                                     h->aasum, h->aesum, h->unsum, s)))
            return rc;
    }
    // at least one sample per alive ray and round, so max_steps rounds always suffice (step += n_step >= 1, renderer.py:270): with every round enqueued there is no tail
    if (rounds < max_steps &&
        (rc = mf_nerf_tail_launch(f->fused_w, x3, f->emb, c.offsets, c.log2_per_level_scale, c.base_resolution, c.bound, enc_a, ind_code, c.individual_dim, eye, c.exp_eye,
                                  density_scale, h->eye_dev, h->sig, h->rgb, h->aa, h->ae, h->un, h->ctl, N, max_steps, rounds, T_thresh, dt_gamma, cascades, grid_size,
                                  h->alive[0], h->alive[1], h->rays_t, rays_o, rays_d, h->fars, density_bitfield, h->xyzs, h->dirs, h->deltas, ws, depth, image,
                                  h->aasum, h->aesum, h->unsum, s)))
        return rc;
    if (bg_per_ray < 0) return MF_OK;                                                       // finished later by mf_nerf_head_finish
    return mf_nerf_finish(image, depth, ws, h->nears, h->fars, bg_color, bg_per_ray, bg_const, N, frame_u8, stream);
}

extern "C" int mf_nerf_head_finish(mf_nerf_head* h, int n_rays, const float* bg_color, int bg_per_ray, float bg_const, float* image, float* depth,
                                   const float* weights_sum, uint8_t* frame_u8, void* stream) {
    MF_REQUIRE(h && image && depth && n_rays > 0 && n_rays <= h->cap && bg_per_ray >= 0, "nerf_head_finish: bad argument");
    return mf_nerf_finish(image, depth, weights_sum ? weights_sum : h->wsum, h->nears, h->fars, bg_color, bg_per_ray, bg_const, n_rays, frame_u8, stream);
}

extern "C" int mf_nerf_head_set_eye(mf_nerf_head* h, const float* eye_dev) {
    MF_REQUIRE(h, "nerf_head_set_eye: null handle");
    h->eye_dev = eye_dev;
    return MF_OK;
}

extern "C" int mf_nerf_head_plan_rounds(mf_nerf_head* h, int max_steps, int* rounds) {
    MF_REQUIRE(h && rounds && max_steps > 0 && max_steps <= HEAD_MAX_ROUNDS, "nerf_head_plan_rounds: bad argument");
    *rounds = head_plan(h, max_steps);
    return MF_OK;
}

extern "C" int mf_nerf_head_set_rounds(mf_nerf_head* h, int rounds) {
    MF_REQUIRE(h && rounds >= -1 && rounds <= HEAD_MAX_ROUNDS, "nerf_head_set_rounds: bad argument");
    h->fixed_rounds = rounds;
    return MF_OK;
}

extern "C" int mf_nerf_head_last_rounds(mf_nerf_head* h, int* rounds, int* error_flag) {
    MF_REQUIRE(h && h->fb, "nerf_head_last_rounds: null handle");
    if (rounds) *rounds = h->fb[0];
    if (error_flag) *error_flag = h->fb[1];
    return MF_OK;
}

extern "C" int mf_nerf_head_ctl_snapshot(mf_nerf_head* h, int* out, int n_ints) {
    MF_REQUIRE(h && out && n_ints > 0 && (size_t)n_ints <= (size_t)LOOP_CTL_TAIL + (size_t)LOOP_TAIL_ARRAYS * (HEAD_MAX_ROUNDS + 1), "nerf_head_ctl_snapshot: bad argument");
    MF_HIP(hipMemcpy(out, h->ctl, (size_t)n_ints * sizeof(int), hipMemcpyDeviceToHost));
    return MF_OK;
}

extern "C" int mf_nerf_head_sums(mf_nerf_head* h, int n_rays, float* ambient_aud, float* ambient_eye, float* uncertainty, void* stream) {
    MF_REQUIRE(h && n_rays > 0 && n_rays <= h->cap, "nerf_head_sums: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t bytes = (size_t)n_rays * sizeof(float);
    if (ambient_aud) MF_HIP(hipMemcpyAsync(ambient_aud, h->aasum, bytes, hipMemcpyDeviceToDevice, s));
    if (ambient_eye) MF_HIP(hipMemcpyAsync(ambient_eye, h->aesum, bytes, hipMemcpyDeviceToDevice, s));
    if (uncertainty) MF_HIP(hipMemcpyAsync(uncertainty, h->unsum, bytes, hipMemcpyDeviceToDevice, s));
    return MF_OK;
}

extern "C" void mf_nerf_head_destroy(mf_nerf_head* h) { delete h; }

// ER-NeRF's audio front-end on gfx950 (on the per-chunk path of BASELINE configs[4]): the wav2vec2 / HuBERT CTC
// network that `NerfASR.__frame_to_text` runs on every (l + m + r) x 20 ms window (nerfasr.py:128-143) --
//   processor(frame)            -> zero-mean / unit-variance waveform            (transformers Wav2Vec2FeatureExtractor, do_normalize)
//   model(input_values).logits  -> 7 x [Conv1d -> LayerNorm -> GELU], LayerNorm + Linear, grouped positional Conv1d (k = 128, weight-normed)
//                                  + GELU, N transformer layers (pre-LN "stable" or post-LN), final LayerNorm, lm_head
// (HubertModel: the same stack, `last_hidden_state` instead of logits, nerfasr.py:133-134).  The network is a third-party dependency of the
// reference (transformers, requirements.txt); names below are its state-dict keys so a real checkpoint loads unchanged.
//
// Same construction as the Whisper stage: a token sequence is an ActBuf{C, H = 1, W = T}; Conv1d(k, s) is a (1 x k) strided convolution and every
// Linear a 1 x 1 convolution on the MFMA implicit-GEMM kernel (bias / GELU / residual epilogues), LayerNorm(+GELU) one wave per token, attention
// the fused kernel of mf_attn.hip (head dim 64).  The positional convolution (16 groups of 64 channels, 128 taps, padding 64, last output
// dropped) runs as 16 convolutions over channel slices of ONE hidden-state buffer with a 64-token zero halo and a 129th zero tap, which makes the
// padding symmetric and the output length T; its GELU and the `hidden + pos` add are the conv epilogue (residual after activation).
#include "mf_nn.h"
#include "mf_aux.h"
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

// one workgroup per window: (x - mean) / sqrt(var + 1e-7) (numpy's population variance), written to channel 0 of the conv0 input
__global__ __launch_bounds__(1024) void k_w2v_normalize(const float* __restrict__ wav, int n, int normalize, bf16_t* hi, bf16_t* lo, int C,
                                                        int64_t per_batch) {
    __shared__ double s_a[1024], s_b[1024];
    const float* x = wav + (int64_t)blockIdx.x * n;
    const int tid = threadIdx.x;
    double mean = 0.0, rstd = 1.0;
    if (normalize) {
        double a = 0.0;
        for (int i = tid; i < n; i += 1024) a += (double)x[i];
        s_a[tid] = a;
        __syncthreads();
        for (int k = 512; k > 0; k >>= 1) { if (tid < k) s_a[tid] += s_a[tid + k]; __syncthreads(); }
        mean = s_a[0] / n;
        double q = 0.0;
        for (int i = tid; i < n; i += 1024) { const double d = (double)x[i] - mean; q += d * d; }
        s_b[tid] = q;
        __syncthreads();
        for (int k = 512; k > 0; k >>= 1) { if (tid < k) s_b[tid] += s_b[tid + k]; __syncthreads(); }
        rstd = 1.0 / sqrt(s_b[0] / n + 1e-7);
    }
    hi += (int64_t)blockIdx.x * per_batch;
    if (lo) lo += (int64_t)blockIdx.x * per_batch;
    for (int i = tid; i < n; i += 1024) {
        const float v = (float)(((double)x[i] - mean) * rstd);
        unsigned u = __float_as_uint(v);
        u += 0x7fffu + ((u >> 16) & 1u);
        const unsigned h = u >> 16;
        hi[(int64_t)i * C] = (bf16_t)h;
        if (lo) {
            unsigned l = __float_as_uint(v - __uint_as_float(h << 16));
            l += 0x7fffu + ((l >> 16) & 1u);
            lo[(int64_t)i * C] = (bf16_t)(l >> 16);
        }
    }
}

ActView V(ActBuf* b) { return ActView{b, 0, b->C}; }

}  // namespace

struct mf_wav2vec2 {
    mf_wav2vec2_config cfg{};
    int precision = MF_PREC_BF16X3;
    int n = 0, cap = 1, T = 0;
    std::vector<int> frames;                       // sequence length after each conv layer
    std::vector<std::unique_ptr<ActBuf>> bufs;
    std::vector<std::unique_ptr<ConvPlan>> plans;
    std::vector<float*> dev_f32;
    ActBuf* wav_in = nullptr;
    std::vector<ActBuf*> conv_out, conv_ln;
    std::vector<ConvPlan*> convs;
    std::vector<std::pair<float*, float*>> conv_ln_w;
    float *fp_g = nullptr, *fp_b = nullptr, *enc_g = nullptr, *enc_b = nullptr;
    ActBuf *fp_ln = nullptr, *hid = nullptr, *xa = nullptr, *xb = nullptr, *ln = nullptr, *qkv = nullptr, *ao = nullptr, *h1 = nullptr, *logit = nullptr;
    ConvPlan *proj = nullptr, *head = nullptr;
    int vocab_pad = 0;                             // lm_head rows rounded up to a channel quad
    std::vector<ConvPlan*> pos;
    struct Layer { ConvPlan *qkv, *out, *fc1, *fc2; float *g1, *b1, *g2, *b2; };
    std::vector<Layer> layers;

    ~mf_wav2vec2() {
        for (auto& p : plans) mf_conv_plan_destroy(p.get());
        for (auto& b : bufs) { if (b->hi) (void)hipFree(b->hi); if (b->lo) (void)hipFree(b->lo); }
        for (float* f : dev_f32) (void)hipFree(f);
        for (auto& g : graphs) if (g.second) (void)hipGraphExecDestroy(g.second);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (ev_in) (void)hipEventDestroy(ev_in);
        if (ev_out) (void)hipEventDestroy(ev_out);
    }
    ActBuf* seq(int C, int T_, int halo = 0) {
        bufs.emplace_back(new ActBuf());
        ActBuf* b = bufs.back().get();
        b->C = (C + 7) / 8 * 8; b->H = 1; b->W = T_; b->halo = halo;
        return b;
    }
    int alloc() {
        for (auto& b : bufs) {
            const size_t bytes = ((size_t)cap * b->per_batch() + 64) * sizeof(bf16_t);
            MF_HIP(hipMalloc(&b->hi, bytes)); MF_HIP(hipMemset(b->hi, 0, bytes));
            if (precision == MF_PREC_BF16X3) { MF_HIP(hipMalloc(&b->lo, bytes)); MF_HIP(hipMemset(b->lo, 0, bytes)); }
        }
        return MF_OK;
    }
    ConvPlan* new_plan() { plans.emplace_back(new ConvPlan()); return plans.back().get(); }
    int upload(const float* host, size_t cnt, float** dev) {
        MF_HIP(hipMalloc(dev, cnt * sizeof(float)));
        MF_HIP(hipMemcpy(*dev, host, cnt * sizeof(float), hipMemcpyHostToDevice));
        dev_f32.push_back(*dev);
        return MF_OK;
    }
    int forward(const float* wav, int S, float* out, hipStream_t s);
    int body(int S, hipStream_t s, ActBuf** last);
    // the ~240 launches between the input normalisation and the output copy replay as one hipGraph per window count (first call eager: split-K
    // workspaces are sized there; second call captures)
    std::map<int, hipGraphExec_t> graphs;
    std::map<int, ActBuf*> graph_last;
    hipStream_t cap_stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    bool use_graph = true;
};

namespace {

const mf_tensor* get(const std::map<std::string, const mf_tensor*>& sd, const std::string& k, int64_t numel, bool required = true) {
    auto it = sd.find(k);
    if (it == sd.end()) { if (required) mf_set_error("wav2vec2: state dict has no tensor '%s'", k.c_str()); return nullptr; }
    int64_t cnt = 1;
    for (int i = 0; i < it->second->ndim; ++i) cnt *= it->second->shape[i];
    if (cnt != numel) { mf_set_error("wav2vec2: tensor '%s' has %lld elements, expected %lld", k.c_str(), (long long)cnt, (long long)numel); return nullptr; }
    return it->second;
}

int conv1d_plan(ConvPlan* p, const float* w, const float* b, int cin, int cout, int k, int stride, int pad, int T_in, int act, int residual, int precision) {
    mf_conv2d_desc d{};
    d.cin = cin; d.cout = cout; d.kh = 1; d.kw = k; d.stride_h = 1; d.stride_w = stride; d.pad_h = 0; d.pad_w = pad;
    d.act = act; d.residual = residual; d.in_h = 1; d.in_w = T_in;
    return mf_conv_plan_create(p, d, w, b, nullptr, nullptr, nullptr, nullptr, precision);
}

}  // namespace

int mf_wav2vec2::forward(const float* wav, int S, float* out, hipStream_t s) {
    MF_REQUIRE(S >= 1 && S <= cap, "wav2vec2: %d windows exceed the handle's capacity %d", S, cap);
    int rc;
    hipLaunchKernelGGL(k_w2v_normalize, dim3(S), dim3(1024), 0, s, wav, n, cfg.do_normalize, wav_in->hi + mf_interior(*wav_in),
                       wav_in->lo ? wav_in->lo + mf_interior(*wav_in) : nullptr, wav_in->C, wav_in->per_batch());
    MF_HIP(hipGetLastError());
    ActBuf* cur = nullptr;
    if (!use_graph) {
        if ((rc = body(S, s, &cur))) return rc;
    } else {
        auto it = graphs.find(S);
        if (it == graphs.end()) {
            graphs.emplace(S, nullptr);
            if ((rc = body(S, s, &cur))) return rc;
            graph_last[S] = cur;
        } else {
            if (!it->second) {
                hipGraph_t graph = nullptr;
                MF_HIP(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
                rc = body(S, cap_stream, &cur);
                hipError_t e = hipStreamEndCapture(cap_stream, &graph);
                if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
                MF_HIP(e);
                MF_HIP(hipGraphInstantiate(&it->second, graph, nullptr, nullptr, 0));
                (void)hipGraphDestroy(graph);
            }
            cur = graph_last[S];
            MF_HIP(hipEventRecord(ev_in, s));
            MF_HIP(hipStreamWaitEvent(cap_stream, ev_in, 0));
            MF_HIP(hipGraphLaunch(it->second, cap_stream));
            MF_HIP(hipEventRecord(ev_out, cap_stream));
            MF_HIP(hipStreamWaitEvent(s, ev_out, 0));
        }
    }
    if (cfg.out_hidden) return mf_rows_to_f32(V(cur), out, S, s);
    return mf_rows_to_f32(ActView{logit, 0, cfg.vocab}, out, S, s);
}

int mf_wav2vec2::body(int S, hipStream_t s, ActBuf** last) {
    int rc;
    // feature extractor: conv -> LayerNorm over channels -> GELU (Wav2Vec2LayerNormConvLayer)
    ActBuf* x = wav_in;
    for (int i = 0; i < cfg.n_conv; ++i) {
        if ((rc = mf_conv_launch(convs[i], ActView{x, 0, i == 0 ? 8 : cfg.conv_dim[i - 1]}, V(conv_out[i]), ActView{}, S, s))) return rc;
        if ((rc = mf_layernorm(V(conv_out[i]), V(conv_ln[i]), conv_ln_w[i].first, conv_ln_w[i].second, cfg.layer_norm_eps, S, s, 0, 3))) return rc;
        x = conv_ln[i];
    }
    // feature projection: LayerNorm -> Linear (Wav2Vec2FeatureProjection)
    if ((rc = mf_layernorm(V(x), V(fp_ln), fp_g, fp_b, cfg.layer_norm_eps, S, s))) return rc;
    if ((rc = mf_conv_launch(proj, V(fp_ln), V(hid), ActView{}, S, s))) return rc;
    // hidden + gelu(pos_conv(hidden))  (Wav2Vec2PositionalConvEmbedding + SamePad)
    const int gc = cfg.hidden / cfg.pos_groups;
    for (int g = 0; g < cfg.pos_groups; ++g)
        if ((rc = mf_conv_launch(pos[g], ActView{hid, g * gc, gc}, ActView{xa, g * gc, gc}, ActView{hid, g * gc, gc}, S, s))) return rc;
    ActBuf *cur = xa, *oth = xb;
    if (!cfg.stable_ln) {       // post-LN encoder: LayerNorm right after the positional add (Wav2Vec2Encoder.forward)
        if ((rc = mf_layernorm(V(cur), V(oth), enc_g, enc_b, cfg.layer_norm_eps, S, s))) return rc;
        std::swap(cur, oth);
    }
    for (auto& L : layers) {
        if (cfg.stable_ln) {
            // h = x + attn(LN(x)); x = h + ff(LN(h))   (Wav2Vec2EncoderLayerStableLayerNorm)
            if ((rc = mf_layernorm(V(cur), V(ln), L.g1, L.b1, cfg.layer_norm_eps, S, s))) return rc;
            if ((rc = mf_conv_launch(L.qkv, V(ln), V(qkv), ActView{}, S, s))) return rc;
            if ((rc = mf_attention(ActView{qkv, 0, cfg.hidden}, ActView{qkv, cfg.hidden, cfg.hidden}, ActView{qkv, 2 * cfg.hidden, cfg.hidden}, V(ao), cfg.n_head, S,
                                   precision, s))) return rc;
            if ((rc = mf_conv_launch(L.out, V(ao), V(oth), V(cur), S, s))) return rc;
            if ((rc = mf_layernorm(V(oth), V(ln), L.g2, L.b2, cfg.layer_norm_eps, S, s))) return rc;
            if ((rc = mf_conv_launch(L.fc1, V(ln), V(h1), ActView{}, S, s))) return rc;
            if ((rc = mf_conv_launch(L.fc2, V(h1), V(cur), V(oth), S, s))) return rc;
        } else {
            // h = LN(x + attn(x)); x = LN(h + ff(h))   (Wav2Vec2EncoderLayer)
            if ((rc = mf_conv_launch(L.qkv, V(cur), V(qkv), ActView{}, S, s))) return rc;
            if ((rc = mf_attention(ActView{qkv, 0, cfg.hidden}, ActView{qkv, cfg.hidden, cfg.hidden}, ActView{qkv, 2 * cfg.hidden, cfg.hidden}, V(ao), cfg.n_head, S,
                                   precision, s))) return rc;
            if ((rc = mf_conv_launch(L.out, V(ao), V(oth), V(cur), S, s))) return rc;
            if ((rc = mf_layernorm(V(oth), V(cur), L.g1, L.b1, cfg.layer_norm_eps, S, s))) return rc;
            if ((rc = mf_conv_launch(L.fc1, V(cur), V(h1), ActView{}, S, s))) return rc;
            if ((rc = mf_conv_launch(L.fc2, V(h1), V(oth), V(cur), S, s))) return rc;
            if ((rc = mf_layernorm(V(oth), V(cur), L.g2, L.b2, cfg.layer_norm_eps, S, s))) return rc;
        }
    }
    if (cfg.stable_ln) {
        if ((rc = mf_layernorm(V(cur), V(oth), enc_g, enc_b, cfg.layer_norm_eps, S, s))) return rc;
        std::swap(cur, oth);
    }
    *last = cur;
    if (cfg.out_hidden) return MF_OK;
    return mf_conv_launch(head, V(cur), ActView{logit, 0, vocab_pad}, ActView{}, S, s);
}

extern "C" int mf_wav2vec2_create(const mf_wav2vec2_config* cfg, const mf_tensor* weights, int n_weights, int n_samples, int max_windows, int precision,
                                  mf_wav2vec2** out) {
    MF_REQUIRE(cfg && weights && out && n_weights > 0, "wav2vec2_create: bad argument");
    MF_REQUIRE(precision == MF_PREC_BF16 || precision == MF_PREC_BF16X3, "wav2vec2_create: unknown precision %d", precision);
    *out = nullptr;
    const mf_wav2vec2_config& c = *cfg;
    MF_REQUIRE(c.n_conv >= 1 && c.n_conv <= 8 && c.hidden > 0 && c.n_layer >= 0 && c.n_head > 0 && c.ffn > 0, "wav2vec2_create: bad config");
    MF_REQUIRE(c.feat_norm_layer == 1, "wav2vec2_create: only feat_extract_norm = \"layer\" is built (the xlsr-53 / large checkpoints nerfasr.py names); "
                                       "\"group\" (GroupNorm on the first conv layer of the base models) is not");
    MF_REQUIRE(c.hidden % c.n_head == 0 && mf_attention_supported(c.hidden / c.n_head), "wav2vec2_create: head dim %d has no fused attention kernel",
               c.hidden / c.n_head);
    MF_REQUIRE(c.pos_groups > 0 && c.hidden % c.pos_groups == 0 && (c.hidden / c.pos_groups) % 8 == 0 && c.pos_k >= 2 && c.pos_k % 2 == 0,
               "wav2vec2_create: positional conv needs an even kernel and channel groups that are multiples of 8");
    MF_REQUIRE(c.out_hidden || c.vocab > 0, "wav2vec2_create: vocab size missing");
    MF_REQUIRE(max_windows >= 1 && max_windows <= 64, "wav2vec2_create: 1..64 windows per call");
    std::map<std::string, const mf_tensor*> sd;
    for (int i = 0; i < n_weights; ++i) {
        MF_REQUIRE(weights[i].name && weights[i].data, "wav2vec2_create: tensor %d has no name/data", i);
        std::string k = weights[i].name;
        for (const char* pre : {"wav2vec2.", "hubert."})
            if (k.rfind(pre, 0) == 0) k = k.substr(strlen(pre));
        sd[k] = &weights[i];
    }
    std::unique_ptr<mf_wav2vec2> h(new mf_wav2vec2());
    h->cfg = c; h->precision = precision; h->n = n_samples; h->cap = max_windows;
    MF_HIP(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    MF_HIP(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
    MF_HIP(hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
    { const char* e = getenv("MF_NO_GRAPH"); h->use_graph = !(e && atoi(e) != 0); }
    int L = n_samples;
    for (int i = 0; i < c.n_conv; ++i) {
        MF_REQUIRE(c.conv_kernel[i] >= 1 && c.conv_stride[i] >= 1 && c.conv_dim[i] % 8 == 0, "wav2vec2_create: bad conv layer %d", i);
        MF_REQUIRE(L >= c.conv_kernel[i], "wav2vec2_create: %d samples are too few for the feature extractor", n_samples);
        L = (L - c.conv_kernel[i]) / c.conv_stride[i] + 1;
        h->frames.push_back(L);
    }
    const int T = h->T = L, C = c.hidden, Cf = c.conv_dim[c.n_conv - 1], half = c.pos_k / 2;
    h->wav_in = h->seq(8, n_samples);
    for (int i = 0; i < c.n_conv; ++i) { h->conv_out.push_back(h->seq(c.conv_dim[i], h->frames[i])); h->conv_ln.push_back(h->seq(c.conv_dim[i], h->frames[i])); }
    h->fp_ln = h->seq(Cf, T); h->hid = h->seq(C, T, half);
    h->xa = h->seq(C, T); h->xb = h->seq(C, T); h->ln = h->seq(C, T); h->qkv = h->seq(3 * C, T); h->ao = h->seq(C, T); h->h1 = h->seq(c.ffn, T);
    if (!c.out_hidden) h->logit = h->seq(c.vocab, T);
    int rc = h->alloc();
    if (rc) return rc;

    auto ln_pair = [&](const std::string& p, int ch, float** g, float** b) -> int {
        const mf_tensor *gw = get(sd, p + ".weight", ch), *gb = get(sd, p + ".bias", ch);
        if (!gw || !gb) return MF_ERR_INVALID;
        int r = h->upload(gw->data, ch, g);
        return r ? r : h->upload(gb->data, ch, b);
    };
    // ---- feature extractor --------------------------------------------------------------------------------------------------------
    for (int i = 0; i < c.n_conv; ++i) {
        const std::string p = "feature_extractor.conv_layers." + std::to_string(i);
        const int cin = i == 0 ? 1 : c.conv_dim[i - 1], cout = c.conv_dim[i], k = c.conv_kernel[i];
        const mf_tensor* w = get(sd, p + ".conv.weight", (int64_t)cout * cin * k);
        const mf_tensor* b = c.conv_bias ? get(sd, p + ".conv.bias", cout) : nullptr;
        if (!w || (c.conv_bias && !b)) return MF_ERR_INVALID;
        ConvPlan* pl = h->new_plan();
        if ((rc = conv1d_plan(pl, w->data, b ? b->data : nullptr, cin, cout, k, c.conv_stride[i], 0, i == 0 ? n_samples : h->frames[i - 1], 0, 0, precision))) return rc;
        if ((rc = mf_conv_bind(pl, i == 0 ? *h->wav_in : *h->conv_ln[i - 1]))) return rc;
        h->convs.push_back(pl);
        float *g = nullptr, *bb = nullptr;
        if ((rc = ln_pair(p + ".layer_norm", cout, &g, &bb))) return rc;
        h->conv_ln_w.push_back({g, bb});
    }
    // ---- feature projection ---------------------------------------------------------------------------------------------------------
    if ((rc = ln_pair("feature_projection.layer_norm", Cf, &h->fp_g, &h->fp_b))) return rc;
    {
        const mf_tensor *w = get(sd, "feature_projection.projection.weight", (int64_t)C * Cf), *b = get(sd, "feature_projection.projection.bias", C);
        if (!w || !b) return MF_ERR_INVALID;
        h->proj = h->new_plan();
        if ((rc = conv1d_plan(h->proj, w->data, b->data, Cf, C, 1, 1, 0, T, 0, 0, precision))) return rc;
        if ((rc = mf_conv_bind(h->proj, *h->fp_ln))) return rc;
    }
    // ---- positional convolution: weight_norm(dim = 2) folded on the host, one plan per channel group, a zero tap appended -----------
    {
        const int gc = C / c.pos_groups, K = c.pos_k;
        const std::string p = "encoder.pos_conv_embed.conv.";
        std::vector<float> wfull((size_t)C * gc * K);
        const mf_tensor* plain = get(sd, p + "weight", (int64_t)C * gc * K, false);
        if (plain) {
            std::copy(plain->data, plain->data + wfull.size(), wfull.begin());
        } else {
            const mf_tensor* g = get(sd, p + "parametrizations.weight.original0", K, false);
            const mf_tensor* v = get(sd, p + "parametrizations.weight.original1", (int64_t)C * gc * K, false);
            if (!g) g = get(sd, p + "weight_g", K, false);
            if (!v) v = get(sd, p + "weight_v", (int64_t)C * gc * K, false);
            MF_REQUIRE(g && v, "wav2vec2: positional conv weight missing (weight, weight_g / weight_v or parametrizations.weight.original0 / 1)");
            std::vector<double> nrm(K, 0.0);
            for (int64_t i = 0; i < (int64_t)C * gc; ++i)
                for (int k = 0; k < K; ++k) { const double x = v->data[i * K + k]; nrm[k] += x * x; }
            for (int k = 0; k < K; ++k) nrm[k] = std::sqrt(nrm[k]);
            for (int64_t i = 0; i < (int64_t)C * gc; ++i)
                for (int k = 0; k < K; ++k) wfull[i * K + k] = (float)((double)g->data[k] * (double)v->data[i * K + k] / nrm[k]);
        }
        const mf_tensor* b = get(sd, p + "bias", C);
        if (!b) return MF_ERR_INVALID;
        std::vector<float> wg((size_t)gc * gc * (K + 1));
        for (int g = 0; g < c.pos_groups; ++g) {
            for (int o = 0; o < gc; ++o)
                for (int i = 0; i < gc; ++i) {
                    const float* src = wfull.data() + ((size_t)(g * gc + o) * gc + i) * K;
                    float* dst = wg.data() + ((size_t)o * gc + i) * (K + 1);
                    std::copy(src, src + K, dst);
                    dst[K] = 0.f;                                  // tap K sits at offset +K/2: the output SamePad drops never sees it
                }
            ConvPlan* pl = h->new_plan();
            // GELU in the epilogue, then + hidden (residual after the activation)
            if ((rc = conv1d_plan(pl, wg.data(), b->data + g * gc, gc, gc, K + 1, 1, half, T, 3, 2, precision))) return rc;
            if ((rc = mf_conv_bind(pl, *h->hid))) return rc;
            h->pos.push_back(pl);
        }
    }
    // ---- encoder ----------------------------------------------------------------------------------------------------------------------
    if ((rc = ln_pair("encoder.layer_norm", C, &h->enc_g, &h->enc_b))) return rc;
    for (int l = 0; l < c.n_layer; ++l) {
        const std::string p = "encoder.layers." + std::to_string(l) + ".";
        const mf_tensor *qw = get(sd, p + "attention.q_proj.weight", (int64_t)C * C), *qb = get(sd, p + "attention.q_proj.bias", C);
        const mf_tensor *kw = get(sd, p + "attention.k_proj.weight", (int64_t)C * C), *kb = get(sd, p + "attention.k_proj.bias", C);
        const mf_tensor *vw = get(sd, p + "attention.v_proj.weight", (int64_t)C * C), *vb = get(sd, p + "attention.v_proj.bias", C);
        const mf_tensor *ow = get(sd, p + "attention.out_proj.weight", (int64_t)C * C), *ob = get(sd, p + "attention.out_proj.bias", C);
        const mf_tensor *f1w = get(sd, p + "feed_forward.intermediate_dense.weight", (int64_t)c.ffn * C), *f1b = get(sd, p + "feed_forward.intermediate_dense.bias", c.ffn);
        const mf_tensor *f2w = get(sd, p + "feed_forward.output_dense.weight", (int64_t)c.ffn * C), *f2b = get(sd, p + "feed_forward.output_dense.bias", C);
        if (!qw || !qb || !kw || !kb || !vw || !vb || !ow || !ob || !f1w || !f1b || !f2w || !f2b) return MF_ERR_INVALID;
        std::vector<float> w((size_t)3 * C * C), b((size_t)3 * C);
        std::copy(qw->data, qw->data + (size_t)C * C, w.begin());
        std::copy(kw->data, kw->data + (size_t)C * C, w.begin() + (size_t)C * C);
        std::copy(vw->data, vw->data + (size_t)C * C, w.begin() + (size_t)2 * C * C);
        std::copy(qb->data, qb->data + C, b.begin());
        std::copy(kb->data, kb->data + C, b.begin() + C);
        std::copy(vb->data, vb->data + C, b.begin() + 2 * C);
        mf_wav2vec2::Layer Ly{};
        Ly.qkv = h->new_plan(); Ly.out = h->new_plan(); Ly.fc1 = h->new_plan(); Ly.fc2 = h->new_plan();
        if ((rc = conv1d_plan(Ly.qkv, w.data(), b.data(), C, 3 * C, 1, 1, 0, T, 0, 0, precision))) return rc;
        if ((rc = conv1d_plan(Ly.out, ow->data, ob->data, C, C, 1, 1, 0, T, 0, 1, precision))) return rc;
        if ((rc = conv1d_plan(Ly.fc1, f1w->data, f1b->data, C, c.ffn, 1, 1, 0, T, 3, 0, precision))) return rc;
        if ((rc = conv1d_plan(Ly.fc2, f2w->data, f2b->data, c.ffn, C, 1, 1, 0, T, 0, 1, precision))) return rc;
        if ((rc = mf_conv_bind(Ly.qkv, *h->ln)) || (rc = mf_conv_bind(Ly.out, *h->ao)) || (rc = mf_conv_bind(Ly.fc1, *h->ln)) || (rc = mf_conv_bind(Ly.fc2, *h->h1))) return rc;
        if ((rc = ln_pair(p + "layer_norm", C, &Ly.g1, &Ly.b1)) || (rc = ln_pair(p + "final_layer_norm", C, &Ly.g2, &Ly.b2))) return rc;
        h->layers.push_back(Ly);
    }
    if (!c.out_hidden) {
        const mf_tensor *w = get(sd, "lm_head.weight", (int64_t)c.vocab * C), *b = get(sd, "lm_head.bias", c.vocab);
        if (!w || !b) return MF_ERR_INVALID;
        const int vp = (c.vocab + 3) / 4 * 4;                      // the epilogue stores channel quads
        std::vector<float> wp((size_t)vp * C, 0.f), bp(vp, 0.f);
        std::copy(w->data, w->data + (size_t)c.vocab * C, wp.begin());
        std::copy(b->data, b->data + c.vocab, bp.begin());
        h->head = h->new_plan();
        if ((rc = conv1d_plan(h->head, wp.data(), bp.data(), C, vp, 1, 1, 0, T, 0, 0, precision))) return rc;
        if ((rc = mf_conv_bind(h->head, *h->xa))) return rc;
        h->vocab_pad = vp;
    }
    *out = h.release();
    return MF_OK;
}

extern "C" int mf_wav2vec2_frames(const mf_wav2vec2* h, int* n_frames, int* width) {
    MF_REQUIRE(h && n_frames && width, "wav2vec2_frames: null argument");
    *n_frames = h->T;
    *width = h->cfg.out_hidden ? h->cfg.hidden : h->cfg.vocab;
    return MF_OK;
}

extern "C" int mf_wav2vec2_forward(mf_wav2vec2* h, const float* wav, int n_samples, int n_windows, float* out, void* stream) {
    MF_REQUIRE(h && wav && out, "wav2vec2_forward: null argument");
    MF_REQUIRE(n_samples == h->n, "wav2vec2_forward: the handle was built for windows of %d samples, got %d (NerfASR feeds (l + m + r) * 320 every step)", h->n,
               n_samples);
    return h->forward(wav, n_windows, out, (hipStream_t)stream);
}

extern "C" void mf_wav2vec2_destroy(mf_wav2vec2* h) { delete h; }

// MuseTalk's Whisper audio features (H3) on gfx950: log-mel spectrogram + Whisper AudioEncoder returning the
// embeddings of every layer (musetalk/whisper/whisper/audio.py:92-125, model.py:143-171, transcribe.py:103-126).
//
// The encoder is a static schedule over token-sequence buffers (mf_nn.h): Conv1d = (1 x 3) convolution,
// every Linear = 1x1 convolution on the MFMA implicit-GEMM kernel with GELU / residual epilogues, attention =
// QK^T and PV as the same GEMM with the key / value operand packed on the device, row softmax in fp32.
// FLOP-wise this stage is <1 % of a MuseTalk step, so it is built for exactness (bf16x3) and simplicity.
#include "mf_nn.h"
#include "mf_aux.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

constexpr int W_NFFT = 400, W_HOP = 160, W_BINS = 201, W_MELS = 80, W_FRAMES = 3000;

// ---- log-mel ------------------------------------------------------------------------------------------
struct WTables { double* win = nullptr; double* tw = nullptr; float* basis = nullptr; bool ready = false; };
WTables g_wt[16];

double w_hz_to_mel(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double w_mel_to_hz(double m) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

int w_tables(int dev) {
    WTables& t = g_wt[dev];
    if (t.ready) return MF_OK;
    const double PI = 3.14159265358979323846;
    std::vector<double> win(W_NFFT), tw(2 * W_NFFT);
    for (int j = 0; j < W_NFFT; ++j) {
        win[j] = 0.5 - 0.5 * std::cos(2.0 * PI * j / W_NFFT);   // torch.hann_window(400), periodic
        tw[2 * j] = std::cos(2.0 * PI * j / W_NFFT);
        tw[2 * j + 1] = std::sin(2.0 * PI * j / W_NFFT);
    }
    // assets/mel_filters.npz == librosa.filters.mel(sr=16000, n_fft=400, n_mels=80) (audio.py:80-87)
    std::vector<double> mel_f(W_MELS + 2);
    const double m_lo = w_hz_to_mel(0.0), m_hi = w_hz_to_mel(8000.0);
    for (int i = 0; i < W_MELS + 2; ++i) mel_f[i] = w_mel_to_hz(m_lo + (m_hi - m_lo) * i / (W_MELS + 1));
    std::vector<float> basis((size_t)W_MELS * W_BINS);
    for (int i = 0; i < W_MELS; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < W_BINS; ++k) {
            const double fk = 8000.0 * k / (W_BINS - 1);
            const double lower = (fk - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - fk) / (mel_f[i + 2] - mel_f[i + 1]);
            const float w = (float)std::fmax(0.0, std::fmin(lower, upper));
            basis[(size_t)i * W_BINS + k] = (float)((double)w * enorm);
        }
    }
    MF_HIP(hipMalloc(&t.win, win.size() * sizeof(double)));
    MF_HIP(hipMalloc(&t.tw, tw.size() * sizeof(double)));
    MF_HIP(hipMalloc(&t.basis, basis.size() * sizeof(float)));
    MF_HIP(hipMemcpy(t.win, win.data(), win.size() * sizeof(double), hipMemcpyHostToDevice));
    MF_HIP(hipMemcpy(t.tw, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
    MF_HIP(hipMemcpy(t.basis, basis.data(), basis.size() * sizeof(float), hipMemcpyHostToDevice));
    t.ready = true;
    return MF_OK;
}

__device__ __forceinline__ unsigned f2ord(float f) {   // order-preserving float -> uint for atomicMax
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// one workgroup per STFT frame: reflect-padded centred frame, fp64 DFT, |.|^2, mel, log10(max(1e-10, .));
// raw[m][t] fp32 and the global maximum (audio.py:117-123)
// grid (frames, windows): window w reads wav + w * n, writes raw + w * raw_stride and gmax[w]
__global__ __launch_bounds__(256) void k_wlogmel_raw(const float* __restrict__ wav, int n, int T, const double* win,
                                                     const double* tw, const float* basis, float* raw, unsigned* gmax, int64_t raw_stride) {
    __shared__ double s_x[W_NFFT];
    __shared__ double s_tw[2 * W_NFFT];
    __shared__ double s_pow[W_BINS + 7];
    const int t = blockIdx.x, tid = threadIdx.x;
    wav += (int64_t)blockIdx.y * n; raw += (int64_t)blockIdx.y * raw_stride; gmax += blockIdx.y;
    for (int j = tid; j < W_NFFT; j += 256) {
        int p = t * W_HOP - W_NFFT / 2 + j;
        if (p < 0) p = -p;                       // torch.stft(center=True, pad_mode="reflect")
        if (p >= n) p = 2 * (n - 1) - p;
        s_x[j] = (double)wav[p] * win[j];
        s_tw[2 * j] = tw[2 * j];
        s_tw[2 * j + 1] = tw[2 * j + 1];
    }
    __syncthreads();
    if (tid < W_BINS) {
        double re = 0.0, im = 0.0;
        int k = 0;
        for (int j = 0; j < W_NFFT; ++j) {
            const double x = s_x[j];
            re = fma(x, s_tw[2 * k], re);
            im = fma(x, s_tw[2 * k + 1], im);
            k += tid;
            if (k >= W_NFFT) k -= W_NFFT;
        }
        s_pow[tid] = re * re + im * im;
    }
    __syncthreads();
    if (tid < W_MELS) {
        const float* row = basis + (size_t)tid * W_BINS;
        double acc = 0.0;
        for (int k = 0; k < W_BINS; ++k) acc = fma((double)row[k], s_pow[k], acc);
        const float v = (float)log10(fmax(acc, 1e-10));
        raw[(size_t)tid * T + t] = v;
        atomicMax(gmax, f2ord(v));
    }
}

// log_spec = max(log_spec, max - 8); (log_spec + 4) / 4; optional fp32 [80][T] copy and/or the padded NHWC
// conv1 input (pad_or_trim to 3000 frames pads the NORMALISED spectrogram with zeros, transcribe.py:108)
__global__ __launch_bounds__(256) void k_wlogmel_fin(const float* raw, const unsigned* gmax, int T, float* out_f32,
                                                     bf16_t* mel_hi, bf16_t* mel_lo, int64_t tok0, int C, int total, int64_t raw_stride,
                                                     int64_t mel_stride) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    raw += (int64_t)blockIdx.y * raw_stride; gmax += blockIdx.y; tok0 += (int64_t)blockIdx.y * mel_stride;
    if (out_f32) out_f32 += (int64_t)blockIdx.y * total;
    const int m = idx / T, t = idx - m * T;
    const float floor_v = ord2f(*gmax) - 8.0f;
    const float v = (fmaxf(raw[idx], floor_v) + 4.0f) / 4.0f;
    if (out_f32) out_f32[idx] = v;
    if (mel_hi) {
        unsigned u = __float_as_uint(v);
        u += 0x7fffu + ((u >> 16) & 1u);
        const unsigned h = u >> 16;
        const int64_t o = tok0 + (int64_t)t * C + m;
        mel_hi[o] = (bf16_t)h;
        if (mel_lo) {
            unsigned l = __float_as_uint(v - __uint_as_float(h << 16));
            l += 0x7fffu + ((l >> 16) & 1u);
            mel_lo[o] = (bf16_t)(l >> 16);
        }
    }
}

}  // namespace

struct mf_whisper {
    int precision = MF_PREC_BF16X3;
    int n_mels = 80, n_ctx = 1500, C = 384, n_head = 6, n_layer = 4;
    std::vector<std::unique_ptr<ActBuf>> bufs;
    std::vector<std::unique_ptr<ConvPlan>> plans;
    std::vector<float*> dev_f32;
    ActBuf *mel_in, *c1, *pos, *xa, *xb, *ln, *qkv, *sc, *pm, *ao, *h1;
    ConvPlan *conv1, *conv2, *scores, *pv;
    struct Layer { ConvPlan *qkv, *out, *fc1, *fc2; float *ln1_g, *ln1_b, *ln2_g, *ln2_b; };
    std::vector<Layer> layers;
    float* raw = nullptr;       // [cap][80][3000] fp32 scratch of the log-mel
    unsigned* gmax = nullptr;   // [cap]
    int cap = 1;                // windows one call can encode (mf_whisper_set_batch)
    std::vector<float> pos_host;   // sinusoids as NCHW [1][C][1][T], re-uploaded into every batch slot of `pos` when the workspace grows
    bool fused_attn = true;     // one k_attention launch per layer; false (MF_ATTN=composite or an unsupported head dim):
                                // per-head pack + GEMM + softmax + pack + GEMM

    ~mf_whisper() {
        for (auto& p : plans) mf_conv_plan_destroy(p.get());
        for (auto& b : bufs) { if (b->hi) (void)hipFree(b->hi); if (b->lo) (void)hipFree(b->lo); }
        for (float* f : dev_f32) (void)hipFree(f);
        if (raw) (void)hipFree(raw);
        if (gmax) (void)hipFree(gmax);
    }
    ActBuf* seq(int C_, int T) {
        bufs.emplace_back(new ActBuf());
        ActBuf* b = bufs.back().get();
        b->C = C_; b->H = 1; b->W = T; b->halo = 1;
        return b;
    }
    int alloc() {
        for (auto& b : bufs) {
            if (b->hi) { (void)hipFree(b->hi); b->hi = nullptr; }
            if (b->lo) { (void)hipFree(b->lo); b->lo = nullptr; }
            const size_t bytes = ((size_t)cap * b->per_batch() + 64) * sizeof(bf16_t);
            MF_HIP(hipMalloc(&b->hi, bytes)); MF_HIP(hipMemset(b->hi, 0, bytes));
            if (precision == MF_PREC_BF16X3) { MF_HIP(hipMalloc(&b->lo, bytes)); MF_HIP(hipMemset(b->lo, 0, bytes)); }
        }
        if (raw) (void)hipFree(raw);
        if (gmax) (void)hipFree(gmax);
        raw = nullptr; gmax = nullptr;
        MF_HIP(hipMalloc(&raw, (size_t)cap * W_MELS * W_FRAMES * sizeof(float)));
        MF_HIP(hipMalloc(&gmax, (size_t)cap * sizeof(unsigned)));
        if (!pos_host.empty()) {
            // the positional embedding is conv2's residual operand: one copy per batch slot (a residual view has a batch stride)
            float* d = nullptr;
            const size_t n1 = pos_host.size();
            MF_HIP(hipMalloc(&d, (size_t)cap * n1 * sizeof(float)));
            for (int b = 0; b < cap; ++b) MF_HIP(hipMemcpy(d + (size_t)b * n1, pos_host.data(), n1 * sizeof(float), hipMemcpyHostToDevice));
            const int rc = mf_nchw_to_act(d, C, *pos, cap, nullptr);
            MF_HIP(hipDeviceSynchronize());
            (void)hipFree(d);
            if (rc) return rc;
        }
        return MF_OK;
    }
    ConvPlan* new_plan() { plans.emplace_back(new ConvPlan()); return plans.back().get(); }
    int upload(const float* host, size_t n, float** dev) {
        MF_HIP(hipMalloc(dev, n * sizeof(float)));
        MF_HIP(hipMemcpy(*dev, host, n * sizeof(float), hipMemcpyHostToDevice));
        dev_f32.push_back(*dev);
        return MF_OK;
    }
    int log_mel_into_input(const float* wav, int n, float* out_f32, bool to_input, hipStream_t s, int S = 1);
    int encode(float* emb, hipStream_t s, int S = 1, int ctx = 0, int keep = 0);
};

namespace {

ActView W(ActBuf* b) { return ActView{b, 0, b->C}; }

const mf_tensor* get(const std::map<std::string, const mf_tensor*>& sd, const std::string& k, int64_t numel) {
    auto it = sd.find(k);
    if (it == sd.end()) { mf_set_error("whisper: state dict has no tensor '%s'", k.c_str()); return nullptr; }
    int64_t n = 1;
    for (int i = 0; i < it->second->ndim; ++i) n *= it->second->shape[i];
    if (n != numel) { mf_set_error("whisper: tensor '%s' has %lld elements, expected %lld", k.c_str(), (long long)n, (long long)numel); return nullptr; }
    return it->second;
}

int linear_plan(ConvPlan* p, const float* w, const float* b, int cin, int cout, int T, int act, int residual, int precision) {
    mf_conv2d_desc d{};
    d.cin = cin; d.cout = cout; d.kh = d.kw = 1; d.stride_h = d.stride_w = 1; d.act = act; d.residual = residual;
    d.in_h = 1; d.in_w = T;
    return mf_conv_plan_create(p, d, w, b, nullptr, nullptr, nullptr, nullptr, precision);
}

}  // namespace

int mf_whisper::log_mel_into_input(const float* wav, int n, float* out_f32, bool to_input, hipStream_t s, int S) {
    const int T = n / W_HOP;   // stft gives 1 + n/160 frames, the last one is dropped (audio.py:117)
    MF_REQUIRE(T >= 1 && T <= W_FRAMES, "whisper: %d samples give %d frames; one segment holds 1..3000", n, T);
    MF_REQUIRE(n > W_NFFT / 2, "whisper: reflect padding needs more than %d samples", W_NFFT / 2);
    MF_REQUIRE(S >= 1 && S <= cap, "whisper: %d windows exceed the handle's batch capacity %d (mf_whisper_set_batch)", S, cap);
    int dev = 0;
    MF_HIP(hipGetDevice(&dev));
    int rc = w_tables(dev);
    if (rc) return rc;
    const WTables& t = g_wt[dev];
    const int64_t raw_stride = (int64_t)W_MELS * W_FRAMES;
    MF_HIP(hipMemsetAsync(gmax, 0, (size_t)S * sizeof(unsigned), s));   // 0 orders below every float
    hipLaunchKernelGGL(k_wlogmel_raw, dim3(T, S), dim3(256), 0, s, wav, n, T, t.win, t.tw, t.basis, raw, gmax, raw_stride);
    MF_HIP(hipGetLastError());
    if (to_input) {
        // frames >= T of the 3000-frame segment are zero (pad_or_trim)
        const size_t bytes = ((size_t)S * mel_in->per_batch()) * sizeof(bf16_t);
        MF_HIP(hipMemsetAsync(mel_in->hi, 0, bytes, s));
        if (mel_in->lo) MF_HIP(hipMemsetAsync(mel_in->lo, 0, bytes, s));
    }
    const int total = W_MELS * T;
    hipLaunchKernelGGL(k_wlogmel_fin, dim3((total + 255) / 256, S), dim3(256), 0, s, raw, gmax, T, out_f32,
                       to_input ? mel_in->hi : nullptr, to_input ? mel_in->lo : nullptr, mf_interior(*mel_in), mel_in->C, total, raw_stride,
                       mel_in->per_batch());
    MF_HIP(hipGetLastError());
    return MF_OK;
}

// S windows at once.  ctx = 0: the reference's full 1500-token context (transcribe.py:108 pads every window to 30 s); ctx > 0: only the
// first ctx tokens exist (an APPROXIMATION: Whisper's encoder attention is global and unmasked, so the pad tokens a shorter context drops
// do shift the result -- the bench reports by how much).  keep = 0: all tokens of all n_layer + 1 hidden states as [layer][T][C]
// (one window only, the layout of `encoder_embeddings`); keep > 0: the first `keep` tokens as [S][keep][n_layer + 1][C] -- what
// Audio2Feature.audio2feat slices out (audio2feature.py:103-110) -- and the LAST block then evaluates only those queries: its keys and
// values still come from every token, so the kept outputs are exactly those of the full evaluation (no later layer reads the rest).
int mf_whisper::encode(float* emb, hipStream_t s, int S, int ctx, int keep) {
    int rc;
    const int T = n_ctx;
    const int tc = (ctx > 0 && ctx < T) ? ctx : 0;         // token prefix of every op (0 = the whole sequence)
    const int Tc = tc ? tc : T;
    MF_REQUIRE(S >= 1 && S <= cap, "whisper: %d windows exceed the handle's batch capacity %d", S, cap);
    MF_REQUIRE(keep >= 0 && keep <= Tc, "whisper: keep = %d tokens exceeds the context (%d)", keep, Tc);
    MF_REQUIRE(keep > 0 || S == 1, "whisper: the full-embedding layout is defined for one window");
    MF_REQUIRE(fused_attn || (S == 1 && !tc && !keep), "whisper: batched / pruned encoding needs the fused attention kernel");
    const int64_t per_emb = (int64_t)T * C;
    auto emit = [&](ActBuf* x, int layer) {
        return keep > 0 ? mf_rows_to_f32_layered(W(x), emb, S, keep, layer, n_layer + 1, s) : mf_rows_to_f32(W(x), emb + layer * per_emb, 1, s);
    };
    if ((rc = mf_conv_launch(conv1, W(mel_in), W(c1), ActView{}, S, s, tc ? std::min(2 * tc + 2, W_FRAMES) : 0))) return rc;   // gelu(conv1(x))
    if ((rc = mf_conv_launch(conv2, W(c1), W(xa), W(pos), S, s, tc))) return rc;                 // gelu(conv2(x)) + pos
    if ((rc = emit(xa, 0))) return rc;
    ActBuf* x = xa;
    ActBuf* y = xb;
    const int dh = C / n_head;
    const float scale = 1.0f / std::sqrt((float)dh);        // (dh^-0.25 on q) * (dh^-0.25 on k), model.py:91-93
    for (int l = 0; l < n_layer; ++l) {
        Layer& L = layers[l];
        const int tq = (keep > 0 && l == n_layer - 1) ? keep : tc;      // rows the rest of this block computes (0 = all)
        if ((rc = mf_layernorm(W(x), W(ln), L.ln1_g, L.ln1_b, 1e-5f, S, s, tc))) return rc;
        if ((rc = mf_conv_launch(L.qkv, W(ln), W(qkv), ActView{}, S, s, tc))) return rc;
        // softmax(q k^T * dh^-0.5) v, all heads in one fused launch (mf_attn.hip); model.py:91-93 applies the same
        // scale as dh^-0.25 on q and on k
        if (fused_attn) {
            if ((rc = mf_attention(ActView{qkv, 0, C}, ActView{qkv, C, C}, ActView{qkv, 2 * C, C}, ActView{ao, 0, C}, n_head, S, precision, s, tq, tc))) return rc;
        } else {
            const int64_t q0 = mf_interior(*qkv);
            for (int h = 0; h < n_head; ++h) {
                // scores[i][j] = q_i . k_j : keys packed as the GEMM's weight operand
                if ((rc = mf_pack_b(scores, qkv->hi + q0 + C + h * dh, qkv->lo ? qkv->lo + q0 + C + h * dh : nullptr, qkv->C, 1, T, dh, s))) return rc;
                if ((rc = mf_conv_launch(scores, ActView{qkv, h * dh, dh}, ActView{sc, 0, T}, ActView{}, 1, s))) return rc;
                if ((rc = mf_softmax_rows(ActView{sc, 0, T}, W(pm), T, scale, 1, s))) return rc;
                // out[i][d] = sum_j p[i][j] v[j][d] : V^T packed as the weight operand
                if ((rc = mf_pack_b(pv, qkv->hi + q0 + 2 * C + h * dh, qkv->lo ? qkv->lo + q0 + 2 * C + h * dh : nullptr, 1, qkv->C, dh, T, s))) return rc;
                if ((rc = mf_conv_launch(pv, W(pm), ActView{ao, h * dh, dh}, ActView{}, 1, s))) return rc;
            }
        }
        if ((rc = mf_conv_launch(L.out, W(ao), W(y), W(x), S, s, tq))) return rc;                 // x + attn(ln(x))
        if ((rc = mf_layernorm(W(y), W(ln), L.ln2_g, L.ln2_b, 1e-5f, S, s, tq))) return rc;
        if ((rc = mf_conv_launch(L.fc1, W(ln), W(h1), ActView{}, S, s, tq))) return rc;            // gelu(fc1)
        if ((rc = mf_conv_launch(L.fc2, W(h1), W(x), W(y), S, s, tq))) return rc;                  // y + fc2(.) -> x
        if ((rc = emit(x, l + 1))) return rc;
    }
    return MF_OK;
}

// ------------------------------------------------------------------------------------------------------
extern "C" int mf_whisper_create(const mf_tensor* weights, int n_weights, int n_head, int precision, mf_whisper** out) {
    MF_REQUIRE(weights && out && n_weights > 0 && n_head > 0, "whisper_create: bad argument");
    MF_REQUIRE(precision == MF_PREC_BF16 || precision == MF_PREC_BF16X3, "whisper_create: unknown precision %d", precision);
    *out = nullptr;
    std::map<std::string, const mf_tensor*> sd;
    for (int i = 0; i < n_weights; ++i) {
        MF_REQUIRE(weights[i].name && weights[i].data, "whisper_create: tensor %d has no name/data", i);
        std::string k = weights[i].name;
        if (k.rfind("encoder.", 0) == 0) k = k.substr(8);   // keys of a full Whisper checkpoint
        sd[k] = &weights[i];
    }
    auto it = sd.find("conv1.weight");
    MF_REQUIRE(it != sd.end() && it->second->ndim == 3 && it->second->shape[2] == 3, "whisper_create: conv1.weight [C, n_mels, 3] missing");
    std::unique_ptr<mf_whisper> h(new mf_whisper());
    h->precision = precision;
    h->C = (int)it->second->shape[0];
    h->n_mels = (int)it->second->shape[1];
    h->n_head = n_head;
    MF_REQUIRE(h->n_mels == W_MELS, "whisper_create: n_mels=%d, only 80 is supported (audio.py:76)", h->n_mels);
    MF_REQUIRE(h->C % n_head == 0 && (h->C / n_head) % 8 == 0, "whisper_create: head dim must be a multiple of 8");
    h->fused_attn = mf_attention_supported(h->C / n_head);
    int L = 0;
    while (sd.count("blocks." + std::to_string(L) + ".attn.query.weight")) ++L;
    MF_REQUIRE(L > 0, "whisper_create: no encoder blocks in the state dict");
    h->n_layer = L;
    h->n_ctx = W_FRAMES / 2;
    const int C = h->C, T = h->n_ctx, Tp = (T + 63) / 64 * 64;

    h->mel_in = h->seq(W_MELS, W_FRAMES); h->c1 = h->seq(C, W_FRAMES); h->pos = h->seq(C, T);
    h->xa = h->seq(C, T); h->xb = h->seq(C, T); h->ln = h->seq(C, T); h->qkv = h->seq(3 * C, T);
    h->sc = h->seq((T + 7) / 8 * 8, T); h->pm = h->seq(Tp, T); h->ao = h->seq(C, T); h->h1 = h->seq(4 * C, T);
    // sinusoids(n_ctx, C) (model.py:48-54), uploaded through the NCHW->planes pass by alloc()
    {
        std::vector<float>& pe = h->pos_host;
        pe.assign((size_t)T * C, 0.f);          // laid out as NCHW [1][C][1][T]
        const double inc = std::log(10000.0) / (C / 2 - 1);
        for (int t = 0; t < T; ++t)
            for (int c = 0; c < C / 2; ++c) {
                // torch computes inv_timescales and scaled_time in float32
                const float inv = std::exp((float)(-inc) * (float)c);
                const float st = (float)t * inv;
                pe[(size_t)c * T + t] = std::sin(st);
                pe[(size_t)(c + C / 2) * T + t] = std::cos(st);
            }
    }
    int rc = h->alloc();
    if (rc) return rc;

    // conv1 / conv2: Conv1d(k=3) as a (1 x 3) convolution over the frame axis
    {
        const mf_tensor *w1 = get(sd, "conv1.weight", (int64_t)C * W_MELS * 3), *b1 = get(sd, "conv1.bias", C);
        const mf_tensor *w2 = get(sd, "conv2.weight", (int64_t)C * C * 3), *b2 = get(sd, "conv2.bias", C);
        if (!w1 || !b1 || !w2 || !b2) return MF_ERR_INVALID;
        mf_conv2d_desc d{};
        d.cin = W_MELS; d.cout = C; d.kh = 1; d.kw = 3; d.stride_h = d.stride_w = 1; d.pad_h = 0; d.pad_w = 1;
        d.act = 3; d.in_h = 1; d.in_w = W_FRAMES;
        h->conv1 = h->new_plan();
        if ((rc = mf_conv_plan_create(h->conv1, d, w1->data, b1->data, nullptr, nullptr, nullptr, nullptr, precision))) return rc;
        if ((rc = mf_conv_bind(h->conv1, *h->mel_in))) return rc;
        d.cin = C; d.stride_w = 2; d.residual = 2;   // x = gelu(conv2(x)) + positional_embedding (model.py:152-156)
        h->conv2 = h->new_plan();
        if ((rc = mf_conv_plan_create(h->conv2, d, w2->data, b2->data, nullptr, nullptr, nullptr, nullptr, precision))) return rc;
        if ((rc = mf_conv_bind(h->conv2, *h->c1))) return rc;
    }
    for (int l = 0; l < L; ++l) {
        const std::string p = "blocks." + std::to_string(l) + ".";
        const mf_tensor *qw = get(sd, p + "attn.query.weight", (int64_t)C * C), *qb = get(sd, p + "attn.query.bias", C);
        const mf_tensor *kw = get(sd, p + "attn.key.weight", (int64_t)C * C);
        const mf_tensor *vw = get(sd, p + "attn.value.weight", (int64_t)C * C), *vb = get(sd, p + "attn.value.bias", C);
        const mf_tensor *ow = get(sd, p + "attn.out.weight", (int64_t)C * C), *ob = get(sd, p + "attn.out.bias", C);
        const mf_tensor *g1 = get(sd, p + "attn_ln.weight", C), *b1 = get(sd, p + "attn_ln.bias", C);
        const mf_tensor *f1w = get(sd, p + "mlp.0.weight", (int64_t)4 * C * C), *f1b = get(sd, p + "mlp.0.bias", 4 * C);
        const mf_tensor *f2w = get(sd, p + "mlp.2.weight", (int64_t)4 * C * C), *f2b = get(sd, p + "mlp.2.bias", C);
        const mf_tensor *g2 = get(sd, p + "mlp_ln.weight", C), *b2 = get(sd, p + "mlp_ln.bias", C);
        if (!qw || !qb || !kw || !vw || !vb || !ow || !ob || !g1 || !b1 || !f1w || !f1b || !f2w || !f2b || !g2 || !b2) return MF_ERR_INVALID;
        // one GEMM for q | k | v (key has no bias, model.py:62)
        std::vector<float> w((size_t)3 * C * C), b((size_t)3 * C, 0.f);
        std::copy(qw->data, qw->data + (size_t)C * C, w.begin());
        std::copy(kw->data, kw->data + (size_t)C * C, w.begin() + (size_t)C * C);
        std::copy(vw->data, vw->data + (size_t)C * C, w.begin() + (size_t)2 * C * C);
        std::copy(qb->data, qb->data + C, b.begin());
        std::copy(vb->data, vb->data + C, b.begin() + 2 * C);
        mf_whisper::Layer Ly{};
        Ly.qkv = h->new_plan(); Ly.out = h->new_plan(); Ly.fc1 = h->new_plan(); Ly.fc2 = h->new_plan();
        if ((rc = linear_plan(Ly.qkv, w.data(), b.data(), C, 3 * C, T, 0, 0, precision))) return rc;
        if ((rc = linear_plan(Ly.out, ow->data, ob->data, C, C, T, 0, 1, precision))) return rc;
        if ((rc = linear_plan(Ly.fc1, f1w->data, f1b->data, C, 4 * C, T, 3, 0, precision))) return rc;
        if ((rc = linear_plan(Ly.fc2, f2w->data, f2b->data, 4 * C, C, T, 0, 1, precision))) return rc;
        if ((rc = mf_conv_bind(Ly.qkv, *h->ln)) || (rc = mf_conv_bind(Ly.out, *h->ao)) || (rc = mf_conv_bind(Ly.fc1, *h->ln)) ||
            (rc = mf_conv_bind(Ly.fc2, *h->h1))) return rc;
        if ((rc = h->upload(g1->data, C, &Ly.ln1_g)) || (rc = h->upload(b1->data, C, &Ly.ln1_b)) ||
            (rc = h->upload(g2->data, C, &Ly.ln2_g)) || (rc = h->upload(b2->data, C, &Ly.ln2_b))) return rc;
        h->layers.push_back(Ly);
    }
    h->scores = h->new_plan(); h->pv = h->new_plan();
    if ((rc = mf_gemm_plan_create(h->scores, C / n_head, T, T, precision))) return rc;
    if ((rc = mf_gemm_plan_create(h->pv, Tp, C / n_head, T, precision))) return rc;
    if ((rc = mf_conv_bind(h->scores, *h->qkv)) || (rc = mf_conv_bind(h->pv, *h->pm))) return rc;
    *out = h.release();
    return MF_OK;
}

extern "C" int mf_whisper_dims(const mf_whisper* h, int* n_layer, int* n_ctx, int* n_state) {
    MF_REQUIRE(h && n_layer && n_ctx && n_state, "whisper_dims: null argument");
    *n_layer = h->n_layer; *n_ctx = h->n_ctx; *n_state = h->C;
    return MF_OK;
}

extern "C" int mf_whisper_log_mel(mf_whisper* h, const float* wav, int n, float* out, void* stream) {
    MF_REQUIRE(h && wav && out, "whisper_log_mel: null argument");
    return h->log_mel_into_input(wav, n, out, false, (hipStream_t)stream);
}

extern "C" int mf_whisper_encode_audio(mf_whisper* h, const float* wav, int n, float* emb, void* stream) {
    MF_REQUIRE(h && wav && emb, "whisper_encode_audio: null argument");
    hipStream_t s = (hipStream_t)stream;
    int rc = h->log_mel_into_input(wav, n, nullptr, true, s);
    if (rc) return rc;
    return h->encode(emb, s);
}

extern "C" int mf_whisper_set_batch(mf_whisper* h, int max_windows) {
    MF_REQUIRE(h && max_windows >= 1 && max_windows <= 64, "whisper_set_batch: 1..64 windows");
    if (max_windows == h->cap) return MF_OK;
    MF_HIP(hipDeviceSynchronize());
    h->cap = max_windows;
    return h->alloc();
}

extern "C" int mf_whisper_encode_windows(mf_whisper* h, const float* wav, int n, int n_windows, int ctx_tokens, float* feat, void* stream) {
    MF_REQUIRE(h && wav && feat, "whisper_encode_windows: null argument");
    MF_REQUIRE(n_windows >= 1 && n_windows <= h->cap, "whisper_encode_windows: %d windows exceed the handle's capacity %d (mf_whisper_set_batch)",
               n_windows, h->cap);
    const int frames = n / W_HOP, keep = frames / 2;                          // audio2feature.py:104-106: int(end_frame - start_frame) / 2
    MF_REQUIRE(keep >= 1, "whisper_encode_windows: %d samples hold no 20 ms feature row", n);
    MF_REQUIRE(ctx_tokens == 0 || (ctx_tokens >= keep && ctx_tokens <= h->n_ctx), "whisper_encode_windows: context of %d tokens must cover the %d kept ones", ctx_tokens, keep);
    hipStream_t s = (hipStream_t)stream;
    int rc = h->log_mel_into_input(wav, n, nullptr, true, s, n_windows);
    if (rc) return rc;
    return h->encode(feat, s, n_windows, ctx_tokens, keep);
}

extern "C" void mf_whisper_destroy(mf_whisper* h) { delete h; }

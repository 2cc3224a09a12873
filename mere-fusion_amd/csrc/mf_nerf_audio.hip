// ER-NeRF audio feature path on gfx950: AudioNet (4 strided Conv1d + 2 Linear) on each of the 8 windows, then AudioAttNet
// (5 Conv1d + Linear(8, 8) + softmax) pooling them into one 32-vector.
//
// Replaces `NeRFNetwork.encode_audio` (reference: ernerf/nerf_triplane/network.py:222-237 -> AudioNet :40-66, AudioAttNet :9-36),
// which runs once per frame on an [8, audio_in_dim, 16] window: ~0.3 MFLOP, pure launch latency in the reference (about 25
// kernels).  Here it is one launch: a workgroup per window through AudioNet, the last one to finish runs AudioAttNet; fp32, every
// intermediate in LDS.
#include "mf_common.h"
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

constexpr int SEQ = 8, WIN = 16, AUD_DIM = 32;
constexpr int ACT_A = 64 * 16;           // one window's largest intermediate: the input, in_dim <= 64 channels x 16 steps
constexpr int ACT_B = 32 * 8;            // conv[0]'s output, 32 channels x 8 steps (every later one is smaller)

// Every layer's weights and bias live in ONE arena (device copy made at create time; offsets in floats, each block padded to 4): a workgroup copies the
// whole arena -- 132 KB at audio_in_dim 29, 145 KB at 64 -- into LDS with one burst of 16-byte loads before its first layer, so the 13 layers pay ONE
// memory round trip between them.  (Rounds 3 - 4 streamed a layer's weights one layer ahead through registers: each of the 13 commits still waited for
// its own L2 / HBM round trip, ~2 - 3 us against ~1 us of MACs -- most of the kernel's 46 us.)
struct Layer { int woff, boff, cin, cout; };
struct AudioArgs {
    Layer conv[4];       // AudioNet.encoder_conv, Conv1d(k3, s2, p1) + LeakyReLU(0.02)
    Layer fc[2];         // AudioNet.encoder_fc1: Linear + LeakyReLU, Linear
    Layer att[5];        // AudioAttNet.attentionConvNet, Conv1d(k3, s1, p1) + LeakyReLU(0.02)
    Layer att_fc;        // AudioAttNet.attentionNet: Linear(8, 8) + Softmax
    const float* arena;
    int arena_n;         // floats, a multiple of 4
    int in_dim, use_att;
};

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : 0.02f * v; }

// lanes that share one output's reduction: as many as keep the workgroup's 1024 threads busy -- at most a wave, at most the layer's input channels.  The small
// layers were the kernel: conv[3] has 64 outputs of 192 MACs each, so ONE wave walked 192-step chains of LDS reads (~7 us) while fifteen waves waited; split 16
// ways and folded by butterfly it is 12 MACs + 4 shuffles.  (fp32 sums in another order than a serial loop: ~1e-7 relative, the golden's gate is 2e-5.)
__device__ __forceinline__ int lanes_per_output(int nout, int cin) {
    int P = 1;
    while (P < 64 && nout * P * 2 <= (int)blockDim.x && P * 2 <= cin) P <<= 1;
    return P;
}
__device__ __forceinline__ float group_sum(float acc, int P) {
    for (int off = P >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    return acc;
}

// out[n][co][t] = act(b[co] + sum_{ci,k} w[co][ci][k] * in[n][ci][t*stride + k - 1]), zero padding 1; weights already in wbuf / bbuf
// (inlined, the layer by value: a reference to a member of the kernel's argument struct put the struct into scratch -- 160 B per lane -- and a kernel with scratch
// costs more to dispatch and reads its loop bounds from memory)
__device__ __forceinline__ void conv1d(const float* in, float* out, const Layer L, int n, int tin, int stride, bool act, const float* wbuf, const float* bbuf) {
    const int tout = (tin + 2 - 3) / stride + 1;
    const int nout = n * L.cout * tout;
    const int P = lanes_per_output(nout, L.cin), sub = threadIdx.x & (P - 1), groups = blockDim.x / P;
    for (int idx = threadIdx.x / P; idx < nout; idx += groups) {
        const int t = idx % tout, co = (idx / tout) % L.cout, b = idx / (tout * L.cout);
        float acc = 0.f;
        for (int ci = sub; ci < L.cin; ci += P) {
            const float* wr = wbuf + ((size_t)co * L.cin + ci) * 3;
            const float* xr = in + ((size_t)b * L.cin + ci) * tin;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ti = t * stride + k - 1;
                if (ti >= 0 && ti < tin) acc += wr[k] * xr[ti];
            }
        }
        acc = group_sum(acc, P) + bbuf[co];
        if (sub == 0) out[idx] = act ? lrelu(acc) : acc;
    }
    __syncthreads();
}

// out[n][o] = act(b[o] + sum_i w[o][i] * in[n][i])
__device__ __forceinline__ void linear(const float* in, float* out, const Layer L, int n, bool act, const float* wbuf, const float* bbuf) {
    const int nout = n * L.cout;
    const int P = lanes_per_output(nout, L.cin), sub = threadIdx.x & (P - 1), groups = blockDim.x / P;
    for (int idx = threadIdx.x / P; idx < nout; idx += groups) {
        const int o = idx % L.cout, b = idx / L.cout;
        float acc = 0.f;
        for (int i = sub; i < L.cin; i += P) acc += wbuf[(size_t)o * L.cin + i] * in[(size_t)b * L.cin + i];
        acc = group_sum(acc, P) + bbuf[o];
        if (sub == 0) out[idx] = act ? lrelu(acc) : acc;
    }
    __syncthreads();
}

// One workgroup per window through AudioNet (the windows are independent there, and one CU's issue rate was the bound: ~10 instructions
// per MAC); the last workgroup to finish pools the 8 feature vectors through AudioAttNet.
// prev != null: the lip-smoothing EMA of renderer.py:190-194 applied on the way out, enc_a = 0.35 * prev + (1 - 0.35) * enc_a in torch's own fp32 order (two
// rounded products, one rounded sum; no FMA contraction) -- three torch elementwise launches per frame otherwise.  prev may alias enc_a.
__device__ __forceinline__ float ema_out(const float* prev, int c, float v) {
    return prev ? __fadd_rn(__fmul_rn(0.35f, prev[c]), __fmul_rn(0.65f, v)) : v;
}

__global__ __launch_bounds__(1024) void k_audio_encode(const AudioArgs a, const float* __restrict__ auds_all, int n_win_all, float* enc_a, float* feat_g,
                                                       int* done, const float* prev) {
    const float* auds = auds_all + (size_t)blockIdx.x * a.in_dim * WIN;
    constexpr int n_win = 1;
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float* bufA = dyn;                        // ACT_A
    float* bufB = bufA + ACT_A;               // ACT_B
    float* feat = bufB + ACT_B;               // SEQ * AUD_DIM
    float* wts = feat + SEQ * AUD_DIM;        // the arena
    const bool att = a.use_att && n_win_all == SEQ;
    {
        // the arena in one burst: <= 9 independent 16-byte loads per thread, then the LDS stores
        // (spelled out nine times: as `float4 r[9]` filled and drained by two unrolled loops the array stayed in scratch -- 160 B per lane, dynamic offsets -- and a kernel
        // with scratch costs more to dispatch)
        const float4* src = reinterpret_cast<const float4*>(a.arena);
        const int n4 = a.arena_n >> 2;
#define MF_AUD_LD(k) const int i##k = threadIdx.x + k * 1024; const float4 r##k = src[i##k < n4 ? i##k : n4 - 1];
        MF_AUD_LD(0) MF_AUD_LD(1) MF_AUD_LD(2) MF_AUD_LD(3) MF_AUD_LD(4) MF_AUD_LD(5) MF_AUD_LD(6) MF_AUD_LD(7) MF_AUD_LD(8)
#undef MF_AUD_LD
        // network.py:61-62: the centre 16 steps of the window (win_size 16 -> all of them)
        for (int i = threadIdx.x; i < n_win * a.in_dim * WIN; i += blockDim.x) bufA[i] = auds[i];
#define MF_AUD_ST(k) if (i##k < n4) reinterpret_cast<float4*>(wts)[i##k] = r##k;
        MF_AUD_ST(0) MF_AUD_ST(1) MF_AUD_ST(2) MF_AUD_ST(3) MF_AUD_ST(4) MF_AUD_ST(5) MF_AUD_ST(6) MF_AUD_ST(7) MF_AUD_ST(8)
#undef MF_AUD_ST
        __syncthreads();
    }
    auto W = [&](const Layer& L) { return wts + L.woff; };
    auto B = [&](const Layer& L) { return wts + L.boff; };
    conv1d(bufA, bufB, a.conv[0], n_win, 16, 2, true, W(a.conv[0]), B(a.conv[0]));
    conv1d(bufB, bufA, a.conv[1], n_win, 8, 2, true, W(a.conv[1]), B(a.conv[1]));
    conv1d(bufA, bufB, a.conv[2], n_win, 4, 2, true, W(a.conv[2]), B(a.conv[2]));
    conv1d(bufB, bufA, a.conv[3], n_win, 2, 2, true, W(a.conv[3]), B(a.conv[3]));        // [n, 64, 1]
    linear(bufA, bufB, a.fc[0], n_win, true, W(a.fc[0]), B(a.fc[0]));
    linear(bufB, feat, a.fc[1], n_win, false, W(a.fc[1]), B(a.fc[1]));               // [n, 32]
    if (!att) {
        // att == 0: encode_audio returns audio_net's output as is (network.py:230-235); callers pass one window then
        for (int i = threadIdx.x; i < AUD_DIM; i += blockDim.x) enc_a[i] = ema_out(prev, i, feat[i]);
        return;
    }
    // hand this window's features over; the last workgroup gathers all eight
    for (int i = threadIdx.x; i < AUD_DIM; i += blockDim.x) feat_g[blockIdx.x * AUD_DIM + i] = feat[i];
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(done, 1) == SEQ - 1;
        if (s_last) *done = 0;
        __threadfence();
    }
    __syncthreads();
    if (!s_last) return;
    for (int i = threadIdx.x; i < SEQ * AUD_DIM; i += blockDim.x) feat[i] = __builtin_nontemporal_load(feat_g + i);
    __syncthreads();
    // AudioAttNet: y = x.permute(0, 2, 1) -> [1, 32, 8]
    for (int i = threadIdx.x; i < SEQ * AUD_DIM; i += blockDim.x) { const int t = i % SEQ, c = i / SEQ; bufA[c * SEQ + t] = feat[t * AUD_DIM + c]; }
    __syncthreads();
    conv1d(bufA, bufB, a.att[0], 1, SEQ, 1, true, W(a.att[0]), B(a.att[0]));
    conv1d(bufB, bufA, a.att[1], 1, SEQ, 1, true, W(a.att[1]), B(a.att[1]));
    conv1d(bufA, bufB, a.att[2], 1, SEQ, 1, true, W(a.att[2]), B(a.att[2]));
    conv1d(bufB, bufA, a.att[3], 1, SEQ, 1, true, W(a.att[3]), B(a.att[3]));
    conv1d(bufA, bufB, a.att[4], 1, SEQ, 1, true, W(a.att[4]), B(a.att[4]));           // [1, 1, 8]
    linear(bufB, bufA, a.att_fc, 1, false, W(a.att_fc), B(a.att_fc));                  // [1, 8]
    if (threadIdx.x == 0) {
        float m = bufA[0];
        for (int t = 1; t < SEQ; ++t) m = fmaxf(m, bufA[t]);
        float s = 0.f;
        for (int t = 0; t < SEQ; ++t) { bufA[t] = expf(bufA[t] - m); s += bufA[t]; }
        for (int t = 0; t < SEQ; ++t) bufA[t] /= s;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < AUD_DIM; c += blockDim.x) {    // torch.sum(y * x, dim=1), network.py:36
        float acc = 0.f;
        for (int t = 0; t < SEQ; ++t) acc += bufA[t] * feat[t * AUD_DIM + c];
        enc_a[c] = ema_out(prev, c, acc);
    }
}

}  // namespace

struct mf_audio_encoder {
    AudioArgs a{};
    float* feat_g = nullptr;
    int* done = nullptr;
    size_t lds = 0;
    std::vector<float*> dev;
    ~mf_audio_encoder() { for (float* d : dev) (void)hipFree(d); }
};

extern "C" int mf_audio_encoder_create(const mf_tensor* weights, int n_weights, int use_att, mf_audio_encoder** out) {
    MF_REQUIRE(weights && out && n_weights > 0, "audio_encoder_create: bad argument");
    *out = nullptr;
    std::map<std::string, const mf_tensor*> sd;
    for (int i = 0; i < n_weights; ++i) {
        MF_REQUIRE(weights[i].name && weights[i].data, "audio_encoder_create: tensor %d has no name/data", i);
        sd[weights[i].name] = &weights[i];
    }
    std::unique_ptr<mf_audio_encoder> h(new mf_audio_encoder());
    std::vector<float> arena;                                   // every layer's [weight | bias], each padded to 4 floats
    auto up = [&](const std::string& k, int64_t n, int* off) -> int {
        auto it = sd.find(k);
        if (it == sd.end()) { mf_set_error("audio_encoder_create: tensor '%s' missing", k.c_str()); return MF_ERR_INVALID; }
        int64_t have = 1;
        for (int d = 0; d < it->second->ndim; ++d) have *= it->second->shape[d];
        if (have != n) { mf_set_error("audio_encoder_create: '%s' has %lld elements, expected %lld", k.c_str(), (long long)have, (long long)n); return MF_ERR_INVALID; }
        *off = (int)arena.size();
        const float* src = static_cast<const float*>(it->second->data);
        arena.insert(arena.end(), src, src + n);
        arena.resize((arena.size() + 3) / 4 * 4, 0.f);
        return MF_OK;
    };
    auto layer = [&](const std::string& p, int cin, int cout, int k, Layer* L) -> int {
        L->cin = cin; L->cout = cout;
        int rc = up(p + ".weight", (int64_t)cout * cin * k, &L->woff);
        return rc ? rc : up(p + ".bias", cout, &L->boff);
    };
    auto it = sd.find("audio_net.encoder_conv.0.weight");
    MF_REQUIRE(it != sd.end() && it->second->ndim == 3 && it->second->shape[0] == 32 && it->second->shape[2] == 3,
               "audio_encoder_create: audio_net.encoder_conv.0.weight [32, in_dim, 3] missing");
    const int in_dim = (int)it->second->shape[1];
    MF_REQUIRE(in_dim >= 1 && in_dim <= 64, "audio_encoder_create: audio_in_dim %d (1..64 built: esperanto 44, deepspeech 29, default 32; hubert's 1024 is not)", in_dim);
    h->a.in_dim = in_dim;
    h->a.use_att = use_att ? 1 : 0;
    int rc;
    const int cc[5] = {in_dim, 32, 32, 64, 64};                 // network.py:46-53 (Sequential indices 0, 2, 4, 6)
    for (int i = 0; i < 4; ++i)
        if ((rc = layer("audio_net.encoder_conv." + std::to_string(2 * i), cc[i], cc[i + 1], 3, &h->a.conv[i]))) return rc;
    if ((rc = layer("audio_net.encoder_fc1.0", 64, 64, 1, &h->a.fc[0])) || (rc = layer("audio_net.encoder_fc1.2", 64, AUD_DIM, 1, &h->a.fc[1]))) return rc;
    if (use_att) {
        const int ac[6] = {AUD_DIM, 16, 8, 4, 2, 1};            // network.py:14-24
        for (int i = 0; i < 5; ++i)
            if ((rc = layer("audio_att_net.attentionConvNet." + std::to_string(2 * i), ac[i], ac[i + 1], 3, &h->a.att[i]))) return rc;
        if ((rc = layer("audio_att_net.attentionNet.0", SEQ, SEQ, 1, &h->a.att_fc))) return rc;
    }
    {
        float* d = nullptr;
        MF_HIP(hipMalloc(&d, arena.size() * sizeof(float)));
        h->dev.push_back(d);
        MF_HIP(hipMemcpy(d, arena.data(), arena.size() * sizeof(float), hipMemcpyHostToDevice));
        h->a.arena = d; h->a.arena_n = (int)arena.size();
        h->lds = (size_t)(ACT_A + ACT_B + SEQ * AUD_DIM + h->a.arena_n) * sizeof(float);
        MF_REQUIRE(h->a.arena_n <= 9 * 4 * 1024 && h->lds <= 160 * 1024 - 256, "audio_encoder_create: %d floats of weights do not fit the kernel's LDS arena", h->a.arena_n);
    }
    MF_HIP(hipMalloc(&h->feat_g, (SEQ * AUD_DIM + 1) * sizeof(float)));
    h->dev.push_back(h->feat_g);
    h->done = reinterpret_cast<int*>(h->feat_g + SEQ * AUD_DIM);
    MF_HIP(hipMemset(h->feat_g, 0, (SEQ * AUD_DIM + 1) * sizeof(float)));
    *out = h.release();
    return MF_OK;
}

static int audio_encoder_launch(mf_audio_encoder* h, const float* auds, int n_windows, float* enc_a, const float* prev, void* stream) {
    MF_REQUIRE(h && auds && enc_a, "audio_encoder_forward: null argument");
    MF_REQUIRE(h->a.use_att ? n_windows == SEQ : n_windows == 1,
               "audio_encoder_forward: %d windows (the attention net pools exactly 8, network.py:10; without it one window)", n_windows);
    static bool attr_done = false;
    if (!attr_done) {
        MF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_audio_encode), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));   // (the kernel's static s_last shares the 160 KB)
        attr_done = true;
    }
    hipLaunchKernelGGL(k_audio_encode, dim3(n_windows), dim3(1024), h->lds, (hipStream_t)stream, h->a, auds, n_windows, enc_a, h->feat_g, h->done, prev);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

extern "C" int mf_audio_encoder_forward(mf_audio_encoder* h, const float* auds, int n_windows, float* enc_a, void* stream) {
    return audio_encoder_launch(h, auds, n_windows, enc_a, nullptr, stream);
}

extern "C" int mf_audio_encoder_forward_smooth(mf_audio_encoder* h, const float* auds, int n_windows, const float* prev_enc_a, float* enc_a, void* stream) {
    return audio_encoder_launch(h, auds, n_windows, enc_a, prev_enc_a, stream);
}

extern "C" void mf_audio_encoder_destroy(mf_audio_encoder* h) { delete h; }

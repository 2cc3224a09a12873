// Wav2Lip generator as a static schedule of fused MFMA convolutions (wav2lip/models/wav2lip.py:12-125).
//
// Buffers: every layer output is a padded-NHWC (hi, lo) plane pair.  The U-Net skip
// `torch.cat((x, feats[-1]), dim=1)` (wav2lip.py:108) costs nothing: the last layer of face-encoder
// block b and the last layer of decoder block 6-b write into disjoint channel slices of ONE buffer
// (cat[k] = [decoder k | encoder 6-k]), and the next layers read slices or the whole of it.
// The audio encoder runs on a forked stream beside the face encoder; the whole schedule is
// captured once per batch size into a hipGraph and replayed.
#include "mf_conv.h"
#include "mf_aux.h"
#include <map>
#include <set>
#include <memory>
#include <string>
#include <vector>
#include <cstring>
#include <cstdlib>

namespace {

struct LayerSpec {
    const char* prefix;
    int cin, cout, k, sh, sw, pad, transposed, outpad, residual;
};

// wav2lip.py:15-36
const LayerSpec kFaceEnc[] = {
    {"face_encoder_blocks.0.0", 6, 16, 7, 1, 1, 3, 0, 0, 0},
    {"face_encoder_blocks.1.0", 16, 32, 3, 2, 2, 1, 0, 0, 0},
    {"face_encoder_blocks.1.1", 32, 32, 3, 1, 1, 1, 0, 0, 1},
    {"face_encoder_blocks.1.2", 32, 32, 3, 1, 1, 1, 0, 0, 1},
    {"face_encoder_blocks.2.0", 32, 64, 3, 2, 2, 1, 0, 0, 0},
    {"face_encoder_blocks.2.1", 64, 64, 3, 1, 1, 1, 0, 0, 1},
    {"face_encoder_blocks.2.2", 64, 64, 3, 1, 1, 1, 0, 0, 1},
    {"face_encoder_blocks.2.3", 64, 64, 3, 1, 1, 1, 0, 0, 1},
    {"face_encoder_blocks.3.0", 64, 128, 3, 2, 2, 1, 0, 0, 0},
    {"face_encoder_blocks.3.1", 128, 128, 3, 1, 1, 1, 0, 0, 1},
    {"face_encoder_blocks.3.2", 128, 128, 3, 1, 1, 1, 0, 0, 1},
    {"face_encoder_blocks.4.0", 128, 256, 3, 2, 2, 1, 0, 0, 0},
    {"face_encoder_blocks.4.1", 256, 256, 3, 1, 1, 1, 0, 0, 1},
    {"face_encoder_blocks.4.2", 256, 256, 3, 1, 1, 1, 0, 0, 1},
    {"face_encoder_blocks.5.0", 256, 512, 3, 2, 2, 1, 0, 0, 0},
    {"face_encoder_blocks.5.1", 512, 512, 3, 1, 1, 1, 0, 0, 1},
    {"face_encoder_blocks.6.0", 512, 512, 3, 1, 1, 0, 0, 0, 0},
    {"face_encoder_blocks.6.1", 512, 512, 1, 1, 1, 0, 0, 0, 0},
};
const int kFaceEncBlockLast[7] = {0, 3, 7, 10, 13, 15, 17};   // index of each block's last layer

// wav2lip.py:38-55
const LayerSpec kAudioEnc[] = {
    {"audio_encoder.0", 1, 32, 3, 1, 1, 1, 0, 0, 0},
    {"audio_encoder.1", 32, 32, 3, 1, 1, 1, 0, 0, 1},
    {"audio_encoder.2", 32, 32, 3, 1, 1, 1, 0, 0, 1},
    {"audio_encoder.3", 32, 64, 3, 3, 1, 1, 0, 0, 0},
    {"audio_encoder.4", 64, 64, 3, 1, 1, 1, 0, 0, 1},
    {"audio_encoder.5", 64, 64, 3, 1, 1, 1, 0, 0, 1},
    {"audio_encoder.6", 64, 128, 3, 3, 3, 1, 0, 0, 0},
    {"audio_encoder.7", 128, 128, 3, 1, 1, 1, 0, 0, 1},
    {"audio_encoder.8", 128, 128, 3, 1, 1, 1, 0, 0, 1},
    {"audio_encoder.9", 128, 256, 3, 3, 2, 1, 0, 0, 0},
    {"audio_encoder.10", 256, 256, 3, 1, 1, 1, 0, 0, 1},
    {"audio_encoder.11", 256, 512, 3, 1, 1, 0, 0, 0, 0},
    {"audio_encoder.12", 512, 512, 1, 1, 1, 0, 0, 0, 0},
};

// wav2lip.py:57-81 ; cin of each block's first layer = decoder channels + skip channels
const LayerSpec kFaceDec[] = {
    {"face_decoder_blocks.0.0", 512, 512, 1, 1, 1, 0, 0, 0, 0},
    {"face_decoder_blocks.1.0", 1024, 512, 3, 1, 1, 0, 1, 0, 0},
    {"face_decoder_blocks.1.1", 512, 512, 3, 1, 1, 1, 0, 0, 1},
    {"face_decoder_blocks.2.0", 1024, 512, 3, 2, 2, 1, 1, 1, 0},
    {"face_decoder_blocks.2.1", 512, 512, 3, 1, 1, 1, 0, 0, 1},
    {"face_decoder_blocks.2.2", 512, 512, 3, 1, 1, 1, 0, 0, 1},
    {"face_decoder_blocks.3.0", 768, 384, 3, 2, 2, 1, 1, 1, 0},
    {"face_decoder_blocks.3.1", 384, 384, 3, 1, 1, 1, 0, 0, 1},
    {"face_decoder_blocks.3.2", 384, 384, 3, 1, 1, 1, 0, 0, 1},
    {"face_decoder_blocks.4.0", 512, 256, 3, 2, 2, 1, 1, 1, 0},
    {"face_decoder_blocks.4.1", 256, 256, 3, 1, 1, 1, 0, 0, 1},
    {"face_decoder_blocks.4.2", 256, 256, 3, 1, 1, 1, 0, 0, 1},
    {"face_decoder_blocks.5.0", 320, 128, 3, 2, 2, 1, 1, 1, 0},
    {"face_decoder_blocks.5.1", 128, 128, 3, 1, 1, 1, 0, 0, 1},
    {"face_decoder_blocks.5.2", 128, 128, 3, 1, 1, 1, 0, 0, 1},
    {"face_decoder_blocks.6.0", 160, 64, 3, 2, 2, 1, 1, 1, 0},
    {"face_decoder_blocks.6.1", 64, 64, 3, 1, 1, 1, 0, 0, 1},
    {"face_decoder_blocks.6.2", 64, 64, 3, 1, 1, 1, 0, 0, 1},
};
const int kFaceDecBlockLast[7] = {0, 2, 5, 8, 11, 14, 17};
const int kDecC[7] = {512, 512, 512, 384, 256, 128, 64};     // decoder block output channels
const int kSkipC[7] = {512, 512, 256, 128, 64, 32, 16};      // skip joined after decoder block k
const int kCatHW[7] = {1, 3, 6, 12, 24, 48, 96};

const LayerSpec kOut0 = {"output_block.0", 80, 32, 3, 1, 1, 1, 0, 0, 0};

struct Step {
    ConvPlan plan;
    ActView in, out, res;
    int group;   // 0 = audio encoder (side stream), 1 = face encoder, 2 = decoder + output_block.0
    std::string name;
};

}  // namespace

struct mf_wav2lip {
    int precision = MF_PREC_BF16;
    int cap = 0;   // batch capacity of the workspace
    std::vector<std::unique_ptr<ActBuf>> bufs;
    std::vector<std::unique_ptr<Step>> steps;
    ActBuf* mel_in = nullptr;
    ActBuf* face_in = nullptr;
    ActBuf* out0 = nullptr;
    float* head_w = nullptr;
    float* head_b = nullptr;
    std::map<std::string, ActView> taps;
    std::map<int, hipGraphExec_t> graphs;
    std::set<int> looked_up;                     // batch sizes whose launch configurations have been looked up (no-graph path)
    hipStream_t side = nullptr;       // audio-encoder lane
    hipStream_t cap_stream = nullptr; // capture origin
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_gin = nullptr, ev_gout = nullptr;
    bool use_graph = true;

    ~mf_wav2lip() {
        for (auto& g : graphs) if (g.second) (void)hipGraphExecDestroy(g.second);
        for (auto& s : steps) mf_conv_plan_destroy(&s->plan);
        free_bufs();
        if (head_w) (void)hipFree(head_w);
        if (head_b) (void)hipFree(head_b);
        if (side) (void)hipStreamDestroy(side);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_gin) (void)hipEventDestroy(ev_gin);
        if (ev_gout) (void)hipEventDestroy(ev_gout);
    }
    void free_bufs() {
        for (auto& b : bufs) {
            if (b->hi) (void)hipFree(b->hi);
            if (b->lo) (void)hipFree(b->lo);
            b->hi = b->lo = nullptr;
        }
    }
    ActBuf* new_buf(int C, int H, int W, int halo) {
        bufs.emplace_back(new ActBuf());
        ActBuf* b = bufs.back().get();
        b->C = C; b->H = H; b->W = W; b->halo = halo;
        return b;
    }
    int add_step(const LayerSpec& ls, int group, const std::map<std::string, const mf_tensor*>& sd,
                 ActView in, ActView out);
    int ensure_capacity(int batch);
    int run_body(int batch, hipStream_t s);
    int run(int batch, hipStream_t s);
    int measure(int batch, hipStream_t s);
    int tune(int batch, hipStream_t s);
};

namespace {

const float* find(const std::map<std::string, const mf_tensor*>& sd, const std::string& key, int64_t numel) {
    auto it = sd.find(key);
    if (it == sd.end()) { mf_set_error("wav2lip: state dict has no tensor '%s'", key.c_str()); return nullptr; }
    int64_t n = 1;
    for (int i = 0; i < it->second->ndim; ++i) n *= it->second->shape[i];
    if (n != numel) {
        mf_set_error("wav2lip: tensor '%s' has %lld elements, expected %lld", key.c_str(), (long long)n, (long long)numel);
        return nullptr;
    }
    return it->second->data;
}

ActView whole(ActBuf* b) { return ActView{b, 0, b->C}; }
ActView slice(ActBuf* b, int coff, int C) { return ActView{b, coff, C}; }

}  // namespace

int mf_wav2lip::add_step(const LayerSpec& ls, int group, const std::map<std::string, const mf_tensor*>& sd,
                         ActView in, ActView out) {
    const std::string p = ls.prefix;
    const int64_t wn = (int64_t)ls.cin * ls.cout * ls.k * ls.k;
    const float* w = find(sd, p + ".conv_block.0.weight", wn);
    const float* b = find(sd, p + ".conv_block.0.bias", ls.cout);
    const float* g = find(sd, p + ".conv_block.1.weight", ls.cout);
    const float* be = find(sd, p + ".conv_block.1.bias", ls.cout);
    const float* mu = find(sd, p + ".conv_block.1.running_mean", ls.cout);
    const float* var = find(sd, p + ".conv_block.1.running_var", ls.cout);
    if (!w || !b || !g || !be || !mu || !var) return MF_ERR_INVALID;
    mf_conv2d_desc d{};
    d.cin = ls.cin; d.cout = ls.cout; d.kh = d.kw = ls.k;
    d.stride_h = ls.sh; d.stride_w = ls.sw; d.pad_h = d.pad_w = ls.pad;
    d.transposed = ls.transposed; d.output_padding = ls.outpad; d.residual = ls.residual;
    d.act = 1;   // every Conv2d / Conv2dTranspose block ends in ReLU (conv.py:12,40)
    d.in_h = in.buf->H; d.in_w = in.buf->W;
    steps.emplace_back(new Step());
    Step* st = steps.back().get();
    st->group = group; st->name = p; st->in = in; st->out = out;
    st->res = ls.residual ? in : ActView{};
    int rc = mf_conv_plan_create(&st->plan, d, w, b, g, be, mu, var, precision);
    if (rc != MF_OK) return rc;
    MF_REQUIRE(st->plan.out_h == out.buf->H && st->plan.out_w == out.buf->W,
               "wav2lip: %s produces %dx%d, buffer is %dx%d", ls.prefix, st->plan.out_h, st->plan.out_w, out.buf->H, out.buf->W);
    return mf_conv_bind(&st->plan, *in.buf);
}

int mf_wav2lip::ensure_capacity(int batch) {
    if (batch <= cap) return MF_OK;
    for (auto& g : graphs) if (g.second) (void)hipGraphExecDestroy(g.second);
    graphs.clear();
    MF_HIP(hipDeviceSynchronize());
    free_bufs();
    for (auto& b : bufs) {
        // +64 elements of slack past the last pixel
        const size_t bytes = ((size_t)batch * b->per_batch() + 64) * sizeof(bf16_t);
        MF_HIP(hipMalloc(&b->hi, bytes));
        MF_HIP(hipMemset(b->hi, 0, bytes));   // halo ring and padded channels stay zero forever
        if (precision == MF_PREC_BF16X3) {
            MF_HIP(hipMalloc(&b->lo, bytes));
            MF_HIP(hipMemset(b->lo, 0, bytes));
        }
    }
    MF_HIP(hipDeviceSynchronize());
    cap = batch;
    return MF_OK;
}

int mf_wav2lip::run_body(int batch, hipStream_t s) {
    const bool fork = true;                                           // (one serial chain measured 11 % slower: 1.12 vs 1.00 ms at batch 16)
    if (!fork) {
        for (int g = 0; g < 3; ++g)
            for (auto& st : steps)
                if (st->group == g) { int rc = mf_conv_launch(&st->plan, st->in, st->out, st->res, batch, s); if (rc) return rc; }
        return MF_OK;
    }
    // fork: audio encoder on the side stream beside the face encoder, join before the decoder
    MF_HIP(hipEventRecord(ev_fork, s));
    MF_HIP(hipStreamWaitEvent(side, ev_fork, 0));
    for (auto& st : steps)
        if (st->group == 0) { int rc = mf_conv_launch(&st->plan, st->in, st->out, st->res, batch, side); if (rc) return rc; }
    MF_HIP(hipEventRecord(ev_join, side));
    for (auto& st : steps)
        if (st->group == 1) { int rc = mf_conv_launch(&st->plan, st->in, st->out, st->res, batch, s); if (rc) return rc; }
    MF_HIP(hipStreamWaitEvent(s, ev_join, 0));
    for (auto& st : steps)
        if (st->group == 2) { int rc = mf_conv_launch(&st->plan, st->in, st->out, st->res, batch, s); if (rc) return rc; }
    return MF_OK;
}

int mf_wav2lip::measure(int batch, hipStream_t s) {
    MF_HIP(hipStreamSynchronize(side));
    for (auto& st : steps) {
        int rc = mf_conv_tune(&st->plan, st->in, st->out, st->res, batch, s);
        if (rc) return rc;
    }
    return MF_OK;
}

// mf_wav2lip_tune: the explicit warm-up -- time every layer's launch configurations on the buffers the last forward at this batch size filled, drop the
// graph captured with the old ones
int mf_wav2lip::tune(int batch, hipStream_t s) {
    auto it = graphs.find(batch);
    if (use_graph && it == graphs.end()) { mf_set_error("wav2lip_tune: run one forward at batch %d first (the layers are timed on its buffers)", batch); return MF_ERR_INVALID; }
    MF_HIP(hipStreamSynchronize(cap_stream));
    MF_HIP(hipStreamSynchronize(s));
    int rc = measure(batch, s);
    if (rc) return rc;
    MF_HIP(hipStreamSynchronize(s));
    // the next forward at this batch runs eagerly again (split-K workspaces of the new configurations), then re-captures
    if (use_graph) { if (it->second) (void)hipGraphExecDestroy(it->second); graphs.erase(it); }
    return MF_OK;
}

int mf_wav2lip::run(int batch, hipStream_t s) {
    if (!use_graph) {
        if (looked_up.insert(batch).second)                                          // (MF_NO_GRAPH: one table lookup per layer and batch size, not one per forward)
            for (auto& st : steps) mf_conv_tune_lookup(&st->plan, st->in, batch);
        return run_body(batch, s);
    }
    auto it = graphs.find(batch);
    if (it == graphs.end()) {
        // first forward at this batch size runs eagerly (it also sets the kernels' LDS attributes,
        // which must not happen inside a capture); the second one captures
        graphs.emplace(batch, nullptr);
        // launch configurations: a table lookup per layer (MF_TUNE_CACHE / the table shipped beside the library), never a measurement
        for (auto& st : steps) mf_conv_tune_lookup(&st->plan, st->in, batch);
        int rc = run_body(batch, s);
        if (rc || !mf_autotune_enabled()) return rc;
        // MF_AUTOTUNE=1 (development): the buffers hold real data now -- measure every implicit-GEMM layer in place, then run once more so that the
        // outputs belong to the configurations the graph will capture
        if ((rc = measure(batch, s))) return rc;
        return run_body(batch, s);
    }
    if (it->second == nullptr) {
        hipGraph_t graph = nullptr;
        MF_HIP(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
        int rc = run_body(batch, cap_stream);
        hipError_t e = hipStreamEndCapture(cap_stream, &graph);
        if (rc != MF_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) { mf_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return MF_ERR_HIP; }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { mf_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return MF_ERR_HIP; }
        it->second = exec;
    }
    // replay on the handle's own stream, fenced by events against the caller's (usually the legacy NULL) stream
    MF_HIP(hipEventRecord(ev_gin, s));
    MF_HIP(hipStreamWaitEvent(cap_stream, ev_gin, 0));
    MF_HIP(hipGraphLaunch(it->second, cap_stream));
    MF_HIP(hipEventRecord(ev_gout, cap_stream));
    MF_HIP(hipStreamWaitEvent(s, ev_gout, 0));
    return MF_OK;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" int mf_wav2lip_create(const mf_tensor* weights, int n_weights, int precision, mf_wav2lip** out) {
    MF_REQUIRE(weights && out && n_weights > 0, "wav2lip_create: null argument");
    MF_REQUIRE(precision == MF_PREC_BF16 || precision == MF_PREC_BF16X3, "wav2lip_create: unknown precision %d", precision);
    *out = nullptr;
    std::map<std::string, const mf_tensor*> sd;
    for (int i = 0; i < n_weights; ++i) {
        MF_REQUIRE(weights[i].name && weights[i].data, "wav2lip_create: tensor %d has no name/data", i);
        std::string k = weights[i].name;
        size_t pos;
        while ((pos = k.find("module.")) != std::string::npos) k.erase(pos, 7);   // lipreal.py:48-49
        sd[k] = &weights[i];
    }
    std::unique_ptr<mf_wav2lip> h(new mf_wav2lip());
    h->precision = precision;
    const char* ng = std::getenv("MF_NO_GRAPH");
    h->use_graph = !(ng && ng[0] == '1');
    MF_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    MF_HIP(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    MF_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    MF_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    MF_HIP(hipEventCreateWithFlags(&h->ev_gin, hipEventDisableTiming));
    MF_HIP(hipEventCreateWithFlags(&h->ev_gout, hipEventDisableTiming));

    // ---- buffers ---------------------------------------------------------------------------
    h->mel_in = h->new_buf(8, 80, 16, 1);     // 1 real channel, padded to an 8-channel group
    h->face_in = h->new_buf(8, 96, 96, 3);    // 6 real channels; 7x7 pad 3
    ActBuf* cat[7];
    for (int k = 0; k < 7; ++k) cat[k] = h->new_buf(kDecC[k] + kSkipC[k], kCatHW[k], kCatHW[k], 1);
    h->out0 = h->new_buf(32, 96, 96, 0);

    int rc;
    // ---- audio encoder: 80x16 -> 27x16 -> 9x6 -> 3x3 -> 1x1 ----------------------------------
    {
        const int hs[13] = {80, 80, 80, 27, 27, 27, 9, 9, 9, 3, 3, 1, 1};
        const int ws[13] = {16, 16, 16, 16, 16, 16, 6, 6, 6, 3, 3, 1, 1};
        ActView cur = whole(h->mel_in);
        for (int i = 0; i < 13; ++i) {
            ActBuf* ob = h->new_buf(kAudioEnc[i].cout, hs[i], ws[i], 1);
            rc = h->add_step(kAudioEnc[i], 0, sd, cur, whole(ob));
            if (rc) return rc;
            cur = whole(ob);
        }
        h->taps["audio_embedding"] = cur;
        // ---- decoder block 0 consumes the audio embedding (wav2lip.py:104) ------------------
        // (added to group 2 below, after the face encoder steps, to keep launch order simple)
        // ---- face encoder ----------------------------------------------------------------------
        ActView x = whole(h->face_in);
        int li = 0;
        for (int b = 0; b < 7; ++b) {
            const int hw = kCatHW[6 - b];
            for (; li <= kFaceEncBlockLast[b]; ++li) {
                ActView o;
                if (li == kFaceEncBlockLast[b]) o = slice(cat[6 - b], kDecC[6 - b], kSkipC[6 - b]);
                else o = whole(h->new_buf(kFaceEnc[li].cout, hw, hw, 1));
                rc = h->add_step(kFaceEnc[li], 1, sd, x, o);
                if (rc) return rc;
                x = o;
            }
            h->taps["face_encoder_blocks." + std::to_string(b)] = x;
        }
        // ---- decoder ---------------------------------------------------------------------------
        ActView y = cur;   // audio embedding
        li = 0;
        for (int b = 0; b < 7; ++b) {
            const int hw = kCatHW[b];
            for (; li <= kFaceDecBlockLast[b]; ++li) {
                ActView o;
                if (li == kFaceDecBlockLast[b]) o = slice(cat[b], 0, kDecC[b]);
                else o = whole(h->new_buf(kFaceDec[li].cout, hw, hw, 1));
                rc = h->add_step(kFaceDec[li], 2, sd, y, o);
                if (rc) return rc;
                y = o;
            }
            h->taps["face_decoder_blocks." + std::to_string(b)] = y;
            y = whole(cat[b]);   // torch.cat((x, feats[-1]), dim=1)
        }
        rc = h->add_step(kOut0, 2, sd, y, whole(h->out0));
        if (rc) return rc;
    }
    // ---- output_block.1 (plain conv 32->3) ------------------------------------------------------
    const float* hw_ = find(sd, "output_block.1.weight", 3 * 32);
    const float* hb_ = find(sd, "output_block.1.bias", 3);
    if (!hw_ || !hb_) return MF_ERR_INVALID;
    MF_HIP(hipMalloc(&h->head_w, 96 * sizeof(float)));
    MF_HIP(hipMalloc(&h->head_b, 3 * sizeof(float)));
    MF_HIP(hipMemcpy(h->head_w, hw_, 96 * sizeof(float), hipMemcpyHostToDevice));
    MF_HIP(hipMemcpy(h->head_b, hb_, 3 * sizeof(float), hipMemcpyHostToDevice));
    *out = h.release();
    return MF_OK;
}

extern "C" int mf_wav2lip_tune(mf_wav2lip* h, int batch, void* stream) {
    MF_REQUIRE(h && batch >= 1, "wav2lip_tune: bad argument");
    MF_REQUIRE(batch <= h->cap, "wav2lip_tune: batch %d exceeds the handle's workspace (%d frames): run one forward at this batch first", batch, h->cap);
    return h->tune(batch, (hipStream_t)stream);
}

extern "C" int mf_wav2lip_forward(mf_wav2lip* h, const float* mel, const float* face, float* out, int batch,
                                  void* stream) {
    MF_REQUIRE(h && mel && face && out, "wav2lip_forward: null argument");
    MF_REQUIRE(batch > 0, "wav2lip_forward: batch must be positive (got %d)", batch);
    hipStream_t s = (hipStream_t)stream;
    int rc = h->ensure_capacity(batch);
    if (rc) return rc;
    if ((rc = mf_nchw_to_act(mel, 1, *h->mel_in, batch, s))) return rc;
    if ((rc = mf_nchw_to_act(face, 6, *h->face_in, batch, s))) return rc;
    if ((rc = h->run(batch, s))) return rc;
    return mf_head_1x1_sigmoid(ActView{h->out0, 0, 32}, h->head_w, h->head_b, out, 0, batch, s);
}

extern "C" int mf_wav2lip_forward_u8(mf_wav2lip* h, const float* mel, const uint8_t* faces_u8, float* frames_hwc,
                                     int batch, void* stream) {
    MF_REQUIRE(h && mel && faces_u8 && frames_hwc, "wav2lip_forward_u8: null argument");
    MF_REQUIRE(batch > 0, "wav2lip_forward_u8: batch must be positive (got %d)", batch);
    hipStream_t s = (hipStream_t)stream;
    int rc = h->ensure_capacity(batch);
    if (rc) return rc;
    if ((rc = mf_nchw_to_act(mel, 1, *h->mel_in, batch, s))) return rc;
    if ((rc = mf_faces_u8_to_act(faces_u8, *h->face_in, batch, s))) return rc;
    if ((rc = h->run(batch, s))) return rc;
    return mf_head_1x1_sigmoid(ActView{h->out0, 0, 32}, h->head_w, h->head_b, frames_hwc, 1, batch, s);
}

extern "C" int mf_wav2lip_read_tap(mf_wav2lip* h, const char* tap, float* dst, int batch, void* stream) {
    MF_REQUIRE(h && tap && dst, "wav2lip_read_tap: null argument");
    MF_REQUIRE(batch > 0 && batch <= h->cap, "wav2lip_read_tap: batch %d exceeds the last forward's workspace (%d)", batch, h->cap);
    auto it = h->taps.find(tap);
    MF_REQUIRE(it != h->taps.end(), "wav2lip_read_tap: unknown tap '%s'", tap);
    return mf_act_to_nchw(it->second, dst, batch, (hipStream_t)stream);
}

// ---- measurement seam ----------------------------------------------------------------------------
// kernel launches of a forward in execution order: prep_mel, prep_face, [audio encoder], [face encoder],
// [decoder], head; a split-K layer contributes two rows (the MFMA kernel and its combine pass)
namespace {
const char* kAuxNames[3] = {"prep_mel(nchw->nhwc)", "prep_face(nchw->nhwc)", "output_block.1+sigmoid"};
const char* kAuxKernels[3] = {"k_nchw_to_act", "k_nchw_to_act", "k_head<32>"};

struct Row { int step; int part; };   // step -1/-2/-3 = aux kernels; part 1 = split-K combine
std::vector<Row> launch_rows(const mf_wav2lip* h, int batch) {
    std::vector<Row> rows{{-1, 0}, {-2, 0}};
    for (size_t k = 0; k < h->steps.size(); ++k) {
        rows.push_back({(int)k, 0});
        const ConvPlan& pl = h->steps[k]->plan;
        if (!pl.halo && mf_conv_pick_tile(&pl, batch).nsplit > 1) rows.push_back({(int)k, 1});
    }
    rows.push_back({-3, 0});
    return rows;
}
}  // namespace

extern "C" int mf_wav2lip_num_launches(const mf_wav2lip* h, int batch) {
    return h && batch > 0 ? (int)launch_rows(h, batch).size() : 0;
}

extern "C" int mf_wav2lip_launch_info(const mf_wav2lip* h, int index, int batch, char* name, int name_cap,
                                      char* kernel, int kernel_cap, double* flops) {
    MF_REQUIRE(h && name && kernel && flops && batch > 0, "launch_info: bad argument");
    const std::vector<Row> rows = launch_rows(h, batch);
    MF_REQUIRE(index >= 0 && index < (int)rows.size(), "launch_info: index %d out of range", index);
    const Row r = rows[index];
    if (r.step < 0) {
        const int a = -r.step - 1;
        snprintf(name, name_cap, "%s", kAuxNames[a]);
        snprintf(kernel, kernel_cap, "%s", kAuxKernels[a]);
        *flops = a == 2 ? 2.0 * batch * 96 * 96 * 32 * 3 : 0.0;
        return MF_OK;
    }
    const Step& st = *h->steps[r.step];
    if (r.part == 1) {
        snprintf(name, name_cap, "%s (split-K combine)", st.name.c_str());
        snprintf(kernel, kernel_cap, "k_splitk_epilogue");
        *flops = 0.0;
        return MF_OK;
    }
    snprintf(name, name_cap, "%s", st.name.c_str());
    mf_conv_kernel_name(&st.plan, batch, kernel, kernel_cap);
    *flops = mf_conv_flops(&st.plan, batch);
    return MF_OK;
}

extern "C" int mf_wav2lip_profile(mf_wav2lip* h, const float* mel, const float* face, float* out, int batch,
                                  int iters, float* ms_per_launch, void* stream) {
    MF_REQUIRE(h && mel && face && out && ms_per_launch, "profile: null argument");
    MF_REQUIRE(batch > 0 && iters > 0, "profile: batch and iters must be positive");
    hipStream_t s = (hipStream_t)stream;
    int rc = h->ensure_capacity(batch);
    if (rc) return rc;
    const std::vector<Row> rows = launch_rows(h, batch);
    const int n = (int)rows.size();
    // one event before every launch + one after the last: duration(i) = ev[i] -> ev[i+1]; everything is
    // on ONE stream, strictly serial (the audio lane is not forked here), so each kernel is timed alone
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) MF_HIP(hipEventCreate(&e));
    std::vector<double> acc(n, 0.0);
    for (int it = 0; it < iters; ++it) {
        int i = 0;
        MF_HIP(hipEventRecord(ev[i++], s));
        if ((rc = mf_nchw_to_act(mel, 1, *h->mel_in, batch, s))) return rc;
        MF_HIP(hipEventRecord(ev[i++], s));
        if ((rc = mf_nchw_to_act(face, 6, *h->face_in, batch, s))) return rc;
        for (size_t k = 0; k < h->steps.size(); ++k) {
            Step& st = *h->steps[k];
            MF_HIP(hipEventRecord(ev[i++], s));
            const bool split = i < n && rows[i].part == 1 && rows[i].step == (int)k;
            st.plan.prof_mid = split ? ev[i++] : nullptr;   // recorded between the MFMA kernel and its combine pass
            rc = mf_conv_launch(&st.plan, st.in, st.out, st.res, batch, s);
            st.plan.prof_mid = nullptr;
            if (rc) return rc;
        }
        MF_HIP(hipEventRecord(ev[i++], s));
        if ((rc = mf_head_1x1_sigmoid(ActView{h->out0, 0, 32}, h->head_w, h->head_b, out, 0, batch, s))) return rc;
        MF_HIP(hipEventRecord(ev[i++], s));
        MF_REQUIRE(i == n + 1, "profile: launch table out of sync (%d vs %d)", i, n + 1);
        MF_HIP(hipStreamSynchronize(s));
        for (int j = 0; j < n; ++j) {
            float ms = 0.f;
            MF_HIP(hipEventElapsedTime(&ms, ev[j], ev[j + 1]));
            acc[j] += ms;
        }
    }
    for (int j = 0; j < n; ++j) ms_per_launch[j] = (float)(acc[j] / iters);
    for (auto& e : ev) (void)hipEventDestroy(e);
    return MF_OK;
}

extern "C" void mf_wav2lip_destroy(mf_wav2lip* h) { delete h; }

// MuseTalk UNet (one conditional evaluation at t = 0) and SD-VAE decoder on gfx950 (H4, H5):
// musereal.py:102-108 -> musetalk/models/unet.py:29-44, vae.py:96-108 -> diffusers UNet2DConditionModel /
// AutoencoderKL (un-vendored; architecture per include/merefusion.h mf_unet_config / mf_vae_config).
//
// Both networks are static schedules of the same fused building blocks as the Wav2Lip generator:
//   * every Conv2d / Linear is an MFMA convolution (3x3 halo-tile or implicit GEMM, 1x1 = GEMM) with
//     bias / residual / SiLU epilogues; the t = 0 time embedding is a per-channel constant folded into
//     each resnet's conv1 bias at create time;
//   * `torch.cat([hidden, skip])` of the up path is free: producer layers write channel slices of one buffer;
//   * nearest-2x upsample + conv3x3 runs as 4 sub-pixel phases of pre-summed 2x2 taps (2.25x fewer FLOPs);
//   * attention = grouped (batch x head) GEMMs with device-packed K / V^T operands + fp32 row softmax;
//   * GroupNorm(+SiLU) = fp64 statistics pass + one apply pass.
#include "mf_nn.h"
#include "mf_aux.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

namespace {

typedef std::function<int(int, hipStream_t)> Op;

// GroupNorm -> SiLU -> the f16 + FP6 format in front of a conv of that format.  `fuse` (maps of a multiple of 64 pixels): the conversion kernel forms the per-channel
// affine itself from the statistics -- the k_gn_affine launch in front of each of the VAE decoder's 28 conversions goes (same bits: mf_gn_affine_pair).
// MF_GN_AFFINE_FUSE=0 keeps the two launches (A/B, tests).
static int gn_silu_to_q(const ActView& x, const float* dg, const float* db, int groups, float eps, double* st, float* scale, float* shift, const ActBuf& tq, int B,
                        hipStream_t s, bool have_stats, const float* post, bool fuse) {
    const ActBuf& xb = *x.buf;
    if (fuse && (xb.H * xb.W) % 64 == 0) {
        if (!have_stats) { const int rc = mf_groupnorm_stats(x, groups, st, B, s); if (rc) return rc; }
        return mf_affine_silu_to_act_q(x, scale, shift, 1, tq, B, s, post, st, dg, db, groups, eps);
    }
    const int r1 = mf_groupnorm_affine(x, dg, db, groups, eps, st, scale, shift, B, s, have_stats);
    return r1 ? r1 : mf_affine_silu_to_act_q(x, scale, shift, 1, tq, B, s, post);
}

struct Net {
    int precision = MF_PREC_BF16X3;
    int cap = 1;
    std::map<std::string, const mf_tensor*> sd;
    std::vector<std::unique_ptr<ActBuf>> bufs;
    std::vector<std::unique_ptr<ConvPlan>> plans;
    std::vector<void*> dev;
    std::vector<Op> ops;
    struct OpInfo { std::string name, kernel; double flops_per_frame; };
    std::vector<OpInfo> info;   // parallel to ops (measurement seam)
    std::map<std::tuple<std::string, int, int, int, int>, ActBuf*> scratch;
    std::map<std::tuple<int, int, int, int>, std::pair<ConvPlan*, ConvPlan*>> attn_plans;   // (dh, Tq, Tk, heads)
    double* gn_stats = nullptr;
    static constexpr int GN_MAX_OPS = 96;
    size_t gn_slice = 0;
    int gn_count = 0;
    std::map<int, hipGraphExec_t> graphs;
    hipStream_t cap_stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    bool use_graph = true;
    // implicit-GEMM layers with their bound views: measured launch configurations (mf_conv_tune) on the first forward at a batch size
    struct Tunable { ConvPlan* p; ActView in, out, res; int op; ConvPlan* pq = nullptr; };   // pq: the f16 + FP6 plan that takes the launches of >= q_dual_min() frames (gn_conv)
    std::vector<Tunable> tunables;
    std::set<int> looked_up;                     // (graph-less mode: batch sizes whose table lookup is done)
    // Side branches of the schedule: an op whose result is not needed by its successors in the list -- the hoisted k | v GEMM, a resnet's 1x1
    // shortcut conv -- runs on a second stream beside the chain (a parallel branch of the captured graph) and is joined right before its first
    // consumer.  The UNet's batch-8 launches leave most CUs idle, so the branch costs the chain nothing (Wav2Lip's forked audio encoder is worth
    // 11 % of its step the same way).  MF_UNET_FORK=0: everything in line.
    std::map<int, int> side_ops;                 // op index -> branch number
    std::multimap<int, int> join_before;         // op index -> branch to wait for before that op
    std::vector<hipEvent_t> ev_fork, ev_join;
    hipStream_t side_stream = nullptr;
    bool fork_on = true;
    void mark_side(int op) { side_ops[op] = (int)side_ops.size(); }
    void join_here(int op) { join_before.insert({(int)ops.size(), side_ops.at(op)}); }   // ... before the NEXT op pushed
    // the last conv and the view it filled: a GroupNorm that reads exactly that view next takes its statistics from the conv's launch
    // (ConvPlan::out_stats: epilogue, split-K combine, or a statistics pass behind the conv) instead of running its own pass over the tensor.
    // Cleared by any other op that writes the buffer.
    // (A few recent producers are remembered: a resnet's shortcut conv runs between conv1 and the norm2 that reads conv1's output.)
    struct StatsSrc { ConvPlan* p; ActView v; };
    std::vector<StatsSrc> stats_srcs;
    void stats_forget(const ActBuf* b) {
        stats_srcs.erase(std::remove_if(stats_srcs.begin(), stats_srcs.end(), [b](const StatsSrc& e) { return e.v.buf == b; }), stats_srcs.end());
    }
    void stats_remember(ConvPlan* p, const ActView& v) {
        stats_forget(v.buf);
        stats_srcs.push_back(StatsSrc{p, v});
        if (stats_srcs.size() > 4) stats_srcs.erase(stats_srcs.begin());
    }
    // Cross-attention keys / values depend on the audio tokens only: the k | v projections of EVERY transformer block as one GEMM at the head of
    // the schedule (kv_all = [ctx_len][sum of 2 C]) instead of one small launch per block inside the chain (16 x ~14 us in the UNet at batch 8).
    ActBuf* kv_all = nullptr; int kv_off = 0, kv_op = -1; std::vector<float> kv_w; ActView kv_ctx{}; bool kv_joined = false;
    bool q_allowed = false;     // the f16 + FP6 conv format: the VAE decoder's resnets (set by the builder of a network whose parity was established with it)
    bool q_dual_allowed = false;   // ... and, from q_dual_min() frames per step, the UNet's 320-channel 3x3 convs on its 32 x 32 maps (gn_conv builds both paths, a launch picks by its batch)
    static int q_dual_min() { return 16; }   // frames per step from which the dual-path convs take the f16 + FP6 tile (profiles/r04_unet_q_dual.md)
    // (whole step, same box, bf16x3 -> f16 + FP6 on those twenty layers: 8 frames 18.78 -> 18.78 ms, 16: 32.0 -> 31.6, 24: 45.5 -> 44.7, 32: 58.7 -> 57.4, 64: 109.05 -> 107.66)
    int next_pad_hi = 0;        // consumed by the next conv(): extra zero rows / columns bottom-right (the VAE encoder's Downsample2D)
    std::string err;

    ~Net() {
        for (auto& g : graphs) if (g.second) (void)hipGraphExecDestroy(g.second);
        for (auto& p : plans) mf_conv_plan_destroy(p.get());
        for (auto& t : tails) mf_tail_conv_destroy(t.get());
        for (auto& b : bufs) { if (b->hi) (void)hipFree(b->hi); if (b->lo) (void)hipFree(b->lo); }
        for (void* d : dev) (void)hipFree(d);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        for (hipEvent_t e : ev_fork) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_join) (void)hipEventDestroy(e);
        if (ev_in) (void)hipEventDestroy(ev_in);
        if (ev_out) (void)hipEventDestroy(ev_out);
    }

    // ---- resources ----------------------------------------------------------------------------------------
    ActBuf* buf(int C, int H, int W, int halo) {
        bufs.emplace_back(new ActBuf());
        ActBuf* b = bufs.back().get();
        b->C = (C + 7) / 8 * 8; b->H = H; b->W = W; b->halo = halo;
        const size_t bytes = ((size_t)cap * b->per_batch() + 64) * sizeof(bf16_t);
        if (hipMalloc(&b->hi, bytes) != hipSuccess || hipMemset(b->hi, 0, bytes) != hipSuccess) { err = "hipMalloc failed for an activation buffer"; return nullptr; }
        if (precision == MF_PREC_BF16X3 && (hipMalloc(&b->lo, bytes) != hipSuccess || hipMemset(b->lo, 0, bytes) != hipSuccess)) { err = "hipMalloc failed for an activation buffer"; return nullptr; }
        return b;
    }
    // scratch reused by every block that asks for the same (slot, shape): blocks run one after another
    ActBuf* tmp(const std::string& slot, int C, int H, int W, int halo) {
        auto key = std::make_tuple(slot, C, H, W, halo);
        auto it = scratch.find(key);
        if (it != scratch.end()) return it->second;
        ActBuf* b = buf(C, H, W, halo);
        scratch[key] = b;
        return b;
    }
    float* upload(const float* host, size_t n) {
        float* d = nullptr;
        if (hipMalloc(&d, n * sizeof(float)) != hipSuccess || hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { err = "upload failed"; return nullptr; }
        dev.push_back(d);
        return d;
    }
    const float* T(const std::string& k, int64_t numel) {
        auto it = sd.find(k);
        if (it == sd.end()) { err = "state dict has no tensor '" + k + "'"; return nullptr; }
        int64_t n = 1;
        for (int i = 0; i < it->second->ndim; ++i) n *= it->second->shape[i];
        if (n != numel) { err = "tensor '" + k + "' has " + std::to_string(n) + " elements, expected " + std::to_string(numel); return nullptr; }
        return it->second->data;
    }
    bool has(const std::string& k) const { return sd.count(k) != 0; }
    ConvPlan* new_plan() { plans.emplace_back(new ConvPlan()); return plans.back().get(); }

    void push(const std::string& name, const std::string& kernel, double flops, Op op) {
        ops.push_back(std::move(op));
        info.push_back(OpInfo{name, kernel, flops});
    }

    // ---- ops ----------------------------------------------------------------------------------------------
    // Conv2d / Linear `name` (k x k, stride, pad), optional SiLU etc., optional residual view, optional
    // nearest-2x upsample in front, optional per-channel constant added to the bias, optional input scale.
    // ---- LayerNorm folded into the GEMMs either side of it (ConvArgs::ln_*): per LayerNorm one fp64 (sum, sum of squares) pair per token, zeroed by the
    // statistics-reset kernel at the head of the list, filled by the producing layer's epilogue, consumed by the epilogue of the layer that follows the norm ----
    double* ln_stats = nullptr;              // [ln_total] doubles behind gn_stats (one allocation, one reset launch)
    size_t ln_used = 0, ln_cap_doubles = 0;
    ConvPlan* last_plan = nullptr;           // the plan of the op conv() / linear_raw() pushed last
    const float* ln_fold_g = nullptr; const float* ln_fold_b = nullptr; const double* ln_fold_in = nullptr;   // set for the NEXT conv() / linear_raw() call only
    double* ln_slot(const ActBuf* t) {       // statistics of one LayerNorm over token buffer t
        const size_t n = (size_t)2 * cap * t->H * t->W;
        if (!ln_stats || ln_used + n > ln_cap_doubles) return nullptr;
        double* p = ln_stats + ln_used;
        ln_used += n;
        return p;
    }
    int ln_fold_next(const std::string& name, int C, const double* stats) {
        ln_fold_g = T(name + ".weight", C);
        ln_fold_b = T(name + ".bias", C);
        ln_fold_in = stats;
        return (ln_fold_g && ln_fold_b && stats) ? MF_OK : MF_ERR_INVALID;
    }
    void ln_apply_fold(ConvPlan* p) { p->ln_gamma = ln_fold_g; p->ln_beta = ln_fold_b; }
    void ln_finish_fold(ConvPlan* p) { if (ln_fold_in) { p->ln_in = ln_fold_in; p->ln_eps = 1e-5f; } ln_fold_g = ln_fold_b = nullptr; ln_fold_in = nullptr; }

    int conv(const std::string& name, ActView in, ActView out, int cin, int cout, int k, int stride, int pad, int act,
             ActView res, int upsample = 0, const std::vector<float>* extra_bias = nullptr, float w_scale = 1.f, bool bias = true,
             ConvPlan** plan_out = nullptr) {
        const float* w = T(name + ".weight", (int64_t)cin * cout * k * k);
        const float* b = bias ? T(name + ".bias", cout) : nullptr;
        if (!w || (bias && !b)) return MF_ERR_INVALID;
        std::vector<float> bb(cout, 0.f), ws;
        for (int i = 0; i < cout; ++i) bb[i] = (b ? b[i] : 0.f) + (extra_bias ? (*extra_bias)[i] : 0.f);
        // A 3x3 layer with a channel count that is not a multiple of 4 (the decoder's conv_out, 128 -> 3) would fall to the
        // implicit-GEMM kernel and re-gather its input once per tap; one zero-weight channel makes it eligible for the LDS
        // halo-tile kernel, which reads the input once (330 -> ~90 us at 256x256).
        if (k == 3 && stride == 1 && pad == 1 && !upsample && cout % 4 && cin >= 16 && cin <= 256 && in.buf->H >= 16 && in.buf->W >= 16 &&
            !res.buf && out.coff % 4 == 0 && out.coff + (cout + 3) / 4 * 4 <= out.buf->C) {
            const int cp = (cout + 3) / 4 * 4;
            const size_t row = (size_t)cin * k * k;
            std::vector<float> wp(row * cp, 0.f);
            for (int o = 0; o < cout; ++o)
                for (size_t i = 0; i < row; ++i) wp[row * o + i] = w[row * o + i] * w_scale;
            bb.resize(cp, 0.f);
            ws.swap(wp);
            w = ws.data();
            w_scale = 1.f;
            cout = cp;
            out.C = cp;
        }
        if (w_scale != 1.f) {
            ws.assign(w, w + (size_t)cin * cout * k * k);
            for (auto& v : ws) v *= w_scale;
            w = ws.data();
        }
        mf_conv2d_desc d{};
        d.cin = cin; d.cout = cout; d.kh = d.kw = k; d.stride_h = d.stride_w = stride; d.pad_h = d.pad_w = pad;
        d.act = act; d.residual = res.buf ? 1 : 0; d.in_h = in.buf->H; d.in_w = in.buf->W; d.upsample = upsample;
        d.pad_hi = next_pad_hi; next_pad_hi = 0;
        stats_forget(out.buf);
        ConvPlan* p = new_plan();
        ln_apply_fold(p);
        int rc = mf_conv_plan_create(p, d, w, bb.data(), nullptr, nullptr, nullptr, nullptr, precision);
        ln_finish_fold(p);
        if (rc) return rc;
        if ((rc = mf_conv_bind(p, *in.buf))) return rc;
        last_plan = p;
        if (plan_out) { *plan_out = p; return MF_OK; }     // the caller pushes its own op around this plan (gn_conv)
        char kn[96];
        mf_conv_kernel_name(p, cap, kn, sizeof(kn));
        push(name, kn, mf_conv_flops(p, 1), [p, in, out, res](int B, hipStream_t s) { return mf_conv_launch(p, in, out, res, B, s); });
        tunables.push_back(Tunable{p, in, out, res, (int)ops.size() - 1});
        if (act != 5) stats_remember(p, out);
        return MF_OK;
    }
    // statistics for GroupNorm(groups) of `x` come from its producer's epilogue?
    bool take_stats(const ActView& x, int groups, double* st) {
        if (x.C % groups || groups > 64 || x.C % 8 || x.coff % 8) return false;
        for (size_t i = 0; i < stats_srcs.size(); ++i) {
            const StatsSrc& e = stats_srcs[i];
            if (e.v.buf != x.buf || e.v.coff != x.coff || e.v.C != x.C || e.p->d.cout != x.C || e.p->out_stats) continue;
            e.p->out_stats = st; e.p->out_stats_groups = groups;
            stats_srcs.erase(stats_srcs.begin() + i);
            return true;
        }
        return false;
    }
    int gn_conv(const std::string& gname, const std::string& cname, ActView x, ActBuf* t, ActView out, int cin, int cout, int groups, float eps,
                ActView res, const std::vector<float>* extra_bias = nullptr) {
        const ActView tv{t, 0, cin};
        // f16 + FP6 operand format for the wide 3x3 convs on large maps (MF_CONV_Q=0: bf16x3 everywhere): GroupNorm-apply writes the conv's input in the new
        // format, the conv runs one f16 + half a block-scaled FP6 MFMA per tap where bf16x3 runs three; outputs and residuals stay bf16 (hi, lo).
        static const bool q_on = [] { const char* e = getenv("MF_CONV_Q"); return !e || atoi(e) != 0; }();
        const bool gn_fuse = [] { const char* e = getenv("MF_GN_AFFINE_FUSE"); return !e || atoi(e) != 0; }();      // (read per handle: tests build both)
        // (maps below 64 x 64 -- the 512-channel 32 x 32 levels -- run it with the channel slices split over two workgroups per tile, mf_q_split_count)
        const int q_minpx = 32 * 32;
        // The UNet's 320-channel convs on its 32 x 32 maps (cout = 2.5 tiles of 128): from 16 frames per step (q_dual_min) the f16 + FP6 tile runs them at 640-730 TF where the
        // bf16x3 LDS-weights tile reaches 405-440 (conv alone, 64 frames: 320 -> 320 298 -> 222 us, 640 -> 320 570 -> 412, 960 -> 320 824 -> 599).  A handle serves
        // steps of every size up to its capacity, so BOTH paths are built and a launch picks by its batch.
        // ... and its 640-channel convs on the 16 x 16 maps (one patch per image, five channel tiles): 640 -> 640 360 -> 250 us, 1280 -> 640 651 -> 455, 1920 -> 640 934 -> 675
        // at 64 frames against the bf16x3 implicit GEMM.
        // (bounds as mf_conv_plan_create's q_small / odd_wide predicates: a layer outside them gets no f16 + FP6 plan and must stay on the bf16x3 path -- ADVICE r04:
        // 1280-channel convs on 16 x 16 maps at sample_size 64, or 320-out convs with cin > 1024, failed the whole handle instead)
        const bool q_dual32 = cout >= 256 && cout % 128 != 0 && cout % 64 == 0 && cin <= 1024 && t->H * t->W >= q_minpx;
        const bool q_dual16 = cout >= 512 && cout <= 1024 && cout % 128 == 0 && cin <= 2048 && t->H == 16 && t->W == 16;
        const bool q_dual = q_on && q_dual_allowed && precision == MF_PREC_BF16X3 && (q_dual32 || q_dual16) && cin % 32 == 0 &&
                            t->C == cin && t->H % 16 == 0 && t->W % 16 == 0 && cap >= q_dual_min();
        if (q_dual || (q_on && q_allowed && precision == MF_PREC_BF16X3 && cin % 32 == 0 && cout % 128 == 0 && t->C == cin && t->H * t->W >= q_minpx && cin >= 128 &&
            (int64_t)cap * ((t->H + 15) / 16) * ((t->W + 15) / 16) * (cout / 128) >= 64)) {
            const float* g = T(gname + ".weight", cin);
            const float* b = T(gname + ".bias", cin);
            const float* w = T(cname + ".weight", (int64_t)cin * cout * 9);
            const float* cb = T(cname + ".bias", cout);
            if (!g || !b || !w || !cb) return MF_ERR_INVALID;
            float* dg = upload(g, cin);
            float* db = upload(b, cin);
            if (!dg || !db) return MF_ERR_HIP;
            if (gn_count >= GN_MAX_OPS) { err = "more GroupNorm layers than GN_MAX_OPS"; return MF_ERR_INVALID; }
            double* st = gn_stats + (size_t)(gn_count++) * gn_slice;
            float* aff = nullptr;
            if (hipMalloc(&aff, (size_t)2 * cap * cin * sizeof(float)) != hipSuccess) { err = "hipMalloc failed for a GroupNorm affine"; return MF_ERR_HIP; }
            dev.push_back(aff);
            float *scale = aff, *shift = aff + (size_t)cap * cin;
            std::vector<float> bb(cout);
            for (int i = 0; i < cout; ++i) bb[i] = cb[i] + (extra_bias ? (*extra_bias)[i] : 0.f);
            // Channel equalisation of the MX blocks (MF_Q_EQUALIZE=0: off).  The FP6 correction planes share one E8M0 scale per 32 CHANNELS: a block's scale follows
            // its largest member and a channel more than ~60 x below it loses its correction altogether (e2m3 spans 0.125 ... 7.5) -- with the random-init weights
            // of the test suite all channels are alike, in a trained decoder they are not (tools/vae_stress_probe.py: image L-inf 1.4e-4 -> 1.5e-3 with channel
            // scales over four decades, bf16x3 7e-5 throughout).  Both operands are known per channel at load time: the activation's expected magnitude
            // m_c = E|silu(gamma_c z + beta_c)|, z ~ N(0, 1) (GroupNorm's output is standardised), and the weights' rms over (cout, taps) w_c.  Channel c of the
            // activation is divided by e_c = 2^round(log2 sqrt(m_c / w_c)) (median-normalised) and input channel c of the weights multiplied by it: a power of two
            // on both sides, so the product and every f16 / FP6 split are unchanged and only the composition of the blocks is -- both operands end up with the
            // per-channel profile sqrt(m_c w_c) (the SmoothQuant balance).
            static const bool eq_on = [] { const char* e = getenv("MF_Q_EQUALIZE"); return !e || atoi(e) != 0; }();
            std::vector<float> wq;
            const float* w_use = w;
            float* d_post = nullptr;
            if (eq_on) {
                std::vector<double> tt(cin);
                for (int c = 0; c < cin; ++c) {
                    double m = 0.0, wsum = 0.0;
                    for (int k = 0; k < 64; ++k) {                                        // midpoint rule over z in [-4, 4] against the normal density
                        const double z = -4.0 + (k + 0.5) * 0.125, u = (double)g[c] * z + (double)b[c];
                        m += std::fabs(u / (1.0 + std::exp(-u))) * std::exp(-0.5 * z * z) * 0.125 * 0.3989422804014327;
                    }
                    for (int o = 0; o < cout; ++o)
                        for (int k = 0; k < 9; ++k) { const double v = w[((int64_t)o * cin + c) * 9 + k]; wsum += v * v; }
                    const double wr = std::sqrt(wsum / (9.0 * cout));
                    tt[c] = (m > 1e-30 && wr > 1e-30) ? std::sqrt(m / wr) : 0.0;
                }
                std::vector<double> srt;
                for (double v : tt) if (v > 0.0) srt.push_back(v);
                std::sort(srt.begin(), srt.end());
                const double med = srt.empty() ? 1.0 : srt[srt.size() / 2];
                std::vector<float> post(cin, 1.f);
                bool any = false;
                wq.assign(w, w + (int64_t)cin * cout * 9);
                for (int c = 0; c < cin; ++c) {
                    if (tt[c] <= 0.0) continue;
                    int ex = (int)std::lround(std::log2(tt[c] / med));
                    ex = std::max(-12, std::min(12, ex));
                    if (ex == 0) continue;
                    any = true;
                    post[c] = (float)std::ldexp(1.0, -ex);
                    const float e = (float)std::ldexp(1.0, ex);
                    for (int o = 0; o < cout; ++o)
                        for (int k = 0; k < 9; ++k) wq[((int64_t)o * cin + c) * 9 + k] *= e;
                }
                if (any) {
                    d_post = upload(post.data(), cin);
                    if (!d_post) return MF_ERR_HIP;
                    w_use = wq.data();
                }
            }
            mf_conv2d_desc d{};
            d.cin = cin; d.cout = cout; d.kh = d.kw = 3; d.stride_h = d.stride_w = 1; d.pad_h = d.pad_w = 1; d.residual = res.buf ? 1 : 0; d.in_h = t->H; d.in_w = t->W;
            ConvPlan* p = new_plan();
            p->q_small_maps = q_dual;
            int rc = mf_conv_plan_create(p, d, w_use, bb.data(), nullptr, nullptr, nullptr, nullptr, MF_PREC_F16Q);
            if (rc) return rc;
            if (!p->q) { err = cname + ": no kernel in the f16 + FP6 format for this layer"; return MF_ERR_INVALID; }
            if (q_dual) {
                // the bf16x3 path of the same layer for steps below q_dual_min() frames (its own plan, LDS-weights tile / implicit-GEMM twin as before); both
                // paths read the normalised tensor from `t`, whose second plane holds bf16 lo values or FP6 blocks as the path writes them
                ConvPlan* p3 = nullptr;
                if ((rc = conv(cname, tv, out, cin, cout, 3, 1, 1, 0, res, 0, extra_bias, 1.f, true, &p3))) return rc;
                if ((rc = mf_conv_bind(p, *t))) return rc;
                const ActBuf* tq = t;
                const bool epi = take_stats(x, groups, st);
                push(gname, epi ? "k_gn_apply | k_affine_silu_to_q from q_dual_min frames (statistics from the producer's epilogue)" : "k_gn_stats+k_gn_apply | +k_affine_silu_to_q from q_dual_min frames", 0.0,
                     [=](int B, hipStream_t s) {
                         if (B < q_dual_min()) return mf_groupnorm(x, tv, dg, db, groups, eps, true, st, B, s, epi);
                         return gn_silu_to_q(x, dg, db, groups, eps, st, scale, shift, *tq, B, s, epi, d_post, gn_fuse);
                     });
                char kn[96];
                mf_conv_kernel_name(cap >= q_dual_min() ? p : p3, cap, kn, sizeof(kn));
                push(cname, kn, mf_conv_flops(p3, 1), [=](int B, hipStream_t s) {
                    if (B < q_dual_min()) return mf_conv_launch(p3, tv, out, res, B, s);
                    p->out_stats = p3->out_stats; p->out_stats_groups = p3->out_stats_groups;      // (the consumer GroupNorm asked the remembered plan)
                    return mf_conv_launch(p, tv, out, res, B, s);
                });
                tunables.push_back(Tunable{p3, tv, out, res, (int)ops.size() - 1, p});
                stats_remember(p3, out);
                return MF_OK;
            }
            // (GroupNorm-apply + SiLU + the format conversion INSIDE this conv -- the producer waves rewriting each landed halo image -- was built in round 4,
            // bit-identical and 2 x slower than the pass it replaces, and removed in round 5: profiles/r04_gn_fuse_q.md keeps the measurement)
            const bool epi = take_stats(x, groups, st);
            if ((rc = mf_conv_bind(p, *t))) return rc;
            const ActBuf* tq = t;
            push(gname, epi ? "k_affine_silu_to_q (statistics from the producer's epilogue)" : "k_gn_stats+k_affine_silu_to_q", 0.0, [=](int B, hipStream_t s) {
                return gn_silu_to_q(x, dg, db, groups, eps, st, scale, shift, *tq, B, s, epi, d_post, gn_fuse);
            });
            char kn[96];
            mf_conv_kernel_name(p, cap, kn, sizeof(kn));
            push(cname, kn, mf_conv_flops(p, 1), [=](int B, hipStream_t s) { return mf_conv_launch(p, tv, out, res, B, s); });
            stats_remember(p, out);
            return MF_OK;
        }
        int rc = gn(gname, x, tv, groups, eps, true);
        return rc ? rc : conv(cname, tv, out, cin, cout, 3, 1, 1, 0, res, 0, extra_bias);
    }
    int gn(const std::string& name, ActView in, ActView out, int groups, float eps, bool silu) {
        const float* g = T(name + ".weight", in.C);
        const float* b = T(name + ".bias", in.C);
        if (!g || !b) return MF_ERR_INVALID;
        float* dg = upload(g, in.C);
        float* db = upload(b, in.C);
        if (!dg || !db) return MF_ERR_HIP;
        if (gn_count >= GN_MAX_OPS) { err = "more GroupNorm layers than GN_MAX_OPS"; return MF_ERR_INVALID; }
        double* st = gn_stats + (size_t)(gn_count++) * gn_slice;
        const bool epi = take_stats(in, groups, st);
        stats_forget(out.buf);
        push(name, epi ? "k_gn_apply (statistics from the producer's epilogue)" : "k_gn_stats+k_gn_apply", 0.0,
             [=](int B, hipStream_t s) { return mf_groupnorm(in, out, dg, db, groups, eps, silu, st, B, s, epi); });
        return MF_OK;
    }
    // `gname` (GroupNorm + SiLU) followed by `cname` (Conv2d 3x3 to <= 16 channels) as ONE pass (mf_conv_tail.hip) where that kernel applies, else the
    // two-op chain through a scratch buffer.  MF_TAIL_FUSE=0: always the chain (A/B).
    std::vector<std::unique_ptr<TailConv>> tails;
    int gn_conv_tail(const std::string& gname, const std::string& cname, ActView x, ActView out, int cin, int cout, int groups, float eps) {
        static const bool fuse = !(getenv("MF_TAIL_FUSE") && atoi(getenv("MF_TAIL_FUSE")) == 0);
        // (maps that give a batch of 8 at least 256 patches of 16 x 16: on the UNet's 32 x 32 map the 32 workgroups of the fused kernel measured 63 us against 35)
        if (!fuse || !mf_tail_conv_supported(cin, cout, precision) || x.buf->H % 16 || x.buf->W % 16 || (x.buf->H / 16) * (x.buf->W / 16) < 32) {
            ActBuf* t = buf(cin, x.buf->H, x.buf->W, 1);     // the normalised tensor, only materialised on this path
            if (!t) return MF_ERR_HIP;
            const ActView tv{t, 0, cin};
            const int rc = gn(gname, x, tv, groups, eps, true);
            return rc ? rc : conv(cname, tv, out, cin, cout, 3, 1, 1, 0, ActView{});
        }
        const float* g = T(gname + ".weight", cin);
        const float* b = T(gname + ".bias", cin);
        const float* w = T(cname + ".weight", (int64_t)cout * cin * 9);
        const float* cb = T(cname + ".bias", cout);
        if (!g || !b || !w || !cb) return MF_ERR_INVALID;
        float* dg = upload(g, cin);
        float* db = upload(b, cin);
        if (!dg || !db) return MF_ERR_HIP;
        if (gn_count >= GN_MAX_OPS) { err = "more GroupNorm layers than GN_MAX_OPS"; return MF_ERR_INVALID; }
        double* st = gn_stats + (size_t)(gn_count++) * gn_slice;
        tails.emplace_back(new TailConv());
        TailConv* tp = tails.back().get();
        const int rc = mf_tail_conv_create(tp, w, cb, cin, cout);
        if (rc) return rc;
        const bool epi = take_stats(x, groups, st);
        stats_forget(out.buf);
        push(gname + " + " + cname, epi ? "k_gn_conv3_tail (statistics from the producer's epilogue)" : "k_gn_stats+k_gn_conv3_tail",
             2.0 * 9 * cin * cout * x.buf->H * x.buf->W,
             [=](int B, hipStream_t s) { return mf_gn_conv3_tail(*tp, x, dg, db, groups, eps, true, st, epi, out, B, s); });
        return MF_OK;
    }
    int ln(const std::string& name, ActView in, ActView out) {
        const float* g = T(name + ".weight", in.C);
        const float* b = T(name + ".bias", in.C);
        if (!g || !b) return MF_ERR_INVALID;
        float* dg = upload(g, in.C);
        float* db = upload(b, in.C);
        if (!dg || !db) return MF_ERR_HIP;
        push(name, "k_layernorm", 0.0, [=](int B, hipStream_t s) { return mf_layernorm(in, out, dg, db, 1e-5f, B, s); });
        return MF_OK;
    }
    // softmax(q k^T * dh^-0.5) v for `heads` heads; q / k / v / out are views of contiguous (halo 0) token buffers
    int attention(ActView q, ActView k, ActView v, ActView out, int heads) {
        const int C = q.C, dh = C / heads;
        const int Tq = q.buf->H * q.buf->W, Tk = k.buf->H * k.buf->W;
        if (q.buf->halo || k.buf->halo || v.buf->halo || out.buf->halo || dh % 8 || k.C != C || v.C != C || out.C != C) {
            err = "attention: needs contiguous token buffers and a head dim that is a multiple of 8";
            return MF_ERR_INVALID;
        }
        // one fused kernel (mf_attn.hip) for the UNet's head dims; the five-launch composite below serves the others (the VAE mid-block's dh 512)
        if (mf_attention_supported(dh)) {
            const int prec = precision;
            push("attention " + std::to_string(Tq) + "x" + std::to_string(Tk) + " heads " + std::to_string(heads) + " dh " + std::to_string(dh),
                 "k_attention", 4.0 * Tq * Tk * C, [=](int B, hipStream_t s) { return mf_attention(q, k, v, out, heads, B, prec, s); });
            return MF_OK;
        }
        const int Tk8 = (Tk + 7) / 8 * 8, Tk64 = (Tk + 63) / 64 * 64;
        auto key = std::make_tuple(dh, Tq, Tk, heads);
        if (!attn_plans.count(key)) {
            ConvPlan* ps = new_plan();
            ConvPlan* pv = new_plan();
            int rc;
            if ((rc = mf_gemm_plan_create_grouped(ps, dh, Tk, Tq, cap * heads, precision))) return rc;
            if ((rc = mf_gemm_plan_create_grouped(pv, Tk64, dh, Tq, cap * heads, precision))) return rc;
            // linear rows: the k-group table is just cg*8
            for (ConvPlan* p : {ps, pv}) {
                std::vector<int> goff(p->goff_total);
                for (int g = 0; g < p->goff_total; ++g) goff[g] = (g < p->cin_pad / 8 ? g : 0) * 8;
                MF_HIP(hipMemcpy(p->goff, goff.data(), goff.size() * sizeof(int), hipMemcpyHostToDevice));
                p->bound_in_ld = 1; p->bound_in_wp = 1;
            }
            attn_plans[key] = {ps, pv};
        }
        ConvPlan* ps = attn_plans[key].first;
        ConvPlan* pv = attn_plans[key].second;
        ActBuf* sc = tmp("attn.scores", Tk8, heads, Tq, 0);
        ActBuf* pm = tmp("attn.probs", Tk64, heads, Tq, 0);
        if (!sc || !pm) return MF_ERR_HIP;
        const float scale = 1.0f / std::sqrt((float)dh);
        const bool x3 = precision == MF_PREC_BF16X3;
        push("attention " + std::to_string(Tq) + "x" + std::to_string(Tk) + " heads " + std::to_string(heads) + " dh " + std::to_string(dh),
             "pack+gemm+softmax+pack+gemm", 4.0 * Tq * Tk * C, [=](int B, hipStream_t s) {
            int rc;
            if ((rc = mf_pack_b_grouped(ps, k.buf->hi + k.coff, x3 ? k.buf->lo + k.coff : nullptr, k.buf->per_batch(), dh,
                                        k.buf->C, 1, Tk, dh, B * heads, heads, s))) return rc;
            GroupedGemm g{};
            g.x_hi = q.buf->hi + q.coff; g.x_lo = x3 ? q.buf->lo + q.coff : nullptr;
            g.zx_b = q.buf->per_batch(); g.zx_h = dh; g.x_row = q.buf->C;
            g.y_hi = sc->hi; g.y_lo = sc->lo; g.zy_b = sc->per_batch(); g.zy_h = (int64_t)Tq * sc->C; g.y_row = sc->C;
            g.M = Tq; g.groups = B * heads; g.heads = heads;
            if ((rc = mf_gemm_grouped_launch(ps, g, s))) return rc;
            if ((rc = mf_softmax_rows(ActView{sc, 0, Tk}, ActView{pm, 0, pm->C}, Tk, scale, B, s))) return rc;
            if ((rc = mf_pack_b_grouped(pv, v.buf->hi + v.coff, x3 ? v.buf->lo + v.coff : nullptr, v.buf->per_batch(), dh,
                                        1, v.buf->C, dh, Tk, B * heads, heads, s))) return rc;
            GroupedGemm o{};
            o.x_hi = pm->hi; o.x_lo = pm->lo; o.zx_b = pm->per_batch(); o.zx_h = (int64_t)Tq * pm->C; o.x_row = pm->C;
            o.y_hi = out.buf->hi + out.coff; o.y_lo = x3 ? out.buf->lo + out.coff : nullptr;
            o.zy_b = out.buf->per_batch(); o.zy_h = dh; o.y_row = out.buf->C;
            o.M = Tq; o.groups = B * heads; o.heads = heads;
            return mf_gemm_grouped_launch(pv, o, s);
        });
        return MF_OK;
    }

    // ---- blocks -------------------------------------------------------------------------------------------
    // diffusers ResnetBlock2D; temb_act = silu(time_embedding) on the host (nullptr: VAE)
    int resnet(const std::string& p, ActView x, ActView y, int cin, int cout, int groups, float eps, const std::vector<float>* temb_act) {
        const int H = x.buf->H, W = x.buf->W;
        ActBuf *t1 = tmp("rn.t1", cin, H, W, 1), *c1 = tmp("rn.c1", cout, H, W, 1), *t2 = tmp("rn.t2", cout, H, W, 1);
        if (!t1 || !c1 || !t2) return MF_ERR_HIP;
        int rc;
        std::vector<float> tb;
        if (temb_act && has(p + ".time_emb_proj.weight")) {
            const int td = (int)temb_act->size();
            const float* w = T(p + ".time_emb_proj.weight", (int64_t)cout * td);
            const float* b = T(p + ".time_emb_proj.bias", cout);
            if (!w || !b) return MF_ERR_INVALID;
            tb.resize(cout);
            for (int o = 0; o < cout; ++o) {
                double a = b[o];
                for (int i = 0; i < td; ++i) a += (double)w[(size_t)o * td + i] * (*temb_act)[i];
                tb[o] = (float)a;
            }
        }
        // the 1x1 shortcut reads only the block's input: first in the list, on the side branch, joined where conv2 adds it
        ActView res = x;
        int sc_op = -1;
        if (has(p + ".conv_shortcut.weight")) {
            ActBuf* sc = tmp("rn.sc", cout, H, W, 1);
            if (!sc) return MF_ERR_HIP;
            if ((rc = conv(p + ".conv_shortcut", x, ActView{sc, 0, cout}, cin, cout, 1, 1, 0, 0, ActView{}))) return rc;
            res = ActView{sc, 0, cout};
            sc_op = (int)ops.size() - 1;
            mark_side(sc_op);
        } else if (cin != cout) {
            err = p + ": cin != cout but no conv_shortcut in the state dict";
            return MF_ERR_INVALID;
        }
        if ((rc = gn_conv(p + ".norm1", p + ".conv1", x, t1, ActView{c1, 0, cout}, cin, cout, groups, eps, ActView{}, tb.empty() ? nullptr : &tb))) return rc;
        if (sc_op >= 0) join_here(sc_op);
        return gn_conv(p + ".norm2", p + ".conv2", ActView{c1, 0, cout}, t2, y, cout, cout, groups, eps, res);
    }

    // diffusers Transformer2DModel (conv projections) with one BasicTransformerBlock; ctx = audio tokens
    int transformer(const std::string& p, ActView x, ActView y, int C, int heads, int groups, ActView ctx) {
        const int H = x.buf->H, W = x.buf->W, X = ctx.C;
        ActBuf *g0 = tmp("xf.gn", C, H, W, 0), *hA = tmp("xf.hA", C, H, W, 0), *hB = tmp("xf.hB", C, H, W, 0);
        ActBuf *nb = tmp("xf.ln", C, H, W, 0), *qkv = tmp("xf.qkv", 3 * C, H, W, 0), *ao = tmp("xf.ao", C, H, W, 0);
        ActBuf *kv = tmp("xf.kv", 2 * C, ctx.buf->H, ctx.buf->W, 0), *gg = tmp("xf.geglu", 4 * C, H, W, 0);
        if (!g0 || !hA || !hB || !nb || !qkv || !ao || !kv || !gg) return MF_ERR_HIP;
        const std::string t = p + ".transformer_blocks.0";
        int rc;
        if ((rc = gn(p + ".norm", x, ActView{g0, 0, C}, groups, 1e-6f, false))) return rc;
        if ((rc = conv(p + ".proj_in", ActView{g0, 0, C}, ActView{hA, 0, C}, C, C, 1, 1, 0, 0, ActView{}))) return rc;
        // The three LayerNorms are FOLDED into the GEMMs around them where the statistics buffer has room (it has, for the shipped configurations): the producer of
        // the residual stream (proj_in, attn1.to_out + x, attn2.to_out + x) leaves every token's (sum, sum of squares), the consumer (q | k | v, attn2.to_q, the
        // GEGLU projection) multiplies the RAW stream by gamma-scaled weights and its epilogue applies rstd * (acc - mean * colsum) + (bias + W beta).
        double *st1 = ln_slot(hA), *st2 = ln_slot(hB), *st3 = ln_slot(hA);
        const bool fold = st1 && st2 && st3;
        // self attention: q | k | v in one GEMM (no bias)
        if (fold) { last_plan->ln_out = st1; if ((rc = ln_fold_next(t + ".norm1", C, st1))) return rc; }
        else if ((rc = ln(t + ".norm1", ActView{hA, 0, C}, ActView{nb, 0, C}))) return rc;
        {
            const float *wq = T(t + ".attn1.to_q.weight", (int64_t)C * C), *wk = T(t + ".attn1.to_k.weight", (int64_t)C * C),
                        *wv = T(t + ".attn1.to_v.weight", (int64_t)C * C);
            if (!wq || !wk || !wv) return MF_ERR_INVALID;
            std::vector<float> w((size_t)3 * C * C);
            std::copy(wq, wq + (size_t)C * C, w.begin());
            std::copy(wk, wk + (size_t)C * C, w.begin() + (size_t)C * C);
            std::copy(wv, wv + (size_t)C * C, w.begin() + (size_t)2 * C * C);
            if ((rc = linear_raw(w.data(), nullptr, fold ? ActView{hA, 0, C} : ActView{nb, 0, C}, ActView{qkv, 0, 3 * C}, C, 3 * C, ActView{}))) return rc;
        }
        if ((rc = attention(ActView{qkv, 0, C}, ActView{qkv, C, C}, ActView{qkv, 2 * C, C}, ActView{ao, 0, C}, heads))) return rc;
        if ((rc = conv(t + ".attn1.to_out.0", ActView{ao, 0, C}, ActView{hB, 0, C}, C, C, 1, 1, 0, 0, ActView{hA, 0, C}))) return rc;
        // cross attention over the audio tokens: k | v in one GEMM
        if (fold) { last_plan->ln_out = st2; if ((rc = ln_fold_next(t + ".norm2", C, st2))) return rc; }
        else if ((rc = ln(t + ".norm2", ActView{hB, 0, C}, ActView{nb, 0, C}))) return rc;
        if ((rc = conv(t + ".attn2.to_q", fold ? ActView{hB, 0, C} : ActView{nb, 0, C}, ActView{qkv, 0, C}, C, C, 1, 1, 0, 0, ActView{}, 0, nullptr, 1.f, false))) return rc;
        ActView kk{kv, 0, C}, vv{kv, C, C};
        {
            const float *wk = T(t + ".attn2.to_k.weight", (int64_t)C * X), *wv = T(t + ".attn2.to_v.weight", (int64_t)C * X);
            if (!wk || !wv) return MF_ERR_INVALID;
            if (kv_all && kv_off + 2 * C <= kv_all->C && ctx.buf == kv_ctx.buf) {
                // rows [kv_off, kv_off + 2 C) of the hoisted GEMM (hoist_kv_finish)
                kv_w.insert(kv_w.end(), wk, wk + (size_t)C * X);
                kv_w.insert(kv_w.end(), wv, wv + (size_t)C * X);
                kk = ActView{kv_all, kv_off, C}; vv = ActView{kv_all, kv_off + C, C};
                kv_off += 2 * C;
            } else {
                std::vector<float> w((size_t)2 * C * X);
                std::copy(wk, wk + (size_t)C * X, w.begin());
                std::copy(wv, wv + (size_t)C * X, w.begin() + (size_t)C * X);
                if ((rc = linear_raw(w.data(), nullptr, ctx, ActView{kv, 0, 2 * C}, X, 2 * C, ActView{}))) return rc;
            }
        }
        if (kk.buf == kv_all && kv_op >= 0 && !kv_joined) { join_here(kv_op); kv_joined = true; }
        if ((rc = attention(ActView{qkv, 0, C}, kk, vv, ActView{ao, 0, C}, heads))) return rc;
        if ((rc = conv(t + ".attn2.to_out.0", ActView{ao, 0, C}, ActView{hA, 0, C}, C, C, 1, 1, 0, 0, ActView{hB, 0, C}))) return rc;
        // GEGLU feed-forward
        if (fold) { last_plan->ln_out = st3; if ((rc = ln_fold_next(t + ".norm3", C, st3))) return rc; }
        else if ((rc = ln(t + ".norm3", ActView{hA, 0, C}, ActView{nb, 0, C}))) return rc;
        // GEGLU in the GEMM epilogue (act 5): the 8C-wide projection never reaches HBM, only value * gelu(gate)
        if ((rc = conv(t + ".ff.net.0.proj", fold ? ActView{hA, 0, C} : ActView{nb, 0, C}, ActView{gg, 0, 4 * C}, C, 8 * C, 1, 1, 0, 5, ActView{}))) return rc;
        if ((rc = conv(t + ".ff.net.2", ActView{gg, 0, 4 * C}, ActView{hB, 0, C}, 4 * C, C, 1, 1, 0, 0, ActView{hA, 0, C}))) return rc;
        return conv(p + ".proj_out", ActView{hB, 0, C}, y, C, C, 1, 1, 0, 0, x);
    }

    // diffusers Upsample2D (nearest 2x + conv 3x3) in the f16 + FP6 operand format where the layer fills the chip that way: a converter pass writes
    // the input in the format (identity affine, no SiLU), then four 2 x 2-tap phase launches of the f16 + FP6 halo tile, which also leave the
    // consumer GroupNorm's statistics.  Elsewhere (small maps, other precisions, MF_CONV_Q=0): the 4-phase implicit GEMM on bf16x3.
#ifndef MF_UP_Q_MIN_WG
#define MF_UP_Q_MIN_WG 256            // workgroups per phase launch from which the f16 + FP6 phases beat the bf16x3 implicit GEMM (A/B build with 128 -- the 512-channel 32^2 -> 64^2 upsampler at batch 8 as four 128-workgroup phase launches: 18.29 vs 18.32 ms per step, no gain)
#endif
    int upsample_conv(const std::string& name, ActView x, ActView out, int C) {
        static const bool on = [] { const char* e = getenv("MF_CONV_Q"); return !e || atoi(e) != 0; }();
        const int H = x.buf->H, W = x.buf->W;
        if (!(on && q_allowed && precision == MF_PREC_BF16X3 && C % 128 == 0 && x.C == C && x.coff % 8 == 0 && H * W >= 32 * 32 &&
              (int64_t)cap * ((H + 15) / 16) * ((W + 15) / 16) * (C / 128) >= MF_UP_Q_MIN_WG))
            return conv(name, x, out, C, C, 3, 1, 1, 0, ActView{}, 1);
        const float* w = T(name + ".weight", (int64_t)C * C * 9);
        const float* b = T(name + ".bias", C);
        if (!w || !b) return MF_ERR_INVALID;
        ActBuf* tq = tmp("up.q", C, H, W, 1);
        if (!tq) return MF_ERR_HIP;
        std::vector<float> ones((size_t)cap * C, 1.f), zeros((size_t)cap * C, 0.f);
        float *d1 = upload(ones.data(), ones.size()), *d0 = upload(zeros.data(), zeros.size());
        if (!d1 || !d0) return MF_ERR_HIP;
        mf_conv2d_desc d{};
        d.cin = C; d.cout = C; d.kh = d.kw = 3; d.stride_h = d.stride_w = 1; d.pad_h = d.pad_w = 1; d.in_h = H; d.in_w = W; d.upsample = 1;
        ConvPlan* p = new_plan();
        int rc = mf_conv_plan_create(p, d, w, b, nullptr, nullptr, nullptr, nullptr, MF_PREC_F16Q);
        if (rc) return rc;
        if ((rc = mf_conv_bind(p, *tq))) return rc;
        stats_forget(out.buf);
        const ActBuf* tqc = tq;
        const ActView tv{tq, 0, C};
        push(name + " (input -> f16 + FP6)", "k_affine_silu_to_q (identity)", 0.0, [=](int B, hipStream_t s) { return mf_affine_silu_to_act_q(x, d1, d0, 0, *tqc, B, s); });
        char kn[96];
        mf_conv_kernel_name(p, cap, kn, sizeof(kn));
        push(name, kn, mf_conv_flops(p, 1), [=](int B, hipStream_t s) { return mf_conv_launch(p, tv, out, ActView{}, B, s); });
        stats_remember(p, out);
        return MF_OK;
    }

    // hoisted cross-attention k | v: reserve the op slot before the blocks are built ...
    int hoist_kv_begin(ActView ctx, int total) {
        if (total <= 0) return MF_OK;
        kv_all = buf(total, ctx.buf->H, ctx.buf->W, 0);
        if (!kv_all) return MF_ERR_HIP;
        kv_ctx = ctx; kv_off = 0; kv_w.clear();
        push("cross-attention k | v of every block", "(placeholder)", 0.0, [](int, hipStream_t) { return MF_OK; });
        kv_op = (int)ops.size() - 1;
        mark_side(kv_op);                      // depends on the audio tokens only: a side branch until the first cross-attention
        kv_joined = false;
        return MF_OK;
    }
    // ... and fill it once every block has handed in its rows
    int hoist_kv_finish() {
        if (!kv_all) return MF_OK;
        if (kv_off != kv_all->C) { err = "hoisted k | v width does not match the blocks that were built"; return MF_ERR_INVALID; }
        const int X = kv_ctx.C;
        int rc = linear_raw(kv_w.data(), nullptr, kv_ctx, ActView{kv_all, 0, kv_off}, X, kv_off, ActView{});
        if (rc) return rc;
        ops[kv_op] = std::move(ops.back()); ops.pop_back();
        info[kv_op].kernel = info.back().kernel; info[kv_op].flops_per_frame = info.back().flops_per_frame;
        info[kv_op].name = "fused linear " + std::to_string(X) + "->" + std::to_string(kv_off) + " (cross-attention k | v of every block)";
        info.pop_back();
        tunables.back().op = kv_op;
        kv_w.clear(); kv_w.shrink_to_fit();
        return MF_OK;
    }

    int linear_raw(const float* w, const float* b, ActView in, ActView out, int cin, int cout, ActView res) {
        mf_conv2d_desc d{};
        d.cin = cin; d.cout = cout; d.kh = d.kw = 1; d.stride_h = d.stride_w = 1; d.residual = res.buf ? 1 : 0;
        d.in_h = in.buf->H; d.in_w = in.buf->W;
        ConvPlan* p = new_plan();
        ln_apply_fold(p);
        int rc = mf_conv_plan_create(p, d, w, b, nullptr, nullptr, nullptr, nullptr, precision);
        ln_finish_fold(p);
        if (rc) return rc;
        if ((rc = mf_conv_bind(p, *in.buf))) return rc;
        last_plan = p;
        char kn[96];
        mf_conv_kernel_name(p, cap, kn, sizeof(kn));
        push("fused linear " + std::to_string(cin) + "->" + std::to_string(cout), kn, mf_conv_flops(p, 1),
             [p, in, out, res](int B, hipStream_t s) { return mf_conv_launch(p, in, out, res, B, s); });
        tunables.push_back(Tunable{p, in, out, res, (int)ops.size() - 1});
        return MF_OK;
    }

    // measurement seam: every op alone between two hipEvents on `s`, no graph
    int profile(int B, int iters, float* ms, hipStream_t s) {
        const int n = (int)ops.size();
        std::vector<hipEvent_t> ev(n + 1);
        for (auto& e : ev) MF_HIP(hipEventCreate(&e));
        std::vector<double> acc(n, 0.0);
        for (int it = 0; it < iters; ++it) {
            for (int i = 0; i < n; ++i) {
                MF_HIP(hipEventRecord(ev[i], s));
                int rc = ops[i](B, s);
                if (rc) return rc;
            }
            MF_HIP(hipEventRecord(ev[n], s));
            MF_HIP(hipStreamSynchronize(s));
            for (int i = 0; i < n; ++i) { float t = 0.f; MF_HIP(hipEventElapsedTime(&t, ev[i], ev[i + 1])); acc[i] += t; }
        }
        for (int i = 0; i < n; ++i) ms[i] = (float)(acc[i] / iters);
        for (auto& e : ev) (void)hipEventDestroy(e);
        return MF_OK;
    }
    int op_info(int i, char* name, int ncap, char* kernel, int kcap, double* flops) const {
        MF_REQUIRE(i >= 0 && i < (int)info.size() && name && kernel && flops, "op_info: bad argument");
        snprintf(name, ncap, "%s", info[i].name.c_str());
        snprintf(kernel, kcap, "%s", info[i].kernel.c_str());
        *flops = info[i].flops_per_frame;
        return MF_OK;
    }

    // ---- execution ----------------------------------------------------------------------------------------
    int run_body(int B, hipStream_t s) {
        if (!fork_on || side_ops.empty()) {
            for (auto& op : ops) { int rc = op(B, s); if (rc) return rc; }
            return MF_OK;
        }
        while (ev_fork.size() < side_ops.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            MF_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
            MF_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
            ev_fork.push_back(a); ev_join.push_back(b);
        }
        std::vector<char> open(side_ops.size(), 0);
        for (size_t i = 0; i < ops.size(); ++i) {
            auto jr = join_before.equal_range((int)i);
            for (auto it = jr.first; it != jr.second; ++it)
                if (open[it->second]) { MF_HIP(hipStreamWaitEvent(s, ev_join[it->second], 0)); open[it->second] = 0; }
            auto so = side_ops.find((int)i);
            if (so == side_ops.end()) {
                int rc = ops[i](B, s);
                if (rc) return rc;
                continue;
            }
            const int k = so->second;
            MF_HIP(hipEventRecord(ev_fork[k], s));
            MF_HIP(hipStreamWaitEvent(side_stream, ev_fork[k], 0));
            int rc = ops[i](B, side_stream);
            if (rc) return rc;
            MF_HIP(hipEventRecord(ev_join[k], side_stream));
            open[k] = 1;
        }
        for (size_t k = 0; k < open.size(); ++k)                       // a branch nobody consumed inside the list still ends inside it
            if (open[k]) MF_HIP(hipStreamWaitEvent(s, ev_join[k], 0));
        return MF_OK;
    }
    void name_kernel(const Tunable& t, int B) {
        char kn[96];
        mf_conv_kernel_name(t.pq && B >= q_dual_min() ? t.pq : t.p, B, kn, sizeof(kn));   // the measurement seam names the kernel that actually runs
        info[t.op].kernel = kn;
    }
    // every buffer holds real data (a forward at this batch size has run): time each implicit-GEMM layer's launch configurations in place
    int measure(int B, hipStream_t s) {
        for (auto& t : tunables) {
            int rc = mf_conv_tune(t.p, t.in, t.out, t.res, B, s);
            if (rc) return rc;
            name_kernel(t, B);
        }
        return MF_OK;
    }
    // The explicit warm-up behind mf_unet_tune / mf_vae_tune: measure at batch B on the data the last forward at B left in the buffers, then drop
    // the graph captured with the old configurations (the next forward re-captures).  Seconds per full-size network: call it at start-up for every
    // batch size the serving loop can emit, or ship MF_TUNE_CACHE.
    int tune(int B, hipStream_t s) {
        auto it = graphs.find(B);
        if (use_graph && it == graphs.end()) { mf_set_error("tune: run one forward at batch %d first (the layers are timed on its buffers)", B); return MF_ERR_INVALID; }
        MF_HIP(hipStreamSynchronize(cap_stream));
        MF_HIP(hipStreamSynchronize(s));
        int rc = measure(B, s);
        if (rc) return rc;
        MF_HIP(hipStreamSynchronize(s));
        // the next forward at B runs eagerly again (layers that share a measured signature size their split-K workspaces there), then re-captures
        if (use_graph) { if (it->second) (void)hipGraphExecDestroy(it->second); graphs.erase(it); }
        return MF_OK;
    }
    int run(int B, hipStream_t s) {
        if (!use_graph) {
            if (!looked_up.count(B)) {
                looked_up.insert(B);
                for (auto& t : tunables)
                    if (mf_conv_tune_lookup(t.p, t.in, B)) name_kernel(t, B);
            }
            return run_body(B, s);
        }
        auto it = graphs.find(B);
        if (it == graphs.end()) {                                                        // first call eager
            graphs.emplace(B, nullptr);
            // launch configurations: a table lookup per implicit-GEMM layer (MF_TUNE_CACHE / the shipped table), never a measurement -- a serving loop
            // that meets a new batch size pays one eager forward and one capture, nothing more.  (MF_AUTOTUNE=1, development: measure here.)
            for (auto& t : tunables)
                if (mf_conv_tune_lookup(t.p, t.in, B)) name_kernel(t, B);
            int rc = run_body(B, s);
            if (rc || !mf_autotune_enabled()) return rc;
            if ((rc = measure(B, s))) return rc;
            return run_body(B, s);
        }
        if (!it->second) {
            hipGraph_t graph = nullptr;
            MF_HIP(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
            int rc = run_body(B, cap_stream);
            hipError_t e = hipStreamEndCapture(cap_stream, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            if (e != hipSuccess) { mf_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return MF_ERR_HIP; }
            hipGraphExec_t exec = nullptr;
            e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess) { mf_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return MF_ERR_HIP; }
            it->second = exec;
        }
        // the graph replays on the handle's own stream, fenced by events against the caller's stream: the
        // caller usually hands in the legacy NULL stream, whose implicit ordering a graph launch does not inherit
        MF_HIP(hipEventRecord(ev_in, s));
        MF_HIP(hipStreamWaitEvent(cap_stream, ev_in, 0));
        MF_HIP(hipGraphLaunch(it->second, cap_stream));
        MF_HIP(hipEventRecord(ev_out, cap_stream));
        MF_HIP(hipStreamWaitEvent(s, ev_out, 0));
        return MF_OK;
    }
    int init(const mf_tensor* weights, int n, int prec, int max_batch, int max_groups, int ln_tokens_per_sample = 0) {
        precision = prec; cap = max_batch;
        for (int i = 0; i < n; ++i) {
            if (!weights[i].name || !weights[i].data) { mf_set_error("tensor %d has no name/data", i); return MF_ERR_INVALID; }
            sd[weights[i].name] = &weights[i];
        }
        // one slice of fp64 (sum, sum of squares) per GroupNorm op; the whole array is zeroed by ONE kernel at the head
        // of the op list, so a GroupNorm is two launches (statistics, apply) instead of four
        gn_slice = (size_t)max_batch * max_groups * 2;
        // ... and behind them the LayerNorm statistics: two doubles per token and LayerNorm (ln_slot); the caller says how many token slots per sample its
        // transformer blocks need (mf_unet_create counts them from the config; 0: no folding)
        ln_cap_doubles = (size_t)max_batch * ln_tokens_per_sample * 2;
        MF_HIP(hipMalloc(&gn_stats, (GN_MAX_OPS * gn_slice + ln_cap_doubles) * sizeof(double)));
        dev.push_back(gn_stats);
        ln_stats = gn_stats + GN_MAX_OPS * gn_slice;
        {
            double* st = gn_stats;
            const int n = (int)(GN_MAX_OPS * gn_slice + ln_cap_doubles);
            push("groupnorm / layernorm statistics reset", "k_zero_f64", 0.0, [=](int, hipStream_t s) { return mf_zero_f64(st, n, s); });
        }
        MF_HIP(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
        MF_HIP(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
        MF_HIP(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming));
        MF_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
        // MF_NO_GRAPH=1: every forward an eager launch chain (side branches still forked onto the second stream).  MF_NO_GRAPH=2: eager AND single-stream, no
        // side branches -- the host-lean mode of a serving rank: on ROCm 7.2 any hipGraphLaunch or cross-stream event wait keeps a runtime thread spinning for
        // as long as the GPU is busy (0.8 - 0.9 host core per process, tools/host_wait_probe2.py), a plain launch chain on one stream does not.
        const char* ng = std::getenv("MF_NO_GRAPH");
        use_graph = !(ng && (ng[0] == '1' || ng[0] == '2'));
        fork_on = !(ng && ng[0] == '2');
        return MF_OK;
    }
};

#define NET_TRY(expr)                                                                      \
    do {                                                                                   \
        int rc_ = (expr);                                                                  \
        if (rc_ != MF_OK) {                                                                \
            if (!net.err.empty()) mf_set_error("%s", net.err.c_str());                     \
            return rc_;                                                                    \
        }                                                                                  \
    } while (0)

std::vector<float> silu_vec(const std::vector<float>& v) {
    std::vector<float> o(v.size());
    for (size_t i = 0; i < v.size(); ++i) o[i] = (float)((double)v[i] / (1.0 + std::exp(-(double)v[i])));
    return o;
}

}  // namespace

// ==========================================================================================================
struct mf_unet {
    mf_unet_config cfg{};
    Net net;
    ActBuf* in_lat = nullptr;
    ActBuf* ctx = nullptr;
    ActBuf* out_buf = nullptr;
    float* pe = nullptr;       // [ctx_len][cross_dim] positional encoding (unet.py:12-27)
};

extern "C" int mf_unet_create(const mf_unet_config* c, const mf_tensor* weights, int n_weights, int precision, int max_batch,
                              mf_unet** out) {
    MF_REQUIRE(c && weights && out && n_weights > 0 && max_batch > 0, "unet_create: bad argument");
    MF_REQUIRE(c->n_blocks >= 1 && c->n_blocks <= 4 && c->layers_per_block >= 1, "unet_create: bad block config");
    *out = nullptr;
    std::unique_ptr<mf_unet> h(new mf_unet());
    h->cfg = *c;
    Net& net = h->net;
    // token slots per sample of the folded LayerNorms: three per transformer block, each over the block's map
    int ln_tokens = 0;
    {
        int sz = c->sample_size;
        for (int b = 0; b < c->n_blocks; ++b) {
            if (c->down_attn[b]) ln_tokens += 3 * c->layers_per_block * sz * sz;
            if (b < c->n_blocks - 1) sz /= 2;
        }
        ln_tokens += 3 * sz * sz;                                                  // mid block
        for (int b = 0; b < c->n_blocks; ++b) {
            if (c->up_attn[b]) ln_tokens += 3 * (c->layers_per_block + 1) * sz * sz;
            if (b < c->n_blocks - 1) sz *= 2;
        }
    }
    int rc = net.init(weights, n_weights, precision, max_batch, c->norm_num_groups, ln_tokens);
    if (rc) return rc;
    net.q_dual_allowed = true;   // (tests/test_musetalk_full.py holds the 40-frame handle to the batch-8 parity gate with it)
    const int nb = c->n_blocks, L = c->layers_per_block, G = c->norm_num_groups, heads = c->attention_heads;
    const int* boc = c->block_out_channels;
    const int S = c->sample_size, X = c->cross_attention_dim;

    // ---- time embedding at t = 0 (musereal.py:59): cos(0) = 1 for the first half, sin(0) = 0 for the second -----
    std::vector<float> temb_act;
    {
        const int d0 = boc[0], td = 4 * boc[0];
        const float *w1 = net.T("time_embedding.linear_1.weight", (int64_t)td * d0), *b1 = net.T("time_embedding.linear_1.bias", td);
        const float *w2 = net.T("time_embedding.linear_2.weight", (int64_t)td * td), *b2 = net.T("time_embedding.linear_2.bias", td);
        if (!w1 || !b1 || !w2 || !b2) { mf_set_error("%s", net.err.c_str()); return MF_ERR_INVALID; }
        std::vector<float> e1(td), e2(td);
        for (int o = 0; o < td; ++o) {
            double a = b1[o];
            for (int i = 0; i < d0 / 2; ++i) a += (double)w1[(size_t)o * d0 + i];   // emb = [1]*half + [0]*half
            e1[o] = (float)a;
        }
        e1 = silu_vec(e1);
        for (int o = 0; o < td; ++o) {
            double a = b2[o];
            for (int i = 0; i < td; ++i) a += (double)w2[(size_t)o * td + i] * e1[i];
            e2[o] = (float)a;
        }
        temb_act = silu_vec(e2);
    }

    // ---- positional encoding table -------------------------------------------------------------------------------
    {
        std::vector<float> pe((size_t)c->ctx_len * X);
        for (int t = 0; t < c->ctx_len; ++t)
            for (int i = 0; i < X; i += 2) {
                const float div = std::exp((float)i * (float)(-std::log(10000.0) / X));
                pe[(size_t)t * X + i] = std::sin((float)t * div);
                if (i + 1 < X) pe[(size_t)t * X + i + 1] = std::cos((float)t * div);
            }
        h->pe = net.upload(pe.data(), pe.size());
        if (!h->pe) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
    }

    // ---- skip bookkeeping: skip k is consumed by up-resnet (n_skips-1-k); both live in one concat buffer ------------
    std::vector<int> skip_c, skip_s;   // channels / spatial size of each skip, in production order
    {
        int s = S;
        skip_c.push_back(boc[0]); skip_s.push_back(s);
        for (int b = 0; b < nb; ++b) {
            for (int i = 0; i < L; ++i) { skip_c.push_back(boc[b]); skip_s.push_back(s); }
            if (b < nb - 1) { s /= 2; skip_c.push_back(boc[b]); skip_s.push_back(s); }
        }
    }
    const int n_skips = (int)skip_c.size();
    // hidden channels entering each up-resnet
    std::vector<int> up_h(n_skips), up_out(n_skips);
    {
        int ch = boc[nb - 1], j = 0;
        for (int b = 0; b < nb; ++b)
            for (int i = 0; i < L + 1; ++i, ++j) { up_h[j] = ch; up_out[j] = boc[nb - 1 - b]; ch = boc[nb - 1 - b]; }
    }
    std::vector<ActBuf*> cat(n_skips);   // cat[j] = [hidden | skip] read by up-resnet j
    for (int j = 0; j < n_skips; ++j) {
        const int k = n_skips - 1 - j;
        cat[j] = net.buf(up_h[j] + skip_c[k], skip_s[k], skip_s[k], 1);
        if (!cat[j]) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
    }
    auto skip_view = [&](int k) { const int j = n_skips - 1 - k; return ActView{cat[j], up_h[j], skip_c[k]}; };

    h->in_lat = net.buf(c->in_channels, S, S, 1);
    h->ctx = net.buf(X, 1, c->ctx_len, 0);
    h->out_buf = net.buf(c->out_channels, S, S, 1);
    if (!h->in_lat || !h->ctx || !h->out_buf) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
    const ActView ctx{h->ctx, 0, X};

    {
        int kv_total = 0;
        for (int b = 0; b < nb; ++b) {
            if (c->down_attn[b]) kv_total += L * 2 * boc[b];
            if (c->up_attn[b]) kv_total += (L + 1) * 2 * boc[nb - 1 - b];
        }
        kv_total += 2 * boc[nb - 1];                                   // mid block
        NET_TRY(net.hoist_kv_begin(ctx, kv_total));
    }
    // ---- down path ------------------------------------------------------------------------------------------------
    int k = 0, s = S, ch = boc[0];
    NET_TRY(net.conv("conv_in", ActView{h->in_lat, 0, h->in_lat->C}, skip_view(k), c->in_channels, boc[0], 3, 1, 1, 0, ActView{}));
    ActView x = skip_view(k++);
    for (int b = 0; b < nb; ++b) {
        for (int i = 0; i < L; ++i) {
            const std::string rp = "down_blocks." + std::to_string(b) + ".resnets." + std::to_string(i);
            if (c->down_attn[b]) {
                ActBuf* r = net.tmp("down.r", boc[b], s, s, 1);
                if (!r) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
                NET_TRY(net.resnet(rp, x, ActView{r, 0, boc[b]}, ch, boc[b], G, 1e-5f, &temb_act));
                NET_TRY(net.transformer("down_blocks." + std::to_string(b) + ".attentions." + std::to_string(i),
                                        ActView{r, 0, boc[b]}, skip_view(k), boc[b], heads, G, ctx));
            } else {
                NET_TRY(net.resnet(rp, x, skip_view(k), ch, boc[b], G, 1e-5f, &temb_act));
            }
            ch = boc[b];
            x = skip_view(k++);
        }
        if (b < nb - 1) {
            // Downsample2D: conv 3x3 stride 2 padding 1
            ActView o = skip_view(k);
            NET_TRY(net.conv("down_blocks." + std::to_string(b) + ".downsamplers.0.conv", x, o, ch, ch, 3, 2, 1, 0, ActView{}));
            s /= 2;
            x = skip_view(k++);
        }
    }
    // ---- mid ------------------------------------------------------------------------------------------------------
    {
        ActBuf *m0 = net.buf(ch, s, s, 1), *m1 = net.buf(ch, s, s, 1);
        if (!m0 || !m1) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
        NET_TRY(net.resnet("mid_block.resnets.0", x, ActView{m0, 0, ch}, ch, ch, G, 1e-5f, &temb_act));
        NET_TRY(net.transformer("mid_block.attentions.0", ActView{m0, 0, ch}, ActView{m1, 0, ch}, ch, heads, G, ctx));
        // the second mid resnet writes the hidden slice of the first up-resnet's concat buffer
        NET_TRY(net.resnet("mid_block.resnets.1", ActView{m1, 0, ch}, ActView{cat[0], 0, up_h[0]}, ch, ch, G, 1e-5f, &temb_act));
    }
    // ---- up path --------------------------------------------------------------------------------------------------
    int j = 0;
    ActView last{};
    for (int b = 0; b < nb; ++b) {
        const int co = boc[nb - 1 - b];
        for (int i = 0; i < L + 1; ++i, ++j) {
            const std::string rp = "up_blocks." + std::to_string(b) + ".resnets." + std::to_string(i);
            const int cin = cat[j]->C;
            const bool tail = (i == L);                               // last resnet of the block
            const bool has_up = b < nb - 1;
            // where this sub-block's result goes: next concat buffer (same resolution), or a staging buffer in
            // front of the upsampler, or the final buffer
            ActView dst;
            ActBuf* stage = nullptr;
            if (!tail) dst = ActView{cat[j + 1], 0, up_h[j + 1]};
            else {
                stage = net.buf(co, s, s, 1);
                if (!stage) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
                dst = ActView{stage, 0, co};
            }
            if (c->up_attn[b]) {
                ActBuf* r = net.tmp("up.r", co, s, s, 1);
                if (!r) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
                NET_TRY(net.resnet(rp, ActView{cat[j], 0, cin}, ActView{r, 0, co}, cin, co, G, 1e-5f, &temb_act));
                NET_TRY(net.transformer("up_blocks." + std::to_string(b) + ".attentions." + std::to_string(i), ActView{r, 0, co}, dst, co, heads, G, ctx));
            } else {
                NET_TRY(net.resnet(rp, ActView{cat[j], 0, cin}, dst, cin, co, G, 1e-5f, &temb_act));
            }
            if (tail && has_up) {
                // Upsample2D: nearest 2x + conv 3x3, written into the hidden slice of the next concat buffer
                NET_TRY(net.conv("up_blocks." + std::to_string(b) + ".upsamplers.0.conv", dst, ActView{cat[j + 1], 0, up_h[j + 1]}, co, co, 3, 1, 1, 0, ActView{}, 1));
                s *= 2;
            }
            last = dst;
        }
    }
    {
        NET_TRY(net.gn_conv_tail("conv_norm_out", "conv_out", last, ActView{h->out_buf, 0, c->out_channels}, boc[0], c->out_channels, G, 1e-5f));
    }
    NET_TRY(net.hoist_kv_finish());
    *out = h.release();
    return MF_OK;
}

extern "C" int mf_unet_forward(mf_unet* h, const float* latents, const float* audio, int add_pe, float* out, int batch, void* stream) {
    MF_REQUIRE(h && latents && audio && out, "unet_forward: null argument");
    MF_REQUIRE(batch > 0 && batch <= h->net.cap, "unet_forward: batch %d exceeds the handle's max_batch %d", batch, h->net.cap);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = mf_nchw_to_act(latents, h->cfg.in_channels, *h->in_lat, batch, s))) return rc;
    if ((rc = mf_rows_from_f32(audio, add_pe ? h->pe : nullptr, ActView{h->ctx, 0, h->cfg.cross_attention_dim}, batch, s))) return rc;
    if ((rc = h->net.run(batch, s))) return rc;
    return mf_act_to_nchw(ActView{h->out_buf, 0, h->cfg.out_channels}, out, batch, s);
}

extern "C" int mf_unet_num_ops(const mf_unet* h) { return h ? (int)h->net.ops.size() : 0; }
extern "C" int mf_unet_op_info(const mf_unet* h, int i, char* name, int ncap, char* kernel, int kcap, double* flops_per_frame) {
    MF_REQUIRE(h, "unet_op_info: null handle");
    return h->net.op_info(i, name, ncap, kernel, kcap, flops_per_frame);
}
extern "C" int mf_unet_profile(mf_unet* h, int batch, int iters, float* ms_per_op, void* stream) {
    MF_REQUIRE(h && ms_per_op && batch > 0 && batch <= h->net.cap && iters > 0, "unet_profile: bad argument");
    return h->net.profile(batch, iters, ms_per_op, (hipStream_t)stream);
}
extern "C" int mf_unet_tune(mf_unet* h, int batch, void* stream) {
    MF_REQUIRE(h && batch >= 1 && batch <= h->net.cap, "unet_tune: batch %d exceeds the capacity %d", batch, h ? h->net.cap : 0);
    return h->net.tune(batch, (hipStream_t)stream);
}
extern "C" void mf_unet_destroy(mf_unet* h) { delete h; }

// ==========================================================================================================
struct mf_vae {
    mf_vae_config cfg{};
    Net net;
    ActBuf* in_lat = nullptr;
    ActBuf* out_buf = nullptr;
};

extern "C" int mf_vae_create(const mf_vae_config* c, const mf_tensor* weights, int n_weights, int precision, int max_batch,
                             mf_vae** out) {
    MF_REQUIRE(c && weights && out && n_weights > 0 && max_batch > 0, "vae_create: bad argument");
    MF_REQUIRE(c->n_blocks >= 1 && c->n_blocks <= 4 && c->scaling_factor > 0.f, "vae_create: bad config");
    *out = nullptr;
    std::unique_ptr<mf_vae> h(new mf_vae());
    h->cfg = *c;
    Net& net = h->net;
    int rc = net.init(weights, n_weights, precision, max_batch, c->norm_num_groups);
    if (rc) return rc;
    net.q_allowed = true;        // decoder resnet convs on maps >= 64 x 64 in the f16 + FP6 format (tests/test_musetalk_full.py holds the parity bound with it)
    const int nb = c->n_blocks, L = c->layers_per_block, G = c->norm_num_groups, Z = c->latent_channels;
    const int* boc = c->block_out_channels;
    int s = c->sample_size, ch = boc[nb - 1];
    h->in_lat = net.buf(Z, s, s, 1);
    ActBuf *pq = net.buf(Z, s, s, 1), *x0 = net.buf(ch, s, s, 1), *x1 = net.buf(ch, s, s, 1);
    if (!h->in_lat || !pq || !x0 || !x1) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
    // latents = (1 / scaling_factor) * latents (vae.py:102) folded into post_quant_conv's weights
    NET_TRY(net.conv("post_quant_conv", ActView{h->in_lat, 0, h->in_lat->C}, ActView{pq, 0, Z}, Z, Z, 1, 1, 0, 0, ActView{}, 0, nullptr, 1.f / c->scaling_factor));
    NET_TRY(net.conv("decoder.conv_in", ActView{pq, 0, pq->C}, ActView{x0, 0, ch}, Z, ch, 3, 1, 1, 0, ActView{}));
    NET_TRY(net.resnet("decoder.mid_block.resnets.0", ActView{x0, 0, ch}, ActView{x1, 0, ch}, ch, ch, G, 1e-6f, nullptr));
    {   // single-head attention over all channels, with residual
        const std::string a = "decoder.mid_block.attentions.0";
        ActBuf *g0 = net.tmp("va.gn", ch, s, s, 0), *qkv = net.tmp("va.qkv", 3 * ch, s, s, 0), *ao = net.tmp("va.ao", ch, s, s, 0);
        if (!g0 || !qkv || !ao) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
        NET_TRY(net.gn(a + ".group_norm", ActView{x1, 0, ch}, ActView{g0, 0, ch}, G, 1e-6f, false));
        const float *wq = net.T(a + ".to_q.weight", (int64_t)ch * ch), *wk = net.T(a + ".to_k.weight", (int64_t)ch * ch), *wv = net.T(a + ".to_v.weight", (int64_t)ch * ch);
        const float *bq = net.T(a + ".to_q.bias", ch), *bk = net.T(a + ".to_k.bias", ch), *bv = net.T(a + ".to_v.bias", ch);
        if (!wq || !wk || !wv || !bq || !bk || !bv) { mf_set_error("%s", net.err.c_str()); return MF_ERR_INVALID; }
        std::vector<float> w((size_t)3 * ch * ch), b((size_t)3 * ch);
        std::copy(wq, wq + (size_t)ch * ch, w.begin()); std::copy(wk, wk + (size_t)ch * ch, w.begin() + (size_t)ch * ch);
        std::copy(wv, wv + (size_t)ch * ch, w.begin() + (size_t)2 * ch * ch);
        std::copy(bq, bq + ch, b.begin()); std::copy(bk, bk + ch, b.begin() + ch); std::copy(bv, bv + ch, b.begin() + 2 * ch);
        NET_TRY(net.linear_raw(w.data(), b.data(), ActView{g0, 0, ch}, ActView{qkv, 0, 3 * ch}, ch, 3 * ch, ActView{}));
        NET_TRY(net.attention(ActView{qkv, 0, ch}, ActView{qkv, ch, ch}, ActView{qkv, 2 * ch, ch}, ActView{ao, 0, ch}, 1));
        NET_TRY(net.conv(a + ".to_out.0", ActView{ao, 0, ch}, ActView{x0, 0, ch}, ch, ch, 1, 1, 0, 0, ActView{x1, 0, ch}));
    }
    NET_TRY(net.resnet("decoder.mid_block.resnets.1", ActView{x0, 0, ch}, ActView{x1, 0, ch}, ch, ch, G, 1e-6f, nullptr));
    ActView x{x1, 0, ch};
    for (int b = 0; b < nb; ++b) {
        const int co = boc[nb - 1 - b];
        for (int i = 0; i < L + 1; ++i) {
            ActBuf* y = net.buf(co, s, s, 1);
            if (!y) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
            NET_TRY(net.resnet("decoder.up_blocks." + std::to_string(b) + ".resnets." + std::to_string(i), x, ActView{y, 0, co}, ch, co, G, 1e-6f, nullptr));
            ch = co;
            x = ActView{y, 0, co};
        }
        if (b < nb - 1) {
            ActBuf* y = net.buf(ch, 2 * s, 2 * s, 1);
            if (!y) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
            NET_TRY(net.upsample_conv("decoder.up_blocks." + std::to_string(b) + ".upsamplers.0.conv", x, ActView{y, 0, ch}, ch));
            s *= 2;
            x = ActView{y, 0, ch};
        }
    }
    h->out_buf = net.buf(c->out_channels, s, s, 1);
    if (!h->out_buf) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
    NET_TRY(net.gn_conv_tail("decoder.conv_norm_out", "decoder.conv_out", x, ActView{h->out_buf, 0, c->out_channels}, ch, c->out_channels, G, 1e-6f));
    *out = h.release();
    return MF_OK;
}

extern "C" int mf_vae_decode_latents(mf_vae* h, const float* latents, uint8_t* frames, float* image_f32, int batch, void* stream) {
    MF_REQUIRE(h && latents && (frames || image_f32), "vae_decode_latents: null argument");
    MF_REQUIRE(batch > 0 && batch <= h->net.cap, "vae_decode_latents: batch %d exceeds the handle's max_batch %d", batch, h->net.cap);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = mf_nchw_to_act(latents, h->cfg.latent_channels, *h->in_lat, batch, s))) return rc;
    if ((rc = h->net.run(batch, s))) return rc;
    const ActView o{h->out_buf, 0, h->cfg.out_channels};
    if (image_f32 && (rc = mf_act_to_nchw(o, image_f32, batch, s))) return rc;
    if (frames && (rc = mf_vae_post_u8(o, frames, batch, s))) return rc;
    return MF_OK;
}

extern "C" int mf_vae_num_ops(const mf_vae* h) { return h ? (int)h->net.ops.size() : 0; }
extern "C" int mf_vae_op_info(const mf_vae* h, int i, char* name, int ncap, char* kernel, int kcap, double* flops_per_frame) {
    MF_REQUIRE(h, "vae_op_info: null handle");
    return h->net.op_info(i, name, ncap, kernel, kcap, flops_per_frame);
}
extern "C" int mf_vae_profile(mf_vae* h, int batch, int iters, float* ms_per_op, void* stream) {
    MF_REQUIRE(h && ms_per_op && batch > 0 && batch <= h->net.cap && iters > 0, "vae_profile: bad argument");
    return h->net.profile(batch, iters, ms_per_op, (hipStream_t)stream);
}
extern "C" int mf_vae_tune(mf_vae* h, int batch, void* stream) {
    MF_REQUIRE(h && batch >= 1 && batch <= h->net.cap, "vae_tune: batch %d exceeds the capacity %d", batch, h ? h->net.cap : 0);
    return h->net.tune(batch, (hipStream_t)stream);
}
extern "C" void mf_vae_destroy(mf_vae* h) { delete h; }


// ==========================================================================================================
// sd-vae ENCODER (avatar preparation): `vae.encode(image).latent_dist` of musetalk/models/vae.py:84-94 -- diffusers
// AutoencoderKL.encode = Encoder (conv_in, 4 DownEncoderBlock2D of 2 resnets (+ Downsample2D: F.pad(x, (0, 1, 0, 1)), conv k3 s2 p0),
// mid resnet - attention - resnet, GroupNorm + SiLU, conv_out -> 2 * latent channels) followed by quant_conv; the result holds the
// distribution's (mean | logvar) "moments".  Sampling (mean + std * noise, then * scaling_factor) stays with the caller, who owns the RNG.
struct mf_vae_encoder {
    mf_vae_config cfg{};
    Net net;
    ActBuf* in_img = nullptr;
    ActBuf* out_buf = nullptr;
    int image_size = 0;
};

extern "C" int mf_vae_encoder_create(const mf_vae_config* c, const mf_tensor* weights, int n_weights, int precision, int max_batch,
                                     mf_vae_encoder** out) {
    MF_REQUIRE(c && weights && out && n_weights > 0 && max_batch > 0, "vae_encoder_create: bad argument");
    MF_REQUIRE(c->n_blocks >= 1 && c->n_blocks <= 4, "vae_encoder_create: bad config");
    *out = nullptr;
    std::unique_ptr<mf_vae_encoder> h(new mf_vae_encoder());
    h->cfg = *c;
    Net& net = h->net;
    int rc = net.init(weights, n_weights, precision, max_batch, c->norm_num_groups);
    if (rc) return rc;
    const int nb = c->n_blocks, L = c->layers_per_block, G = c->norm_num_groups, Z = c->latent_channels;
    const int* boc = c->block_out_channels;
    int s = c->sample_size << (nb - 1);                     // image size: the latent grid times 2 per downsampler
    h->image_size = s;
    h->in_img = net.buf(c->out_channels, s, s, 1);          // RGB image, channels padded to 8
    ActBuf* x0 = net.buf(boc[0], s, s, 1);
    if (!h->in_img || !x0) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
    NET_TRY(net.conv("encoder.conv_in", ActView{h->in_img, 0, h->in_img->C}, ActView{x0, 0, boc[0]}, c->out_channels, boc[0], 3, 1, 1, 0, ActView{}));
    ActView x{x0, 0, boc[0]};
    int ch = boc[0];
    for (int b = 0; b < nb; ++b) {
        for (int i = 0; i < L; ++i) {
            ActBuf* y = net.buf(boc[b], s, s, 1);
            if (!y) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
            NET_TRY(net.resnet("encoder.down_blocks." + std::to_string(b) + ".resnets." + std::to_string(i), x, ActView{y, 0, boc[b]}, ch, boc[b], G, 1e-6f, nullptr));
            ch = boc[b];
            x = ActView{y, 0, ch};
        }
        if (b < nb - 1) {
            ActBuf* y = net.buf(ch, s / 2, s / 2, 1);
            if (!y) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
            net.next_pad_hi = 1;                            // F.pad(x, (0, 1, 0, 1)) + conv(k3, s2, p0): the buffer's own zero ring is the pad
            NET_TRY(net.conv("encoder.down_blocks." + std::to_string(b) + ".downsamplers.0.conv", x, ActView{y, 0, ch}, ch, ch, 3, 2, 0, 0, ActView{}));
            s /= 2;
            x = ActView{y, 0, ch};
        }
    }
    ActBuf *m0 = net.buf(ch, s, s, 1), *m1 = net.buf(ch, s, s, 1);
    if (!m0 || !m1) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
    NET_TRY(net.resnet("encoder.mid_block.resnets.0", x, ActView{m0, 0, ch}, ch, ch, G, 1e-6f, nullptr));
    {   // single-head attention over all channels, with residual (as the decoder's mid block)
        const std::string a = "encoder.mid_block.attentions.0";
        ActBuf *g0 = net.tmp("va.gn", ch, s, s, 0), *qkv = net.tmp("va.qkv", 3 * ch, s, s, 0), *ao = net.tmp("va.ao", ch, s, s, 0);
        if (!g0 || !qkv || !ao) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
        NET_TRY(net.gn(a + ".group_norm", ActView{m0, 0, ch}, ActView{g0, 0, ch}, G, 1e-6f, false));
        const float *wq = net.T(a + ".to_q.weight", (int64_t)ch * ch), *wk = net.T(a + ".to_k.weight", (int64_t)ch * ch), *wv = net.T(a + ".to_v.weight", (int64_t)ch * ch);
        const float *bq = net.T(a + ".to_q.bias", ch), *bk = net.T(a + ".to_k.bias", ch), *bv = net.T(a + ".to_v.bias", ch);
        if (!wq || !wk || !wv || !bq || !bk || !bv) { mf_set_error("%s", net.err.c_str()); return MF_ERR_INVALID; }
        std::vector<float> w((size_t)3 * ch * ch), b((size_t)3 * ch);
        std::copy(wq, wq + (size_t)ch * ch, w.begin()); std::copy(wk, wk + (size_t)ch * ch, w.begin() + (size_t)ch * ch);
        std::copy(wv, wv + (size_t)ch * ch, w.begin() + (size_t)2 * ch * ch);
        std::copy(bq, bq + ch, b.begin()); std::copy(bk, bk + ch, b.begin() + ch); std::copy(bv, bv + ch, b.begin() + 2 * ch);
        NET_TRY(net.linear_raw(w.data(), b.data(), ActView{g0, 0, ch}, ActView{qkv, 0, 3 * ch}, ch, 3 * ch, ActView{}));
        NET_TRY(net.attention(ActView{qkv, 0, ch}, ActView{qkv, ch, ch}, ActView{qkv, 2 * ch, ch}, ActView{ao, 0, ch}, 1));
        NET_TRY(net.conv(a + ".to_out.0", ActView{ao, 0, ch}, ActView{m1, 0, ch}, ch, ch, 1, 1, 0, 0, ActView{m0, 0, ch}));
    }
    NET_TRY(net.resnet("encoder.mid_block.resnets.1", ActView{m1, 0, ch}, ActView{m0, 0, ch}, ch, ch, G, 1e-6f, nullptr));
    ActBuf *t = net.buf(ch, s, s, 1), *co = net.buf(2 * Z, s, s, 1);
    h->out_buf = net.buf(2 * Z, s, s, 1);
    if (!t || !co || !h->out_buf) { mf_set_error("%s", net.err.c_str()); return MF_ERR_HIP; }
    NET_TRY(net.gn("encoder.conv_norm_out", ActView{m0, 0, ch}, ActView{t, 0, ch}, G, 1e-6f, true));
    NET_TRY(net.conv("encoder.conv_out", ActView{t, 0, ch}, ActView{co, 0, 2 * Z}, ch, 2 * Z, 3, 1, 1, 0, ActView{}));
    NET_TRY(net.conv("quant_conv", ActView{co, 0, 2 * Z}, ActView{h->out_buf, 0, 2 * Z}, 2 * Z, 2 * Z, 1, 1, 0, 0, ActView{}));
    *out = h.release();
    return MF_OK;
}

extern "C" int mf_vae_encode(mf_vae_encoder* h, const float* image, const uint8_t* image_u8_bgr, int half_mask, float* moments, int batch, void* stream) {
    MF_REQUIRE(h && moments && (image || image_u8_bgr) && !(image && image_u8_bgr), "vae_encode: give exactly one of image / image_u8_bgr, and moments");
    MF_REQUIRE(batch > 0 && batch <= h->net.cap, "vae_encode: batch %d exceeds the handle's max_batch %d", batch, h->net.cap);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (image) { if ((rc = mf_nchw_to_act(image, h->cfg.out_channels, *h->in_img, batch, s))) return rc; }
    else if ((rc = mf_vae_image_u8_to_act(image_u8_bgr, *h->in_img, half_mask, batch, s))) return rc;
    if ((rc = h->net.run(batch, s))) return rc;
    return mf_act_to_nchw(ActView{h->out_buf, 0, 2 * h->cfg.latent_channels}, moments, batch, s);
}

extern "C" int mf_vae_encoder_image_size(const mf_vae_encoder* h) { return h ? h->image_size : 0; }
extern "C" void mf_vae_encoder_destroy(mf_vae_encoder* h) { delete h; }

// C ABI of the single fused convolution layer (mf_conv2d_*): NCHW fp32 in/out around the
// padded-NHWC MFMA kernel.  This is the per-geometry test seam and the building block the
// network schedules (mf_wav2lip.hip) are made of.
#include "mf_conv.h"
#include "mf_aux.h"
#include <memory>
#include <algorithm>

static int out_channels(const mf_conv2d_desc& d) { return d.act == 5 ? d.cout / 2 : d.cout; }   // GEGLU halves the channels

struct mf_conv2d {
    ConvPlan plan;
    ActBuf in, out;
    int cap = 0;
    ~mf_conv2d() {
        mf_conv_plan_destroy(&plan);
        for (ActBuf* b : {&in, &out}) {
            if (b->hi) (void)hipFree(b->hi);
            if (b->lo) (void)hipFree(b->lo);
        }
    }
};

extern "C" int mf_conv2d_create(const mf_conv2d_desc* desc, const float* weight, const float* bias,
                                const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                                const float* bn_var, int precision, mf_conv2d** out) {
    MF_REQUIRE(desc && weight && out, "conv2d_create: null argument");
    MF_REQUIRE((bn_gamma != nullptr) == (bn_beta != nullptr) && (bn_gamma != nullptr) == (bn_mean != nullptr) &&
               (bn_gamma != nullptr) == (bn_var != nullptr), "conv2d_create: BatchNorm tensors must be all set or all NULL");
    *out = nullptr;
    std::unique_ptr<mf_conv2d> h(new mf_conv2d());
    int rc = mf_conv_plan_create(&h->plan, *desc, weight, bias, bn_gamma, bn_beta, bn_mean, bn_var, precision);
    if (rc) return rc;
    if (desc->residual)
        MF_REQUIRE(desc->cin == desc->cout && h->plan.out_h == desc->in_h && h->plan.out_w == desc->in_w,
                   "conv2d_create: residual needs matching input/output shapes");
    MF_REQUIRE(precision != MF_PREC_F16Q || (h->plan.q && !desc->residual), "conv2d_create: MF_PREC_F16Q serves wide 3x3 stride-1 layers only (cin %% 32 == 0, cout %% 128 == 0, no residual from the input)");
    h->in.C = h->plan.q ? (h->plan.cin_pad + 31) / 32 * 32 : h->plan.cin_pad; h->in.H = desc->in_h; h->in.W = desc->in_w;
    h->in.halo = std::max(1, h->plan.in_halo_need);
    h->out.C = (out_channels(*desc) + 7) / 8 * 8; h->out.H = h->plan.out_h; h->out.W = h->plan.out_w; h->out.halo = 1;
    if ((rc = mf_conv_bind(&h->plan, h->in))) return rc;
    *out = h.release();
    return MF_OK;
}

extern "C" int mf_conv2d_forward(mf_conv2d* h, const float* x, float* y, int batch, void* stream) {
    MF_REQUIRE(h && x && y, "conv2d_forward: null argument");
    MF_REQUIRE(batch > 0, "conv2d_forward: batch must be positive (got %d)", batch);
    hipStream_t s = (hipStream_t)stream;
    if (batch > h->cap) {
        MF_HIP(hipDeviceSynchronize());
        for (ActBuf* b : {&h->in, &h->out}) {
            if (b->hi) (void)hipFree(b->hi);
            if (b->lo) (void)hipFree(b->lo);
            b->hi = b->lo = nullptr;
            const size_t bytes = ((size_t)batch * b->per_batch() + 64) * sizeof(bf16_t);
            MF_HIP(hipMalloc(&b->hi, bytes));
            MF_HIP(hipMemset(b->hi, 0, bytes));
            if (h->plan.precision != MF_PREC_BF16) {
                MF_HIP(hipMalloc(&b->lo, bytes));
                MF_HIP(hipMemset(b->lo, 0, bytes));
            }
        }
        MF_HIP(hipDeviceSynchronize());
        h->cap = batch;
    }
    int rc;
    if (h->plan.q) { if ((rc = mf_nchw_to_act_q(x, h->plan.d.cin, h->in, batch, s))) return rc; }
    else if ((rc = mf_nchw_to_act(x, h->plan.d.cin, h->in, batch, s))) return rc;
    ActView in{&h->in, 0, h->in.C}, out{&h->out, 0, out_channels(h->plan.d)};
    ActView res = h->plan.d.residual ? ActView{&h->in, 0, h->plan.d.cout} : ActView{};
    if ((rc = mf_conv_launch(&h->plan, in, out, res, batch, s))) return rc;
    return mf_act_to_nchw(out, y, batch, s);
}

// Test seam of ConvPlan::out_stats: the layer as the producer of a GroupNorm(groups) -- the launch also leaves the (sum, sum of squares) of
// the stored output per (sample, group) in stats[batch][groups][2] (device fp64), whichever way the chosen kernel configuration provides them.
extern "C" int mf_conv2d_forward_stats(mf_conv2d* h, const float* x, float* y, int groups, double* stats, int batch, void* stream) {
    MF_REQUIRE(h && stats && groups > 0 && groups <= 64 && h->plan.d.act != 5 && h->plan.d.cout % groups == 0 && h->plan.d.cout % 8 == 0,
               "conv2d_forward_stats: needs cout %% groups == 0, cout %% 8 == 0, groups <= 64, no GEGLU");
    MF_HIP(hipMemsetAsync(stats, 0, (size_t)batch * groups * 2 * sizeof(double), (hipStream_t)stream));
    h->plan.out_stats = stats; h->plan.out_stats_groups = groups;
    const int rc = mf_conv2d_forward(h, x, y, batch, stream);
    h->plan.out_stats = nullptr; h->plan.out_stats_groups = 0;
    return rc;
}

extern "C" int mf_conv2d_time(mf_conv2d* h, int batch, int iters, float* ms, void* stream) {
    MF_REQUIRE(h && ms, "conv2d_time: null argument");
    MF_REQUIRE(batch > 0 && batch <= h->cap && iters > 0, "conv2d_time: run mf_conv2d_forward at this batch first");
    hipStream_t s = (hipStream_t)stream;
    ActView in{&h->in, 0, h->in.C}, out{&h->out, 0, out_channels(h->plan.d)};
    ActView res = h->plan.d.residual ? ActView{&h->in, 0, h->plan.d.cout} : ActView{};
    hipEvent_t e0, e1;
    MF_HIP(hipEventCreate(&e0)); MF_HIP(hipEventCreate(&e1));
    MF_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) {
        int rc = mf_conv_launch(&h->plan, in, out, res, batch, s);
        if (rc) return rc;
    }
    MF_HIP(hipEventRecord(e1, s));
    MF_HIP(hipEventSynchronize(e1));
    float t = 0.f;
    MF_HIP(hipEventElapsedTime(&t, e0, e1));
    *ms = t / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return MF_OK;
}

extern "C" int mf_conv2d_out_shape(const mf_conv2d* h, int* out_h, int* out_w) {
    MF_REQUIRE(h && out_h && out_w, "conv2d_out_shape: null argument");
    *out_h = h->plan.out_h; *out_w = h->plan.out_w;
    return MF_OK;
}

extern "C" void mf_conv2d_destroy(mf_conv2d* h) { delete h; }

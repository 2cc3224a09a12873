// A small static-graph executor for the plain CNNs of avatar preparation: the S3FD face detector
// (face_detection/detection/sfd/net_s3fd.py:22-129, used by genavatar.py:61-99 and musetalk/utils/preprocessing.py:23,63,104) and the
// BiSeNet face parser (musetalk/utils/face_parsing/model.py:236-262 + resnet.py, called at mere_musetalk.py:250-317).
//
// The reference defines these networks in Python, so the host side (mere-fusion_amd/avatar/*.py) walks the same module trees and emits ops
// through this C ABI; everything numeric runs here.  Convolutions are the MFMA kernels of the hot path (BatchNorm folded, bias / ReLU / sigmoid /
// residual epilogues, torch.cat as channel slices of one buffer); the rest are one-pass HBM kernels on the padded NHWC (hi, lo) planes:
//   max-pool, per-pixel L2Norm, global average pool, y = x * s[b][c] + t (+ v[b][c]) (channel attention, broadcast adds), nearest upsample,
//   bilinear resize with align_corners (the parser's output head).
// The op list is replayed as a hipGraph per batch size like the other stages.
#include "mf_nn.h"
#include "mf_aux.h"
#include <cfloat>
#include <cmath>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

__device__ __forceinline__ float nbf(uint32_t h16) { return __uint_as_float(h16 << 16); }
__device__ __forceinline__ uint32_t nfb(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
struct Pl {   // one buffer's geometry for the elementwise kernels
    bf16_t* hi; bf16_t* lo; int C, H, W, halo;
    __device__ int64_t at(int b, int y, int x) const { return (((int64_t)b * (H + 2 * halo) + y + halo) * (W + 2 * halo) + x + halo) * C; }
    __device__ float ld(int64_t o) const { float v = nbf(hi[o]); if (lo) v += nbf(lo[o]); return v; }
    __device__ void st(int64_t o, float v) const { const uint32_t h = nfb(v); hi[o] = (bf16_t)h; if (lo) lo[o] = (bf16_t)nfb(v - nbf(h)); }
};
Pl pl_of(const ActBuf& b) { return Pl{b.hi, b.lo, b.C, b.H, b.W, b.halo}; }

// F.max_pool2d(x, k, s, p): taps outside the image are skipped (== -inf padding)
__global__ __launch_bounds__(256) void k_maxpool(Pl X, Pl Y, int k, int s, int p, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % Y.C;
    int64_t t = idx / Y.C;
    const int ox = t % Y.W; t /= Y.W;
    const int oy = t % Y.H;
    const int b = t / Y.H;
    float m = -FLT_MAX;
    for (int dy = 0; dy < k; ++dy) {
        const int iy = oy * s - p + dy;
        if (iy < 0 || iy >= X.H) continue;
        for (int dx = 0; dx < k; ++dx) {
            const int ix = ox * s - p + dx;
            if (ix < 0 || ix >= X.W) continue;
            m = fmaxf(m, X.ld(X.at(b, iy, ix) + c));
        }
    }
    Y.st(Y.at(b, oy, ox) + c, m);
}

// L2Norm (net_s3fd.py:6-19): x / (sqrt(sum_c x^2) + eps) * weight[c]; one wave per pixel
__global__ __launch_bounds__(256) void k_l2norm(Pl X, Pl Y, const float* __restrict__ w, float eps, int C, int64_t pixels) {
    const int64_t px = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (px >= pixels) return;
    const int x = px % X.W;
    int64_t t = px / X.W;
    const int y = t % X.H;
    const int b = t / X.H;
    const int64_t xo = X.at(b, y, x), yo = Y.at(b, y, x);
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = X.ld(xo + c); q += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float inv = 1.f / (sqrtf(q) + eps);
    for (int c = lane; c < C; c += 64) Y.st(yo + c, X.ld(xo + c) * inv * w[c]);
}

// F.avg_pool2d(x, x.size()[2:]): one workgroup per (batch, 64-channel block), 4 pixel lanes per channel
__global__ __launch_bounds__(256) void k_gap(Pl X, int coff, int C, Pl Y) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int p = q; p < X.H * X.W; p += 4) s += X.ld(X.at(b, p / X.W, p % X.W) + coff + c);
    part[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < C) Y.st(Y.at(b, 0, 0) + c, (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) / (float)(X.H * X.W));
}

// out = x * s[b][c] + t + v[b][c]   (s, v: 1 x 1 maps or null; t: a map of x's size or null)
__global__ __launch_bounds__(256) void k_scale_add(Pl X, int xoff, Pl S, Pl T, int toff, Pl V, Pl Y, int yoff, int C, int has_s, int has_t, int has_v, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % C;
    int64_t t = idx / C;
    const int x = t % X.W; t /= X.W;
    const int y = t % X.H;
    const int b = t / X.H;
    float v = X.ld(X.at(b, y, x) + xoff + c);
    if (has_s) v *= S.ld(S.at(b, 0, 0) + c);
    if (has_t) v += T.ld(T.at(b, y, x) + toff + c);
    if (has_v) v += V.ld(V.at(b, 0, 0) + c);
    Y.st(Y.at(b, y, x) + yoff + c, v);
}

// F.interpolate(x, (H, W), mode='nearest'): src = floor(dst * in / out)
__global__ __launch_bounds__(256) void k_up_nearest(Pl X, Pl Y, int C, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % C;
    int64_t t = idx / C;
    const int x = t % Y.W; t /= Y.W;
    const int y = t % Y.H;
    const int b = t / Y.H;
    const int sy = min((int)floorf(y * ((float)X.H / Y.H)), X.H - 1), sx = min((int)floorf(x * ((float)X.W / Y.W)), X.W - 1);
    const int64_t so = X.at(b, sy, sx) + c, d = Y.at(b, y, x) + c;
    Y.hi[d] = X.hi[so];
    if (Y.lo) Y.lo[d] = X.lo[so];
}

// F.interpolate(x, (H, W), mode='bilinear', align_corners=True) -> fp32 NCHW (aten's upsample_bilinear2d: scale = (in - 1) / (out - 1),
// src = scale * dst, lambda in fp32)
__global__ __launch_bounds__(256) void k_bilinear_ac(Pl X, int coff, int C, float* __restrict__ dst, int H, int W, float sh, float sw, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = idx % W;
    int64_t t = idx / W;
    const int y = t % H; t /= H;
    const int c = t % C;
    const int b = t / C;
    const float fy = sh * y, fx = sw * x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < X.H - 1 ? 1 : 0), x1 = x0 + (x0 < X.W - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const float v00 = X.ld(X.at(b, y0, x0) + coff + c), v01 = X.ld(X.at(b, y0, x1) + coff + c);
    const float v10 = X.ld(X.at(b, y1, x0) + coff + c), v11 = X.ld(X.at(b, y1, x1) + coff + c);
    dst[idx] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

// net_s3fd.py:123-126 "max-out background label": [B, 4, h, w] -> [B, 2, h, w] = (max(c0, c1, c2), c3)
__global__ __launch_bounds__(256) void k_maxout_bg(const float* __restrict__ src, float* __restrict__ dst, int hw, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int p = idx % hw;
    const int64_t b = idx / hw;
    const float* s = src + b * 4 * hw + p;
    dst[b * 2 * hw + p] = fmaxf(fmaxf(s[0], s[hw]), s[2 * hw]);
    dst[b * 2 * hw + hw + p] = s[3 * hw];
}

}  // namespace

struct mf_net {
    int precision = MF_PREC_BF16X3, cap = 1;
    std::vector<std::unique_ptr<ActBuf>> bufs;
    std::vector<std::unique_ptr<ConvPlan>> plans;
    std::vector<float*> dev;
    typedef std::function<int(int, hipStream_t)> Op;
    std::vector<Op> ops;
    struct Tunable { ConvPlan* p; ActView in, out, res; };
    std::vector<Tunable> tunables;
    std::vector<std::string> names;
    std::vector<double> flops;
    std::map<int, hipGraphExec_t> graphs;
    hipStream_t cap_stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    bool use_graph = true;

    ~mf_net() {
        for (auto& g : graphs) if (g.second) (void)hipGraphExecDestroy(g.second);
        for (auto& p : plans) mf_conv_plan_destroy(p.get());
        for (auto& b : bufs) { if (b->hi) (void)hipFree(b->hi); if (b->lo) (void)hipFree(b->lo); }
        for (float* d : dev) (void)hipFree(d);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (ev_in) (void)hipEventDestroy(ev_in);
        if (ev_out) (void)hipEventDestroy(ev_out);
    }
    ActBuf* B(int id) { return id >= 0 && id < (int)bufs.size() ? bufs[id].get() : nullptr; }
    int run_body(int batch, hipStream_t s) {
        for (auto& op : ops) { int rc = op(batch, s); if (rc) return rc; }
        return MF_OK;
    }
};

#define NET_BUF(var, id) ActBuf* var = h->B(id); MF_REQUIRE(var, "net: no buffer %d", id)

extern "C" int mf_net_create(int max_batch, int precision, mf_net** out) {
    MF_REQUIRE(out && max_batch >= 1 && max_batch <= 256, "net_create: bad argument");
    MF_REQUIRE(precision == MF_PREC_BF16 || precision == MF_PREC_BF16X3, "net_create: unknown precision %d", precision);
    std::unique_ptr<mf_net> h(new mf_net());
    h->precision = precision; h->cap = max_batch;
    MF_HIP(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    MF_HIP(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
    MF_HIP(hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
    { const char* e = getenv("MF_NO_GRAPH"); h->use_graph = !(e && atoi(e) != 0); }
    *out = h.release();
    return MF_OK;
}

extern "C" int mf_net_buffer(mf_net* h, int C, int H, int W, int halo) {
    MF_REQUIRE(h && C > 0 && H > 0 && W > 0 && halo >= 0, "net_buffer: bad geometry");
    h->bufs.emplace_back(new ActBuf());
    ActBuf* b = h->bufs.back().get();
    b->C = (C + 7) / 8 * 8; b->H = H; b->W = W; b->halo = halo;
    const size_t bytes = ((size_t)h->cap * b->per_batch() + 64) * sizeof(bf16_t);
    if (hipMalloc(&b->hi, bytes) != hipSuccess || hipMemset(b->hi, 0, bytes) != hipSuccess) { mf_set_error("net_buffer: hipMalloc of %zu bytes failed", bytes); return MF_ERR_HIP; }
    if (h->precision == MF_PREC_BF16X3 && (hipMalloc(&b->lo, bytes) != hipSuccess || hipMemset(b->lo, 0, bytes) != hipSuccess)) { mf_set_error("net_buffer: hipMalloc failed"); return MF_ERR_HIP; }
    return (int)h->bufs.size() - 1;
}

extern "C" int mf_net_conv(mf_net* h, const mf_conv2d_desc* d, const float* weight, const float* bias, const float* bn_gamma, const float* bn_beta,
                           const float* bn_mean, const float* bn_var, int in_buf, int in_coff, int out_buf, int out_coff, int res_buf, int res_coff,
                           const char* name) {
    MF_REQUIRE(h && d && weight, "net_conv: null argument");
    NET_BUF(ib, in_buf); NET_BUF(ob, out_buf);
    ActBuf* rb = res_buf >= 0 ? h->B(res_buf) : nullptr;
    MF_REQUIRE(res_buf < 0 || rb, "net_conv: no residual buffer %d", res_buf);
    mf_conv2d_desc dd = *d;
    dd.in_h = ib->H; dd.in_w = ib->W; dd.residual = rb ? (d->residual ? d->residual : 1) : 0;
    std::vector<float> zero;
    if (!bias) { zero.assign(dd.cout, 0.f); bias = zero.data(); }
    h->plans.emplace_back(new ConvPlan());
    ConvPlan* p = h->plans.back().get();
    int rc = mf_conv_plan_create(p, dd, weight, bias, bn_gamma, bn_beta, bn_mean, bn_var, h->precision);
    if (rc) return rc;
    if ((rc = mf_conv_bind(p, *ib))) return rc;
    const ActView in{ib, in_coff, (dd.cin + 7) / 8 * 8 <= ib->C - in_coff ? (dd.cin + 7) / 8 * 8 : dd.cin}, out{ob, out_coff, dd.cout};
    const ActView res = rb ? ActView{rb, res_coff, dd.cout} : ActView{};
    h->ops.push_back([p, in, out, res](int B, hipStream_t s) { return mf_conv_launch(p, in, out, res, B, s); });
    h->tunables.push_back(mf_net::Tunable{p, in, out, res});
    h->names.push_back(name ? name : "conv");
    h->flops.push_back(mf_conv_flops(p, 1));
    return MF_OK;
}

extern "C" int mf_net_maxpool(mf_net* h, int in_buf, int out_buf, int k, int stride, int pad) {
    MF_REQUIRE(h && k >= 1 && stride >= 1 && pad >= 0 && 2 * pad <= k, "net_maxpool: bad window");
    NET_BUF(ib, in_buf); NET_BUF(ob, out_buf);
    MF_REQUIRE(ob->C == ib->C && ob->H == (ib->H + 2 * pad - k) / stride + 1 && ob->W == (ib->W + 2 * pad - k) / stride + 1,
               "net_maxpool: output buffer %dx%dx%d does not match floor((%dx%d + 2*%d - %d) / %d) + 1", ob->C, ob->H, ob->W, ib->H, ib->W, pad, k, stride);
    h->ops.push_back([ib, ob, k, stride, pad](int B, hipStream_t s) {
        const int64_t total = (int64_t)B * ob->H * ob->W * ob->C;
        hipLaunchKernelGGL(k_maxpool, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pl_of(*ib), pl_of(*ob), k, stride, pad, total);
        MF_HIP(hipGetLastError());
        return MF_OK;
    });
    h->names.push_back("maxpool"); h->flops.push_back(0.0);
    return MF_OK;
}

extern "C" int mf_net_l2norm(mf_net* h, int in_buf, int out_buf, const float* weight, int C, float eps) {
    MF_REQUIRE(h && weight && C > 0, "net_l2norm: bad argument");
    NET_BUF(ib, in_buf); NET_BUF(ob, out_buf);
    MF_REQUIRE(ib->H == ob->H && ib->W == ob->W && C <= ib->C && C <= ob->C, "net_l2norm: shape mismatch");
    float* dw = nullptr;
    MF_HIP(hipMalloc(&dw, C * sizeof(float)));
    MF_HIP(hipMemcpy(dw, weight, C * sizeof(float), hipMemcpyHostToDevice));
    h->dev.push_back(dw);
    h->ops.push_back([ib, ob, dw, C, eps](int B, hipStream_t s) {
        const int64_t px = (int64_t)B * ib->H * ib->W;
        hipLaunchKernelGGL(k_l2norm, dim3((unsigned)((px + 3) / 4)), dim3(256), 0, s, pl_of(*ib), pl_of(*ob), dw, eps, C, px);
        MF_HIP(hipGetLastError());
        return MF_OK;
    });
    h->names.push_back("l2norm"); h->flops.push_back(0.0);
    return MF_OK;
}

extern "C" int mf_net_global_avgpool(mf_net* h, int in_buf, int in_coff, int C, int out_buf) {
    MF_REQUIRE(h && C > 0 && in_coff >= 0, "net_global_avgpool: bad argument");
    NET_BUF(ib, in_buf); NET_BUF(ob, out_buf);
    MF_REQUIRE(ob->H == 1 && ob->W == 1 && C <= ob->C && in_coff + C <= ib->C, "net_global_avgpool: the output must be a 1x1 map with >= %d channels", C);
    h->ops.push_back([ib, ob, in_coff, C](int B, hipStream_t s) {
        hipLaunchKernelGGL(k_gap, dim3((C + 63) / 64, B), dim3(256), 0, s, pl_of(*ib), in_coff, C, pl_of(*ob));
        MF_HIP(hipGetLastError());
        return MF_OK;
    });
    h->names.push_back("global_avgpool"); h->flops.push_back(0.0);
    return MF_OK;
}

extern "C" int mf_net_scale_add(mf_net* h, int x_buf, int x_coff, int C, int s_buf, int t_buf, int t_coff, int v_buf, int out_buf, int out_coff) {
    MF_REQUIRE(h && C > 0, "net_scale_add: bad argument");
    NET_BUF(xb, x_buf); NET_BUF(ob, out_buf);
    ActBuf *sb = s_buf >= 0 ? h->B(s_buf) : nullptr, *tb = t_buf >= 0 ? h->B(t_buf) : nullptr, *vb = v_buf >= 0 ? h->B(v_buf) : nullptr;
    MF_REQUIRE((s_buf < 0 || sb) && (t_buf < 0 || tb) && (v_buf < 0 || vb), "net_scale_add: unknown operand buffer");
    MF_REQUIRE(xb->H == ob->H && xb->W == ob->W && x_coff + C <= xb->C && out_coff + C <= ob->C, "net_scale_add: x / out mismatch");
    MF_REQUIRE(!sb || (sb->H == 1 && sb->W == 1 && sb->C >= C), "net_scale_add: the scale must be a 1x1 map");
    MF_REQUIRE(!vb || (vb->H == 1 && vb->W == 1 && vb->C >= C), "net_scale_add: the broadcast addend must be a 1x1 map");
    MF_REQUIRE(!tb || (tb->H == xb->H && tb->W == xb->W && t_coff + C <= tb->C), "net_scale_add: the addend must have x's size");
    h->ops.push_back([=](int B, hipStream_t s) {
        const int64_t total = (int64_t)B * xb->H * xb->W * C;
        const Pl none{};
        hipLaunchKernelGGL(k_scale_add, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pl_of(*xb), x_coff, sb ? pl_of(*sb) : none, tb ? pl_of(*tb) : none,
                           t_coff, vb ? pl_of(*vb) : none, pl_of(*ob), out_coff, C, sb ? 1 : 0, tb ? 1 : 0, vb ? 1 : 0, total);
        MF_HIP(hipGetLastError());
        return MF_OK;
    });
    h->names.push_back("scale_add"); h->flops.push_back(0.0);
    return MF_OK;
}

extern "C" int mf_net_upsample_nearest(mf_net* h, int in_buf, int out_buf) {
    MF_REQUIRE(h, "net_upsample_nearest: null handle");
    NET_BUF(ib, in_buf); NET_BUF(ob, out_buf);
    MF_REQUIRE(ib->C == ob->C, "net_upsample_nearest: channel mismatch");
    h->ops.push_back([ib, ob](int B, hipStream_t s) {
        const int64_t total = (int64_t)B * ob->H * ob->W * ob->C;
        hipLaunchKernelGGL(k_up_nearest, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pl_of(*ib), pl_of(*ob), ob->C, total);
        MF_HIP(hipGetLastError());
        return MF_OK;
    });
    h->names.push_back("upsample_nearest"); h->flops.push_back(0.0);
    return MF_OK;
}

extern "C" int mf_net_num_ops(const mf_net* h) { return h ? (int)h->ops.size() : 0; }

extern "C" double mf_net_flops_per_item(const mf_net* h) {
    double f = 0.0;
    if (h) for (double x : h->flops) f += x;
    return f;
}

extern "C" int mf_net_set_input(mf_net* h, int buf, const float* nchw, int C, int batch, void* stream) {
    MF_REQUIRE(h && nchw && batch >= 1 && batch <= h->cap, "net_set_input: bad argument (batch %d, capacity %d)", batch, h ? h->cap : 0);
    NET_BUF(b, buf);
    return mf_nchw_to_act(nchw, C, *b, batch, (hipStream_t)stream);
}

// the explicit warm-up: time every conv's launch configurations on the buffers the last run at this batch size filled, drop the graph captured with the old ones
extern "C" int mf_net_tune(mf_net* h, int batch, void* stream) {
    MF_REQUIRE(h && batch >= 1 && batch <= h->cap, "net_tune: batch %d exceeds the capacity %d", batch, h ? h->cap : 0);
    hipStream_t s = (hipStream_t)stream;
    auto it = h->graphs.find(batch);
    MF_REQUIRE(!h->use_graph || it != h->graphs.end(), "net_tune: run the net once at batch %d first (the layers are timed on its buffers)", batch);
    MF_HIP(hipStreamSynchronize(h->cap_stream));
    MF_HIP(hipStreamSynchronize(s));
    for (auto& t : h->tunables) {
        int rc = mf_conv_tune(t.p, t.in, t.out, t.res, batch, s);
        if (rc) return rc;
    }
    MF_HIP(hipStreamSynchronize(s));
    if (h->use_graph) { if (it->second) (void)hipGraphExecDestroy(it->second); h->graphs.erase(it); }   // next run: eager (workspaces), then re-capture
    return MF_OK;
}

extern "C" int mf_net_run(mf_net* h, int batch, void* stream) {
    MF_REQUIRE(h && batch >= 1 && batch <= h->cap, "net_run: batch %d exceeds the capacity %d", batch, h ? h->cap : 0);
    hipStream_t s = (hipStream_t)stream;
    if (!h->use_graph) {
        for (auto& t : h->tunables) mf_conv_tune_lookup(t.p, t.in, batch);
        return h->run_body(batch, s);
    }
    auto it = h->graphs.find(batch);
    if (it == h->graphs.end()) {                                                                        // first call eager (split-K workspaces grow here)
        h->graphs.emplace(batch, nullptr);
        for (auto& t : h->tunables) mf_conv_tune_lookup(t.p, t.in, batch);                              // launch configurations: a table lookup, never a measurement
        int rc = h->run_body(batch, s);
        if (rc || !mf_autotune_enabled()) return rc;
        for (auto& t : h->tunables)                                                                     // MF_AUTOTUNE=1 (development): measure here, then the real outputs again
            if ((rc = mf_conv_tune(t.p, t.in, t.out, t.res, batch, s))) return rc;
        return h->run_body(batch, s);
    }
    if (!it->second) {
        hipGraph_t graph = nullptr;
        MF_HIP(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
        int rc = h->run_body(batch, h->cap_stream);
        hipError_t e = hipStreamEndCapture(h->cap_stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        MF_HIP(e);
        MF_HIP(hipGraphInstantiate(&it->second, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
    }
    MF_HIP(hipEventRecord(h->ev_in, s));
    MF_HIP(hipStreamWaitEvent(h->cap_stream, h->ev_in, 0));
    MF_HIP(hipGraphLaunch(it->second, h->cap_stream));
    MF_HIP(hipEventRecord(h->ev_out, h->cap_stream));
    MF_HIP(hipStreamWaitEvent(s, h->ev_out, 0));
    return MF_OK;
}

extern "C" int mf_net_get_output(mf_net* h, int buf, int coff, int C, float* nchw, int batch, void* stream) {
    MF_REQUIRE(h && nchw && batch >= 1 && batch <= h->cap, "net_get_output: bad argument");
    NET_BUF(b, buf);
    MF_REQUIRE(coff >= 0 && C > 0 && coff + C <= b->C, "net_get_output: channel slice [%d, %d) outside the buffer (%d)", coff, coff + C, b->C);
    return mf_act_to_nchw(ActView{b, coff, C}, nchw, batch, (hipStream_t)stream);
}

extern "C" int mf_net_get_output_bilinear(mf_net* h, int buf, int coff, int C, float* nchw, int H, int W, int batch, void* stream) {
    MF_REQUIRE(h && nchw && batch >= 1 && batch <= h->cap && H >= 1 && W >= 1, "net_get_output_bilinear: bad argument");
    NET_BUF(b, buf);
    MF_REQUIRE(coff >= 0 && C > 0 && coff + C <= b->C, "net_get_output_bilinear: channel slice outside the buffer");
    const float sh = H > 1 ? (float)(b->H - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(b->W - 1) / (float)(W - 1) : 0.f;
    const int64_t total = (int64_t)batch * C * H * W;
    hipLaunchKernelGGL(k_bilinear_ac, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pl_of(*b), coff, C, nchw, H, W, sh, sw, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

extern "C" int mf_s3fd_maxout_bg(const float* cls4, float* cls2, int batch, int hw, void* stream) {
    MF_REQUIRE(cls4 && cls2 && batch >= 1 && hw >= 1, "s3fd_maxout_bg: bad argument");
    const int64_t total = (int64_t)batch * hw;
    hipLaunchKernelGGL(k_maxout_bg, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cls4, cls2, hw, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

extern "C" void mf_net_destroy(mf_net* h) { delete h; }

// LayerNorm / softmax / operand packing kernels for token sequences (see mf_nn.h).  All are one-pass,
// latency/HBM-bound helpers around the MFMA GEMMs; math in fp32, storage in bf16 (hi, lo) planes.
#include "mf_nn.h"
#include <vector>

namespace {

__device__ __forceinline__ uint32_t nf2bf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float nbf2f(uint32_t h) { return __uint_as_float(h << 16); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ float ld(const bf16_t* hi, const bf16_t* lo, int64_t o) {
    float v = nbf2f(hi[o]);
    if (lo) v += nbf2f(lo[o]);
    return v;
}
__device__ __forceinline__ void st(bf16_t* hi, bf16_t* lo, int64_t o, float v) {
    const uint32_t h = nf2bf(v);
    hi[o] = (bf16_t)h;
    if (lo) lo[o] = (bf16_t)nf2bf(v - nbf2f(h));
}

constexpr int MAXPL = 32;   // channels per lane held in registers: C <= 2048

// one wave per token; rows are addressed as base + b*batch_stride + t*row_stride
__global__ __launch_bounds__(256) void k_layernorm(const bf16_t* xh, const bf16_t* xl, int64_t xb, int xs,
                                                   bf16_t* yh, bf16_t* yl, int64_t yb, int ys, const float* gamma,
                                                   const float* beta, float eps, int C, int T, int total) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= total) return;
    const int b = row / T, t = row - b * T;
    const int64_t xo = (int64_t)b * xb + (int64_t)t * xs, yo = (int64_t)b * yb + (int64_t)t * ys;
    float v[MAXPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < C ? ld(xh, xl, xo + c) : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
        const int c = lane + 64 * i;
        const float d = c < C ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
        const int c = lane + 64 * i;
        if (c < C) st(yh, yl, yo + c, (v[i] - mean) * rstd * gamma[c] + beta[c]);
    }
}

__global__ __launch_bounds__(256) void k_softmax_rows(const bf16_t* sh, const bf16_t* sl, int64_t sb, int ss,
                                                      bf16_t* ph, bf16_t* pl, int64_t pb, int ps, int n_keys,
                                                      int n_out, float scale, int T, int total) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= total) return;
    const int b = row / T, t = row - b * T;
    const int64_t so = (int64_t)b * sb + (int64_t)t * ss, po = (int64_t)b * pb + (int64_t)t * ps;
    float v[MAXPL];
    float m = -3.0e38f;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < n_keys ? ld(sh, sl, so + c) * scale : -3.0e38f;
        m = fmaxf(m, v[i]);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < n_keys ? expf(v[i] - m) : 0.f;
        s += v[i];
    }
    const float inv = 1.f / wave_sum(s);
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
        const int c = lane + 64 * i;
        if (c < n_out) st(ph, pl, po + c, v[i] * inv);
    }
}

// one thread per packed element: dst[(kt*Npad + n)*64 + e] = src(n, kt*64 + e)
__global__ __launch_bounds__(256) void k_pack_b(const bf16_t* sh, const bf16_t* sl, int64_t stride_n, int64_t stride_k,
                                                int N, int K, int Npad, bf16_t* dh, bf16_t* dl, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx & 63);
    const int64_t r = idx >> 6;
    const int n = (int)(r % Npad);
    const int k = (int)(r / Npad) * 64 + e;
    const bool in = n < N && k < K;
    const int64_t so = (int64_t)n * stride_n + (int64_t)k * stride_k;
    dh[idx] = in ? sh[so] : (bf16_t)0;
    if (dl) dl[idx] = in ? sl[so] : (bf16_t)0;
}

__global__ __launch_bounds__(256) void k_rows_to_f32(const bf16_t* xh, const bf16_t* xl, int64_t xb, int xs, int C,
                                                     int T, float* dst, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int64_t r = idx / C;
    const int t = (int)(r % T), b = (int)(r / T);
    dst[idx] = ld(xh, xl, (int64_t)b * xb + (int64_t)t * xs + c);
}

struct Rows { const bf16_t* hi; const bf16_t* lo; int64_t bstride; int rstride; int T; };
Rows rows_of(const ActView& v) {
    const ActBuf& b = *v.buf;
    const int64_t base = mf_interior(b) + v.coff;
    return Rows{b.hi + base, b.lo ? b.lo + base : nullptr, b.per_batch(), b.C, b.W};
}

}  // namespace

int mf_layernorm(const ActView& x, const ActView& y, const float* gamma, const float* beta, float eps, int batch,
                 hipStream_t s) {
    MF_REQUIRE(x.buf->H == 1 && y.buf->H == 1 && x.buf->W == y.buf->W && x.C == y.C, "layernorm: shape mismatch");
    MF_REQUIRE(x.C <= 64 * MAXPL, "layernorm: C=%d exceeds %d", x.C, 64 * MAXPL);
    const Rows xr = rows_of(x), yr = rows_of(y);
    const int total = batch * xr.T;
    hipLaunchKernelGGL(k_layernorm, dim3((total + 3) / 4), dim3(256), 0, s, xr.hi, xr.lo, xr.bstride, xr.rstride,
                       const_cast<bf16_t*>(yr.hi), const_cast<bf16_t*>(yr.lo), yr.bstride, yr.rstride, gamma, beta,
                       eps, x.C, xr.T, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_softmax_rows(const ActView& scores, const ActView& probs, int n_keys, float scale, int batch, hipStream_t s) {
    MF_REQUIRE(scores.buf->W == probs.buf->W && scores.C >= n_keys && probs.C >= n_keys, "softmax: shape mismatch");
    MF_REQUIRE(probs.C <= 64 * MAXPL, "softmax: row length %d exceeds %d", probs.C, 64 * MAXPL);
    const Rows sr = rows_of(scores), pr = rows_of(probs);
    const int total = batch * sr.T;
    hipLaunchKernelGGL(k_softmax_rows, dim3((total + 3) / 4), dim3(256), 0, s, sr.hi, sr.lo, sr.bstride, sr.rstride,
                       const_cast<bf16_t*>(pr.hi), const_cast<bf16_t*>(pr.lo), pr.bstride, pr.rstride, n_keys, probs.C,
                       scale, sr.T, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_pack_b(ConvPlan* plan, const bf16_t* src_hi, const bf16_t* src_lo, int64_t stride_n, int64_t stride_k, int N,
              int K, hipStream_t s) {
    MF_REQUIRE(N <= plan->Npad && K <= plan->ph[0].KT * 64, "pack_b: %dx%d does not fit the plan", N, K);
    const int64_t total = (int64_t)plan->ph[0].KT * plan->Npad * 64;
    hipLaunchKernelGGL(k_pack_b, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src_hi, src_lo, stride_n,
                       stride_k, N, K, plan->Npad, plan->w_hi, plan->w_lo, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_gemm_plan_create(ConvPlan* p, int K, int N, int T, int precision) {
    MF_REQUIRE(K % 8 == 0 && N % 4 == 0 && K > 0 && N > 0 && T > 0, "gemm plan: K=%d must be a multiple of 8, N=%d of 4", K, N);
    mf_conv2d_desc d{};
    d.cin = K; d.cout = N; d.kh = d.kw = 1; d.stride_h = d.stride_w = 1; d.in_h = 1; d.in_w = T;
    p->d = d;
    p->precision = precision;
    p->cin_pad = K;
    p->out_h = 1; p->out_w = T; p->Hq = 1; p->Wq = T;
    p->nphase = 1; p->out_step = 1; p->in_step_h = p->in_step_w = 1; p->in_halo_need = 0;
    p->phase_taps = {{ConvPlan::Tap{0, 0}}};
    p->phase_oy = {0}; p->phase_ox = {0};
    p->Npad = (N + 15) / 16 * 16;
    p->BK = 64;
    p->halo = false;
    const int KT = (K / 8 + 7) / 8;
    p->ph[0] = ConvPhase{0, KT * 8, KT, 0, 0, 0};
    p->goff_total = KT * 8;
    const int64_t total = (int64_t)KT * p->Npad * 64;
    MF_HIP(hipMalloc(&p->w_hi, total * sizeof(bf16_t)));
    MF_HIP(hipMemset(p->w_hi, 0, total * sizeof(bf16_t)));
    if (precision == MF_PREC_BF16X3) {
        MF_HIP(hipMalloc(&p->w_lo, total * sizeof(bf16_t)));
        MF_HIP(hipMemset(p->w_lo, 0, total * sizeof(bf16_t)));
    }
    MF_HIP(hipMalloc(&p->bias, p->Npad * sizeof(float)));
    MF_HIP(hipMemset(p->bias, 0, p->Npad * sizeof(float)));
    MF_HIP(hipMalloc(&p->goff, p->goff_total * sizeof(int)));
    p->bound_in_ld = p->bound_in_wp = -1;
    return MF_OK;
}

int mf_rows_to_f32(const ActView& x, float* dst, int batch, hipStream_t s) {
    const Rows xr = rows_of(x);
    const int64_t total = (int64_t)batch * xr.T * x.C;
    hipLaunchKernelGGL(k_rows_to_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xr.hi, xr.lo, xr.bstride,
                       xr.rstride, x.C, xr.T, dst, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

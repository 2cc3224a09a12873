// LayerNorm / softmax / operand packing kernels for token sequences (see mf_nn.h).  All are one-pass,
// latency/HBM-bound helpers around the MFMA GEMMs; math in fp32, storage in bf16 (hi, lo) planes.
#include "mf_nn.h"
#include <vector>
#include <algorithm>

namespace {

__device__ __forceinline__ uint32_t nf2bf(float f) {
    // round to nearest even in hardware: gfx950's v_cvt_pk_bf16_f32 (the compiler pairs neighbouring calls), a quarter of the integer form's instructions
    return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)f);
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

// token t of batch b of a (possibly padded) H x W image view
struct Rows {
    const bf16_t* hi; const bf16_t* lo;
    int64_t bstride; int W, Wp, halo, C, T;
    uint32_t w_mul, w_shr;            // t / W by multiply-shift (mf_fastdiv): every kernel here turns a token index into (row, column) per token it touches
    uint32_t g_mul, g_shr;            // channel / channels-per-group, set by the GroupNorm launchers (0, 0 elsewhere)
    __device__ int group_of(int c) const { return mf_fdiv(c, g_mul, g_shr); }
    __host__ __device__ int64_t off(int b, int t) const {
        const int y = w_mul ? (int)((uint32_t)(((uint64_t)(uint32_t)t * w_mul) >> 32) >> w_shr) : t, x = t - y * W;
        return (int64_t)b * bstride + ((int64_t)(y + halo) * Wp + x + halo) * C;
    }
};

// one wave per token
__device__ __forceinline__ float ln_act(float o, int act) {   // 0 none, 3 GELU (erf form): wav2vec2's conv layers are conv -> LayerNorm -> GELU
    return act == 3 ? 0.5f * o * (1.f + erff(o * 0.70710678118654752f)) : o;
}

__global__ __launch_bounds__(256) void k_layernorm(Rows X, Rows Y, const float* gamma, const float* beta, float eps,
                                                   int C, int total, int act) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= total) return;
    const int T = X.T;
    const int b = row / T, t = row - b * T;
    const bf16_t *xh = X.hi, *xl = X.lo;
    bf16_t *yh = const_cast<bf16_t*>(Y.hi), *yl = const_cast<bf16_t*>(Y.lo);
    const int64_t xo = X.off(b, t), yo = Y.off(b, t);
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
        if (c < C) st(yh, yl, yo + c, ln_act((v[i] - mean) * rstd * gamma[c] + beta[c], act));
    }
}

// one wave per token, 8 channels (16 bytes per plane) per lane and step: C % 8 == 0, views 16-byte aligned
template <int NCH>
__global__ __launch_bounds__(256) void k_layernorm_v8(Rows X, Rows Y, const float* gamma, const float* beta, float eps, int C, int total, int act) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= total) return;
    const int T = X.T;
    const int b = row / T, t = row - b * T;
    const int64_t xo = X.off(b, t), yo = Y.off(b, t);
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = (lane + 64 * j) * 8;
        if (c < C) {
            const uint4 h = *reinterpret_cast<const uint4*>(X.hi + xo + c);
            const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[j][2 * e] = nbf2f(hw[e] & 0xffffu); v[j][2 * e + 1] = nbf2f(hw[e] >> 16); }
            if (X.lo) {
                const uint4 l = *reinterpret_cast<const uint4*>(X.lo + xo + c);
                const uint32_t lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[j][2 * e] += nbf2f(lw[e] & 0xffffu); v[j][2 * e + 1] += nbf2f(lw[e] >> 16); }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[j][e];
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
        if ((lane + 64 * j) * 8 < C) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[j][e] - mean; q += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
    bf16_t *yh = const_cast<bf16_t*>(Y.hi), *yl = const_cast<bf16_t*>(Y.lo);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = (lane + 64 * j) * 8;
        if (c < C) {
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + c), g1 = *reinterpret_cast<const float4*>(gamma + c + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(beta + c), b1 = *reinterpret_cast<const float4*>(beta + c + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            uint32_t hb[8], lb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float o = ln_act((v[j][e] - mean) * rstd * gg[e] + bb[e], act);
                hb[e] = nf2bf(o);
                lb[e] = nf2bf(o - nbf2f(hb[e]));
            }
            *reinterpret_cast<uint4*>(yh + yo + c) = make_uint4(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16));
            if (yl) *reinterpret_cast<uint4*>(yl + yo + c) = make_uint4(lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16), lb[4] | (lb[5] << 16), lb[6] | (lb[7] << 16));
        }
    }
}

__global__ __launch_bounds__(256) void k_softmax_rows(Rows S, Rows P, int n_keys, int n_out, float scale, int total) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= total) return;
    const int T = S.T;
    const int b = row / T, t = row - b * T;
    const bf16_t *sh = S.hi, *sl = S.lo;
    bf16_t *ph = const_cast<bf16_t*>(P.hi), *pl = const_cast<bf16_t*>(P.lo);
    const int64_t so = S.off(b, t), po = P.off(b, t);
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

__global__ __launch_bounds__(256) void k_rows_to_f32(Rows X, int C, float* dst, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int64_t r = idx / C;
    const int t = (int)(r % X.T), b = (int)(r / X.T);
    dst[idx] = ld(X.hi, X.lo, X.off(b, t) + c);
}

__global__ __launch_bounds__(256) void k_rows_from_f32(const float* src, const float* addend, Rows Y, int C, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int64_t r = idx / C;
    const int t = (int)(r % Y.T), b = (int)(r / Y.T);
    float v = src[idx];
    if (addend) v += addend[(int64_t)t * C + c];
    st(const_cast<bf16_t*>(Y.hi), const_cast<bf16_t*>(Y.lo), Y.off(b, t) + c, v);
}

// GroupNorm statistics.  grid (pixel blocks, batch); thread (k = tid % c8, pp = tid / c8) owns the 16-byte channel
// chunk k (and k + 256, ... for very wide tensors) and walks the block's pixels pp, pp + PPI, ...: for one pixel
// the active threads read one contiguous row of the NHWC tensor.  Per-channel fp32 partials over <= 64 pixels,
// then fp64: LDS bins per group, one global fp64 atomic per (workgroup, group).
constexpr int GN_MAXCOL = 2;     // chunk columns per thread: C <= 8 * 256 * 2 = 4096
__global__ __launch_bounds__(256) void k_gn_stats(Rows X, int groups, int cpg, int C, int P, double* stats) {
    __shared__ double bins[2 * 64];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int c8 = C / 8;
    const int cols = c8 < 256 ? c8 : 256;
    const int ppi = 256 / cols;                     // pixels walked in parallel
    const int k0 = tid % cols, pp = tid / cols;
    const int t0 = blockIdx.x * P, t1 = min(X.T, t0 + P);
    for (int i = tid; i < 2 * groups; i += 256) bins[i] = 0.0;
    __syncthreads();
    float s[GN_MAXCOL][8], q[GN_MAXCOL][8];
#pragma unroll
    for (int j = 0; j < GN_MAXCOL; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[j][e] = 0.f; q[j][e] = 0.f; }
    if (pp < ppi) {
        // GN_UP pixels per trip with every load issued before the first add: a lane otherwise has two 16-byte loads in flight, and the
        // big VAE maps (268 MB per tensor) streamed at 1.8 TB/s
        auto acc = [&](int j, const uint4& vh, const uint4& vl) __attribute__((always_inline)) {
            const uint32_t hh[4] = {vh.x, vh.y, vh.z, vh.w}, ll[4] = {vl.x, vl.y, vl.z, vl.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = nbf2f(hh[e] & 0xffffu) + nbf2f(ll[e] & 0xffffu);
                const float v1 = nbf2f(hh[e] >> 16) + nbf2f(ll[e] >> 16);
                s[j][2 * e] += v0; q[j][2 * e] += v0 * v0;
                s[j][2 * e + 1] += v1; q[j][2 * e + 1] += v1 * v1;
            }
        };
        constexpr int GN_UP = 4;
        int t = t0 + pp;
        if (c8 <= 256) {
            for (; t + (GN_UP - 1) * ppi < t1; t += GN_UP * ppi) {
                uint4 vh[GN_UP], vl[GN_UP];
#pragma unroll
                for (int u = 0; u < GN_UP; ++u) {
                    const int64_t o = X.off(b, t + u * ppi) + k0 * 8;
                    vh[u] = *reinterpret_cast<const uint4*>(X.hi + o);
                    vl[u] = X.lo ? *reinterpret_cast<const uint4*>(X.lo + o) : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < GN_UP; ++u) acc(0, vh[u], vl[u]);
            }
        }
        for (; t < t1; t += ppi) {
            const int64_t o = X.off(b, t);
#pragma unroll
            for (int j = 0; j < GN_MAXCOL; ++j) {
                const int k = k0 + 256 * j;
                if (k >= c8) break;
                const uint4 vh = *reinterpret_cast<const uint4*>(X.hi + o + k * 8);
                uint4 vl = make_uint4(0, 0, 0, 0);
                if (X.lo) vl = *reinterpret_cast<const uint4*>(X.lo + o + k * 8);
                acc(j, vh, vl);
            }
        }
#pragma unroll
        for (int j = 0; j < GN_MAXCOL; ++j) {
            const int k = k0 + 256 * j;
            if (k >= c8) break;
            // channels of this chunk -> groups (cpg may be smaller or larger than 8, and need not divide it)
            int g_cur = X.group_of(k * 8);
            double as = 0.0, aq = 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int g = X.group_of(k * 8 + e);
                if (g != g_cur) {
                    atomicAdd(&bins[2 * g_cur], as); atomicAdd(&bins[2 * g_cur + 1], aq);
                    g_cur = g; as = 0.0; aq = 0.0;
                }
                as += (double)s[j][e]; aq += (double)q[j][e];
            }
            atomicAdd(&bins[2 * g_cur], as); atomicAdd(&bins[2 * g_cur + 1], aq);
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * groups; i += 256) atomicAdd(&stats[2 * (b * groups) + i], bins[i]);
}

// zeroes the statistics scratch (a kernel rather than hipMemsetAsync: memset nodes inside a captured hipGraph
// were observed to misbehave next to other graphs on ROCm 7.2 -- DESIGN.md "hipGraph notes")
__global__ void k_zero_f64(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

// y = (x - mean) * rstd * gamma + beta [, SiLU].  A thread owns one 8-channel column and GA_U tokens (t, t + TS, ...): the column's
// gamma / beta / (mean, rstd) are formed once, the GA_U (hi, lo) loads are all issued before the first use, and the grid is
// (token blocks, batch) so no 64-bit division is left.  (One token per thread streamed the big VAE maps at 2.5 TB/s.)
template <int GA_U>
__global__ __launch_bounds__(256) void k_gn_apply(Rows X, Rows Y, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  const double* stats, double inv_n, float eps, int groups, int cpg, int C, int silu, int TS) {
    const int c8 = C / 8;
    const int b = blockIdx.y;
    // 256 threads = (256 / cols) token lanes x cols columns, cols = min(c8, 256); wider tensors loop over column blocks
    const int cols = c8 < 256 ? c8 : 256;
    const int tl = threadIdx.x / cols, kk = threadIdx.x - tl * cols;
    const int tpb = 256 / cols;                               // token lanes per block
    const int tbase = blockIdx.x * tpb * GA_U + tl;           // tokens tbase + u * tpb
    if (tl >= tpb) return;
    for (int k = kk; k < c8; k += 256) {
        const int c0 = k * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + c0), g1 = *reinterpret_cast<const float4*>(gamma + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + c0), b1 = *reinterpret_cast<const float4*>(beta + c0 + 4);
        float sc[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        float sh[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float mu[8], rs[8];
        int g_prev = -1;
        float2 st2 = make_float2(0.f, 1.f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = X.group_of(c0 + e);
            if (g != g_prev) {
                // finalise (sum, sum of squares) -> (mean, rstd): a handful of fp64 ops per thread and group
                const double2 sq = *reinterpret_cast<const double2*>(stats + 2 * (b * groups + g));
                const double mean = sq.x * inv_n;
                const double var = fmax(sq.y * inv_n - mean * mean, 0.0);
                st2 = make_float2((float)mean, rsqrtf((float)var + eps));
                g_prev = g;
            }
            mu[e] = st2.x; rs[e] = st2.y;
        }
        uint4 vh[GA_U], vl[GA_U];
        int64_t yo[GA_U];
        bool ok[GA_U];
#pragma unroll
        for (int u = 0; u < GA_U; ++u) {
            const int t = tbase + u * tpb;
            ok[u] = t < X.T;
            const int tt = ok[u] ? t : X.T - 1;
            const int64_t xo = X.off(b, tt) + c0;
            yo[u] = Y.off(b, tt) + c0;
            vh[u] = *reinterpret_cast<const uint4*>(X.hi + xo);
            vl[u] = X.lo ? *reinterpret_cast<const uint4*>(X.lo + xo) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < GA_U; ++u) {
            const uint32_t hh[4] = {vh[u].x, vh[u].y, vh[u].z, vh[u].w}, ll[4] = {vl[u].x, vl[u].y, vl[u].z, vl[u].w};
            uint32_t oh[4], ol[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float out2[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float x = nbf2f(q ? hh[e] >> 16 : hh[e] & 0xffffu) + nbf2f(q ? ll[e] >> 16 : ll[e] & 0xffffu);
                    float v = (x - mu[2 * e + q]) * rs[2 * e + q] * sc[2 * e + q] + sh[2 * e + q];
                    if (silu) v = v * __builtin_amdgcn_rcpf(1.f + __expf(-v));   // (v_rcp_f32, 1 ulp, as the VAE's converter pass: the IEEE division is ~10 instructions per value)
                    out2[q] = v;
                }
                const uint32_t h0 = nf2bf(out2[0]), h1 = nf2bf(out2[1]);
                oh[e] = h0 | (h1 << 16);
                ol[e] = nf2bf(out2[0] - nbf2f(h0)) | (nf2bf(out2[1] - nbf2f(h1)) << 16);
            }
            if (ok[u]) {
                *reinterpret_cast<uint4*>(const_cast<bf16_t*>(Y.hi) + yo[u]) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
                if (Y.lo) *reinterpret_cast<uint4*>(const_cast<bf16_t*>(Y.lo) + yo[u]) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
            }
        }
    }
    (void)TS;
}

__global__ __launch_bounds__(256) void k_geglu(Rows X, Rows Y, int C, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int64_t r = idx / C;
    const int t = (int)(r % X.T), b = (int)(r / X.T);
    const int64_t xo = X.off(b, t);
    const float a = ld(X.hi, X.lo, xo + c), g = ld(X.hi, X.lo, xo + C + c);
    st(const_cast<bf16_t*>(Y.hi), const_cast<bf16_t*>(Y.lo), Y.off(b, t) + c, a * (0.5f * g * (1.f + erff(g * 0.70710678118654752f))));
}

__global__ __launch_bounds__(256) void k_pack_b_grouped(const bf16_t* sh, const bf16_t* sl, int64_t sb, int64_t shd,
                                                        int64_t stride_n, int64_t stride_k, int N, int K, int Npad, int heads,
                                                        int64_t per_group, bf16_t* dh, bf16_t* dl, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int z = (int)(idx / per_group);
    const int64_t w = idx - (int64_t)z * per_group;
    const int e = (int)(w & 63);
    const int64_t r = w >> 6;
    const int n = (int)(r % Npad);
    const int k = (int)(r / Npad) * 64 + e;
    const bool in = n < N && k < K;
    const int zb = z / heads, zh = z - zb * heads;
    const int64_t so = zb * sb + zh * shd + (int64_t)n * stride_n + (int64_t)k * stride_k;
    dh[idx] = in ? sh[so] : (bf16_t)0;
    if (dl) dl[idx] = in ? sl[so] : (bf16_t)0;
}

__global__ __launch_bounds__(256) void k_vae_post(Rows X, int H, int W, uint8_t* dst, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one thread per pixel
    if (idx >= total) return;
    const int t = (int)(idx % X.T), b = (int)(idx / X.T);
    const int64_t o = X.off(b, t);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = ld(X.hi, X.lo, o + c) / 2.f + 0.5f;
        v = fminf(fmaxf(v, 0.f), 1.f);
        dst[idx * 3 + (2 - c)] = (uint8_t)rintf(v * 255.f);   // numpy round (half to even), RGB -> BGR
    }
}

Rows rows_of(const ActView& v, int cpg = 0) {     // cpg: channels per GroupNorm group for the kernels that map channels to groups
    const ActBuf& b = *v.buf;
    Rows r{b.hi + v.coff, b.lo ? b.lo + v.coff : nullptr, b.per_batch(), b.W, b.Wp(), b.halo, b.C, b.H * b.W, 0u, 0u, 0u, 0u};
    mf_fastdiv((uint32_t)b.W, &r.w_mul, &r.w_shr);
    if (cpg > 0) mf_fastdiv((uint32_t)cpg, &r.g_mul, &r.g_shr);
    return r;
}

}  // namespace

int mf_layernorm(const ActView& x, const ActView& y, const float* gamma, const float* beta, float eps, int batch,
                 hipStream_t s, int tokens, int act) {
    MF_REQUIRE(act == 0 || act == 3, "layernorm: activation %d (0 none, 3 GELU)", act);
    MF_REQUIRE(x.buf->H * x.buf->W == y.buf->H * y.buf->W && x.C == y.C, "layernorm: shape mismatch");
    MF_REQUIRE(x.C <= 64 * MAXPL, "layernorm: C=%d exceeds %d", x.C, 64 * MAXPL);
    Rows xr = rows_of(x), yr = rows_of(y);
    if (tokens > 0) {
        MF_REQUIRE(tokens <= xr.T, "layernorm: prefix of %d tokens exceeds the sequence (%d)", tokens, xr.T);
        xr.T = yr.T = tokens;
    }
    const int total = batch * xr.T;
    const bool vec = x.C % 8 == 0 && x.coff % 8 == 0 && y.coff % 8 == 0 && x.buf->C % 8 == 0 && y.buf->C % 8 == 0 && x.C <= 2048;
    const dim3 grid((total + 3) / 4), block(256);
    if (vec && x.C <= 512) hipLaunchKernelGGL(k_layernorm_v8<1>, grid, block, 0, s, xr, yr, gamma, beta, eps, x.C, total, act);
    else if (vec && x.C <= 1024) hipLaunchKernelGGL(k_layernorm_v8<2>, grid, block, 0, s, xr, yr, gamma, beta, eps, x.C, total, act);
    else if (vec) hipLaunchKernelGGL(k_layernorm_v8<4>, grid, block, 0, s, xr, yr, gamma, beta, eps, x.C, total, act);
    else hipLaunchKernelGGL(k_layernorm, grid, block, 0, s, xr, yr, gamma, beta, eps, x.C, total, act);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_softmax_rows(const ActView& scores, const ActView& probs, int n_keys, float scale, int batch, hipStream_t s) {
    MF_REQUIRE(scores.buf->H * scores.buf->W == probs.buf->H * probs.buf->W && scores.C >= n_keys && probs.C >= n_keys, "softmax: shape mismatch");
    MF_REQUIRE(probs.C <= 64 * MAXPL, "softmax: row length %d exceeds %d", probs.C, 64 * MAXPL);
    const Rows sr = rows_of(scores), pr = rows_of(probs);
    const int total = batch * sr.T;
    hipLaunchKernelGGL(k_softmax_rows, dim3((total + 3) / 4), dim3(256), 0, s, sr, pr, n_keys, probs.C, scale, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_rows_from_f32(const float* src, const float* addend, const ActView& y, int batch, hipStream_t s) {
    const Rows yr = rows_of(y);
    const int64_t total = (int64_t)batch * yr.T * y.C;
    hipLaunchKernelGGL(k_rows_from_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, addend, yr, y.C, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_zero_f64(double* p, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_zero_f64, dim3((n + 255) / 256), dim3(256), 0, s, p, n);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

namespace {
// scale[b][c] = rstd * gamma[c], shift[b][c] = beta[c] - mean * rstd * gamma[c] from the (sum, sum of squares) of k_gn_stats
__global__ void k_gn_affine(const double* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, double inv_n, float eps,
                            int groups, int cpg, int C, int total, float* scale, float* shift) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = i / C, c = i - b * C, g = c / cpg;
    const double2 sq = *reinterpret_cast<const double2*>(stats + 2 * (b * groups + g));
    float sc, sh;
    mf_gn_affine_pair(sq.x, sq.y, inv_n, eps, gamma[c], beta[c], sc, sh);
    scale[i] = sc;
    shift[i] = sh;
}
}  // namespace

// GroupNorm folded into the consumer conv (mf_conv_halo2.hip): statistics as in mf_groupnorm, then the per-(sample, channel) affine the
// conv applies (with SiLU) to its halo image in LDS.  scale / shift: [batch][C] fp32 device arrays.
int mf_groupnorm_affine(const ActView& x, const float* gamma, const float* beta, int groups, float eps, double* stats, float* scale, float* shift,
                        int batch, hipStream_t s, bool have_stats) {
    MF_REQUIRE(x.C % groups == 0 && x.C % 8 == 0 && x.coff % 8 == 0, "groupnorm: C=%d groups=%d", x.C, groups);
    const int cpg = x.C / groups;
    const Rows xr = rows_of(x, cpg);
    MF_REQUIRE(groups <= 64 && x.C <= 8 * 256 * GN_MAXCOL, "groupnorm: groups=%d / C=%d beyond the kernel's limits", groups, x.C);
    const int cols = std::min(256, x.C / 8), ppi = 256 / cols;
    int P = std::max(ppi, std::min(64 * ppi, (xr.T * batch + 1023) / 1024));
    if (!have_stats) hipLaunchKernelGGL(k_gn_stats, dim3((xr.T + P - 1) / P, batch), dim3(256), 0, s, xr, groups, cpg, x.C, P, stats);
    const int total = batch * x.C;
    hipLaunchKernelGGL(k_gn_affine, dim3((total + 255) / 256), dim3(256), 0, s, stats, gamma, beta, 1.0 / ((double)xr.T * cpg), eps, groups, cpg, x.C, total,
                       scale, shift);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

// the statistics pass alone: (sum, sum of squares) per (sample, group) ADDED to `stats` (mf_conv_launch behind a conv whose kernel
// configuration cannot accumulate them in its epilogue, ConvPlan::out_stats)
int mf_groupnorm_stats(const ActView& x, int groups, double* stats, int batch, hipStream_t s) {
    MF_REQUIRE(x.C % groups == 0 && x.C % 8 == 0 && x.coff % 8 == 0, "groupnorm: C=%d groups=%d", x.C, groups);
    MF_REQUIRE(groups <= 64 && x.C <= 8 * 256 * GN_MAXCOL, "groupnorm: groups=%d / C=%d beyond the kernel's limits", groups, x.C);
    const Rows xr = rows_of(x, x.C / groups);
    const int cols = std::min(256, x.C / 8), ppi = 256 / cols;
    const int P = std::max(ppi, std::min(64 * ppi, (xr.T * batch + 1023) / 1024));
    hipLaunchKernelGGL(k_gn_stats, dim3((xr.T + P - 1) / P, batch), dim3(256), 0, s, xr, groups, x.C / groups, x.C, P, stats);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_groupnorm(const ActView& x, const ActView& y, const float* gamma, const float* beta, int groups, float eps,
                 bool silu, double* stats, int batch, hipStream_t s, bool have_stats) {
    MF_REQUIRE(x.C == y.C && x.C % groups == 0 && x.C % 8 == 0 && x.coff % 8 == 0 && y.coff % 8 == 0, "groupnorm: C=%d groups=%d", x.C, groups);
    MF_REQUIRE(x.buf->H == y.buf->H && x.buf->W == y.buf->W, "groupnorm: spatial mismatch");
    const int cpg = x.C / groups;
    const Rows xr = rows_of(x, cpg), yr = rows_of(y);
    MF_REQUIRE(groups <= 64 && x.C <= 8 * 256 * GN_MAXCOL, "groupnorm: groups=%d / C=%d beyond the kernel's limits", groups, x.C);
    // pixels per workgroup: enough workgroups to fill the chip, at most 64 pixels per thread column
    const int cols = std::min(256, x.C / 8), ppi = 256 / cols;
    int P = std::max(ppi, std::min(64 * ppi, (xr.T * batch + 1023) / 1024));
    if (!have_stats) hipLaunchKernelGGL(k_gn_stats, dim3((xr.T + P - 1) / P, batch), dim3(256), 0, s, xr, groups, cpg, x.C, P, stats);
    MF_HIP(hipGetLastError());
    {
        // four tokens per thread once the tensor is big enough to fill the chip that way; small maps keep one token per thread
        const int tpb = 256 / cols;
        const bool big = (int64_t)batch * xr.T * (x.C / 8) >= (int64_t)1 << 20;
        const int U = big ? 4 : 1;
        const dim3 grid((unsigned)((xr.T + tpb * U - 1) / (tpb * U)), batch);
        if (big)
            hipLaunchKernelGGL(k_gn_apply<4>, grid, dim3(256), 0, s, xr, yr, gamma, beta, stats, 1.0 / ((double)xr.T * cpg), eps, groups, cpg, x.C, silu ? 1 : 0, 0);
        else
            hipLaunchKernelGGL(k_gn_apply<1>, grid, dim3(256), 0, s, xr, yr, gamma, beta, stats, 1.0 / ((double)xr.T * cpg), eps, groups, cpg, x.C, silu ? 1 : 0, 0);
    }
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_geglu(const ActView& x, const ActView& y, int batch, hipStream_t s) {
    MF_REQUIRE(x.C == 2 * y.C, "geglu: input must have twice the output channels");
    const Rows xr = rows_of(x), yr = rows_of(y);
    const int64_t total = (int64_t)batch * xr.T * y.C;
    hipLaunchKernelGGL(k_geglu, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xr, yr, y.C, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_vae_post_u8(const ActView& x, uint8_t* dst, int batch, hipStream_t s) {
    MF_REQUIRE(x.C >= 3, "vae_post: need 3 channels");
    const Rows xr = rows_of(x);
    const int64_t total = (int64_t)batch * xr.T;
    hipLaunchKernelGGL(k_vae_post, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xr, x.buf->H, x.buf->W, dst, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_pack_b_grouped(ConvPlan* plan, const bf16_t* src_hi, const bf16_t* src_lo, int64_t sb, int64_t sh, int64_t stride_n,
                      int64_t stride_k, int N, int K, int groups, int heads, hipStream_t s) {
    MF_REQUIRE(N <= plan->Npad && K <= plan->ph[0].KT * 64 && groups <= plan->groups_cap, "pack_b: does not fit the plan");
    const int64_t per_group = (int64_t)plan->ph[0].KT * plan->Npad * 64;
    const int64_t total = per_group * groups;
    hipLaunchKernelGGL(k_pack_b_grouped, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src_hi, src_lo, sb, sh,
                       stride_n, stride_k, N, K, plan->Npad, heads, per_group, plan->w_hi, plan->w_lo, total);
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
    return mf_gemm_plan_create_grouped(p, K, N, T, 1, precision);
}

int mf_gemm_plan_create_grouped(ConvPlan* p, int K, int N, int T, int groups, int precision) {
    MF_REQUIRE(K % 8 == 0 && K > 0 && N > 0 && T > 0 && groups > 0, "gemm plan: K=%d must be a multiple of 8", K);
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
    p->groups_cap = groups;
    const int64_t total = (int64_t)KT * p->Npad * 64 * groups;
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

namespace {
__global__ __launch_bounds__(256) void k_rows_to_f32_layered(Rows X, int C, float* dst, int layer, int n_layers, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int64_t r = idx / C;
    const int t = (int)(r % X.T), b = (int)(r / X.T);
    dst[((r * n_layers) + layer) * C + c] = ld(X.hi, X.lo, X.off(b, t) + c);
}
}  // namespace

int mf_rows_to_f32_layered(const ActView& x, float* dst, int batch, int tokens, int layer, int n_layers, hipStream_t s) {
    Rows xr = rows_of(x);
    MF_REQUIRE(tokens > 0 && tokens <= xr.T && layer >= 0 && layer < n_layers, "rows_to_f32_layered: bad argument");
    xr.T = tokens;
    const int64_t total = (int64_t)batch * tokens * x.C;
    hipLaunchKernelGGL(k_rows_to_f32_layered, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xr, x.C, dst, layer, n_layers, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_rows_to_f32(const ActView& x, float* dst, int batch, hipStream_t s) {
    const Rows xr = rows_of(x);
    const int64_t total = (int64_t)batch * xr.T * x.C;
    hipLaunchKernelGGL(k_rows_to_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xr, x.C, dst, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

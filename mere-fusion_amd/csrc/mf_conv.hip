// Implicit-GEMM convolution (+ folded BatchNorm bias, residual add, ReLU/sigmoid) on gfx950 MFMA.
//
// Replaces the unfused Conv2d -> BatchNorm2d -> (+x) -> ReLU module chain of
// wav2lip/models/conv.py:5-19 and the ConvTranspose2d variant of conv.py:33-44.
//
// GEMM view per phase: D[n][m] = sum_k W[n][k] * P[m][k]
//   m : output pixel of the quotient grid (b, i, j)              (MFMA "B" operand / columns)
//   n : output channel                                          (MFMA "A" operand / rows)
//   k : (tap, input channel), enumerated in 8-channel groups    (contraction)
// Weights are the A operand so that one lane of the 16x16 accumulator tile owns 4 CONSECUTIVE
// channels of one pixel: the NHWC epilogue is an 8-byte store per lane, 32 contiguous bytes per
// 4-lane group.  A stride-2 ConvTranspose2d runs as 4 sub-pixel phases (blockIdx.z), each an
// ordinary gather with 1/2/2/4 taps, so no zero-stuffed input is ever multiplied.
//
// One workgroup = 256 threads = 4 wave64; tile BM pixels x BN channels x 32 deep, LDS
// double-buffered, global->register->LDS staging issued one K tile ahead of the MFMAs.
// The padded-halo activation layout means no load in the main loop is predicated.
#include "mf_conv.h"
#include <cmath>
#include <cstring>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// [rows][32] bf16 tile, 64-byte rows, 16-byte slots XOR-swizzled so that every 16-lane service
// group of ds_read_b128 (rows l&15, slot l>>4) touches 16 distinct slots of the 256-byte bank row.
__device__ __forceinline__ int tile_off(int row, int kg) {
    return row * 64 + ((kg ^ ((0 - (row >> 2)) & 3)) << 4);
}

__device__ __forceinline__ float bf2f(uint32_t h16) { return __uint_as_float(h16 << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);   // round to nearest even (inputs are finite)
    return u >> 16;
}

template <int BM, int BN, int WGM, int WGN, bool X3>
__global__ __launch_bounds__(256) void k_conv_igemm(const ConvArgs a) {
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    constexpr int NP = X3 ? 2 : 1;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int FM = WTM / 16, FN = WTN / 16;
    static_assert(FM >= 1 && FN >= 1, "wave tile must hold a 16x16 fragment");
    constexpr int P_BYTES = BM * 64, W_BYTES = BN * 64;
    constexpr int PLANE = P_BYTES + W_BYTES;
    constexpr int STAGE = PLANE * NP;
    constexpr int NPP = (BM * 4 + 255) / 256;   // 16-byte pieces of the pixel tile per thread
    constexpr int NWP = (BN * 4 + 255) / 256;   // ... of the weight tile

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_goff = reinterpret_cast<int*>(smem + 2 * STAGE);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const ConvPhase ph = a.ph[blockIdx.z];

    // XCD-aware tile order: the dispatcher round-robins blockIdx over the 8 XCDs; give each XCD a
    // contiguous run of tiles (n fastest) so the N tiles of one pixel tile share an L2.
    const int nt = a.tiles_m * a.tiles_n;
    const int bid = blockIdx.x;
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tm = t / a.tiles_n, tn = t - tm * a.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    for (int i = tid; i < ph.ngroups; i += 256) s_goff[i] = a.goff[ph.goff_begin + i];

    // ---- staging assignment --------------------------------------------------------------
    const bf16_t* xp_hi[NPP];
    const bf16_t* xp_lo[NPP];
    int p_lds[NPP];
    bool p_on[NPP];
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
        const int p = tid + 256 * i;
        const int row = p >> 2, kg = p & 3;
        p_on[i] = (BM * 4 % 256 == 0) || (p < BM * 4);
        int m = m0 + row;
        m = m < a.M ? m : a.M - 1;
        const int b = m / a.HqWq;
        const int rem = m - b * a.HqWq;
        const int qi = rem / a.Wq, qj = rem - qi * a.Wq;
        const int64_t base = (int64_t)b * a.xb + (int64_t)qi * a.xi + (int64_t)qj * a.xj;
        xp_hi[i] = a.x_hi + base;
        xp_lo[i] = X3 ? a.x_lo + base : nullptr;
        p_lds[i] = tile_off(row, kg);
    }
    const bf16_t* wp_hi[NWP];
    const bf16_t* wp_lo[NWP];
    int w_lds[NWP];
    bool w_on[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int p = tid + 256 * i;
        const int row = p >> 2, kg = p & 3;
        w_on[i] = (BN * 4 % 256 == 0) || (p < BN * 4);
        int n = n0 + row;
        n = n < a.Npad ? n : a.Npad - 1;
        const int64_t off = ph.w_off + (int64_t)n * 32 + kg * 8;
        wp_hi[i] = a.w_hi + off;
        wp_lo[i] = X3 ? a.w_lo + off : nullptr;
        w_lds[i] = P_BYTES + tile_off(row, kg);
    }
    const int kg_me = tid & 3;
    const int64_t w_kstep = (int64_t)a.Npad * 32;

    u32x4 rp[NP][NPP], rw[NP][NWP];

    auto gload = [&](int kt) __attribute__((always_inline)) {
        const int go = s_goff[kt * 4 + kg_me];
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            if (p_on[i]) {
                rp[0][i] = *reinterpret_cast<const u32x4*>(xp_hi[i] + go);
                if (X3) rp[NP - 1][i] = *reinterpret_cast<const u32x4*>(xp_lo[i] + go);
            }
        }
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
            if (w_on[i]) {
                rw[0][i] = *reinterpret_cast<const u32x4*>(wp_hi[i] + kt * w_kstep);
                if (X3) rw[NP - 1][i] = *reinterpret_cast<const u32x4*>(wp_lo[i] + kt * w_kstep);
            }
        }
    };
    auto swrite = [&](int s) __attribute__((always_inline)) {
        char* base = smem + s * STAGE;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int i = 0; i < NPP; ++i)
                if (p_on[i]) *reinterpret_cast<u32x4*>(base + pl * PLANE + p_lds[i]) = rp[pl][i];
#pragma unroll
            for (int i = 0; i < NWP; ++i)
                if (w_on[i]) *reinterpret_cast<u32x4*>(base + pl * PLANE + w_lds[i]) = rw[pl][i];
        }
    };

    // ---- MFMA fragments -------------------------------------------------------------------
    const int wave_m = wave % WGM, wave_n = wave / WGM;
    const int pm0 = wave_m * WTM, cn0 = wave_n * WTN;
    const int fr = lane & 15, fk = lane >> 4;
    int p_rd[FM], w_rd[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) p_rd[i] = tile_off(pm0 + i * 16 + fr, fk);
#pragma unroll
    for (int i = 0; i < FN; ++i) w_rd[i] = P_BYTES + tile_off(cn0 + i * 16 + fr, fk);

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int s) __attribute__((always_inline)) {
        const char* base = smem + s * STAGE;
        bf16x8 pf[NP][FM], wf[NP][FN];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
                pf[pl][i] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE + p_rd[i]);
#pragma unroll
            for (int i = 0; i < FN; ++i)
                wf[pl][i] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE + w_rd[i]);
        }
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                if (X3) {
                    // small cross terms first, the dominant hi*hi product last
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[NP - 1][i], pf[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][i], pf[NP - 1][j], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][i], pf[0][j], acc[i][j], 0, 0, 0);
            }
    };

    __syncthreads();   // s_goff visible
    gload(0);
    swrite(0);
    __syncthreads();
    for (int kt = 0; kt < ph.KT; ++kt) {
        const bool more = kt + 1 < ph.KT;
        if (more) gload(kt + 1);
        compute(kt & 1);
        if (more) swrite((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: + bias, + residual, activation, bf16 (hi, lo) store ---------------------
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = m0 + pm0 + j * 16 + fr;
        if (m >= a.M) continue;
        const int b = m / a.HqWq;
        const int rem = m - b * a.HqWq;
        const int qi = rem / a.Wq, qj = rem - qi * a.Wq;
        const int64_t yo = (int64_t)b * a.yb + (int64_t)qi * a.yi + (int64_t)qj * a.yj + ph.y_off;
        const int64_t ro = (int64_t)b * a.rb + (int64_t)qi * a.ri + (int64_t)qj * a.rj;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int c = n0 + cn0 + i * 16 + fk * 4;
            if (c >= a.N) continue;
            const float4 bv = *reinterpret_cast<const float4*>(a.bias + c);
            float v[4] = {acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w};
            if (a.r_hi) {
                const uint2 rh = *reinterpret_cast<const uint2*>(a.r_hi + ro + c);
                v[0] += bf2f(rh.x & 0xffffu); v[1] += bf2f(rh.x >> 16);
                v[2] += bf2f(rh.y & 0xffffu); v[3] += bf2f(rh.y >> 16);
                if (X3) {
                    const uint2 rl = *reinterpret_cast<const uint2*>(a.r_lo + ro + c);
                    v[0] += bf2f(rl.x & 0xffffu); v[1] += bf2f(rl.x >> 16);
                    v[2] += bf2f(rl.y & 0xffffu); v[3] += bf2f(rl.y >> 16);
                }
            }
            if (a.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (a.act == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
            }
            uint32_t h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = f2bf(v[e]);
            *reinterpret_cast<uint2*>(a.y_hi + yo + c) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            if (X3) {
                uint32_t l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) l[e] = f2bf(v[e] - bf2f(h[e]));
                *reinterpret_cast<uint2*>(a.y_lo + yo + c) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
namespace {

struct TileCfg { int bm, bn; };

template <int BM, int BN, int WGM, int WGN, bool X3>
int launch_cfg(const ConvArgs& a, int nphase, size_t lds, hipStream_t s) {
    static bool attr_done = false;
    auto kern = k_conv_igemm<BM, BN, WGM, WGN, X3>;
    if (!attr_done) {
        MF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, 1, nphase);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

template <int BM, int BN, int WGM, int WGN>
int launch_prec(const ConvArgs& a, int nphase, int goff_max, bool x3, hipStream_t s) {
    const size_t stage = (size_t)(BM + BN) * 64 * (x3 ? 2 : 1);
    const size_t lds = 2 * stage + (size_t)goff_max * 4;
    return x3 ? launch_cfg<BM, BN, WGM, WGN, true>(a, nphase, lds, s)
              : launch_cfg<BM, BN, WGM, WGN, false>(a, nphase, lds, s);
}

int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

int mf_conv_plan_create(ConvPlan* p, const mf_conv2d_desc& d, const float* weight, const float* bias,
                        const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                        const float* bn_var, int precision) {
    MF_REQUIRE(d.cin > 0 && d.cout > 0 && d.kh > 0 && d.kw > 0, "conv: bad channel/kernel size");
    MF_REQUIRE(d.cout % 4 == 0, "conv: cout=%d must be a multiple of 4 for the MFMA path", d.cout);
    MF_REQUIRE(d.stride_h > 0 && d.stride_w > 0 && d.in_h > 0 && d.in_w > 0, "conv: bad stride/input size");
    MF_REQUIRE(precision == MF_PREC_BF16 || precision == MF_PREC_BF16X3, "conv: unknown precision %d", precision);
    p->d = d;
    p->precision = precision;
    p->cin_pad = (d.cin + 7) / 8 * 8;
    const int cpg = p->cin_pad / 8;
    p->phase_taps.clear(); p->phase_oy.clear(); p->phase_ox.clear();

    if (!d.transposed) {
        p->out_h = (d.in_h + 2 * d.pad_h - d.kh) / d.stride_h + 1;
        p->out_w = (d.in_w + 2 * d.pad_w - d.kw) / d.stride_w + 1;
        MF_REQUIRE(p->out_h > 0 && p->out_w > 0, "conv: empty output");
        p->Hq = p->out_h; p->Wq = p->out_w;
        p->out_step = 1; p->in_step_h = d.stride_h; p->in_step_w = d.stride_w;
        std::vector<ConvPlan::Tap> taps;
        for (int ky = 0; ky < d.kh; ++ky)
            for (int kx = 0; kx < d.kw; ++kx) taps.push_back({ky - d.pad_h, kx - d.pad_w});
        p->phase_taps.push_back(taps);
        p->phase_oy.push_back(0); p->phase_ox.push_back(0);
        // last anchor + largest displacement may run past the input by (pad - slack)
        int need = std::max(d.pad_h, d.pad_w);
        const int over_h = (p->out_h - 1) * d.stride_h + d.kh - 1 - d.pad_h - (d.in_h - 1);
        const int over_w = (p->out_w - 1) * d.stride_w + d.kw - 1 - d.pad_w - (d.in_w - 1);
        need = std::max(need, std::max(over_h, over_w));
        p->in_halo_need = std::max(need, 0);
    } else {
        MF_REQUIRE(d.stride_h == d.stride_w && d.kh == d.kw && d.pad_h == d.pad_w, "convT: square only");
        const int s = d.stride_h, k = d.kh, pad = d.pad_h;
        p->out_h = (d.in_h - 1) * s - 2 * pad + k + d.output_padding;
        p->out_w = (d.in_w - 1) * s - 2 * pad + k + d.output_padding;
        if (s == 1) {
            MF_REQUIRE(d.in_h == 1 && d.in_w == 1 && pad == 0,
                       "convT stride 1 is only built for 1x1 inputs without padding (wav2lip.py:60)");
            // out[oy][ox] = in[0][0] * w[oy][ox]: k*k single-tap phases on a 1x1 quotient grid
            p->Hq = p->Wq = 1; p->out_step = 1; p->in_step_h = p->in_step_w = 1;
            for (int ky = 0; ky < k; ++ky)
                for (int kx = 0; kx < k; ++kx) {
                    p->phase_taps.push_back({{0, 0}});
                    p->phase_oy.push_back(ky); p->phase_ox.push_back(kx);
                }
            p->in_halo_need = 0;
        } else {
            MF_REQUIRE(p->out_h % s == 0 && p->out_w % s == 0, "convT: output %dx%d not a multiple of stride", p->out_h, p->out_w);
            p->Hq = p->out_h / s; p->Wq = p->out_w / s;
            p->out_step = s; p->in_step_h = p->in_step_w = 1;
            int dmin = 0, dmax = 0;
            for (int ry = 0; ry < s; ++ry)
                for (int rx = 0; rx < s; ++rx) {
                    std::vector<ConvPlan::Tap> taps;
                    for (int ky = 0; ky < k; ++ky) {
                        if ((ry + pad - ky) % s != 0) continue;
                        for (int kx = 0; kx < k; ++kx) {
                            if ((rx + pad - kx) % s != 0) continue;
                            const int dy = (ry + pad - ky) / s, dx = (rx + pad - kx) / s;
                            taps.push_back({dy, dx});
                            dmin = std::min(dmin, std::min(dy, dx));
                            dmax = std::max(dmax, std::max(dy, dx));
                        }
                    }
                    MF_REQUIRE(!taps.empty(), "convT: phase without taps is not supported");
                    p->phase_taps.push_back(taps);
                    p->phase_oy.push_back(ry); p->phase_ox.push_back(rx);
                }
            const int over = std::max(p->Hq - 1 + dmax - (d.in_h - 1), p->Wq - 1 + dmax - (d.in_w - 1));
            p->in_halo_need = std::max(std::max(-dmin, over), 0);
        }
    }
    p->nphase = (int)p->phase_taps.size();
    MF_REQUIRE(p->nphase <= MF_MAX_PHASE, "conv: too many phases");
    p->Npad = (d.cout + 15) / 16 * 16;

    // ---- fold BatchNorm (eval mode, eps 1e-5: conv.py:10) into weight scale and bias ---------
    std::vector<float> scale(d.cout, 1.f), fbias(p->Npad, 0.f);
    for (int n = 0; n < d.cout; ++n) {
        const float b0 = bias ? bias[n] : 0.f;
        if (bn_gamma) {
            const double sc = (double)bn_gamma[n] / std::sqrt((double)bn_var[n] + 1e-5);
            scale[n] = (float)sc;
            fbias[n] = (float)(((double)b0 - (double)bn_mean[n]) * sc + (double)bn_beta[n]);
        } else {
            fbias[n] = b0;
        }
    }

    // ---- pack: per phase [KT][Npad][32], K groups tap-major ----------------------------------
    int64_t total = 0;
    int goff_total = 0;
    for (int ph = 0; ph < p->nphase; ++ph) {
        const int ngroups = (int)p->phase_taps[ph].size() * cpg;
        const int KT = cdiv(ngroups, 4);
        p->ph[ph].goff_begin = goff_total;
        p->ph[ph].ngroups = KT * 4;
        p->ph[ph].KT = KT;
        p->ph[ph].w_off = total;
        p->ph[ph].y_off = 0;
        total += (int64_t)KT * p->Npad * 32;
        goff_total += KT * 4;
    }
    p->goff_total = goff_total;
    std::vector<bf16_t> hi(total, 0), lo(total, 0);
    const int k = d.kh;  // (transposed: square)
    for (int ph = 0; ph < p->nphase; ++ph) {
        const auto& taps = p->phase_taps[ph];
        for (size_t ti = 0; ti < taps.size(); ++ti) {
            int ky, kx;
            if (!d.transposed) {
                ky = taps[ti].dy + d.pad_h; kx = taps[ti].dx + d.pad_w;
            } else if (d.stride_h == 1) {
                ky = p->phase_oy[ph]; kx = p->phase_ox[ph];
            } else {
                ky = p->phase_oy[ph] + d.pad_h - taps[ti].dy * d.stride_h;
                kx = p->phase_ox[ph] + d.pad_w - taps[ti].dx * d.stride_w;
            }
            for (int c = 0; c < d.cin; ++c) {
                const int g = (int)ti * cpg + c / 8;
                const int kt = g / 4, e = (g % 4) * 8 + c % 8;
                for (int n = 0; n < d.cout; ++n) {
                    const float w = d.transposed
                        ? weight[(((int64_t)c * d.cout + n) * k + ky) * k + kx]
                        : weight[(((int64_t)n * d.cin + c) * d.kh + ky) * d.kw + kx];
                    const float wf = w * scale[n];
                    const int64_t idx = p->ph[ph].w_off + ((int64_t)kt * p->Npad + n) * 32 + e;
                    const bf16_t h = mf_f2bf(wf);
                    hi[idx] = h;
                    lo[idx] = mf_f2bf(wf - mf_bf2f(h));
                }
            }
        }
    }
    MF_HIP(hipMalloc(&p->w_hi, total * sizeof(bf16_t)));
    MF_HIP(hipMemcpy(p->w_hi, hi.data(), total * sizeof(bf16_t), hipMemcpyHostToDevice));
    if (precision == MF_PREC_BF16X3) {
        MF_HIP(hipMalloc(&p->w_lo, total * sizeof(bf16_t)));
        MF_HIP(hipMemcpy(p->w_lo, lo.data(), total * sizeof(bf16_t), hipMemcpyHostToDevice));
    }
    MF_HIP(hipMalloc(&p->bias, p->Npad * sizeof(float)));
    MF_HIP(hipMemcpy(p->bias, fbias.data(), p->Npad * sizeof(float), hipMemcpyHostToDevice));
    MF_HIP(hipMalloc(&p->goff, goff_total * sizeof(int)));
    p->bound_in_ld = p->bound_in_wp = -1;
    return MF_OK;
}

void mf_conv_plan_destroy(ConvPlan* p) {
    if (!p) return;
    if (p->w_hi) (void)hipFree(p->w_hi);
    if (p->w_lo) (void)hipFree(p->w_lo);
    if (p->bias) (void)hipFree(p->bias);
    if (p->goff) (void)hipFree(p->goff);
    p->w_hi = p->w_lo = nullptr; p->bias = nullptr; p->goff = nullptr;
}

int mf_conv_bind(ConvPlan* p, const ActBuf& in) {
    MF_REQUIRE(in.halo >= p->in_halo_need, "conv: input halo %d < required %d", in.halo, p->in_halo_need);
    MF_REQUIRE(in.H == p->d.in_h && in.W == p->d.in_w, "conv: plan built for %dx%d input, bound to %dx%d",
               p->d.in_h, p->d.in_w, in.H, in.W);
    MF_REQUIRE(in.C % 8 == 0 && in.C >= p->cin_pad, "conv: input buffer has %d channels, need >= %d (multiple of 8)", in.C, p->cin_pad);
    if (p->bound_in_ld == in.C && p->bound_in_wp == in.Wp()) return MF_OK;
    const int cpg = p->cin_pad / 8;
    std::vector<int> goff(p->goff_total, 0);
    for (int ph = 0; ph < p->nphase; ++ph) {
        const auto& taps = p->phase_taps[ph];
        const int real = (int)taps.size() * cpg;
        for (int g = 0; g < p->ph[ph].ngroups; ++g) {
            const int gg = g < real ? g : 0;   // padding groups re-read group 0 against zero weights
            const int ti = gg / cpg, cg = gg % cpg;
            goff[p->ph[ph].goff_begin + g] =
                ((taps[ti].dy + in.halo) * in.Wp() + (taps[ti].dx + in.halo)) * in.C + cg * 8;
        }
    }
    MF_HIP(hipMemcpy(p->goff, goff.data(), goff.size() * sizeof(int), hipMemcpyHostToDevice));
    p->bound_in_ld = in.C; p->bound_in_wp = in.Wp();
    return MF_OK;
}

int mf_conv_launch(const ConvPlan* p, const ActView& in, const ActView& out, const ActView& res,
                   int batch, hipStream_t stream) {
    const ActBuf& ib = *in.buf;
    const ActBuf& ob = *out.buf;
    MF_REQUIRE(p->bound_in_ld == ib.C && p->bound_in_wp == ib.Wp(), "conv: plan not bound to this input geometry");
    MF_REQUIRE(in.C >= p->cin_pad && in.coff % 8 == 0 && in.coff + in.C <= ib.C, "conv: bad input view");
    MF_REQUIRE(out.C == p->d.cout && out.coff % 4 == 0 && out.coff + out.C <= ob.C, "conv: bad output view");
    MF_REQUIRE(ob.H == p->out_h && ob.W == p->out_w, "conv: output buffer %dx%d != %dx%d", ob.H, ob.W, p->out_h, p->out_w);
    const bool x3 = p->precision == MF_PREC_BF16X3;
    MF_REQUIRE(!x3 || (ib.lo && ob.lo), "conv: BF16X3 needs lo planes");

    ConvArgs a{};
    a.x_hi = ib.hi + in.coff; a.x_lo = x3 ? ib.lo + in.coff : nullptr;
    a.w_hi = p->w_hi; a.w_lo = p->w_lo; a.bias = p->bias; a.goff = p->goff;
    a.M = batch * p->Hq * p->Wq; a.N = p->d.cout; a.Npad = p->Npad;
    a.HqWq = p->Hq * p->Wq; a.Wq = p->Wq;
    a.xb = ib.per_batch(); a.xi = p->in_step_h * ib.Wp() * ib.C; a.xj = p->in_step_w * ib.C;
    const int64_t ybase = ((int64_t)ob.halo * ob.Wp() + ob.halo) * ob.C + out.coff;
    a.y_hi = ob.hi + ybase; a.y_lo = x3 ? ob.lo + ybase : nullptr;
    a.yb = ob.per_batch(); a.yi = p->out_step * ob.Wp() * ob.C; a.yj = p->out_step * ob.C;
    if (res.buf) {
        const ActBuf& rb = *res.buf;
        MF_REQUIRE(rb.H == p->out_h && rb.W == p->out_w && res.C == p->d.cout && p->out_step == 1,
                   "conv: residual view does not match the output");
        const int64_t rbase = ((int64_t)rb.halo * rb.Wp() + rb.halo) * rb.C + res.coff;
        a.r_hi = rb.hi + rbase; a.r_lo = x3 ? rb.lo + rbase : nullptr;
        a.rb = rb.per_batch(); a.ri = rb.Wp() * rb.C; a.rj = rb.C;
    }
    a.act = p->d.act;
    a.goff_total = p->goff_total;
    int goff_max = 0;
    for (int ph = 0; ph < p->nphase; ++ph) {
        a.ph[ph] = p->ph[ph];
        a.ph[ph].y_off = ((int64_t)p->phase_oy[ph] * ob.Wp() + p->phase_ox[ph]) * ob.C;
        goff_max = std::max(goff_max, p->ph[ph].ngroups);
    }

    // ---- tile selection: largest tile that still yields >= ~2 workgroups per CU ---------------
    const int M = a.M, N = a.N;
    auto tiles = [&](int bm, int bn) { return cdiv(M, bm) * cdiv(N, bn) * p->nphase; };
    int rc;
#define MF_GO(BM, BN, WGM, WGN)                                                            \
    do {                                                                                   \
        a.tiles_m = cdiv(M, BM); a.tiles_n = cdiv(N, BN);                                  \
        rc = launch_prec<BM, BN, WGM, WGN>(a, p->nphase, goff_max, x3, stream);            \
    } while (0)
    if (N <= 16) MF_GO(128, 16, 4, 1);
    else if (N <= 32) MF_GO(128, 32, 4, 1);
    else if (M <= 16) MF_GO(16, 64, 1, 4);
    else if (tiles(128, 128) >= 512 && N % 128 == 0) MF_GO(128, 128, 2, 2);
    else if (tiles(128, 64) >= 512) MF_GO(128, 64, 2, 2);
    else MF_GO(64, 64, 2, 2);
#undef MF_GO
    return rc;
}

// Hardware check of the proposed operand format on a real contraction ("fewer passes per product"):   (GPU box)
//   D[n][m] = sum_k W[n][k] X[m][k],  W = wh + wl, X = xh + xl  with  wh, xh in f16  and the residuals in FP8 e4m3 (x 2^s) or FP6 e2m3 (MX block scales)
//   D ~= wh.xh (v_mfma_f32_16x16x32_f16)  +  q(wh).xl + wl.q(xh)  (v_mfma_scale_f32_16x16x128_f8f6f4)
// against an fp64 evaluation of the fp32 inputs, next to bf16x3 (what ships) on the same data.  One wave per 16 x 16 output tile, operands straight
// from global memory in the lane layouts tools/mx_cross_probe.hip established -- a numerics probe, not a fast kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/mx_gemm_probe.hip -o /tmp/mxg && /tmp/mxg
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// ---- host-side number formats -------------------------------------------------------------------------------------------------
static float bf16_round(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; memcpy(&f, &u, 4); return f; }
static uint16_t bf16_bits(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float f16_round(float f) { return (float)(_Float16)f; }
static uint16_t f16_bits(float f) { _Float16 h = (_Float16)f; uint16_t b; memcpy(&b, &h, 2); return b; }
static double dec_e4m3(uint8_t b) { const int s = b >> 7, e = (b >> 3) & 15, m = b & 7; const double v = e == 0 ? m / 8.0 * std::ldexp(1.0, -6) : (1.0 + m / 8.0) * std::ldexp(1.0, e - 7); return s ? -v : v; }
static uint8_t enc_e4m3(double x) {   // round to nearest (ties to even on the code grid), saturating at 448
    uint8_t best = 0; double bd = 1e300;
    const uint8_t sgn = x < 0 ? 0x80 : 0; const double a = std::fabs(x);
    for (int c = 0; c < 0x7f; ++c) { const double d = std::fabs(dec_e4m3((uint8_t)c) - a); if (d < bd || (d == bd && !(c & 1))) { bd = d; best = (uint8_t)c; } }
    return sgn | best;
}
static double dec_e2m3(uint8_t b) { const int s = (b >> 5) & 1, e = (b >> 3) & 3, m = b & 7; const double v = e == 0 ? m / 8.0 : (1.0 + m / 8.0) * std::ldexp(1.0, e - 1); return s ? -v : v; }
static uint8_t enc_e2m3(double x) {
    uint8_t best = 0; double bd = 1e300;
    const uint8_t sgn = x < 0 ? 0x20 : 0; const double a = std::fabs(x);
    for (int c = 0; c < 0x20; ++c) { const double d = std::fabs(dec_e2m3((uint8_t)c) - a); if (d < bd || (d == bd && !(c & 1))) { bd = d; best = (uint8_t)c; } }
    return sgn | best;
}

// ---- device: one wave = one 16 x 16 tile of D, K in slabs of 128 -------------------------------------------------------------------------
// operand images are pre-arranged per (tile row block, slab, lane): nothing to compute on the device but MFMAs
template <int FMT>   // 0: FP8 corrections, 2: FP6 corrections
__global__ void k_split(const f16x8* wh, const f16x8* xh, const i32x8* w8h, const i32x8* x8l, const i32x8* w8l, const i32x8* x8h, const int* s_w8h, const int* s_x8l,
                        const int* s_w8l, const int* s_x8h, float* D, int M, int N, int slabs) {
    const int l = threadIdx.x, tn = blockIdx.x, tm = blockIdx.y;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < slabs; ++s) {
        for (int q = 0; q < 4; ++q)      // main term: four 32-deep f16 steps
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[((tn * slabs + s) * 4 + q) * 64 + l], xh[((tm * slabs + s) * 4 + q) * 64 + l], acc, 0, 0, 0);
        const int wa = (tn * slabs + s) * 64 + l, xa = (tm * slabs + s) * 64 + l;
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8h[wa], x8l[xa], acc, FMT, FMT, 0, s_w8h[wa], 0, s_x8l[xa]);
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8l[wa], x8h[xa], acc, FMT, FMT, 0, s_w8l[wa], 0, s_x8h[xa]);
    }
    for (int r = 0; r < 4; ++r) D[(size_t)(tn * 16 + (l >> 4) * 4 + r) * M + tm * 16 + (l & 15)] = acc[r];
}
__global__ void k_bf16x3(const bf16x8* wh, const bf16x8* wl, const bf16x8* xh, const bf16x8* xl, float* D, int M, int N, int slabs) {
    const int l = threadIdx.x, tn = blockIdx.x, tm = blockIdx.y;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < slabs * 4; ++s) {
        const int wa = (tn * slabs * 4 + s) * 64 + l, xa = (tm * slabs * 4 + s) * 64 + l;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[wa], xh[xa], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[wa], xl[xa], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[wa], xh[xa], acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(size_t)(tn * 16 + (l >> 4) * 4 + r) * M + tm * 16 + (l & 15)] = acc[r];
}

// K index of element t (0..31) of lane group g in a 128-deep slab: FP8 operands are two 16-element halves 64 apart, FP6 operands 32 consecutive K
static int kmap(int fmt, int g, int t) { return fmt == 0 ? (t < 16 ? 16 * g + t : 64 + 16 * g + (t - 16)) : 32 * g + t; }

struct Packed { std::vector<i32x8> v; std::vector<int> scale; };
// rows x K matrix (row-major fp32, already the VALUES to encode) -> per (row block, slab, lane) register images + E8M0 scale bytes.
// fmt 0: one power-of-two scale for the whole matrix (FP8 has range to spare); fmt 2: a scale per (row, 32-element K block) as OCP MX prescribes.
static Packed pack(const std::vector<double>& a, int rows, int K, int fmt) {
    const int slabs = K / 128, bits = fmt == 0 ? 8 : 6;
    Packed p; p.v.assign((size_t)rows / 16 * slabs * 64, i32x8{0, 0, 0, 0, 0, 0, 0, 0}); p.scale.assign(p.v.size(), 127);
    double gmax = 0; for (double x : a) gmax = std::fmax(gmax, std::fabs(x));
    const int gexp = gmax > 0 ? (int)std::floor(std::log2(256.0 / gmax)) : 0;          // fmt 0: values scaled into [.., 256]
    for (int rb = 0; rb < rows / 16; ++rb)
        for (int s = 0; s < slabs; ++s)
            for (int l = 0; l < 64; ++l) {
                const int row = rb * 16 + (l & 15), g = l >> 4;
                const size_t idx = ((size_t)rb * slabs + s) * 64 + l;
                // the scale byte of lane (i + 16 kb) belongs to K block kb = 32 consecutive K of row i
                int e;
                if (fmt == 0) e = gexp;
                else {
                    double bm = 0; for (int t = 0; t < 32; ++t) bm = std::fmax(bm, std::fabs(a[(size_t)row * K + s * 128 + 32 * g + t]));
                    e = bm > 0 ? -((int)std::floor(std::log2(bm)) - 2) : 0;           // block maximum lands in [4, 8)
                }
                p.scale[idx] = 127 - e;
                uint8_t* dst = reinterpret_cast<uint8_t*>(&p.v[idx]);
                for (int t = 0; t < 32; ++t) {
                    const int k = s * 128 + kmap(fmt, g, t);
                    // the scale that will multiply this element is the one of ITS K block (lane group k / 32 within the slab)
                    int ek = e;
                    if (fmt == 2) {
                        double bm = 0; const int kb = (k % 128) / 32;
                        for (int u = 0; u < 32; ++u) bm = std::fmax(bm, std::fabs(a[(size_t)row * K + s * 128 + 32 * kb + u]));
                        ek = bm > 0 ? -((int)std::floor(std::log2(bm)) - 2) : 0;
                    }
                    const double v = a[(size_t)row * K + k] * std::ldexp(1.0, ek);
                    const uint8_t code = fmt == 0 ? enc_e4m3(v) : enc_e2m3(v);
                    const int bit = bits * t;
                    for (int b = 0; b < bits; ++b) if (code >> b & 1) dst[(bit + b) >> 3] |= 1 << ((bit + b) & 7);
                }
            }
    return p;
}
template <class T, class F>
static std::vector<T> pack16(const std::vector<float>& a, int rows, int K, F bits) {   // f16 / bf16 operands of the 16x16x32 shapes: lane l = row l % 16, K 8 (l / 16) .. +7
    std::vector<T> v((size_t)rows / 16 * (K / 32) * 64);
    for (int rb = 0; rb < rows / 16; ++rb)
        for (int s = 0; s < K / 32; ++s)
            for (int l = 0; l < 64; ++l) {
                uint16_t w[8];
                for (int t = 0; t < 8; ++t) w[t] = bits(a[(size_t)(rb * 16 + (l & 15)) * K + s * 32 + 8 * (l >> 4) + t]);
                memcpy(&v[((size_t)rb * (K / 32) + s) * 64 + l], w, 16);
            }
    return v;
}
template <class T> static T* up(const std::vector<T>& h) { T* d = nullptr; hipMalloc(&d, h.size() * sizeof(T)); hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return d; }

int main() {
    const int M = 256, N = 256, K = 4608;                       // a 3x3 x 512-channel contraction
    std::mt19937 rng(5);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> W((size_t)N * K), X((size_t)M * K);
    for (auto& v : W) v = nd(rng) * 0.02f;                       // weights ~ sqrt(2 / K)
    for (auto& v : X) { const float z = nd(rng) * 1.5f; v = z / (1.f + std::exp(-z)); }   // SiLU-shaped activations
    std::vector<double> ref((size_t)N * M);
    for (int n = 0; n < N; ++n) for (int m = 0; m < M; ++m) { double a = 0; for (int k = 0; k < K; ++k) a += (double)W[(size_t)n * K + k] * X[(size_t)m * K + k]; ref[(size_t)n * M + m] = a; }
    double mag = 0; for (double v : ref) mag = std::fmax(mag, std::fabs(v));
    float* dD; CK(hipMalloc(&dD, (size_t)N * M * 4));
    std::vector<float> D((size_t)N * M);
    auto report = [&](const char* name) {
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, rms = 0; for (size_t i = 0; i < D.size(); ++i) { const double e = D[i] - ref[i]; worst = std::fmax(worst, std::fabs(e)); rms += e * e; }
        printf("%-34s max |err| %.3e  rms %.3e   (max |D| %.3f: %.2e relative)\n", name, worst, std::sqrt(rms / D.size()), mag, worst / mag);
    };
    {   // bf16x3
        std::vector<float> wh(W.size()), wl(W.size()), xh(X.size()), xl(X.size());
        for (size_t i = 0; i < W.size(); ++i) { wh[i] = bf16_round(W[i]); wl[i] = bf16_round(W[i] - wh[i]); }
        for (size_t i = 0; i < X.size(); ++i) { xh[i] = bf16_round(X[i]); xl[i] = bf16_round(X[i] - xh[i]); }
        auto P = [&](const std::vector<float>& a, int rows) { return up(pack16<bf16x8>(a, rows, K, bf16_bits)); };
        hipLaunchKernelGGL(k_bf16x3, dim3(N / 16, M / 16), dim3(64), 0, 0, P(wh, N), P(wl, N), P(xh, M), P(xl, M), dD, M, N, K / 128);
        report("bf16x3 (3 passes, ships)");
    }
    for (int fmt : {0, 2}) {
        std::vector<float> wh(W.size()), xh(X.size());
        std::vector<double> wl(W.size()), xl(X.size()), whd(W.size()), xhd(X.size());
        for (size_t i = 0; i < W.size(); ++i) { wh[i] = f16_round(W[i]); wl[i] = (double)W[i] - wh[i]; whd[i] = wh[i]; }
        for (size_t i = 0; i < X.size(); ++i) { xh[i] = f16_round(X[i]); xl[i] = (double)X[i] - xh[i]; xhd[i] = xh[i]; }
        const Packed w8h = pack(whd, N, K, fmt), x8l = pack(xl, M, K, fmt), w8l = pack(wl, N, K, fmt), x8h = pack(xhd, M, K, fmt);
        f16x8* dwh = up(pack16<f16x8>(wh, N, K, f16_bits)); f16x8* dxh = up(pack16<f16x8>(xh, M, K, f16_bits));
        if (fmt == 0) hipLaunchKernelGGL(k_split<0>, dim3(N / 16, M / 16), dim3(64), 0, 0, dwh, dxh, up(w8h.v), up(x8l.v), up(w8l.v), up(x8h.v), up(w8h.scale), up(x8l.scale), up(w8l.scale), up(x8h.scale), dD, M, N, K / 128);
        else hipLaunchKernelGGL(k_split<2>, dim3(N / 16, M / 16), dim3(64), 0, 0, dwh, dxh, up(w8h.v), up(x8l.v), up(w8l.v), up(x8h.v), up(w8h.scale), up(x8l.scale), up(w8l.scale), up(x8h.scale), dD, M, N, K / 128);
        report(fmt == 0 ? "f16 + FP8 e4m3 corrections (2 passes)" : "f16 + FP6 e2m3 MX corrections (1.5)");
    }
    {   // f16 alone, for scale
        std::vector<float> wh(W.size()), xh(X.size());
        for (size_t i = 0; i < W.size(); ++i) wh[i] = f16_round(W[i]);
        for (size_t i = 0; i < X.size(); ++i) xh[i] = f16_round(X[i]);
        std::vector<float> z(W.size(), 0.f);
        std::vector<double> zd(W.size(), 0.0), zx(X.size(), 0.0);
        const Packed zw = pack(zd, N, K, 0), zxx = pack(zx, M, K, 0);
        hipLaunchKernelGGL(k_split<0>, dim3(N / 16, M / 16), dim3(64), 0, 0, up(pack16<f16x8>(wh, N, K, f16_bits)), up(pack16<f16x8>(xh, M, K, f16_bits)), up(zw.v), up(zxx.v), up(zw.v), up(zxx.v),
                           up(zw.scale), up(zxx.scale), up(zw.scale), up(zxx.scale), dD, M, N, K / 128);
        report("f16 alone (1 pass)");
    }
    return 0;
}

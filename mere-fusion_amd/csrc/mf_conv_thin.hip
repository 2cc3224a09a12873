// Convolutions with a THIN input (cin <= 16) on large maps: Wav2Lip's first face-encoder layers -- Conv2d(6, 16, k7, s1, p3) at 96^2 and Conv2d(16, 32, k3, s2, p1)
// 96^2 -> 48^2 (wav2lip.py:19-21) -- on MFMA with the input patch staged in LDS ONCE.
//
// The implicit GEMM (mf_conv.hip) stages a [pixels x K] operand tile per K step; with 8 or 16 channels per tap a 64-deep K step is 8 or 4 TAPS of every pixel, so a
// 7 x 7 layer pushes every input pixel through the L2 -> LDS path 49 times (231 MB for a 3.5 MB input) and the layer is bound by the issue of those LDS-DMA pieces
// (27.9 us at batch 16 for 1.4 GFLOP).  Here a workgroup owns an 8 x 16 patch of OUTPUT pixels: it loads the input patch with its halo once (14 x 22 pixels for k7 s1,
// 17 x 33 for k3 s2; 10 - 36 KB in both planes), and every MFMA B fragment is read straight out of it: the contraction index of a 32-deep step is (tap, channel) with
// 32 / CINP taps per step, so lane (pixel, k-group) reads the 16 bytes of ITS tap's pixel -- a per-lane LDS address, no data movement between lanes.  Weights are
// packed on the host as lane-ordered A fragments (rows = output channels) in that K order and read from L2 (13 or 5 steps x 1 KiB per 16 channels and plane).
// Same operand roles and numerics as the other conv kernels: (hi, lo) bf16 planes, three MFMAs per product in bf16x3, fp32 accumulate, BatchNorm folded into weights
// and bias, ReLU, one lane owns 4 consecutive channels of one pixel in the epilogue.
#include "mf_conv.h"
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int TPH = 8, TPW = 16;          // output patch of a workgroup: 4 waves x 2 rows x one 16-pixel fragment

__device__ __forceinline__ uint32_t tf2bf(float f) { return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ float tbf2f(uint32_t h16) { return __uint_as_float(h16 << 16); }

// KS: kernel size (square), ST: stride, CINP: channels per pixel the kernel contracts (8 or 16; the input view's first CINP channels, zero weights past cin),
// NF: 16-channel output fragments (1 or 2)
template <int KS, int ST, int CINP, int NF, bool X3>
__global__ __launch_bounds__(256) void k_conv_thin(const ThinArgs a) {
    constexpr int NP = X3 ? 2 : 1;
    constexpr int IH = (TPH - 1) * ST + KS, IW = (TPW - 1) * ST + KS;       // input patch
    constexpr int PIX_B = CINP * 2;                                          // bytes of a pixel in one plane
    constexpr int PLANE = IH * IW * PIX_B;
    constexpr int TPK = 32 / CINP;                                           // taps per 32-deep MFMA step
    constexpr int NKS = (KS * KS + TPK - 1) / TPK;
    constexpr int PPP = CINP / 8;                                            // 16-byte pieces per pixel and plane
    __shared__ __attribute__((aligned(16))) char smem[NP * PLANE];

    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, y0 = blockIdx.y * TPH, x0 = blockIdx.x * TPW;

    // ---- the input patch, both planes: input pixel of patch position (py, px) is (y0 * ST - pad + py, x0 * ST - pad + px); the buffer's zero ring supplies the padding,
    // positions past the buffer are clamped (they only feed masked outputs)
    for (int i = tid; i < IH * IW * PPP; i += 256) {
        const int pix = i / PPP, piece = i - pix * PPP;
        const int py = pix / IW, px = pix - py * IW;
        int iy = y0 * ST - a.pad + py + a.in_halo, ix = x0 * ST - a.pad + px + a.in_halo;
        iy = iy < 0 ? 0 : (iy < a.in_hp ? iy : a.in_hp - 1);
        ix = ix < 0 ? 0 : (ix < a.in_wp ? ix : a.in_wp - 1);
        const int64_t off = (int64_t)b * a.xb + ((int64_t)iy * a.in_wp + ix) * a.x_ld + piece * 8;
        *reinterpret_cast<uint4*>(smem + pix * PIX_B + piece * 16) = *reinterpret_cast<const uint4*>(a.x_hi + off);
        if (X3) *reinterpret_cast<uint4*>(smem + PLANE + pix * PIX_B + piece * 16) = *reinterpret_cast<const uint4*>(a.x_lo + off);
    }
    __syncthreads();

    // ---- this lane's part of every B fragment: k-group g of step ks is tap ks * TPK + (8 g) / CINP, channels (8 g) % CINP .. + 7
    f32x4 acc[NF][2];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f][0] = acc[f][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int row0 = wave * 2;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        int tap = ks * TPK + (8 * g) / CINP;
        tap = tap < KS * KS ? tap : KS * KS - 1;                             // padding slots of the last step: zero weights against any finite value
        const int dy = tap / KS, dx = tap - dy * KS;
        const int ch_b = ((8 * g) % CINP) * 2;
        bf16x8 w_hi[NF], w_lo[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const bf16_t* wp = a.w + ((size_t)((f * NKS + ks) * NP) * 64 + lane) * 8;
            w_hi[f] = *reinterpret_cast<const bf16x8*>(wp);
            if (X3) w_lo[f] = *reinterpret_cast<const bf16x8*>(wp + 64 * 8);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const char* p = smem + (((row0 + j) * ST + dy) * IW + fr * ST + dx) * PIX_B + ch_b;
            const bf16x8 x_hi = *reinterpret_cast<const bf16x8*>(p);
            bf16x8 x_lo;
            if (X3) x_lo = *reinterpret_cast<const bf16x8*>(p + PLANE);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (X3) {
                    acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_lo[f], x_hi, acc[f][j], 0, 0, 0);
                    acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi[f], x_lo, acc[f][j], 0, 0, 0);
                }
                acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi[f], x_hi, acc[f][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane (fr, g) holds channels 16 f + 4 g .. + 3 of pixel (y0 + row0 + j, x0 + fr)
    const int ox = x0 + fr;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int c = f * 16 + 4 * g;
        const float4 bq = *reinterpret_cast<const float4*>(a.bias + c);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oy = y0 + row0 + j;
            if (oy >= a.H || ox >= a.W || c >= a.N) continue;
            float v[4] = {acc[f][j][0] + bq.x, acc[f][j][1] + bq.y, acc[f][j][2] + bq.z, acc[f][j][3] + bq.w};
            if (a.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (a.act == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
            }
            const int64_t yo = (int64_t)b * a.yb + (int64_t)oy * a.yi + (int64_t)ox * a.yj + c;
            uint32_t h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = tf2bf(v[e]);
            *reinterpret_cast<uint2*>(a.y_hi + yo) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            if (X3) {
                uint32_t l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) l[e] = tf2bf(v[e] - tbf2f(h[e]));
                *reinterpret_cast<uint2*>(a.y_lo + yo) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
            }
        }
    }
}

template <int KS, int ST, int CINP, int NF>
int thin_launch_cfg(const ThinArgs& a, bool x3, hipStream_t s) {
    const dim3 grid((a.W + TPW - 1) / TPW, (a.H + TPH - 1) / TPH, a.batch);
    if (x3) hipLaunchKernelGGL((k_conv_thin<KS, ST, CINP, NF, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_conv_thin<KS, ST, CINP, NF, false>), grid, dim3(256), 0, s, a);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

}  // namespace

// the shapes that are built: (kernel, stride, channels contracted per pixel, 16-channel output fragments)
bool mf_thin_supported(int k, int stride, int cin, int cout) {
    const int cinp = cin <= 8 ? 8 : 16;
    if (cin > 16 || cout % 4 || cout > 32) return false;
    if (k == 7 && stride == 1 && cinp == 8 && cout <= 16) return true;
    if (k == 3 && (stride == 1 || stride == 2) && cout <= 32) return true;
    return false;
}

int mf_thin_nks(int k, int cin) {
    const int cinp = cin <= 8 ? 8 : 16, tpk = 32 / cinp;
    return (k * k + tpk - 1) / tpk;
}

// weight[cout][cin][k][k] (already scaled by the folded BatchNorm) -> lane-ordered A fragments [fragment f][step ks][plane][lane][8]:
// lane = (m = output channel in the fragment, g = k-group), element j = K index 8 g + j of the step = (tap ks * TPK + (8 g + j) / CINP, channel (8 g + j) % CINP)
void mf_thin_pack(const float* w, const float* scale, int cout, int cin, int k, bool x3, std::vector<bf16_t>& dst) {
    const int cinp = cin <= 8 ? 8 : 16, tpk = 32 / cinp, nks = mf_thin_nks(k, cin), nf = (cout + 15) / 16, np = x3 ? 2 : 1;
    dst.assign((size_t)nf * nks * np * 64 * 8, 0);
    for (int f = 0; f < nf; ++f)
        for (int ks = 0; ks < nks; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 15, g = lane >> 4, n = f * 16 + m;
                for (int j = 0; j < 8; ++j) {
                    const int kk = 8 * g + j, tap = ks * tpk + kk / cinp, c = kk % cinp;
                    float v = 0.f;
                    if (n < cout && c < cin && tap < k * k) v = w[(((size_t)n * cin + c) * k + tap / k) * k + tap % k] * scale[n];
                    const bf16_t hi = mf_f2bf(v);
                    dst[((size_t)((f * nks + ks) * np) * 64 + lane) * 8 + j] = hi;
                    if (x3) dst[((size_t)((f * nks + ks) * np + 1) * 64 + lane) * 8 + j] = mf_f2bf(v - mf_bf2f(hi));
                }
            }
}

int mf_thin_launch(const ThinArgs& a, int k, int stride, int cin, int cout, bool x3, hipStream_t s) {
    const int cinp = cin <= 8 ? 8 : 16, nf = (cout + 15) / 16;
#define MF_TCASE(KS, ST, CINP, NF) if (k == KS && stride == ST && cinp == CINP && nf == NF) return thin_launch_cfg<KS, ST, CINP, NF>(a, x3, s);
    MF_TCASE(7, 1, 8, 1)
    MF_TCASE(3, 1, 8, 1) MF_TCASE(3, 1, 8, 2) MF_TCASE(3, 1, 16, 1) MF_TCASE(3, 1, 16, 2)
    MF_TCASE(3, 2, 8, 1) MF_TCASE(3, 2, 8, 2) MF_TCASE(3, 2, 16, 1) MF_TCASE(3, 2, 16, 2)
#undef MF_TCASE
    mf_set_error("thin conv: no kernel for k%d s%d cin %d cout %d", k, stride, cin, cout);
    return MF_ERR_INVALID;
}

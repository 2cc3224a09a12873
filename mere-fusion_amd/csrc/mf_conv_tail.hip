// GroupNorm [+ SiLU] -> Conv2d 3x3 (stride 1, pad 1, <= 16 output channels) in ONE pass over the input: the tail of the VAE decoder
// (diffusers AutoencoderKL `decoder.conv_norm_out` -> `conv_act` -> `conv_out`, 128 -> 3 channels at 256 x 256; musetalk/models/vae.py:96-108 runs it
// through `vae.decode`) and of the UNet (`conv_norm_out` -> `conv_act` -> `conv_out`, 320 -> 4 at 32 x 32; musereal.py:105-107).
//
// Why: as two launches the normalised tensor is written and read back in (hi, lo) planes -- 2 x 268 MB at batch 8 on the VAE's 256^2 x 128 map for a
// convolution of 3.6 GFLOP (k_gn_apply<4> 91 us at 5.9 TB/s + k_conv3x3_halo<8,32> 122 us).  Here a workgroup (4 waves, a 16 x 16-pixel patch) reads the RAW
// 18 x 18-pixel halo patch of a 32-channel slice into registers, applies the finalised GroupNorm (one FMA per value) and SiLU, splits into
// (hi, lo) bf16 and parks it in LDS in the halo kernels' swizzled layout; the nine taps are bf16x3 MFMAs against weight fragments that sit in LDS in lane order
// (16 output-channel rows, the real ones first).  Pixels outside the map are zeros AFTER the activation (the padding of the convolution), so they are masked by
// coordinate, not taken from the input's zero ring.  HBM traffic: the input once.  The next slice's raw loads are in flight under this slice's MFMAs.
#include "mf_nn.h"
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // (native vector: an array of HIP's uint4 struct stayed an alloca and was promoted to LDS)

namespace {

__device__ __forceinline__ uint32_t tf2bf(float f) { return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ float tbf2f(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ int tswz(int hx) { return ((hx >> 2) & 1) << 1; }   // hswz<32> of the halo kernels: a function of the halo column only

struct TailArgs {
    const bf16_t* x_hi; const bf16_t* x_lo;
    int64_t xb; int x_wp, x_ld, x_halo;
    int H, W, n_slices, cout4;
    const float* gamma; const float* beta; const double* stats;
    double inv_n; float eps; int groups, cpg, silu;
    const u32x4* w;                 // [slice][tap][plane][lane] 16-byte A fragments
    const float* bias;              // [16]
    bf16_t* y_hi; bf16_t* y_lo;
    int64_t yb; int yi, yj;
    int tiles_x, tiles_per_img;
};

constexpr int TP = 16, THW = TP + 2, TROWS = THW * THW;      // 18 x 18 halo patch
constexpr int TPIX_PLANE = TROWS * 64;                       // one plane of a 32-channel slice
constexpr int TCH = TROWS * 4;                               // 16-byte chunks of a plane
constexpr int TNI = (TCH + 255) / 256;                       // ... per thread
constexpr int TWCH = 9 * 2 * 64;                             // weight chunks of a slice
constexpr int TNW = (TWCH + 255) / 256;

__global__ __launch_bounds__(256) void k_gn_conv3_tail(const TailArgs a) {
    __shared__ __attribute__((aligned(16))) char s_pix[2 * TPIX_PLANE];
    __shared__ u32x4 s_w[TWCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / a.tiles_per_img;
    const int tr = blockIdx.x - b * a.tiles_per_img;
    const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
    const int y0 = ty * TP, x0 = tx * TP;
    const int kg = tid & 3;                                  // this thread's 8-channel group in every chunk it moves (256 % 4 == 0)

    // chunk i of this thread: halo row r = (tid + 256 i) / 4 -> pixel (y0 + hy - 1, x0 + hx - 1)
    int64_t goff[TNI];
    int loff[TNI];
    int inmask = 0;                                           // bit i: chunk i lies inside the map
#pragma unroll
    for (int i = 0; i < TNI; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx < TCH ? idx >> 2 : TROWS - 1;
        const int hy = r / THW, hx = r - hy * THW;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        inmask |= (idx < TCH && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? 1 << i : 0;
        const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
        goff[i] = (int64_t)b * a.xb + ((int64_t)(cy + a.x_halo) * a.x_wp + cx + a.x_halo) * a.x_ld + kg * 8;
        loff[i] = idx < TCH ? r * 64 + ((kg ^ tswz(hx)) << 4) : -1;
    }
    uint4 rh[TNI], rl[TNI];
    u32x4 rw[TNW];
#define TAIL_FETCH(slice_)                                                                                   \
    {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < TNI; ++i) {                                                    \
            rh[i] = *reinterpret_cast<const uint4*>(a.x_hi + goff[i] + (slice_) * 32);                       \
            rl[i] = *reinterpret_cast<const uint4*>(a.x_lo + goff[i] + (slice_) * 32);                       \
        }                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < TNW; ++i) {                                                    \
            const int idx_ = tid + 256 * i;                                                                  \
            rw[i] = a.w[(size_t)(slice_) * TWCH + (idx_ < TWCH ? idx_ : 0)];                                 \
        }                                                                                                    \
    }

    const int fr = lane & 15, fk = lane >> 4;
    const int row0 = wave * 4;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    TAIL_FETCH(0);
    for (int slice = 0; slice < a.n_slices; ++slice) {
        // (mean, rstd, gamma, beta) of this thread's 8 channels in this slice -- as k_gn_apply forms them
        const int c0 = slice * 32 + kg * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(a.gamma + c0), g1 = *reinterpret_cast<const float4*>(a.gamma + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(a.beta + c0), b1 = *reinterpret_cast<const float4*>(a.beta + c0 + 4);
        const float sc[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float sh[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        // y = x * ka + kb with ka = rstd * gamma, kb = beta - mean * ka (one FMA per value; k_gn_apply's (x - mean) * rstd * gamma + beta to ~1 ulp)
        float ka[8], kb[8];
        int g_prev = -1;
        float2 st2 = make_float2(0.f, 1.f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c0 + e) / a.cpg;
            if (g != g_prev) {
                const double2 sq = *reinterpret_cast<const double2*>(a.stats + 2 * (b * a.groups + g));
                const double mean = sq.x * a.inv_n;
                const double var = fmax(sq.y * a.inv_n - mean * mean, 0.0);
                st2 = make_float2((float)mean, rsqrtf((float)var + a.eps));
                g_prev = g;
            }
            ka[e] = st2.y * sc[e];
            kb[e] = sh[e] - st2.x * ka[e];
        }
        __syncthreads();                                     // every wave is done with the previous slice's LDS image
#pragma unroll
        for (int i = 0; i < TNI; ++i) {
            if (loff[i] < 0) continue;
            const uint32_t hh[4] = {rh[i].x, rh[i].y, rh[i].z, rh[i].w}, ll[4] = {rl[i].x, rl[i].y, rl[i].z, rl[i].w};
            uint32_t oh[4], ol[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float o2[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float x = tbf2f(q ? hh[e] >> 16 : hh[e] & 0xffffu) + tbf2f(q ? ll[e] >> 16 : ll[e] & 0xffffu);
                    float v = fmaf(x, ka[2 * e + q], kb[2 * e + q]);
                    if (a.silu) v = v * __builtin_amdgcn_rcpf(1.f + __expf(-v));   // (v_rcp_f32, 1 ulp: the IEEE division is ~10 instructions per value)
                    o2[q] = (inmask >> i) & 1 ? v : 0.f;
                }
                const uint32_t h0 = tf2bf(o2[0]), h1 = tf2bf(o2[1]);
                oh[e] = h0 | (h1 << 16);
                ol[e] = tf2bf(o2[0] - tbf2f(h0)) | (tf2bf(o2[1] - tbf2f(h1)) << 16);
            }
            *reinterpret_cast<uint4*>(s_pix + loff[i]) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
            *reinterpret_cast<uint4*>(s_pix + TPIX_PLANE + loff[i]) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
        }
#pragma unroll
        for (int i = 0; i < TNW; ++i) {
            const int idx = tid + 256 * i;
            if (idx < TWCH) s_w[idx] = rw[i];
        }
        if (slice + 1 < a.n_slices) TAIL_FETCH(slice + 1);   // in flight under this slice's MFMAs
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            const bf16x8 w_hi = *reinterpret_cast<const bf16x8*>(&s_w[(tap * 2 + 0) * 64 + lane]);
            const bf16x8 w_lo = *reinterpret_cast<const bf16x8*>(&s_w[(tap * 2 + 1) * 64 + lane]);
            const int hx = fr + dx;
            const int lo16 = (fk ^ tswz(hx)) << 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* p = s_pix + ((row0 + j + dy) * THW + hx) * 64 + lo16;
                const bf16x8 p_hi = *reinterpret_cast<const bf16x8*>(p);
                const bf16x8 p_lo = *reinterpret_cast<const bf16x8*>(p + TPIX_PLANE);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_lo, p_hi, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi, p_lo, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi, p_hi, acc[j], 0, 0, 0);
            }
        }
    }
    // lane (fr, fk) holds output channels fk * 4 .. + 3 of pixel (row0 + j, fr)
    if (fk * 4 < a.cout4) {
        const float4 bq = *reinterpret_cast<const float4*>(a.bias + fk * 4);
        const int ox = x0 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = y0 + row0 + j;
            if (oy >= a.H || ox >= a.W) continue;
            const float v[4] = {acc[j][0] + bq.x, acc[j][1] + bq.y, acc[j][2] + bq.z, acc[j][3] + bq.w};
            uint32_t h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { h[e] = tf2bf(v[e]); l[e] = tf2bf(v[e] - tbf2f(h[e])); }
            const int64_t yo = (int64_t)b * a.yb + (int64_t)oy * a.yi + (int64_t)ox * a.yj + fk * 4;
            *reinterpret_cast<uint2*>(a.y_hi + yo) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            *reinterpret_cast<uint2*>(a.y_lo + yo) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
        }
    }
}

#undef TAIL_FETCH

}  // namespace

bool mf_tail_conv_supported(int cin, int cout, int precision) { return precision == MF_PREC_BF16X3 && cin % 32 == 0 && cin >= 32 && cout >= 1 && cout <= 16; }

int mf_tail_conv_create(TailConv* p, const float* weight, const float* bias, int cin, int cout) {
    MF_REQUIRE(p && weight && mf_tail_conv_supported(cin, cout, MF_PREC_BF16X3), "tail conv: cin %d (multiple of 32) -> cout %d (<= 16)", cin, cout);
    p->cin = cin; p->cout = cout;
    const int ns = cin / 32;
    // A fragment of v_mfma_f32_16x16x32_bf16: lane l holds row l & 15 (output channel), K elements (l >> 4) * 8 .. + 7 (input channels of the slice)
    std::vector<bf16_t> w((size_t)ns * 9 * 2 * 64 * 8, 0);
    for (int s = 0; s < ns; ++s)
        for (int tap = 0; tap < 9; ++tap)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int n = l & 15, c = s * 32 + (l >> 4) * 8 + e;
                    if (n >= cout) continue;
                    const float wf = weight[(((size_t)n * cin + c) * 3 + tap / 3) * 3 + tap % 3];
                    const bf16_t h = mf_f2bf(wf);
                    const size_t base = ((size_t)(s * 9 + tap) * 2) * 512 + (size_t)l * 8 + e;
                    w[base] = h;
                    w[base + 512] = mf_f2bf(wf - mf_bf2f(h));
                }
    std::vector<float> bz(16, 0.f);
    for (int n = 0; n < cout; ++n) bz[n] = bias ? bias[n] : 0.f;
    MF_HIP(hipMalloc(&p->w, w.size() * sizeof(bf16_t)));
    MF_HIP(hipMemcpy(p->w, w.data(), w.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
    MF_HIP(hipMalloc(&p->bias, 16 * sizeof(float)));
    MF_HIP(hipMemcpy(p->bias, bz.data(), 16 * sizeof(float), hipMemcpyHostToDevice));
    return MF_OK;
}

void mf_tail_conv_destroy(TailConv* p) {
    if (!p) return;
    if (p->w) (void)hipFree(p->w);
    if (p->bias) (void)hipFree(p->bias);
    p->w = nullptr; p->bias = nullptr;
}

int mf_gn_conv3_tail(const TailConv& p, const ActView& x, const float* gamma, const float* beta, int groups, float eps, bool silu, double* stats,
                     bool have_stats, const ActView& out, int batch, hipStream_t s) {
    const ActBuf& xb = *x.buf;
    const ActBuf& ob = *out.buf;
    MF_REQUIRE(x.C == p.cin && x.coff % 8 == 0 && xb.lo && ob.lo && xb.H == ob.H && xb.W == ob.W, "gn + conv tail: views do not match the plan");
    MF_REQUIRE(x.C % groups == 0, "gn + conv tail: %d channels, %d groups", x.C, groups);
    const int cout4 = (p.cout + 3) / 4 * 4;
    MF_REQUIRE(out.coff % 4 == 0 && out.coff + cout4 <= ob.C, "gn + conv tail: output view needs room for %d channels", cout4);
    if (!have_stats) {
        const int rc = mf_groupnorm_stats(x, groups, stats, batch, s);
        if (rc) return rc;
    }
    TailArgs a{};
    a.x_hi = xb.hi + x.coff; a.x_lo = xb.lo + x.coff;
    a.xb = xb.per_batch(); a.x_wp = xb.Wp(); a.x_ld = xb.C; a.x_halo = xb.halo;
    a.H = xb.H; a.W = xb.W; a.n_slices = p.cin / 32; a.cout4 = cout4;
    a.gamma = gamma; a.beta = beta; a.stats = stats;
    const int cpg = x.C / groups;
    a.inv_n = 1.0 / ((double)xb.H * xb.W * cpg); a.eps = eps; a.groups = groups; a.cpg = cpg; a.silu = silu ? 1 : 0;
    a.w = reinterpret_cast<const u32x4*>(p.w); a.bias = p.bias;
    const int64_t y0 = ((int64_t)ob.halo * ob.Wp() + ob.halo) * ob.C + out.coff;
    a.y_hi = ob.hi + y0; a.y_lo = ob.lo + y0;
    a.yb = ob.per_batch(); a.yi = ob.Wp() * ob.C; a.yj = ob.C;
    a.tiles_x = (xb.W + TP - 1) / TP;
    a.tiles_per_img = a.tiles_x * ((xb.H + TP - 1) / TP);
    hipLaunchKernelGGL(k_gn_conv3_tail, dim3((unsigned)(batch * a.tiles_per_img)), dim3(256), 0, s, a);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

// Layout-conversion and head kernels (HBM-bound elementwise passes; see mf_aux.h).
#include "mf_aux.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ uint32_t f2bf_d(float f) {
    // round to nearest even in hardware: gfx950's v_cvt_pk_bf16_f32 (the compiler pairs neighbouring calls), a quarter of the integer form's instructions
    return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ float bf2f_d(uint32_t h) { return __uint_as_float(h << 16); }

// one thread = one pixel x 8-channel group; consecutive threads walk x so the NCHW reads coalesce
__global__ void k_nchw_to_act(const float* __restrict__ src, int C, int H, int W, bf16_t* hi, bf16_t* lo,
                              int Cbuf, int halo, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int groups = Cbuf / 8;
    const int x = idx % W;
    int64_t t = idx / W;
    const int y = t % H; t /= H;
    const int g = t % groups;
    const int b = t / groups;
    const int Wp = W + 2 * halo, Hp = H + 2 * halo;
    uint32_t h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        const float v = c < C ? src[(((int64_t)b * C + c) * H + y) * W + x] : 0.f;
        h[e] = f2bf_d(v);
        l[e] = f2bf_d(v - bf2f_d(h[e]));
    }
    const int64_t o = (((int64_t)b * Hp + y + halo) * Wp + x + halo) * Cbuf + g * 8;
    *reinterpret_cast<uint4*>(hi + o) = make_uint4(h[0] | h[1] << 16, h[2] | h[3] << 16, h[4] | h[5] << 16, h[6] | h[7] << 16);
    if (lo) *reinterpret_cast<uint4*>(lo + o) = make_uint4(l[0] | l[1] << 16, l[2] | l[3] << 16, l[4] | l[5] << 16, l[6] | l[7] << 16);
}

__global__ void k_act_to_nchw(const bf16_t* hi, const bf16_t* lo, int Cbuf, int coff, int C, int H, int W,
                              int halo, float* __restrict__ dst, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = idx % W;
    int64_t t = idx / W;
    const int y = t % H; t /= H;
    const int c = t % C;
    const int b = t / C;
    const int Wp = W + 2 * halo, Hp = H + 2 * halo;
    const int64_t o = (((int64_t)b * Hp + y + halo) * Wp + x + halo) * Cbuf + coff + c;
    float v = bf2f_d(hi[o]);
    if (lo) v += bf2f_d(lo[o]);
    dst[idx] = v;
}

__global__ void k_faces_u8(const uint8_t* __restrict__ faces, int H, int W, bf16_t* hi, bf16_t* lo, int halo,
                           int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = idx % W;
    int64_t t = idx / W;
    const int y = t % H;
    const int b = t / H;
    const uint8_t* px = faces + (((int64_t)b * H + y) * W + x) * 3;
    float v[8];
    const bool keep = y < H / 2;   // img_masked[:, face.shape[0]//2:] = 0  (lipreal.py:116)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float f = (float)px[c] / 255.f;
        v[c] = keep ? f : 0.f;
        v[3 + c] = f;
    }
    v[6] = v[7] = 0.f;
    uint32_t h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { h[e] = f2bf_d(v[e]); l[e] = f2bf_d(v[e] - bf2f_d(h[e])); }
    const int Wp = W + 2 * halo, Hp = H + 2 * halo;
    const int64_t o = (((int64_t)b * Hp + y + halo) * Wp + x + halo) * 8;
    *reinterpret_cast<uint4*>(hi + o) = make_uint4(h[0] | h[1] << 16, h[2] | h[3] << 16, h[4] | h[5] << 16, h[6] | h[7] << 16);
    if (lo) *reinterpret_cast<uint4*>(lo + o) = make_uint4(l[0] | l[1] << 16, l[2] | l[3] << 16, l[4] | l[5] << 16, l[6] | l[7] << 16);
}

// VAE.preprocess_img (vae.py:52-82) on an in-memory uint8 BGR crop: RGB order, x = fp32(u8 / 255.) (numpy divides in double, FloatTensor rounds
// once), rows >= H/2 zeroed when half_mask (x * (mask > 0.5), vae.py:75-76), then transforms.Normalize: (x - 0.5) / 0.5 in fp32
__global__ void k_vae_image_u8(const uint8_t* __restrict__ img, int H, int W, int half_mask, bf16_t* hi, bf16_t* lo, int halo, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = idx % W;
    int64_t t = idx / W;
    const int y = t % H;
    const int b = t / H;
    const uint8_t* px = img + (((int64_t)b * H + y) * W + x) * 3;
    const bool keep = !half_mask || y < H / 2;
    float v[8];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float f = (float)((double)px[2 - c] / 255.0);      // BGR -> RGB
        f = keep ? f : 0.f;
        v[c] = (f - 0.5f) / 0.5f;
    }
#pragma unroll
    for (int c = 3; c < 8; ++c) v[c] = 0.f;
    uint32_t h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { h[e] = f2bf_d(v[e]); l[e] = f2bf_d(v[e] - bf2f_d(h[e])); }
    const int Wp = W + 2 * halo, Hp = H + 2 * halo;
    const int64_t o = (((int64_t)b * Hp + y + halo) * Wp + x + halo) * 8;
    *reinterpret_cast<uint4*>(hi + o) = make_uint4(h[0] | h[1] << 16, h[2] | h[3] << 16, h[4] | h[5] << 16, h[6] | h[7] << 16);
    if (lo) *reinterpret_cast<uint4*>(lo + o) = make_uint4(l[0] | l[1] << 16, l[2] | l[3] << 16, l[4] | l[5] << 16, l[6] | l[7] << 16);
}

// one thread per pixel: 32-channel dot products against 3 filters held in registers
template <int CIN>
__global__ void k_head(const bf16_t* hi, const bf16_t* lo, int Cbuf, int coff, int H, int W, int halo,
                       const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ dst,
                       int hwc255, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = idx % W;
    int64_t t = idx / W;
    const int y = t % H;
    const int b = t / H;
    const int Wp = W + 2 * halo, Hp = H + 2 * halo;
    const int64_t o = (((int64_t)b * Hp + y + halo) * Wp + x + halo) * Cbuf + coff;
    float acc[3] = {bias[0], bias[1], bias[2]};
#pragma unroll
    for (int g = 0; g < CIN / 8; ++g) {
        const uint4 vh = *reinterpret_cast<const uint4*>(hi + o + g * 8);
        uint4 vl = make_uint4(0, 0, 0, 0);
        if (lo) vl = *reinterpret_cast<const uint4*>(lo + o + g * 8);
        const uint32_t hh[4] = {vh.x, vh.y, vh.z, vh.w}, ll[4] = {vl.x, vl.y, vl.z, vl.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = bf2f_d(hh[e] & 0xffffu) + bf2f_d(ll[e] & 0xffffu);
            const float a1 = bf2f_d(hh[e] >> 16) + bf2f_d(ll[e] >> 16);
            const int c = g * 8 + e * 2;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[k] = fmaf(a1, w[k * CIN + c + 1], fmaf(a0, w[k * CIN + c], acc[k]));
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float s = 1.f / (1.f + expf(-acc[k]));
        if (hwc255) dst[idx * 3 + k] = s * 255.f;
        else dst[(((int64_t)b * 3 + k) * H + y) * W + x] = s;
    }
}

inline unsigned blocks_for(int64_t total, int bs) { return (unsigned)((total + bs - 1) / bs); }

}  // namespace

int mf_nchw_to_act(const float* src, int C, const ActBuf& dst, int batch, hipStream_t s) {
    MF_REQUIRE(dst.C % 8 == 0 && dst.C >= C, "nchw_to_act: destination has %d channels for %d", dst.C, C);
    const int64_t total = (int64_t)batch * (dst.C / 8) * dst.H * dst.W;
    hipLaunchKernelGGL(k_nchw_to_act, dim3(blocks_for(total, 256)), dim3(256), 0, s, src, C, dst.H, dst.W,
                       dst.hi, dst.lo, dst.C, dst.halo, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_act_to_nchw(const ActView& src, float* dst, int batch, hipStream_t s) {
    const ActBuf& b = *src.buf;
    const int64_t total = (int64_t)batch * src.C * b.H * b.W;
    hipLaunchKernelGGL(k_act_to_nchw, dim3(blocks_for(total, 256)), dim3(256), 0, s, b.hi, b.lo, b.C, src.coff,
                       src.C, b.H, b.W, b.halo, dst, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_faces_u8_to_act(const uint8_t* faces, const ActBuf& dst, int batch, hipStream_t s) {
    MF_REQUIRE(dst.C == 8, "faces_u8_to_act: destination must have 8 channels");
    const int64_t total = (int64_t)batch * dst.H * dst.W;
    hipLaunchKernelGGL(k_faces_u8, dim3(blocks_for(total, 256)), dim3(256), 0, s, faces, dst.H, dst.W, dst.hi,
                       dst.lo, dst.halo, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_vae_image_u8_to_act(const uint8_t* img, const ActBuf& dst, int half_mask, int batch, hipStream_t s) {
    MF_REQUIRE(dst.C == 8, "vae_image_u8_to_act: destination must have 8 channels");
    const int64_t total = (int64_t)batch * dst.H * dst.W;
    hipLaunchKernelGGL(k_vae_image_u8, dim3(blocks_for(total, 256)), dim3(256), 0, s, img, dst.H, dst.W, half_mask, dst.hi, dst.lo, dst.halo, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

int mf_head_1x1_sigmoid(const ActView& src, const float* w, const float* b, float* dst, int hwc255, int batch,
                        hipStream_t s) {
    const ActBuf& sb = *src.buf;
    MF_REQUIRE(src.C == 32, "head: built for the 32-channel output_block.0 activation");
    const int64_t total = (int64_t)batch * sb.H * sb.W;
    hipLaunchKernelGGL(k_head<32>, dim3(blocks_for(total, 256)), dim3(256), 0, s, sb.hi, sb.lo, sb.C, src.coff,
                       sb.H, sb.W, sb.halo, w, b, dst, hwc255, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

namespace {
// e2m3 code of y (already divided by the block scale): sign, 2 exponent bits (bias 1), 3 mantissa bits; round to nearest, saturate at 7.5
__device__ __forceinline__ uint32_t enc_e2m3(float y) {
    const uint32_t sgn = y < 0.f ? 0x20u : 0u;
    float a = fminf(fabsf(y), 7.5f);
    uint32_t code;
    if (a < 1.f) code = (uint32_t)rintf(a * 8.f);                                  // subnormal step 0.125 (8 -> exponent 1, mantissa 0: same bit pattern)
    else {
        const int e = a < 2.f ? 0 : (a < 4.f ? 1 : 2);                             // value = (1 + m / 8) * 2^e
        const uint32_t m = (uint32_t)rintf((a * (e == 0 ? 1.f : (e == 1 ? 0.5f : 0.25f)) - 1.f) * 8.f);   // 0..8
        code = ((uint32_t)(e + 1) << 3) + m;                                       // m == 8 carries into the exponent
        if (code > 0x1fu) code = 0x1fu;
    }
    return sgn | code;
}
// the same code by integer arithmetic on the float's bits (round half up instead of half even: irrelevant at 3 mantissa bits of a residual): |y| < 1 goes
// through 1 + |y|, whose mantissa IS |y| in fixed point; 6 integer ops instead of a compare chain -- the GroupNorm-apply producer encodes 2 codes per element
__device__ __forceinline__ uint32_t enc_e2m3_fast(float y) {
    const uint32_t u = __float_as_uint(y), sgn = (u >> 26) & 0x20u;
    const float a = fminf(__uint_as_float(u & 0x7fffffffu), 7.5f);
    const bool sub = a < 1.f;
    const uint32_t b = __float_as_uint(sub ? a + 1.f : a) + 0x00080000u;
    const uint32_t code = (b >> 20) - (sub ? (127u << 3) : (126u << 3));
    return sgn | (code > 31u ? 31u : code);
}
// one thread = one pixel x one 32-channel block
__global__ void k_nchw_to_act_q(const float* __restrict__ src, int C, int H, int W, bf16_t* hi, bf16_t* lo, int Cbuf, int halo, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int nblk = Cbuf / 32;
    const int x = idx % W;
    int64_t t = idx / W;
    const int y = t % H; t /= H;
    const int g = t % nblk;
    const int b = t / nblk;
    const int Wp = W + 2 * halo, Hp = H + 2 * halo;
    float vh[32], vl[32];
    float mh = 0.f, ml = 0.f;
    for (int e = 0; e < 32; ++e) {
        const int c = g * 32 + e;
        const float v = c < C ? src[(((int64_t)b * C + c) * H + y) * W + x] : 0.f;
        const _Float16 h = (_Float16)v;
        vh[e] = (float)h; vl[e] = v - vh[e];
        mh = fmaxf(mh, fabsf(vh[e])); ml = fmaxf(ml, fabsf(vl[e]));
    }
    const int64_t o = (((int64_t)b * Hp + y + halo) * Wp + x + halo) * Cbuf + g * 32;
    uint16_t* ph = reinterpret_cast<uint16_t*>(hi + o);
    for (int e = 0; e < 32; ++e) { const _Float16 h = (_Float16)vh[e]; ph[e] = __builtin_bit_cast(uint16_t, h); }
    uint8_t* pl = reinterpret_cast<uint8_t*>(lo + o);                               // 64 bytes: [residual block | hi block]
    for (int blk = 0; blk < 2; ++blk) {
        const float* v = blk == 0 ? vl : vh;
        const float m = blk == 0 ? ml : mh;
        int ex = 0;
        if (m > 0.f) { (void)frexpf(m, &ex); ex = 3 - ex; }                        // m * 2^ex in [4, 8)
        const float sc = ldexpf(1.f, ex);
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int e = 0; e < 32; ++e) {
            const uint32_t code = enc_e2m3(v[e] * sc);
            const int bit = 6 * e;
            w[bit >> 5] |= code << (bit & 31);
            if ((bit & 31) > 26) w[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
        }
        w[6] = (uint32_t)(127 - ex) & 0xffu;                                       // E8M0: value = code * 2^(scale - 127)
        uint32_t* d = reinterpret_cast<uint32_t*>(pl + 32 * blk);
        for (int k = 0; k < 8; ++k) d[k] = w[k];
    }
}
}  // namespace

int mf_nchw_to_act_q(const float* src, int C, const ActBuf& dst, int batch, hipStream_t s) {
    MF_REQUIRE(dst.C % 32 == 0 && dst.lo, "nchw_to_act_q: the destination needs 32-channel blocks and a second plane");
    const int64_t total = (int64_t)batch * (dst.C / 32) * dst.H * dst.W;
    hipLaunchKernelGGL(k_nchw_to_act_q, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, s, src, C, dst.H, dst.W, dst.hi, dst.lo, dst.C, dst.halo, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

namespace {
// one thread = one pixel x one 32-channel block (blocks fastest: a wave reads 4 KB of contiguous channels per plane)
__global__ __launch_bounds__(256) void k_affine_silu_to_q(const bf16_t* __restrict__ xh, const bf16_t* __restrict__ xl, int xC, int xcoff, int xhalo, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ post, int C, int silu, int H, int W, bf16_t* yh, bf16_t* yl,
                                                          int yC, int yhalo, int64_t total) {
    // A thread owns one 32-channel block (64 bytes per plane), but a wave moves its 64 blocks as 16-byte pieces with consecutive lanes on consecutive
    // pieces (piece q * 64 + lane belongs to thread (q * 64 + lane) / 4): global requests are whole lines; a wave-private LDS image does the transposition.
    __shared__ uint4 s_img[2][4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // 32-bit index arithmetic throughout (the host checks that both tensors stay below 2^31 elements): 64-bit divisions are emulated
    const int idx_raw = blockIdx.x * 256 + threadIdx.x;
    const bool valid = idx_raw < (int)total;
    const int idx = valid ? idx_raw : (int)total - 1;
    const int nblk = C / 32;
    const int g = idx % nblk;
    int t = idx / nblk;
    const int x = t % W; t /= W;
    const int y = t % H;
    const int b = t / H;
    const int xo = ((b * (H + 2 * xhalo) + y + xhalo) * (W + 2 * xhalo) + x + xhalo) * xC + xcoff + g * 32;
    const int yo = ((b * (H + 2 * yhalo) + y + yhalo) * (W + 2 * yhalo) + x + yhalo) * yC + g * 32;
    const float* sc = scale + b * C + g * 32;
    const float* sh = shift + b * C + g * 32;
    const float* po = post ? post + g * 32 : nullptr;               // per-CHANNEL power of two applied behind the activation (channel equalisation of the MX blocks), or null
    typedef float v16f __attribute__((ext_vector_type(16)));
    typedef _Float16 v32h __attribute__((ext_vector_type(32)));
    typedef unsigned v6u __attribute__((ext_vector_type(6)));
    typedef unsigned v16u __attribute__((ext_vector_type(16)));
    v16f le, lod;                                   // residuals x - f16(x): even / odd channels (the f32 conversion interleaves its two sources)
    float mh = 0.f, ml = 0.f;
    v16u hbits;
    uint4 av[4], cv[4];
    float4 scv[8], shv[8], pov[8];
    const int sub = lane & 3;
    // piece p of the wave's image sits at p ^ ((p >> 4) & 3): 16 consecutive lanes hit 16 different 16-byte bank groups on both the
    // (piece = lane) global side and the (piece = 4 * lane + q) owner side
    {
        // all eight loads in flight before the first LDS write (written as load; store per piece the compiler waits for each load in turn)
        uint4 in_h[4], in_l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = __shfl(xo, (q * 64 + lane) >> 2) + 8 * sub;
            in_h[q] = *reinterpret_cast<const uint4*>(xh + o);
            in_l[q] = *reinterpret_cast<const uint4*>(xl + o);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pc = q * 64 + lane;
            s_img[0][wv][pc ^ ((pc >> 4) & 3)] = in_h[q];
            s_img[1][wv][pc ^ ((pc >> 4) & 3)] = in_l[q];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int pc = lane * 4 + q; av[q] = s_img[0][wv][pc ^ ((pc >> 4) & 3)]; cv[q] = s_img[1][wv][pc ^ ((pc >> 4) & 3)]; }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        scv[q] = *reinterpret_cast<const float4*>(sc + 4 * q); shv[q] = *reinterpret_cast<const float4*>(sh + 4 * q);
        pov[q] = po ? *reinterpret_cast<const float4*>(po + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t aw[4] = {av[q].x, av[q].y, av[q].z, av[q].w}, cw[4] = {cv[q].x, cv[q].y, cv[q].z, cv[q].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * q + e;
            const uint32_t hw = (e & 1) ? aw[e >> 1] >> 16 : aw[e >> 1] & 0xffffu, lw = (e & 1) ? cw[e >> 1] >> 16 : cw[e >> 1] & 0xffffu;
            const float4 s4 = scv[k >> 2], t4 = shv[k >> 2];
            const float sck = (k & 3) == 0 ? s4.x : ((k & 3) == 1 ? s4.y : ((k & 3) == 2 ? s4.z : s4.w));
            const float shk = (k & 3) == 0 ? t4.x : ((k & 3) == 1 ? t4.y : ((k & 3) == 2 ? t4.z : t4.w));
            const float4 p4 = pov[k >> 2];
            const float pok = (k & 3) == 0 ? p4.x : ((k & 3) == 1 ? p4.y : ((k & 3) == 2 ? p4.z : p4.w));
            float v = bf2f_d(hw) + bf2f_d(lw);
            v = v * sck + shk;
            if (silu) v = v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
            v *= pok;                                               // (a power of two: exact)
            const _Float16 h = (_Float16)v;
            const float vhk = (float)h, vlk = v - vhk;
            mh = fmaxf(mh, fabsf(vhk)); ml = fmaxf(ml, fabsf(vlk));
            if (k & 1) lod[k >> 1] = vlk; else le[k >> 1] = vlk;
            const uint32_t hb = __builtin_bit_cast(uint16_t, h);
            if (k & 1) hbits[k >> 1] |= hb << 16; else hbits[k >> 1] = hb;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int pc = lane * 4 + q; s_img[0][wv][pc ^ ((pc >> 4) & 3)] = make_uint4(hbits[4 * q], hbits[4 * q + 1], hbits[4 * q + 2], hbits[4 * q + 3]); }
    // Block scale 2^(floor(log2 max) - 2) puts the block's largest value in [4, 8) (e2m3 tops out at 7.5); the gfx950 conversions divide by
    // 2^exponent(scale operand), round to nearest even and saturate (tools/cvt_fp6_probe.hip), 32 values per instruction, packed 6 bits each in
    // channel order -- the layout the block-scaled MFMA reads.  The E8M0 byte is that exponent.  (A block of zeros / denormals: exponent 0.)
    const uint32_t bl = max((__float_as_uint(ml) >> 23) & 0xffu, 2u) - 2u, bh = max((__float_as_uint(mh) >> 23) & 0xffu, 2u) - 2u;
    const v6u ql = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(le, lod, __uint_as_float(bl << 23));
    const v6u qh = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(v32h, hbits), __uint_as_float(bh << 23));
    {
        const int p0 = lane * 4;
        s_img[1][wv][(p0 + 0) ^ ((p0 >> 4) & 3)] = make_uint4(ql[0], ql[1], ql[2], ql[3]);
        s_img[1][wv][(p0 + 1) ^ ((p0 >> 4) & 3)] = make_uint4(ql[4], ql[5], bl, 0u);
        s_img[1][wv][(p0 + 2) ^ ((p0 >> 4) & 3)] = make_uint4(qh[0], qh[1], qh[2], qh[3]);
        s_img[1][wv][(p0 + 3) ^ ((p0 >> 4) & 3)] = make_uint4(qh[4], qh[5], bh, 0u);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pc = q * 64 + lane, owner = pc >> 2;
        const int o = __shfl(yo, owner) + 8 * sub;
        const bool ok = __shfl((int)valid, owner) != 0;
        const uint4 ph = s_img[0][wv][pc ^ ((pc >> 4) & 3)], pl = s_img[1][wv][pc ^ ((pc >> 4) & 3)];
        if (ok) { *reinterpret_cast<uint4*>(yh + o) = ph; *reinterpret_cast<uint4*>(yl + o) = pl; }
    }
}

// The same conversion with the 32-channel block UNIFORM over a wave: wave `wid` owns block g = wid % nblk of 64 consecutive pixels of one image
// (needs H * W % 64 == 0), so the 96 per-channel parameters are scalar loads into SGPRs instead of 24 vector loads and 96 VGPRs per thread, and the
// four waves of a workgroup cover four neighbouring blocks of the same pixels (their 64-byte pieces are halves of the same lines).  Thread <-> data
// ownership, the LDS transposition and the arithmetic are k_affine_silu_to_q's, bit for bit.
// GN: `scale` / `shift` are not read; the wave forms the affine of its 32 channels itself from the GroupNorm statistics (one channel per lane, mf_gn_affine_pair -- the
// expression k_gn_affine evaluates, same bits) and hands the 64 values to the scalar registers with v_readlane: the k_gn_affine launch in front of every conversion goes
struct GnAffineSrc { const double* stats; const float* gamma; const float* beta; double inv_n; float eps; int groups, cpg; };
template <bool SILU, bool POST, bool GN = false>
__global__ __launch_bounds__(256) void k_affine_silu_to_q_u(const bf16_t* __restrict__ xh, const bf16_t* __restrict__ xl, int xC, int xcoff, int xhalo, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ post, int C, int H, int W, bf16_t* yh, bf16_t* yl,
                                                            int yC, int yhalo, int nchunk, int cpb, int nblk, int xcd_order, const GnAffineSrc gn) {
    __shared__ uint4 s_img[2][4][256];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroup b runs on XCD b % 8.  Each XCD walks its OWN sequence of waves (chunk' = jw / nblk, g = jw % nblk, chunk = 8 * chunk' + xcd): all blocks of a
    // chunk go through one L2, back to back, and every XCD touches every 256-byte piece of the pixel rows -- with wid = b * 4 + wave and C = 512 an XCD would only
    // ever see blocks g = 4 * (xcd % 4) + {0..3}, a quarter of the address interleave (measured 4.2 instead of 5.5 TB/s).
    int chunk, g;
    if (xcd_order) {
        const int jw = (blockIdx.x >> 3) * 4 + wv, cl = __builtin_amdgcn_readfirstlane(jw / nblk);
        g = jw - cl * nblk; chunk = cl * 8 + (blockIdx.x & 7);
    } else {
        const int wid = blockIdx.x * 4 + wv;
        chunk = __builtin_amdgcn_readfirstlane(wid / nblk); g = wid - chunk * nblk;
    }
    if (chunk >= nchunk) return;                                    // (a whole wave; the LDS image is wave-private, no barrier in this kernel)
    const int b = __builtin_amdgcn_readfirstlane(chunk / cpb);
    const int pix = (chunk - b * cpb) * 64 + lane;
    const int y = pix / W, x = pix - y * W;
    const int xo = ((b * (H + 2 * xhalo) + y + xhalo) * (W + 2 * xhalo) + x + xhalo) * xC + xcoff + g * 32;
    const int yo = ((b * (H + 2 * yhalo) + y + yhalo) * (W + 2 * yhalo) + x + yhalo) * yC + g * 32;
    float sc[32], sh[32];                                           // wave-uniform: scalar loads (or, GN, v_readlane of the values formed here)
    if constexpr (GN) {
        const int k = lane & 31, c = g * 32 + k;
        const double2 sq = *reinterpret_cast<const double2*>(gn.stats + 2 * (b * gn.groups + c / gn.cpg));
        float my_sc, my_sh;
        mf_gn_affine_pair(sq.x, sq.y, gn.inv_n, gn.eps, gn.gamma[c], gn.beta[c], my_sc, my_sh);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            sc[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_sc), j));
            sh[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_sh), j));
        }
    } else {
        const float* __restrict__ scp = scale + b * C + g * 32;
        const float* __restrict__ shp = shift + b * C + g * 32;
#pragma unroll
        for (int j = 0; j < 32; ++j) { sc[j] = scp[j]; sh[j] = shp[j]; }
    }
    const float* __restrict__ po = post + g * 32;
    typedef float v16f __attribute__((ext_vector_type(16)));
    typedef _Float16 v32h __attribute__((ext_vector_type(32)));
    typedef unsigned v6u __attribute__((ext_vector_type(6)));
    typedef unsigned v16u __attribute__((ext_vector_type(16)));
    v16f le, lod;
    float mh = 0.f, ml = 0.f;
    v16u hbits;
    uint4 av[4], cv[4];
    const int sub = lane & 3;
    {
        uint4 in_h[4], in_l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = __shfl(xo, (q * 64 + lane) >> 2) + 8 * sub;
            in_h[q] = *reinterpret_cast<const uint4*>(xh + o);
            in_l[q] = *reinterpret_cast<const uint4*>(xl + o);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pc = q * 64 + lane;
            s_img[0][wv][pc ^ ((pc >> 4) & 3)] = in_h[q];
            s_img[1][wv][pc ^ ((pc >> 4) & 3)] = in_l[q];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int pc = lane * 4 + q; av[q] = s_img[0][wv][pc ^ ((pc >> 4) & 3)]; cv[q] = s_img[1][wv][pc ^ ((pc >> 4) & 3)]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t aw[4] = {av[q].x, av[q].y, av[q].z, av[q].w}, cw[4] = {cv[q].x, cv[q].y, cv[q].z, cv[q].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // The operation sequence is spelled out (one fused multiply-add, then separately rounded products): left to -ffp-contract=fast the residual below
            // becomes fma(v, 1 / (1 + e^-v), -f16(v)) in the instantiations without `post`, one rounding less than k_affine_silu_to_q and the GroupNorm-fused
            // convolution (mf_conv_halo2.hip gn_transform) make -- all three must agree bit for bit (tests/test_musetalk_stress.py).
#pragma clang fp contract(off)
            const int k = 8 * q + e;
            const uint32_t hw = (e & 1) ? aw[e >> 1] >> 16 : aw[e >> 1] & 0xffffu, lw = (e & 1) ? cw[e >> 1] >> 16 : cw[e >> 1] & 0xffffu;
            float v = bf2f_d(hw) + bf2f_d(lw);
            v = __builtin_fmaf(v, sc[k], sh[k]);
            if constexpr (SILU) v = v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
            if constexpr (POST) v *= po[k];                         // (a power of two: exact)
            const _Float16 h = (_Float16)v;
            const float vhk = (float)h, vlk = v - vhk;
            mh = fmaxf(mh, fabsf(vhk)); ml = fmaxf(ml, fabsf(vlk));
            if (k & 1) lod[k >> 1] = vlk; else le[k >> 1] = vlk;
            const uint32_t hb = __builtin_bit_cast(uint16_t, h);
            if (k & 1) hbits[k >> 1] |= hb << 16; else hbits[k >> 1] = hb;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int pc = lane * 4 + q; s_img[0][wv][pc ^ ((pc >> 4) & 3)] = make_uint4(hbits[4 * q], hbits[4 * q + 1], hbits[4 * q + 2], hbits[4 * q + 3]); }
    const uint32_t bl = max((__float_as_uint(ml) >> 23) & 0xffu, 2u) - 2u, bh = max((__float_as_uint(mh) >> 23) & 0xffu, 2u) - 2u;
    const v6u ql = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(le, lod, __uint_as_float(bl << 23));
    const v6u qh = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(v32h, hbits), __uint_as_float(bh << 23));
    {
        const int p0 = lane * 4;
        s_img[1][wv][(p0 + 0) ^ ((p0 >> 4) & 3)] = make_uint4(ql[0], ql[1], ql[2], ql[3]);
        s_img[1][wv][(p0 + 1) ^ ((p0 >> 4) & 3)] = make_uint4(ql[4], ql[5], bl, 0u);
        s_img[1][wv][(p0 + 2) ^ ((p0 >> 4) & 3)] = make_uint4(qh[0], qh[1], qh[2], qh[3]);
        s_img[1][wv][(p0 + 3) ^ ((p0 >> 4) & 3)] = make_uint4(qh[4], qh[5], bh, 0u);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pc = q * 64 + lane;
        const int o = __shfl(yo, pc >> 2) + 8 * sub;
        const uint4 ph = s_img[0][wv][pc ^ ((pc >> 4) & 3)], pl = s_img[1][wv][pc ^ ((pc >> 4) & 3)];
        *reinterpret_cast<uint4*>(yh + o) = ph; *reinterpret_cast<uint4*>(yl + o) = pl;
    }
}
}  // namespace

int mf_affine_silu_to_act_q(const ActView& x, const float* scale, const float* shift, int silu, const ActBuf& dst, int batch, hipStream_t s, const float* post,
                            const double* gn_stats, const float* gn_gamma, const float* gn_beta, int gn_groups, float gn_eps) {
    const ActBuf& xb = *x.buf;
    GnAffineSrc gn{};
    if (gn_stats) {
        MF_REQUIRE(gn_gamma && gn_beta && gn_groups > 0 && x.C % gn_groups == 0, "affine_silu_to_act_q: GroupNorm statistics need gamma, beta and a group count that divides C");
        gn.stats = gn_stats; gn.gamma = gn_gamma; gn.beta = gn_beta; gn.groups = gn_groups; gn.cpg = x.C / gn_groups; gn.eps = gn_eps;
        gn.inv_n = 1.0 / ((double)xb.H * xb.W * gn.cpg);
    }
    MF_REQUIRE(x.C % 32 == 0 && x.coff % 8 == 0 && dst.C == x.C && dst.H == xb.H && dst.W == xb.W && dst.lo && xb.lo && scale && shift,
               "affine_silu_to_act_q: needs 32-channel blocks, matching geometry and a second plane on both sides");
    MF_REQUIRE((int64_t)batch * xb.per_batch() < ((int64_t)1 << 31) && (int64_t)batch * dst.per_batch() < ((int64_t)1 << 31),
               "affine_silu_to_act_q: tensors of 2^31 elements or more are not supported (32-bit offsets)");
    const int64_t total = (int64_t)batch * xb.H * xb.W * (x.C / 32);
    // wave-uniform channel blocks, XCD-ordered when a pixel row is wider than one workgroup's four blocks; maps that are not a multiple of 64 pixels keep the
    // per-thread-parameter kernel below (the three implementations measured in round 4: tools/affine_q_probe.hip, profiles/r04_affine_q_probe.txt)
    if ((xb.H * xb.W) % 64 == 0) {
        const int cpb = xb.H * xb.W / 64, nchunk = batch * cpb, nblk = x.C / 32;
        const int64_t waves = (int64_t)nchunk * nblk;
        const int xcd_order = nblk > 4 && waves >= 16384;
        const int64_t waves_per_xcd = (int64_t)((nchunk + 7) / 8) * nblk;
        const dim3 grid(xcd_order ? (unsigned)(8 * ((waves_per_xcd + 3) / 4)) : (unsigned)((waves + 3) / 4));
#define MF_AFFQ_U(S, P, G)                                                                                                                                                  \
    hipLaunchKernelGGL((k_affine_silu_to_q_u<S, P, G>), grid, dim3(256), 0, s, xb.hi, xb.lo, xb.C, x.coff, xb.halo, scale, shift, post, x.C, xb.H, xb.W, dst.hi, dst.lo, dst.C, \
                       dst.halo, nchunk, cpb, nblk, xcd_order, gn)
        if (gn_stats) {        // (the GroupNorm case is always followed by SiLU in the networks that use this format)
            MF_REQUIRE(silu, "affine_silu_to_act_q: the statistics form is built with SiLU");
            if (post) MF_AFFQ_U(true, true, true); else MF_AFFQ_U(true, false, true);
        } else if (silu) { if (post) MF_AFFQ_U(true, true, false); else MF_AFFQ_U(true, false, false); }
        else      { if (post) MF_AFFQ_U(false, true, false); else MF_AFFQ_U(false, false, false); }
#undef MF_AFFQ_U
        MF_HIP(hipGetLastError());
        return MF_OK;
    }
    MF_REQUIRE(!gn_stats, "affine_silu_to_act_q: maps that are not a multiple of 64 pixels take scale / shift arrays (mf_groupnorm_affine first)");
    hipLaunchKernelGGL(k_affine_silu_to_q, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xb.hi, xb.lo, xb.C, x.coff, xb.halo, scale, shift, post, x.C, silu, xb.H, xb.W, dst.hi,
                       dst.lo, dst.C, dst.halo, total);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

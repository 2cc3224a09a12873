// 3x3 stride-1 convolution on MFMA with an LDS-staged activation halo tile (gfx950).
//
// The implicit-GEMM kernel (mf_conv.hip) re-gathers every input pixel once per tap: 9x the
// activation bytes cross the L2 -> LDS path, and for the large-spatial / narrow-channel layers of
// the decoder (64-256 channels at 24^2..96^2) that path, not the MFMA pipe, sets the time.  Here a
// workgroup owns a PH x 16 patch of output pixels of one image and BN output channels; for each
// CK-channel slice of the input it DMAs the (PH+2) x 18 halo patch into LDS ONCE, then runs all
// nine taps out of it: the pixel fragment of tap (dy,dx) is the same LDS image read at a shifted
// row.  The halo image is double-buffered (the next slice's DMA flies under this slice's MFMAs).
// Weights never touch LDS: each wave owns 16 output channels and streams the (slice, tap)
// A-fragments straight from L2 into a 3-deep VGPR ring (coalesced 16-byte-per-lane reads, requested
// two taps ahead), so the only barrier is one per slice.  Global -> LDS bytes per output pixel drop ~6x vs implicit GEMM.
//
// LDS image: row = one pixel, CK bf16 channels (64 or 128 bytes), 16-byte slots XOR-swizzled with
// swz(row) so that a ds_read_b128 of 16 CONSECUTIVE rows starting at ANY row (tap shifts make the
// window unaligned) is bank-conflict free -- exhaustive search in DESIGN.md.
//
// Same operand roles and epilogue as mf_conv.hip: weights = MFMA A (rows = channels), pixels = B,
// one lane owns 4 consecutive channels of one pixel.
#include "mf_conv.h"
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#ifndef MF_HALO_RING9
#define MF_HALO_RING9 0
#endif

namespace {

constexpr int PW = 16;   // patch width = one MFMA pixel fragment

// XOR term of the 16-byte slot index, a function of the halo COLUMN hx only, so that a fragment
// address is (lane-constant per dx) + (compile-time row offset).  Conflict-free for every tap shift.
template <int CK>
__device__ __forceinline__ int hswz(int hx) {
    return CK == 32 ? (((hx >> 2) & 1) << 1) : (((hx >> 1) & 3) << 1);
}

__device__ __forceinline__ float hbf2f(uint32_t h16) { return __uint_as_float(h16 << 16); }
__device__ __forceinline__ uint32_t hf2bf(float f) {
    // round to nearest even in hardware: gfx950's v_cvt_pk_bf16_f32 (the compiler pairs neighbouring calls), a quarter of the integer form's instructions
    return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)f);
}

__device__ __forceinline__ void hglds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

}  // namespace

template <int PH, int BN, int WGM, int WGN, bool X3, int NST>
__global__ __launch_bounds__(WGM * WGN * 64, (WGM * WGN == 4 && (PH / WGM) * (BN / WGN / 16) <= 8) ? 2 : 1) void k_conv3x3_halo(const HaloArgs a) {
    constexpr int NW = WGM * WGN;                           // waves per workgroup
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    constexpr int CK = X3 ? 32 : 64;
    constexpr int KG = CK / 8, ROWB = CK * 2, RPC = 1024 / ROWB;
    constexpr int NP = X3 ? 2 : 1;
    constexpr int KK = CK / 32;                             // MFMA k-steps per slice
    constexpr int HW = PW + 2, HROWS = (PH + 2) * HW;
    constexpr int HCH = (HROWS + RPC - 1) / RPC;           // 1-KiB DMA chunks of the halo image
    constexpr int H_BYTES = HCH * 1024;
    constexpr int NHC = (HCH + NW - 1) / NW;
    constexpr int FM = PH / WGM;                            // patch rows (= pixel fragments) per wave
    constexpr int FN = BN / WGN / 16;                       // 16-channel fragment rows per wave
    static_assert(FN >= 1 && FM >= 1, "wave tile must hold a fragment");
    constexpr int STAGE = NP * H_BYTES;                     // one halo image (hi, lo); two stages

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware order, n tile fastest
    const int nt = a.n_patches * a.tiles_n;
    const int bid = blockIdx.x;
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int patch = t / a.tiles_n, tn = t - patch * a.tiles_n;
    const int b = patch / a.patches_per_img;
    const int pr = patch - b * a.patches_per_img;
    const int py = pr / a.patches_x, px = pr - py * a.patches_x;
    const int y0 = py * PH, x0 = px * PW;      // patch origin (unpadded output == input coordinates)
    const int n0 = tn * BN;

    // ---- halo DMA sources ------------------------------------------------------------------------
    // halo row hr -> input pixel (y0 - 1 + hy, x0 - 1 + hx); with the input buffer's zero ring that is
    // padded coordinate (y0 + hy + halo - 1, ...).  Rows/cols past the buffer are clamped: they only
    // feed output pixels that are masked below.
    const bf16_t* hp[NHC];
    const int64_t x_delta = X3 ? (a.x_lo - a.x_hi) : 0;
#pragma unroll
    for (int i = 0; i < NHC; ++i) {
        int hr = (wave + NW * i) * RPC + lane / KG;
        const int kg = (lane % KG) ^ hswz<CK>(hr % HW);
        hr = hr < HROWS ? hr : HROWS - 1;
        const int hy = hr / HW, hx = hr - hy * HW;
        int iy = y0 + hy + a.in_halo - 1, ix = x0 + hx + a.in_halo - 1;
        iy = iy < a.in_hp ? iy : a.in_hp - 1;
        ix = ix < a.in_wp ? ix : a.in_wp - 1;
        hp[i] = a.x_hi + ((int64_t)b * a.xb + ((int64_t)iy * a.in_wp + ix) * a.x_ld + kg * 8);
    }
    auto load_halo = [&](int slice, int stage) __attribute__((always_inline)) {
        char* base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < NHC; ++i) {
            const int c = wave + NW * i;
            if (HCH % NW == 0 || c < HCH) {
                const bf16_t* src = hp[i] + slice * CK;
                hglds16(src, base + c * 1024);
                if (X3) hglds16(src + x_delta, base + H_BYTES + c * 1024);
            }
        }
    };

    // ---- weights: register-stationary per slice ---------------------------------------------------
    // wave (wave_m, wave_n) owns channels n0 + wave_n*16 .. +15 for FM patch rows; the MFMA A fragment
    // of (tap, k-step) is 16 rows x 32 k = one coalesced 16-byte load per lane straight from L2.
    const int wave_m = wave % WGM, wave_n = wave / WGM;
    const int row0 = wave_m * FM;
    const int cn0 = wave_n * (FN * 16);
    const int fr = lane & 15, fk = lane >> 4;
    const bf16_t* wsrc[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        int nrow = n0 + cn0 + i * 16 + fr;
        nrow = nrow < a.Npad ? nrow : a.Npad - 1;
        wsrc[i] = a.w_hi + ((int64_t)nrow * CK + fk * 8);
    }
    const int64_t w_delta = X3 ? (a.w_lo - a.w_hi) : 0;
    const int64_t w_tap = (int64_t)a.Npad * CK;            // one (slice, tap) tile

    // Rolling 3-deep ring over the flattened (slice, tap) sequence: the fragment for step s+2 is
    // requested while step s computes, so only 3 taps' weights are ever live (24 VGPRs in bf16x3).
    constexpr int RING = (FN <= 2 && MF_HALO_RING9) ? 9 : (FN == 1 ? 9 : 3);   // 9: in-place refill one whole slice ahead
    constexpr int DIST = RING - 1;
    bf16x8 wr[RING][FN][KK][NP];
    auto load_step = [&](int ring, int step) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const bf16_t* src = wsrc[i] + (int64_t)step * w_tap;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                wr[ring][i][kk][0] = *reinterpret_cast<const bf16x8*>(src + kk * 32);
                if (X3) wr[ring][i][kk][NP - 1] = *reinterpret_cast<const bf16x8*>(src + kk * 32 + w_delta);
            }
        }
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int n_steps = a.n_slices * 9;
    // lane part of a pixel-fragment address, one per horizontal tap shift and k-step
    int lane_off[3][KK];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            lane_off[dx][kk] = (row0 * HW + fr + dx) * ROWB + (((kk * 4 + fk) ^ hswz<CK>(fr + dx)) << 4);

    auto compute = [&](int stage, int slice) __attribute__((always_inline)) {
        const char* base = smem + stage * STAGE;
        constexpr int NF = 9 * KK * FM;     // fragment reads of one slice, software-pipelined one ahead
        auto rd = [&](int idx, bf16x8& hi, bf16x8& lo) __attribute__((always_inline)) {
            const int tap = idx / (KK * FM), kk = (idx / FM) % KK, j = idx % FM;
            const int dy = tap / 3, dx = tap % 3;
            const char* p = base + lane_off[dx][kk] + (j + dy) * HW * ROWB;
            hi = *reinterpret_cast<const bf16x8*>(p);
            if (X3) lo = *reinterpret_cast<const bf16x8*>(p + H_BYTES);
        };
        if (a.res_from_halo) {
            // conv.py:17-18 `out += x`: x is this layer's input, whose centre pixels sit in the halo image
            // of the slice that carries the lane's channels -- no second trip to HBM for the residual
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int cb = n0 + cn0 + i * 16;                  // wave-uniform
                if (cb / CK == slice) {
                    const int kg = (cb % CK) / 8 + (fk >> 1);
                    const int hx = fr + 1;
                    const int lo8 = ((kg ^ hswz<CK>(hx)) << 4) + (fk & 1) * 8;
#pragma unroll
                    for (int j = 0; j < FM; ++j) {
                        const char* p = base + ((row0 + j + 1) * HW + hx) * ROWB + lo8;
                        uint2 rh = *reinterpret_cast<const uint2*>(p);
                        acc[i][j][0] += hbf2f(rh.x & 0xffffu); acc[i][j][1] += hbf2f(rh.x >> 16);
                        acc[i][j][2] += hbf2f(rh.y & 0xffffu); acc[i][j][3] += hbf2f(rh.y >> 16);
                        if (X3) {
                            rh = *reinterpret_cast<const uint2*>(p + H_BYTES);
                            acc[i][j][0] += hbf2f(rh.x & 0xffffu); acc[i][j][1] += hbf2f(rh.x >> 16);
                            acc[i][j][2] += hbf2f(rh.y & 0xffffu); acc[i][j][3] += hbf2f(rh.y >> 16);
                        }
                    }
                }
            }
        }
        bf16x8 c_hi, c_lo, n_hi, n_lo;
        rd(0, c_hi, c_lo);
#pragma unroll
        for (int idx = 0; idx < NF; ++idx) {
            const int tap = idx / (KK * FM), kk = (idx / FM) % KK, j = idx % FM;
            if (idx % (KK * FM) == 0) {
                // unconditional (clamped at the tail): a branch here would split the schedule per tap
                const int step2 = slice * 9 + tap + DIST;
                load_step((tap + DIST) % RING, step2 < n_steps ? step2 : n_steps - 1);
            }
            if (idx + 1 < NF) rd(idx + 1, n_hi, n_lo);
            // pin the prefetch ABOVE this fragment's MFMAs (hipcc otherwise sinks it to its first use
            // and every MFMA group eats a full LDS round trip)
            __builtin_amdgcn_sched_barrier(0);
            // product-type outer, channel fragment inner: consecutive MFMAs hit different accumulators
            if (X3) {
#pragma unroll
                for (int i = 0; i < FN; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[tap % RING][i][kk][NP - 1], c_hi, acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < FN; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[tap % RING][i][kk][0], c_lo, acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < FN; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[tap % RING][i][kk][0], c_hi, acc[i][j], 0, 0, 0);
            c_hi = n_hi; c_lo = n_lo;
        }
    };

    load_halo(0, 0);
#pragma unroll
    for (int r0 = 0; r0 < DIST; ++r0) load_step(r0, r0 < n_steps ? r0 : n_steps - 1);
    __syncthreads();                           // drains the DMA (vmcnt) and publishes halo stage 0
    for (int slice = 0; slice < a.n_slices; ++slice) {
        const bool more = slice + 1 < a.n_slices;
        if (NST == 2) {
            const int st = slice & 1;
            if (more) load_halo(slice + 1, st ^ 1);   // flies under this slice's MFMAs
            compute(st, slice);
            if (more) __syncthreads();         // next halo landed; everyone is done with this one
        } else {
            // one halo image: half the LDS, twice the resident workgroups; the reload is exposed to
            // this workgroup and hidden by its neighbours on the CU
            compute(0, slice);
            if (more) {
                __syncthreads();
                load_halo(slice + 1, 0);
                __syncthreads();
            }
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------
    // Loads ahead of the stores they do not depend on (see mf_conv_halo2.hip's epilogue): bias quads once, the residuals of a row group
    // in one burst, then that group's stores -- a load issued behind a store waits for the store's acknowledgement too (one in-order vmcnt).
    float4 bq[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        int c = n0 + cn0 + i * 16 + fk * 4;
        c = c < a.Npad - 3 ? c : a.Npad - 4;
        bq[i] = *reinterpret_cast<const float4*>(a.bias + c);
    }
    const bool has_res = a.r_hi != nullptr;
    const int ox = x0 + fr;
    constexpr int JG = (FM * FN * NP <= 32) ? FM : (32 / (FN * NP) >= 1 ? 32 / (FN * NP) : 1);   // rows per residual burst
#pragma unroll
    for (int j0 = 0; j0 < FM; j0 += JG) {
        uint2 rh[JG][FN], rl[JG][FN];
        if (has_res) {
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) {
                const int j = j0 + jj;
                if (j >= FM) break;
                int oy = y0 + row0 + j, oxc = ox;
                oy = oy < a.H ? oy : a.H - 1; oxc = oxc < a.W ? oxc : a.W - 1;      // clamped, never branched around: the stores are masked
                const int64_t ro = (int64_t)b * a.rb + (int64_t)oy * a.ri + (int64_t)oxc * a.rj;
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    int c = n0 + cn0 + i * 16 + fk * 4;
                    c = c < a.N ? c : 0;
                    rh[jj][i] = *reinterpret_cast<const uint2*>(a.r_hi + ro + c);
                    if (X3) rl[jj][i] = *reinterpret_cast<const uint2*>(a.r_lo + ro + c);
                }
            }
        }
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) {
            const int j = j0 + jj;
            if (j >= FM) break;
            const int oy = y0 + row0 + j;
            const bool row_ok = oy < a.H && ox < a.W;
            const int64_t yo = (int64_t)b * a.yb + (int64_t)oy * a.yi + (int64_t)ox * a.yj;
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int c = n0 + cn0 + i * 16 + fk * 4;
                float v[4] = {acc[i][j][0] + bq[i].x, acc[i][j][1] + bq[i].y, acc[i][j][2] + bq[i].z, acc[i][j][3] + bq[i].w};
                if (has_res) {
                    v[0] += hbf2f(rh[jj][i].x & 0xffffu); v[1] += hbf2f(rh[jj][i].x >> 16);
                    v[2] += hbf2f(rh[jj][i].y & 0xffffu); v[3] += hbf2f(rh[jj][i].y >> 16);
                    if (X3) {
                        v[0] += hbf2f(rl[jj][i].x & 0xffffu); v[1] += hbf2f(rl[jj][i].x >> 16);
                        v[2] += hbf2f(rl[jj][i].y & 0xffffu); v[3] += hbf2f(rl[jj][i].y >> 16);
                    }
                }
                if (a.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (a.act == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
                }
                if (!row_ok || c >= a.N) continue;
                uint32_t h[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = hf2bf(v[e]);
                *reinterpret_cast<uint2*>(a.y_hi + yo + c) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                if (X3) {
                    uint32_t l[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) l[e] = hf2bf(v[e] - hbf2f(h[e]));
                    *reinterpret_cast<uint2*>(a.y_lo + yo + c) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
namespace {

template <int PH, int BN, int WGM, int WGN, bool X3, int NST>
int halo_launch_cfg(const HaloArgs& a, hipStream_t s) {
    static bool attr_done = false;
    auto kern = k_conv3x3_halo<PH, BN, WGM, WGN, X3, NST>;
    if (!attr_done) {
        MF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    constexpr int CK = X3 ? 32 : 64, RPC = 1024 / (CK * 2), NP = X3 ? 2 : 1;
    constexpr int HCH = ((PH + 2) * (PW + 2) + RPC - 1) / RPC;
    const size_t lds = (size_t)NST * NP * HCH * 1024;
    hipLaunchKernelGGL(kern, dim3(a.n_patches * a.tiles_n), dim3(WGM * WGN * 64), lds, s, a);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

template <int PH, int BN, int WGM, int WGN>
int halo_launch_prec(const HaloArgs& a, bool x3, hipStream_t s) {
    return x3 ? halo_launch_cfg<PH, BN, WGM, WGN, true, 2>(a, s) : halo_launch_cfg<PH, BN, WGM, WGN, false, 2>(a, s);
}

}  // namespace

// Patch selection: 8x16 pixels x 64 (or 32) channels; maps too small to give every CU a workgroup fall
// back to 4-row patches and then to 32-channel tiles (4x the workgroups, 1/4 of the MFMAs each).
HaloTile mf_halo_pick_tile(int H, int W, int N, int batch, int cin) {
    auto wgs = [&](int ph, int bn) { return batch * ((H + ph - 1) / ph) * ((W + PW - 1) / PW) * ((N + bn - 1) / bn); };
    if (N <= 32) return wgs(8, 32) >= 256 ? HaloTile{8, 32, 2, 2} : HaloTile{4, 32, 2, 2};
    // (16 x 16 patches and fat 2 x 2-wave tiles measured 4 % slower / no faster inside the networks than the 8-row patch: not instantiated)
    (void)cin;
    if (wgs(8, 64) >= 256) return HaloTile{8, 64, 2, 2};
    if (wgs(4, 64) >= 256) return HaloTile{4, 64, 2, 2};
    return HaloTile{4, 32, 2, 2};
}

int mf_halo_launch(const HaloArgs& a0, const HaloTile& t, bool x3, hipStream_t s) {
    HaloArgs a = a0;
    a.patches_x = (a.W + PW - 1) / PW;
    const int patches_y = (a.H + t.ph - 1) / t.ph;
    a.patches_per_img = a.patches_x * patches_y;
    a.n_patches = a.batch * a.patches_per_img;
    a.tiles_n = (a.N + t.bn - 1) / t.bn;
#define MF_HCASE(PH, BN, WGM, WGN) \
    if (t.ph == PH && t.bn == BN && t.wgm == WGM) return halo_launch_prec<PH, BN, WGM, WGN>(a, x3, s);
    MF_HCASE(8, 64, 2, 2)
    MF_HCASE(4, 64, 2, 2)
    MF_HCASE(4, 32, 2, 2)
    MF_HCASE(8, 32, 2, 2)
#undef MF_HCASE
    mf_set_error("halo conv: no kernel for patch %dx16, BN %d", t.ph, t.bn);
    return MF_ERR_INVALID;
}

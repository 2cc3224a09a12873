// 3x3 stride-1 convolution on MFMA, second generation of the LDS halo-tile kernel (mf_conv_halo.hip): the WEIGHTS also go
// through LDS, shared by every wave of the workgroup.
//
// Why: in mf_conv_halo.hip each wave streams its own (slice, tap) A fragments straight from L2.  rocprofv3 on the VAE's
// 128 -> 128 layer at 256x256 (profiles/r01_pmc_halo_128x128_256.txt): TCC_REQ = 42.2 M x 128 B = 5.4 GB per launch, 86 % of it
// weights (147 KB per workgroup and slice against a 23 KB halo patch) -- the kernel sits on the ~10-12 TB/s L2 -> CU ceiling
// with the MFMA pipe 42 % busy.  Here a 16 x 16-pixel patch (8 waves) shares ONE copy of each tap's weight tile:
// L2 traffic per FLOP drops to about a third.
//
// Pipeline per channel slice (CK = 32 channels in bf16x3, 64 in bf16): the halo patch is double-buffered as before; the nine
// taps' weight tiles arrive in three rows of three taps through a 2-deep LDS ring (row r+1 lands while row r is multiplied);
// one barrier per tap row.  Both images are filled by LDS-DMA (lane-linear), so the bank-conflict swizzle is applied to the DMA
// source address and again on the ds_read side, exactly as in the other two conv kernels.
#include "mf_conv.h"
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

namespace {

constexpr int PW = 16;   // patch width = one MFMA pixel fragment

// XOR term of the 16-byte slot index, a function of the halo COLUMN hx only, so that a fragment
// address is (lane-constant per dx) + (compile-time row offset).  Conflict-free for every tap shift.
template <int CK>
__device__ __forceinline__ int hswz(int hx) {
    return CK == 32 ? (((hx >> 2) & 1) << 1) : (((hx >> 1) & 3) << 1);
}

// Slot permutation of the FP6 plane (f16 + FP6 format): a row holds two 32-byte blocks, and lane group fk reads block fk & 1 as two
// ds_read_b128.  With the f16 plane's XOR (0 / 2) every such read touches only the even or only the odd 16-byte slots of its rows -- half
// the banks, a 2-way conflict on half of the kernel's LDS traffic (SQ_LDS_BANK_CONFLICT = 34 % of SQ_LDS_IDX_ACTIVE).  ds_read_b128 serves
// lanes {0-3, 12-15, 20-27} etc. together: of the four rows r, r+4, r+8, r+12 that share a 256-byte bank window, two read block 0 and two
// block 1; XOR 3 on every other group of four rows sends them to four different slots, for every tap shift.
__device__ __forceinline__ int qswz(int r) { return ((r >> 2) & 1) * 3; }

__device__ __forceinline__ float hbf2f(uint32_t h16) { return __uint_as_float(h16 << 16); }
__device__ __forceinline__ uint32_t hf2bf(float f) {
    // round to nearest even in hardware: gfx950's v_cvt_pk_bf16_f32 (the compiler pairs neighbouring calls), a quarter of the integer form's instructions
    return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)f);
}

// LDS fragment read of the specialised workgroup's compute waves
template <typename T>
__device__ __forceinline__ T ldsr(const char* p) {
    return *reinterpret_cast<const T*>(p);
}

__device__ __forceinline__ void hglds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// 16-byte epilogue stores (as mf_conv.hip's store_pair16): lanes fk / fk ^ 1 own the two 4-channel halves of one 8-channel group in every
// fragment; for a fragment PAIR the even lane takes both halves of p0's group, the odd lane both of p1's -- one exchange, one dwordx4 store
// where two dwordx2 went.  c16 = first channel of p0's 16-channel block; p1's block is 16 channels on.
__device__ __forceinline__ void hstore_pair16(bf16_t* base, int64_t yo, int c16, int fk, uint2 p0, uint2 p1, int N, bool ok) {
    const bool odd = fk & 1;
    const uint2 send = odd ? p0 : p1;
    uint2 recv;
    recv.x = (uint32_t)__shfl_xor((int)send.x, 16);
    recv.y = (uint32_t)__shfl_xor((int)send.y, 16);
    const uint4 out = odd ? make_uint4(recv.x, recv.y, p1.x, p1.y) : make_uint4(p0.x, p0.y, recv.x, recv.y);
    const int cw = c16 + (odd ? 16 : 0) + (fk & ~1) * 4;
    if (ok && cw < N) *reinterpret_cast<uint4*>(base + yo + cw) = out;
}

}  // namespace

// TR = taps per ring slot: 3 (one kernel row, a barrier per row) or 1 (a barrier per tap: the wide tiles, whose 3-tap slot would not fit)
// PHASE = -1: the 3x3 convolution (9 taps).  PHASE = 2 * py + px in 0..3: one phase of `nearest 2x upsample -> 3x3 conv` (mf_conv.hip): a 2 x 2
// convolution on the INPUT grid whose taps sit at halo rows py..py+1, columns px..px+1, writing output pixels (2i + py, 2j + px) -- the
// launcher passes doubled output strides and the phase's weight block.
// HS = halo images in LDS: 2 (the next slice's image lands under this slice's MFMAs) or 1 (half the LDS: two 4-wave workgroups per CU, each
// other's DMA waits and epilogues hidden by the neighbour's MFMAs)
// Q: operands in the f16 + FP6-residual format (MF_PREC_F16Q).  Plane 0 rows are 32 f16 channels, plane 1 rows two 32-byte FP6 blocks
// ([q6(wh) | q6(wl)] for weights, [q6(xl) | q6(xh)] for pixels: 24 B of e2m3 codes + the block's E8M0 byte).  TWO taps share one correction
// instruction (compute_pair): per tap pair and accumulator tile two v_mfma_f32_16x16x32_f16 (wh.xh) + ONE v_mfma_scale_f32_16x16x128_f8f6f4 whose K
// blocks 0 / 1 carry tap A's q6(wh).xl / wl.q6(xh) and blocks 2 / 3 tap B's -- 48 matrix cycles where bf16x3 spends 96, with the ring, DMA and LDS
// swizzle of the bf16x3 kernel unchanged.  The WEIGHT DMA is issued by waves NW/2 .. NW-1 only: waves w and w + NW/2 share a SIMD, and with all
// pieces on the second one the first starts its reads and MFMAs at once (s_memtime: 13.0 k -> 12.3 k cycles per slice).
// Where this kernel's time goes, by ablation (matrix floor 7.7 k cycles per slice, LDS floor 6.3 k, overlapped by half; DMA 10 %), and the
// restructurings that did NOT move it (two workgroups per CU, skewed halves, software pipelining): profiles/r03_halo_q_loop_study.md.
// SP: the workgroup is SPECIALISED -- WGM x WGN = 4 compute waves (one per SIMD: 128 px x 64 ch each, the matrix pipe to itself, 25 % fewer LDS
// fragment reads per MFMA than eight 64 px x 64 ch waves) + 4 producer waves that issue ALL of the LDS-DMA (halo image and weight ring) and
// otherwise only meet the barriers; every barrier is executed by all eight.
template <int PH, int BN, int WGM, int WGN, bool X3, int TR, int PHASE = -1, int HS = 2, bool Q = false, bool SP = false>
__global__ __launch_bounds__((WGM * WGN + (SP ? 4 : 0)) * 64, HS == 1 ? 2 : 1) void k_conv3x3_halo_w(const HaloArgs a) {
    constexpr int NW = WGM * WGN;                           // compute waves per workgroup
    constexpr int NPW = SP ? 4 : 0;                         // producer waves (SP)
    static_assert(NW == 4 || NW == 8, "4 or 8 compute waves per workgroup");
    static_assert(!SP || (Q && NW == 4 && HS == 2), "specialised workgroup: f16 + FP6 format, four compute waves");
    constexpr int CK = X3 ? 32 : 64;
    constexpr int KG = CK / 8, ROWB = CK * 2, RPC = 1024 / ROWB;
    constexpr int NP = X3 ? 2 : 1;
    constexpr int KK = CK / 32;                             // MFMA k-steps per slice
    constexpr int HW = PW + 2, HROWS = (PH + 2) * HW;
    constexpr int HCH = (HROWS + RPC - 1) / RPC;           // 1-KiB DMA chunks of the halo image
    constexpr int H_BYTES = HCH * 1024;
    constexpr int NHW = SP ? NPW : NW;                      // waves that issue halo DMA (non-SP: all of them -- moving it to the lower half as well left the loop unchanged and cost 1 k cycles of prologue)
    constexpr int NHC = (HCH + NHW - 1) / NHW;
    constexpr int FM = PH / WGM;                            // patch rows (= pixel fragments) per wave
    constexpr int FN = BN / WGN / 16;                       // 16-channel fragment rows per wave
    static_assert(FN >= 1 && FM >= 1, "wave tile must hold a fragment");
    constexpr int STAGE = NP * H_BYTES;                     // one halo image (hi, lo); two stages
    constexpr int WT_BYTES = BN * ROWB;                     // one tap's weight tile, one plane: [BN][CK] bf16
    constexpr int WCH = WT_BYTES / 1024;                    // its 1-KiB DMA chunks
    static_assert(TR == 1 || TR == 3, "ring slot = one tap or one kernel row");
    constexpr int NT = PHASE < 0 ? 9 : 4;                   // taps per channel slice
    static_assert(PHASE < 0 || TR == 1, "upsample phases use the one-tap ring");
    constexpr int WROW = TR * NP * WT_BYTES;                // one ring slot (TR taps, planes)
    constexpr int WRC = TR * NP * WCH;                      // DMA chunks per slot
    constexpr bool PROD = Q && !SP;                         // f16 + FP6 tiles: weight DMA by the upper half of the waves only
    constexpr int NWD = SP ? NPW : (PROD ? NW / 2 : NW);    // waves that issue weight DMA
    constexpr int NWR = (WRC + NWD - 1) / NWD;
    static_assert(WT_BYTES % 1024 == 0, "weight tile must be whole DMA chunks");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(HS == 1 || HS == 2, "halo stages");
    char* const wring = smem + HS * STAGE;                  // 2 x WROW

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long* dbg = a.dbg ? a.dbg + 4 * ((size_t)blockIdx.x + gridDim.x * (size_t)blockIdx.y) : nullptr;
    if (dbg && threadIdx.x == 0) dbg[0] = __builtin_amdgcn_s_memtime();

    // XCD-aware order, n tile fastest
    const int nt = a.n_patches * a.tiles_n;
    const int bid = blockIdx.x;
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
#ifndef MF_HALO_ORDER
#define MF_HALO_ORDER 0       // (A/B builds, profiles/r05_halo_refetch_ab.md) 1: patch fastest -- an XCD walks every patch of one channel tile; 2: plain launch order
#endif
#if MF_HALO_ORDER == 1
    const int tn = t / a.n_patches, patch = t - tn * a.n_patches;
#elif MF_HALO_ORDER == 2
    const int patch = bid / a.tiles_n, tn = bid - patch * a.tiles_n;
#else
    const int patch = t / a.tiles_n, tn = t - patch * a.tiles_n;
#endif
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
    int hq[Q ? NHC : 1];                       // f16 + FP6 format: element offset of the FP6 plane's source slot relative to the f16 plane's (qswz)
    const int64_t x_delta = X3 ? (a.x_lo - a.x_hi) : 0;
    const int hwave = SP ? wave - NW : wave;               // index among the halo-DMA waves (negative: a compute wave of a specialised workgroup)
    const bool cons = !SP || wave < NW;                    // this wave multiplies
#pragma unroll
    for (int i = 0; i < NHC; ++i) {
        int hr = ((hwave < 0 ? 0 : hwave) + NHW * i) * RPC + lane / KG;
        const int kg = (lane % KG) ^ hswz<CK>(hr % HW);
        if (Q) hq[i] = (((lane % KG) ^ qswz(hr % HW)) - kg) * 8;
        hr = hr < HROWS ? hr : HROWS - 1;
        const int hy = hr / HW, hx = hr - hy * HW;
        int iy = y0 + hy + a.in_halo - 1, ix = x0 + hx + a.in_halo - 1;
        iy = iy < a.in_hp ? iy : a.in_hp - 1;
        ix = ix < a.in_wp ? ix : a.in_wp - 1;
        iy = iy > 0 ? iy : 0;                   // (a GroupNorm input may have no zero ring: the pixel is masked by coordinate after the transform)
        ix = ix > 0 ? ix : 0;
        hp[i] = a.x_hi + ((int64_t)b * a.xb + ((int64_t)iy * a.in_wp + ix) * a.x_ld + kg * 8);
    }
    auto load_halo = [&](int slice, int stage) __attribute__((always_inline)) {
        if (SP && hwave < 0) return;
        char* base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < NHC; ++i) {
            const int c = hwave + NHW * i;
            if (HCH % NHW == 0 || c < HCH) {
                const bf16_t* src = hp[i] + slice * CK;
                hglds16(src, base + c * 1024);
                if (X3) hglds16(src + x_delta + (Q ? hq[i] : 0), base + H_BYTES + c * 1024);
            }
        }
    };

    // ---- weights: one tile [BN][CK] per (slice, tap, plane), DMA'd by tap rows into the LDS ring ------------------------
    const int cwave = SP ? wave % NW : wave;               // (a producer wave of a specialised workgroup shares the epilogue of compute wave `wave - NW`)
    const int wave_m = cwave % WGM, wave_n = cwave / WGM;
    const int row0 = wave_m * FM;
    const int cn0 = wave_n * (FN * 16);
    const int fr = lane & 15, fk = lane >> 4;
    const int64_t w_delta = X3 ? (a.w_lo - a.w_hi) : 0;
    const int64_t w_tap = (int64_t)a.Npad * CK;            // one (slice, tap) tile in the packed weights
    // chunk c of a tap row = (tap t = c / (NP * WCH), plane, 1-KiB piece q): lane l lands in LDS row q*RPC + l/KG, slot l%KG
    const bf16_t* wsrc[NWR];
    int wtap[NWR];
    const int dwave = SP ? wave - NW : (PROD ? wave - NW / 2 : wave);   // index among the DMA-issuing waves (negative: none of this wave's business)
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
        const int c = (dwave < 0 ? 0 : dwave) + NWD * i;
        const int t3 = c / (NP * WCH), rem = c - t3 * (NP * WCH), pl = rem / WCH, q = rem - pl * WCH;
        const int r = q * RPC + lane / KG;
        const int kg = (lane % KG) ^ ((Q && pl) ? qswz(r) : hswz<CK>(r));
        int nrow = n0 + r;
        nrow = nrow < a.Npad ? nrow : a.Npad - 1;
        wsrc[i] = a.w_hi + (pl ? w_delta : 0) + ((int64_t)nrow * CK + kg * 8);
        wtap[i] = t3;
    }
    // tap row `trow` (0..2) of slice `slice` into ring buffer `buf`
    auto load_wrow = [&](int slice, int trow, int buf) __attribute__((always_inline)) {
        if ((PROD || SP) && dwave < 0) return;
        char* base = wring + buf * WROW;
#pragma unroll
        for (int i = 0; i < NWR; ++i) {
            const int c = dwave + NWD * i;
            if (WRC % NWD == 0 || c < WRC) hglds16(wsrc[i] + (int64_t)(slice * NT + trow * TR + wtap[i]) * w_tap, base + c * 1024);
        }
    };
    // A fragment of (tap-in-row t3, channel block i, k-step kk, plane): rows cn0 + i*16 + fr, 16-byte slot kk*4 + fk
    int wlane[FN][KK];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int r = cn0 + i * 16 + fr;
            wlane[i][kk] = r * ROWB + (((kk * 4 + fk) ^ hswz<CK>(r)) << 4);
        }

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // lane part of a pixel-fragment address, one per horizontal tap shift and k-step
    int lane_off[3][KK];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            lane_off[dx][kk] = (row0 * HW + fr + dx) * ROWB + (((kk * 4 + fk) ^ hswz<CK>(fr + dx)) << 4);

    // residual taken from the halo image (conv.py:17-18 `out += x`), once per slice that carries the lane's channels
    auto add_residual = [&](int stage, int slice) __attribute__((always_inline)) {
        const char* base = smem + stage * STAGE;
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
    };
    // the TR taps of ring slot `step` (taps step*TR ...): pixel fragments from the halo image at the tap's shift, weights from ring buffer `buf`
    constexpr int NSTEP = NT / TR;
    // the next ring slot (of this slice, or slot 0 of the next) into the other ring buffer: everyone left it at the last barrier
    auto issue_next = [&](int slice, int step, bool more, int buf) __attribute__((always_inline)) {
        if (step < NSTEP - 1) load_wrow(slice, step + 1, buf ^ 1);
        else if (more) load_wrow(slice + 1, 0, buf ^ 1);
    };
    auto compute_row = [&](int stage, int step, int buf, int slice, bool more) __attribute__((always_inline)) {
        const char* base = smem + stage * STAGE;
        const char* wb = wring + buf * WROW;
#pragma unroll
        for (int t3 = 0; t3 < TR; ++t3) {
            const int tap = step * TR + t3;
            const int dy = PHASE < 0 ? tap / 3 : (PHASE >> 1) + (tap >> 1), dx = PHASE < 0 ? tap % 3 : (PHASE & 1) + (tap & 1);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                bf16x8 whi[FN], wlo[FN];
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    whi[i] = *reinterpret_cast<const bf16x8*>(wb + (t3 * NP) * WT_BYTES + wlane[i][kk]);
                    if (X3) wlo[i] = *reinterpret_cast<const bf16x8*>(wb + (t3 * NP + 1) * WT_BYTES + wlane[i][kk]);
                }
                // pixel fragments one ahead of the MFMAs that use them (pinned above them: hipcc otherwise sinks the read to its first use) --
                // only where registers allow: the 128 x 64 wave tile already sits at 256 VGPRs and would spill
                constexpr bool PF = FN * FM <= 16;
                bf16x8 n_hi, n_lo, c_hi, c_lo;
                if (PF) {
                    const char* p0 = base + lane_off[dx][kk] + dy * HW * ROWB;
                    c_hi = *reinterpret_cast<const bf16x8*>(p0);
                    if (X3) c_lo = *reinterpret_cast<const bf16x8*>(p0 + H_BYTES);
                }
#pragma unroll
                for (int j = 0; j < FM; ++j) {
                    if (PF) {
                        if (j + 1 < FM) {
                            const char* p = base + lane_off[dx][kk] + (j + 1 + dy) * HW * ROWB;
                            n_hi = *reinterpret_cast<const bf16x8*>(p);
                            if (X3) n_lo = *reinterpret_cast<const bf16x8*>(p + H_BYTES);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        const char* p = base + lane_off[dx][kk] + (j + dy) * HW * ROWB;
                        c_hi = *reinterpret_cast<const bf16x8*>(p);
                        if (X3) c_lo = *reinterpret_cast<const bf16x8*>(p + H_BYTES);
                    }
                    if (X3) {
#pragma unroll
                        for (int i = 0; i < FN; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[i], c_hi, acc[i][j], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < FN; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[i], c_lo, acc[i][j], 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < FN; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[i], c_hi, acc[i][j], 0, 0, 0);
                    if (PF) { c_hi = n_hi; if (X3) c_lo = n_lo; }
                }
            }
        }
    };

    // f16 + FP6 format, TWO taps per correction instruction: K blocks 0 / 1 of the 16x16x128 MFMA carry tap A (q6(wh).xl, wl.q6(xh)), blocks 2 / 3 tap B, so
    // every lane group reads bytes that are used -- 32 operand bytes per fragment and tap, as in bf16x3 -- and a pair of taps costs two f16 MFMAs + one
    // correction MFMA (48 matrix cycles where bf16x3 spends 96).  Lane group fk reads tap (fk >> 1), block (fk & 1); a lone tap (the ninth) leaves blocks
    // 2 / 3 switched off by a zero scale.  tA / tB index the 3x3 taps, sA / sB the ring slots that hold their weights.
    auto compute_pair = [&](int stage, int tA, int tB, int sA, int sB) __attribute__((always_inline)) {
        const char* base = smem + stage * STAGE;
        const char* wA = wring + sA * WROW;
        const char* wB = wring + sB * WROW;
        const bool second = fk >= 2, pairB = tB >= 0;
        const int blk = fk & 1;
        // 3x3: tap t sits at halo row t / 3, column t % 3; upsample phase (py, px): its four taps at rows py .. py + 1, columns px .. px + 1
        const int dyA = PHASE < 0 ? tA / 3 : (PHASE >> 1) + (tA >> 1), dxA = PHASE < 0 ? tA % 3 : (PHASE & 1) + (tA & 1);
        const int dyB = !pairB ? dyA : (PHASE < 0 ? tB / 3 : (PHASE >> 1) + (tB >> 1)), dxB = !pairB ? dxA : (PHASE < 0 ? tB % 3 : (PHASE & 1) + (tB & 1));
        const char* wq = (second && pairB ? wB : wA) + WT_BYTES;
        f16x8 whA[FN], whB[FN];
        i32x8 w6[FN];
        int wsc[FN];
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int r = cn0 + i * 16 + fr;
            whA[i] = *reinterpret_cast<const f16x8*>(wA + wlane[i][0]);
            if (pairB) whB[i] = *reinterpret_cast<const f16x8*>(wB + wlane[i][0]);
            const char* q = wq + r * ROWB;
            const i32x4 q0 = *reinterpret_cast<const i32x4*>(q + (((2 * blk) ^ qswz(r)) << 4)), q1 = *reinterpret_cast<const i32x4*>(q + (((2 * blk + 1) ^ qswz(r)) << 4));
            w6[i] = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, -1, -1);
            wsc[i] = (second && !pairB) ? 0 : q1[2];
        }
        const int hxA = fr + dxA, hxB = fr + dxB;
        const int hxq = second ? hxB : hxA, dyq = second ? dyB : dyA;
        const int offA = (dyA * HW + hxA) * ROWB + ((fk ^ hswz<CK>(hxA)) << 4);
        const int offB = (dyB * HW + hxB) * ROWB + ((fk ^ hswz<CK>(hxB)) << 4);
        const int offq = (dyq * HW + hxq) * ROWB + H_BYTES + (((2 * blk) ^ qswz(hxq)) << 4);
        const int offq1 = (dyq * HW + hxq) * ROWB + H_BYTES + (((2 * blk + 1) ^ qswz(hxq)) << 4);
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const char* prow = base + (row0 + j) * HW * ROWB;
            const f16x8 xhA = *reinterpret_cast<const f16x8*>(prow + offA);
            f16x8 xhB;
            if (pairB) xhB = *reinterpret_cast<const f16x8*>(prow + offB);
            const i32x4 q0 = *reinterpret_cast<const i32x4*>(prow + offq), q1 = *reinterpret_cast<const i32x4*>(prow + offq1);
            const i32x8 x6 = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, -1, -1);
            const int xsc = (second && !pairB) ? 0 : q1[2];
#pragma unroll
            for (int i = 0; i < FN; ++i) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w6[i], x6, acc[i][j], 2, 2, 0, wsc[i], 0, xsc);
#pragma unroll
            for (int i = 0; i < FN; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whA[i], xhA, acc[i][j], 0, 0, 0);
            if (pairB) {
#pragma unroll
                for (int i = 0; i < FN; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whB[i], xhB, acc[i][j], 0, 0, 0);
            }
        }
    };

    // channel slices of this workgroup: all of them, or the blockIdx.y-th share when the layer is split for lack of patches
    const int s_begin = a.nsplit > 1 ? (int)((int64_t)a.n_slices * blockIdx.y / a.nsplit) : 0;
    const int s_end = a.nsplit > 1 ? (int)((int64_t)a.n_slices * (blockIdx.y + 1) / a.nsplit) : a.n_slices;
    if constexpr (SP) {
        // Two instruction streams with the same barrier sequence (the branch is wave-uniform): the DMA pointers live only in the producers' stream, the
        // accumulators and fragment addresses only in the compute waves' -- written as one loop, the producers' pointers were spilled around the MFMAs.
        constexpr int NPAIR = (NT + 1) / 2;
        if (!cons) {
            load_halo(s_begin, 0);
            load_wrow(s_begin, 0, 0);
            load_wrow(s_begin, 1, 1);
            __syncthreads();
            int wb = 0;
            for (int slice = s_begin; slice < s_end; ++slice) {
                const bool more = slice + 1 < s_end;
                const int nst = ((slice - s_begin) & 1) ^ 1;
                if (more) load_halo(slice + 1, nst);
#pragma unroll
                for (int p = 0; p < NPAIR; ++p) {
                    const int nb = 2 * (wb ^ 1);
                    if (p < NPAIR - 1) { load_wrow(slice, 2 * p + 2, nb); if (2 * p + 3 < NT) load_wrow(slice, 2 * p + 3, nb + 1); }
                    else if (more) { load_wrow(slice + 1, 0, nb); load_wrow(slice + 1, 1, nb + 1); }
                    if (p < NPAIR - 1 || more) __syncthreads();
                    wb ^= 1;
                }
            }
        } else {
            // The compute waves' addressing: the lane-dependent part of every fragment address in NINE registers (byte offsets from smem), the halo
            // stage, ring slot, patch row, tap shift and channel block as instruction immediates -- two slices are unrolled so that stage and ring
            // parity are compile-time.  (Left to itself hipcc keeps ~40 per-(pair, fragment) addresses live across the loop: 160 B of scratch.)
            constexpr int RING0 = HS * STAGE;
            int P16[3], PQx[3], Wh, Wq;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                P16[dx] = (row0 * HW + fr) * ROWB + ((fk ^ hswz<CK>(fr + dx)) << 4);
                PQx[dx] = (row0 * HW + fr) * ROWB + H_BYTES + (((2 * (fk & 1)) ^ qswz(fr + dx)) << 4);
                asm volatile("" : "+v"(P16[dx]));
                asm volatile("" : "+v"(PQx[dx]));
            }
            Wh = RING0 + (cn0 + fr) * ROWB + ((fk ^ hswz<CK>(cn0 + fr)) << 4);
            Wq = RING0 + WT_BYTES + (cn0 + fr) * ROWB + (((2 * (fk & 1)) ^ qswz(cn0 + fr)) << 4) + (fk >= 2 ? WROW : 0);   // lane groups 2, 3: tap B's slot (= tap A's + 1)
            asm volatile("" : "+v"(Wh));
            asm volatile("" : "+v"(Wq));
            const bool second = fk >= 2;
            // Tap B's f16 products of a pair's LAST DR patch rows are issued behind the pair's barrier, at the head of the next pair: they need only registers
            // (tap B's weights, the rows' fragments), so they fill the matrix pipe while the next pair's weight fragments are on their way from LDS.
            constexpr int DR = 3;
            f16x8 whB[FN], xb[4];
#pragma unroll
            for (int i = 0; i < FN; ++i) whB[i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) xb[i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            auto sp_deferred = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int j = FM - DR; j < FM; ++j)
#pragma unroll
                    for (int i = 0; i < FN; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whB[i], xb[j % 4], acc[i][j], 0, 0, 0);
            };
            auto sp_pair = [&](auto ST_, auto P_, auto WB_) __attribute__((always_inline)) {
                constexpr int st = decltype(ST_)::value, p = decltype(P_)::value, wb = decltype(WB_)::value;
                constexpr int tA = 2 * p, tB = 2 * p + 1 < NT ? 2 * p + 1 : -1;
                constexpr bool pairB = tB >= 0;
                constexpr int sA = 2 * wb, sB = 2 * wb + 1;
                constexpr int dyA = PHASE < 0 ? tA / 3 : (PHASE >> 1) + (tA >> 1), dxA = PHASE < 0 ? tA % 3 : (PHASE & 1) + (tA & 1);
                constexpr int dyB = !pairB ? dyA : (PHASE < 0 ? tB / 3 : (PHASE >> 1) + (tB >> 1)), dxB = !pairB ? dxA : (PHASE < 0 ? tB % 3 : (PHASE & 1) + (tB & 1));
                // The issue order is pinned (sched_barrier): a compute wave has its SIMD to itself, so nothing hides an operand that is requested late.  Pass 1:
                // per patch row the correction instruction (both taps) + tap A's f16 product, the next row's fragments requested before the row's MFMAs; pass 2:
                // tap B's f16 product, fragments two rows ahead.  (All three operand sets at once do not fit beside 128 accumulators.)
#define MF_SB() __builtin_amdgcn_sched_barrier(0)
                f16x8 whA[FN];
                i32x8 w6[FN];
                int wsc[FN];
                const int wq1 = Wq ^ 16;
                constexpr bool defer_in = NT == 9 ? p >= 1 : true;   // the pair before this one (in issue order) left DR rows of its pass 2 behind (3x3: the ninth tap is alone and has none; upsample phases: every pair, the first finds zeros)
                int pq0 = second ? PQx[dxB] + (dyB * HW + dxB) * ROWB : PQx[dxA] + (dyA * HW + dxA) * ROWB;
                asm volatile("" : "+v"(pq0));            // (computed here, per pair: two VALU operations instead of a register held across the loop)
                const int pq1 = pq0 ^ 16;
                constexpr int IA = st * STAGE + (dyA * HW + dxA) * ROWB, IB = st * STAGE + (dyB * HW + dxB) * ROWB, IQ = st * STAGE;
                i32x4 wq0r[FN], wq1r[FN];
                f16x8 xa[2];
                i32x4 xq0[2], xq1[2];
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    wq0r[i] = ldsr<i32x4>(smem + Wq + (sA * WROW + i * 16 * ROWB));
                    wq1r[i] = ldsr<i32x4>(smem + wq1 + (sA * WROW + i * 16 * ROWB));
                }
                auto ld_q = [&](int j) __attribute__((always_inline)) {
                    xq0[j & 1] = ldsr<i32x4>(smem + pq0 + (IQ + j * HW * ROWB));
                    xq1[j & 1] = ldsr<i32x4>(smem + pq1 + (IQ + j * HW * ROWB));
                };
                auto ld_a = [&](int j) __attribute__((always_inline)) { xa[j & 1] = ldsr<f16x8>(smem + P16[dxA] + (IA + j * HW * ROWB)); };
                auto ld_b = [&](int j) __attribute__((always_inline)) { xb[j % 4] = ldsr<f16x8>(smem + P16[dxB] + (IB + j * HW * ROWB)); };
                ld_q(0);
                ld_a(0);
#pragma unroll
                for (int i = 0; i < FN; ++i) whA[i] = ldsr<f16x8>(smem + Wh + (sA * WROW + i * 16 * ROWB));
                ld_q(1);
                ld_a(1);
                MF_SB();
                if (defer_in) { sp_deferred(); MF_SB(); }
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    w6[i] = __builtin_shufflevector(wq0r[i], wq1r[i], 0, 1, 2, 3, 4, 5, -1, -1);
                    wsc[i] = (second && !pairB) ? 0 : wq1r[i][2];
                }
                // pass 1, fragments 1.5 rows ahead on two buffers: a row's FP6 fragment is requested as soon as the correction MFMAs of the row two above have
                // been issued, its f16 fragment behind that row's f16 MFMAs (a request one row = 128 matrix cycles ahead came back late: 10.8 k cycles per slice)
#pragma unroll
                for (int j = 0; j < FM; ++j) {
                    const int c = j & 1;
                    const i32x8 x6 = __builtin_shufflevector(xq0[c], xq1[c], 0, 1, 2, 3, 4, 5, -1, -1);
                    const int xsc = (second && !pairB) ? 0 : xq1[c][2];
#pragma unroll
                    for (int i = 0; i < FN; ++i) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w6[i], x6, acc[i][j], 2, 2, 0, wsc[i], 0, xsc);
                    MF_SB();
                    if (j + 2 < FM) ld_q(j + 2);
                    else if (pairB) {
                        if (j == FM - 2) {
                            whB[0] = ldsr<f16x8>(smem + Wh + (sB * WROW + 0 * 16 * ROWB));
                            whB[1] = ldsr<f16x8>(smem + Wh + (sB * WROW + 1 * 16 * ROWB));
                        } else { ld_b(0); ld_b(1); }
                    }
                    MF_SB();
#pragma unroll
                    for (int i = 0; i < FN; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whA[i], xa[c], acc[i][j], 0, 0, 0);
                    MF_SB();
                    if (j + 2 < FM) ld_a(j + 2);
                    else if (pairB) {
                        if (j == FM - 2) {
                            whB[2] = ldsr<f16x8>(smem + Wh + (sB * WROW + 2 * 16 * ROWB));
                            whB[3] = ldsr<f16x8>(smem + Wh + (sB * WROW + 3 * 16 * ROWB));
                        } else ld_b(2);
                    }
                    MF_SB();
                }
                static_assert(FN == 4, "specialised workgroup: the pass-2 weight loads are written for four channel blocks");
                if (pairB) {
                    // pass 2: tap B's f16 product, fragments three rows ahead; the last DR rows are left to the head of the next pair (sp_deferred)
#pragma unroll
                    for (int j = 0; j < FM; ++j) {
                        if (j + 3 < FM) ld_b(j + 3);
                        MF_SB();
                        if (j < FM - DR) {
#pragma unroll
                            for (int i = 0; i < FN; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whB[i], xb[j % 4], acc[i][j], 0, 0, 0);
                        }
                        MF_SB();
                    }
                }
#undef MF_SB
            };
#define MF_IC(x) std::integral_constant<int, (x)>{}
            // one channel slice: NPAIR tap pairs from ring parity WB0 on, a barrier behind every pair but the very last of the workgroup
            auto sp_slice = [&](auto ST_, auto WB0_, bool more) __attribute__((always_inline)) {
                constexpr int st = decltype(ST_)::value, wb0 = decltype(WB0_)::value;
                sp_pair(MF_IC(st), MF_IC(0), MF_IC(wb0));
                __syncthreads();
                sp_pair(MF_IC(st), MF_IC(1), MF_IC(wb0 ^ 1));
                if (NPAIR > 2 || more) __syncthreads();
                if constexpr (NPAIR == 5) {
                    sp_pair(MF_IC(st), MF_IC(2), MF_IC(wb0));
                    __syncthreads();
                    sp_pair(MF_IC(st), MF_IC(3), MF_IC(wb0 ^ 1));
                    __syncthreads();
                    sp_pair(MF_IC(st), MF_IC(4), MF_IC(wb0));
                    if (more) __syncthreads();
                }
            };
            __syncthreads();
            if (dbg && threadIdx.x == 0) dbg[1] = __builtin_amdgcn_s_memtime();
            int slice = s_begin;
            for (; slice + 1 < s_end; slice += 2) {
                sp_slice(MF_IC(0), MF_IC(0), true);
                sp_slice(MF_IC(1), MF_IC(NPAIR & 1), slice + 2 < s_end);
            }
            if (slice < s_end) sp_slice(MF_IC(0), MF_IC(0), false);
            if (NT != 9) sp_deferred();                              // (3x3: the last pair is the lone ninth tap)
#undef MF_IC
        }
    }
    if (!SP) {
        load_halo(s_begin, 0);
        load_wrow(s_begin, 0, 0);
        if (Q) load_wrow(s_begin, 1, 1);
        __syncthreads();                           // drains the DMA (vmcnt) and publishes halo stage 0 + weight row 0
    }
    if (dbg && threadIdx.x == 0 && !SP) dbg[1] = __builtin_amdgcn_s_memtime();
    int wbuf = 0;
    for (int slice = s_begin; slice < (SP ? s_begin : s_end); ++slice) {
        const bool more = slice + 1 < s_end;
        const int st = HS == 2 ? ((slice - s_begin) & 1) : 0;
        if (HS == 2 && more) load_halo(slice + 1, st ^ 1); // flies under this slice's MFMAs
        if (a.res_from_halo) add_residual(st, slice);
        if constexpr (Q) {
            {
                // taps in pairs (0,1) (2,3) (4,5) (6,7) (8): four ring slots, the pair being multiplied in slots {2 wbuf, 2 wbuf + 1} while the next pair
                // lands in the other two (everyone left those at the previous barrier); one barrier per PAIR
                constexpr int NPAIR = (NT + 1) / 2;                  // 5 for the 3x3 layer (the ninth tap alone), 2 for an upsample phase
#pragma unroll
                for (int p = 0; p < NPAIR; ++p) {
                    const int nb = 2 * (wbuf ^ 1);
                    if (p < NPAIR - 1) { load_wrow(slice, 2 * p + 2, nb); if (2 * p + 3 < NT) load_wrow(slice, 2 * p + 3, nb + 1); }
                    else if (more) { load_wrow(slice + 1, 0, nb); load_wrow(slice + 1, 1, nb + 1); }
                    if (cons) compute_pair(st, 2 * p, 2 * p + 1 < NT ? 2 * p + 1 : -1, 2 * wbuf, 2 * wbuf + 1);
                    if (p < NPAIR - 1 || more) __syncthreads();
                    wbuf ^= 1;
                }
                continue;
            }
        }
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            issue_next(slice, step, more, wbuf);
            compute_row(st, step, wbuf, slice, more);
            if (step < NSTEP - 1 || more) __syncthreads();   // next slot (and, at the last step, the next halo image) landed; this one is released
            if (HS == 1 && step == NSTEP - 1 && more) {      // single image: reload it now that every wave is done with it
                load_halo(slice + 1, 0);
                __syncthreads();
            }
            wbuf ^= 1;
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------
    if (dbg && threadIdx.x == 0) dbg[2] = __builtin_amdgcn_s_memtime();
    // GroupNorm statistics of the OUTPUT for the layer's consumer (a.gn_out): per-thread fp32 (sum, sum of squares) of its channel quads
    float gs[Q ? FN : 1], gq[Q ? FN : 1];
#pragma unroll
    for (int i = 0; i < (Q ? FN : 1); ++i) { gs[i] = 0.f; gq[i] = 0.f; }
    // Specialised workgroup: the compute waves hand the lower half of their patch rows to the producer waves through LDS (free now), so that all
    // eight waves issue the epilogue's loads and stores -- with four, the epilogue took 29 k cycles instead of 11 k (store issue is per wave).
    constexpr int FME = SP ? FM / 2 : FM;       // patch rows per wave in the epilogue
    const int erow0 = row0 + (SP && !cons ? FM / 2 : 0);
    if constexpr (SP) {
        f32x4* xch = reinterpret_cast<f32x4*>(smem);
        __syncthreads();                        // every compute wave is done with the halo images and the ring
        if (cons) {
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int jj = 0; jj < FME; ++jj) xch[((wave * FN + i) * FME + jj) * 64 + lane] = acc[i][FME + jj];
        }
        __syncthreads();
        if (!cons) {
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int jj = 0; jj < FME; ++jj) acc[i][jj] = xch[(((wave - NW) * FN + i) * FME + jj) * 64 + lane];
        }
    }
    if (a.ws) {
        // split over channel slices: fp32 partial tile [split][B][H][W][N]; bias, residual, activation and the (hi, lo) store happen in
        // k_splitk_epilogue (mf_conv.hip)
#pragma unroll
        for (int j = 0; j < FME; ++j) {
            const int oy = y0 + erow0 + j, ox = x0 + fr;
            if (oy >= a.H || ox >= a.W) continue;
            float* wo = a.ws + (int64_t)blockIdx.y * a.ws_split + (((int64_t)b * a.H + oy) * a.W + ox) * a.N;
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int c = n0 + cn0 + i * 16 + fk * 4;
                if (c >= a.N) continue;
                *reinterpret_cast<float4*>(wo + c) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        return;
    }
    // Every global LOAD of the epilogue is issued ahead of the stores it does not depend on.  Written the obvious way -- per (row, fragment):
    // bias quad, residual (hi, lo), store (hi, lo) -- each load sits behind the previous fragment's stores: the output may alias the
    // residual, so the compiler cannot hoist it, and on gfx9 loads and stores retire through ONE in-order counter (vmcnt), so the wait for
    // that load is also a wait for the store acknowledgements in front of it.  32 fragments x (store ack + load) was ~95 us of a 205 us
    // tile on the VAE's 256-channel layers (fit over the 8- and 16-slice launches of the r02a profile; 42 us without a residual: the
    // bias quads alone).  Now: bias quads once, residuals of a whole row group (<= 64 VGPRs) in one burst, then that group's stores.
    float4 bq[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        int c = n0 + cn0 + i * 16 + fk * 4;
        c = c < a.Npad - 3 ? c : a.Npad - 4;
        bq[i] = *reinterpret_cast<const float4*>(a.bias + c);
    }
    const bool has_res = a.r_hi != nullptr;
    const int ox = x0 + fr;
    constexpr int JG = (FME * FN * NP <= 32) ? FME : (32 / (FN * NP) >= 1 ? 32 / (FN * NP) : 1);   // rows per residual burst
#pragma unroll
    for (int j0 = 0; j0 < FME; j0 += JG) {
        uint2 rh[JG][FN], rl[JG][FN];
        if (has_res) {
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) {
                const int j = j0 + jj;
                if (j >= FME) break;
                int oy = y0 + erow0 + j, oxc = ox;
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
            if (j >= FME) break;
            const int oy = y0 + erow0 + j;
            const bool row_ok = oy < a.H && ox < a.W;
            const int64_t yo = (int64_t)b * a.yb + (int64_t)oy * a.yi + (int64_t)ox * a.yj;
            constexpr bool W16 = Q && FN % 2 == 0;               // f16 + FP6 tiles: 16-byte stores (a.wide_store)
            uint2 pk_hi[W16 ? FN : 1], pk_lo[W16 ? FN : 1];
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
                if constexpr (W16) {
                    if (a.wide_store) {                                   // (wave-uniform; every lane takes part in the exchange, the store is masked)
                        if (row_ok && c < a.N) {
                            gs[i] += mf_sum4(v[0], v[1], v[2], v[3]);      // sums of neighbours of one register pair: see mf_opaque (mf_common.h)
                            gq[i] += mf_sum4(v[0] * v[0], v[1] * v[1], v[2] * v[2], v[3] * v[3]);
                        }
                        uint32_t h[4], l[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { h[e] = hf2bf(v[e]); l[e] = hf2bf(v[e] - hbf2f(h[e])); }
                        pk_hi[i] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                        pk_lo[i] = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
                        continue;
                    }
                }
                if (!row_ok || c >= a.N) continue;
                if constexpr (Q) {
                    gs[i] += mf_sum4(v[0], v[1], v[2], v[3]);      // sums of neighbours of one register pair: see mf_opaque (mf_common.h)
                    gq[i] += mf_sum4(v[0] * v[0], v[1] * v[1], v[2] * v[2], v[3] * v[3]);
                }
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
            if constexpr (W16) {
                if (a.wide_store) {
#pragma unroll
                    for (int i = 0; i + 1 < FN; i += 2) {
                        hstore_pair16(a.y_hi, yo, n0 + cn0 + i * 16, fk, pk_hi[i], pk_hi[i + 1], a.N, row_ok);
                        if (X3) hstore_pair16(a.y_lo, yo, n0 + cn0 + i * 16, fk, pk_lo[i], pk_lo[i + 1], a.N, row_ok);
                    }
                }
            }
        }
    }
    if constexpr (Q) {
        // The consumer's GroupNorm statistics from the accumulators instead of a second pass over the tensor (k_gn_stats re-reads 4 bytes per
        // value: 268 MB on the 256^2 maps).  A thread's quad lies in one group (channels per group 4, 8 or 16); sum over the wave's 16 pixel
        // columns, park per (wave, quad) in LDS, then one fp64 atomic per (workgroup, group, moment) -- the granularity k_gn_stats has.
        if (a.gn_out) {
            __shared__ float s_gn[NW + NPW][FN * 4][2];
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) { gs[i] += __shfl_xor(gs[i], off); gq[i] += __shfl_xor(gq[i], off); }
            if (fr == 0) {
#pragma unroll
                for (int i = 0; i < FN; ++i) { s_gn[wave][i * 4 + fk][0] = gs[i]; s_gn[wave][i * 4 + fk][1] = gq[i]; }
            }
            __syncthreads();
            const int qpg = a.gn_out_cpg >> 2;                     // quads per group
            const int ng = BN / a.gn_out_cpg;                      // groups this workgroup's channel tile covers
            if (tid < 2 * ng) {
                const int gl = tid >> 1, m = tid & 1;
                double acc_d = 0.0;
                for (int k = 0; k < qpg; ++k) {
                    const int qd = gl * qpg + k;                   // quad within the channel tile: wave_n * (FN * 4) + i * 4 + fk
                    const int wn = qd / (FN * 4), sl = qd - wn * (FN * 4);
#pragma unroll
                    for (int wm = 0; wm < WGM; ++wm) acc_d += (double)s_gn[wn * WGM + wm][sl][m] + (SP ? (double)s_gn[NW + wn * WGM + wm][sl][m] : 0.0);
                }
                const int g = n0 / a.gn_out_cpg + gl;
                if (g < a.gn_out_groups) atomicAdd(a.gn_out + 2 * ((size_t)b * a.gn_out_groups + g) + m, acc_d);
            }
        }
    }
    if (dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // measurement only: the stamp includes the store acknowledgements
        if (threadIdx.x == 0) dbg[3] = __builtin_amdgcn_s_memtime();
    }
}

// ------------------------------------------------------------------------------------------
namespace {

template <int PH, int BN, int WGM, int WGN, bool X3, int TR, int PHASE = -1, int HS = 2, bool Q = false, bool SP = false>
int halo_w_launch_cfg(const HaloArgs& a, hipStream_t s) {
    static bool attr_done = false;
    auto kern = k_conv3x3_halo_w<PH, BN, WGM, WGN, X3, TR, PHASE, HS, Q, SP>;
    if (!attr_done) {
        MF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (Q ? 158 : 160) * 1024));   // Q: 1-2 KiB of static LDS (s_gn) beside the dynamic block
        attr_done = true;
    }
    constexpr int CK = X3 ? 32 : 64, RPC = 1024 / (CK * 2), NP = X3 ? 2 : 1;
    constexpr int HCH = ((PH + 2) * (PW + 2) + RPC - 1) / RPC;
    const size_t lds = (size_t)HS * NP * HCH * 1024 + (size_t)(Q ? 4 : 2) * TR * NP * BN * CK * 2;   // (the two-taps-per-instruction loop of the f16 + FP6 format: four ring slots)
    static const bool dbg_times = mf_debug_has("times");
    HaloArgs aa = a;
    const size_t nwg = (size_t)a.n_patches * a.tiles_n * (a.nsplit > 1 ? a.nsplit : 1);
    if (dbg_times && nwg <= 65536) {
        static unsigned long long* dbg_buf = nullptr;
        if (!dbg_buf) MF_HIP(hipMalloc(&dbg_buf, (size_t)4 * 65536 * sizeof(unsigned long long)));
        aa.dbg = dbg_buf;
    }
    hipLaunchKernelGGL(kern, dim3(a.n_patches * a.tiles_n, a.nsplit > 1 ? a.nsplit : 1), dim3((WGM * WGN + (SP ? 4 : 0)) * 64), lds, s, aa);
    MF_HIP(hipGetLastError());
    if (aa.dbg) {
        static int reports = 0;
        if (++reports > 3 && reports <= 6) {   // skip the warm-up launches
            MF_HIP(hipStreamSynchronize(s));
            std::vector<unsigned long long> t(4 * nwg);
            MF_HIP(hipMemcpy(t.data(), aa.dbg, t.size() * sizeof(t[0]), hipMemcpyDeviceToHost));
            unsigned long long lo = ~0ull, hi = 0;
            std::vector<double> d[3], start, end;
            for (size_t w = 0; w < nwg; ++w) {
                lo = std::min(lo, t[4 * w]); hi = std::max(hi, t[4 * w + 3]);
                for (int k = 0; k < 3; ++k) d[k].push_back((double)(t[4 * w + k + 1] - t[4 * w + k]));
            }
            for (size_t w = 0; w < nwg; ++w) { start.push_back((double)(t[4 * w] - lo)); end.push_back((double)(t[4 * w + 3] - lo)); }
            auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            auto mx = [](const std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
            // s_memtime ticks at 100 MHz on gfx9: 10 ns per tick
            fprintf(stderr, "[MF_DEBUG=times halo_w<%d,%d,%d,%d>] %zu WGs, %d slices: span %.1f; prologue med %.1f max %.1f; loop med %.1f max %.1f; "
                            "epilogue (to store acks) med %.1f max %.1f; WG start med %.1f max %.1f; WG end med %.1f (s_memtime ticks)\n",
                    PH, BN, WGM, WGN, nwg, a.n_slices, (hi - lo) * 1.0, med(d[0]), mx(d[0]), med(d[1]), mx(d[1]), med(d[2]),
                    mx(d[2]), med(start), mx(start), med(end));
        }
    }
    return MF_OK;
}

template <int PH, int BN, int WGM, int WGN, int TR, int PHASE = -1>
int halo_w_launch_prec(const HaloArgs& a, bool x3, hipStream_t s) {
    return x3 ? halo_w_launch_cfg<PH, BN, WGM, WGN, true, TR, PHASE>(a, s) : halo_w_launch_cfg<PH, BN, WGM, WGN, false, TR, PHASE>(a, s);
}
}  // namespace

// Same contract as mf_halo_launch; `t` comes from mf_halo_w_pick_tile.
int mf_halo_w_launch(const HaloArgs& a0, const HaloTile& t, bool x3, hipStream_t s, int phase) {
    HaloArgs a = a0;
    a.patches_x = (a.W + PW - 1) / PW;
    const int patches_y = (a.H + t.ph - 1) / t.ph;
    a.patches_per_img = a.patches_x * patches_y;
    a.n_patches = a.batch * a.patches_per_img;
    a.tiles_n = (a.N + t.bn - 1) / t.bn;
    if (a.q) {
        // the f16 + FP6 format has ONE tile: 16 x 16 pixels x 128 channels, 8 waves of 64 px x 64 ch (the 256-channel tile does not fit its registers,
        // four waves of 128 px x 64 ch measured 6 % slower); nearest 2x upsample + 3x3 runs it as four 2 x 2-tap phases on the same four-slot ring
        if (t.ph != 16 || t.bn != 128 || t.wgm != 4 || (phase >= 0 && a.nsplit > 1)) { mf_set_error("halo conv (f16 + FP6 format): plain 3x3 layers and unsplit upsample phases on the 16 x 16 x 128 tile only"); return MF_ERR_INVALID; }
        // the specialised workgroup (4 compute + 4 producer waves): 25 % fewer LDS fragment reads per MFMA than eight compute waves, 4-7 % faster on every VAE grid
        // in same-box A/B (profiles/r04_halo_sp_study.md; the eight-compute-wave instantiation of round 3 left the library in round 5)
        switch (phase) {
            case 0: return halo_w_launch_cfg<16, 128, 2, 2, true, 1, 0, 2, true, true>(a, s);
            case 1: return halo_w_launch_cfg<16, 128, 2, 2, true, 1, 1, 2, true, true>(a, s);
            case 2: return halo_w_launch_cfg<16, 128, 2, 2, true, 1, 2, 2, true, true>(a, s);
            case 3: return halo_w_launch_cfg<16, 128, 2, 2, true, 1, 3, 2, true, true>(a, s);
            default: return halo_w_launch_cfg<16, 128, 2, 2, true, 1, -1, 2, true, true>(a, s);
        }
    }
    if (phase >= 0) { mf_set_error("halo conv (LDS weights): upsample phases exist in the f16 + FP6 format only"); return MF_ERR_INVALID; }
    // the fat tiles
    if (t.ph == 16 && t.bn == 256 && t.wgm == 2) return halo_w_launch_prec<16, 256, 2, 4, 1>(a, x3, s);   // wave tile 128 px x 64 ch (FM 8, FN 4)
    if (t.ph == 16 && t.bn == 128 && t.wgm == 4) return halo_w_launch_prec<16, 128, 4, 2, 1>(a, x3, s);   // 64 px x 64 ch
    // two 4-wave workgroups per CU (single halo image): wave tile 128 px x 64 ch like the 256-channel tile
    if (t.ph == 16 && t.bn == 128 && t.wgm == 2 && t.wgn == 2)
        return x3 ? halo_w_launch_cfg<16, 128, 2, 2, true, 1, -1, 1>(a, s) : halo_w_launch_cfg<16, 128, 2, 2, false, 1, -1, 1>(a, s);
    mf_set_error("halo conv (LDS weights): no kernel for patch %dx16, BN %d", t.ph, t.bn);
    return MF_ERR_INVALID;
}

// The LDS-weights kernel pays off where the weight stream dominates the patch: 64-channel tiles on maps large enough to give every
// CU a 16 x 16 (or 8 x 16) patch.  Returns ph == 0 when the first-generation kernel should be used.
HaloTile mf_halo_w_pick_tile(int H, int W, int N, int batch, int cin) {
    auto wgs = [&](int ph, int bn) { return batch * ((H + ph - 1) / ph) * ((W + PW - 1) / PW) * ((N + bn - 1) / bn); };
    if (N < 64) return HaloTile{0, 0, 0, 0};
    // fat wave tiles: 16 x 16 pixels x 256 / 128 channels, a ring slot per tap, on maps of at least 64 x 64 (on Wav2Lip's 24^2 / 48^2 layers at batch
    // 128 they measured 2 % slower than the register-weights kernel); 512+ -> 256k-channel layers gain on small maps too once there are 256 patches x
    // channel tiles (512 -> 512 @32^2: 464 -> 521 TF at batch 64, 459 -> 498 at 32; 374 -> 298 at 16, hence the workgroup floor)
    const bool big_map = H * W >= 64 * 64;
    if (N % 256 == 0 && (big_map || cin >= 512) && wgs(16, 256) >= 256) return HaloTile{16, 256, 2, 4};
    // 128-channel tile: two 4-wave workgroups per CU on one halo image each (wave 128 px x 64 ch) wherever that still gives every CU its pair -- each
    // other's DMA waits and epilogues are hidden (128 -> 128 @256^2 401 -> 373 us) -- else one 8-wave workgroup of 4 x 2 waves (64 px x 64 ch each)
    if (N % 128 == 0 && big_map && wgs(16, 128) >= 512) return HaloTile{16, 128, 2, 2};
    if (N % 128 == 0 && big_map && wgs(16, 128) >= 256) return HaloTile{16, 128, 4, 2};
    // the UNet's 320-channel layers at 32 x 32 once a step carries >= 40 frames (mf_conv_plan_create's `odd_wide`; 160+ patches x 3 channel tiles).  The conv
    // alone wins from 24 frames on, but these layers feed GroupNorms with 10 channels per group, whose statistics this epilogue cannot leave (quads straddle
    // groups): the statistics pass behind the conv eats the gain below ~40 frames (whole step at 32 frames: 59.4 -> 59.7 ms)
    if (N >= 256 && N % 128 != 0 && N % 64 == 0 && H * W >= 32 * 32 && wgs(16, 128) >= 480) return HaloTile{16, 128, 4, 2};
    return HaloTile{0, 0, 0, 0};
}

// ER-NeRF radiance field as ONE kernel on gfx950: tri-plane hash-grid gathers, SH, and the nine Linear layers of
// `NeRFNetwork.forward` / `density` (reference: ernerf/nerf_triplane/network.py:249-308, 23 184 MACs per sample) without a
// single intermediate in HBM.
//
// Why: as separate launches (mf_nerf_net.hip) each Linear moves a [M, 64] (hi, lo) tensor out and back -- at M = 262 144 that
// is ~70 MB per layer and the nine GEMMs are memory-bound at ~45 us each.  Fused, a sample costs 12 B in and 28 B out.
//
// How (wave64, MFMA 16x16x32 bf16, fp32 accumulate, bf16x3 = (hi, lo) operands and 3 MFMAs per product):
//   * one wave owns 16 samples (one MFMA fragment column block); D[out channel][sample] = W * X^T, so the weights are the MFMA A
//     operand and a lane of the accumulator tile holds 4 consecutive OUT channels of ONE sample (lane % 16).
//   * a layer's accumulators ARE the next layer's B operand: the contraction index of a 32-deep step is defined as
//     (8g + j) <-> channel 16*(2s + j/4) + 4g + j%4, i.e. exactly what lane (sample, g) already holds from blocks 2s and 2s+1
//     of the previous layer.  Activations never move between lanes or through LDS; the weights are packed on the host in that
//     K order (mf_nerf_fused_pack).
//   * torch.cat is an ordering of K blocks: sigma_net reads [enc_x | enc_a * aud_ch_att | e * eye_att], colour_net reads
//     [geo_feat (sigma_net blocks 0..4, its row 0 = log sigma has zero weights) | SH | individual code].
//   * every weight fragment (63 x 1 KB x planes) sits in LDS, lane-linear, loaded once per workgroup; workgroups are persistent
//     over 256-sample tiles (16 waves x 16 samples).
//   * each lane gathers only the grid features its own B fragments need (channels 4g..4g+3 of each 16-block): 36 bilinear
//     lookups per sample spread over its 4 lanes, straight from the L2-resident tables.
#include "mf_nn.h"
#include "mf_nerf_march.h"
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

namespace {

constexpr int NLEV = 12;
constexpr int NSF = 1;          // 16-sample fragments per wave (2 halves the weight reads per MFMA but spills past 256 VGPRs)
constexpr int NWAVE = 16;            // waves per workgroup: ONE workgroup fits a CU (126 KB of weight fragments), so its size IS the occupancy --
                                     // 16 waves = 4 per SIMD hide the gather latency twice as well as 8 did (the kernel needs 120 VGPRs <= 128)
constexpr int TILE = NWAVE * 16 * NSF;   // samples per workgroup tile
// fragment table: (first fragment, out blocks, k-steps) per layer, fragment = blk * nks + ks
enum { L_AUD0, L_AUD1, L_EYE0, L_EYE1, L_SIG0, L_SIG1, L_SIG2, L_COL0, L_COL1, NLAYER };
constexpr int L_NBLK[NLAYER] = {4, 2, 1, 1, 4, 4, 5, 4, 1};
constexpr int L_NKS[NLAYER] = {2, 2, 2, 1, 3, 2, 2, 4, 2};
constexpr int frag_base(int l) { int b = 0; for (int i = 0; i < l; ++i) b += L_NBLK[i] * L_NKS[i]; return b; }
constexpr int NFRAG = frag_base(NLAYER);      // 63

struct FusedArgs {
    const float *xyzs, *dirs, *enc_a, *ind;
    const float* deltas;             // the march's (dt, t) per sample slot, or null: a slot with dt == 0 holds no sample (the ray ended, or missed) and is not evaluated
    const float* emb[3];
    const bf16_t* w;                 // packed fragments [NFRAG][planes][64 lanes][8]
    float scale[NLEV];
    uint32_t resolution[NLEV], offset[NLEV], hashmap_size[NLEV];
    float bound, eye;
    int n_ind, has_eye, M, ntiles;
    const int* M_dev;                // device-side sample count (sync-free render loop), or null
    const float* eye_dev;            // the eye feature read from device memory (mf_nerf_head_set_eye: no host copy of a value that lives on the device), or null
    float sigma_scale;               // NeRFRenderer.density_scale (renderer.py:261)
    float *sigmas, *rgbs, *amb_aud, *amb_eye, *unc;
};

__device__ __forceinline__ uint32_t fbf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bff(uint32_t h) { return __uint_as_float(h << 16); }

// a B fragment half (4 channels of one 16-block): bf16 (hi, lo) pairs of 4 fp32 values -> 2 + 2 dwords
struct Half { uint32_t h[2], l[2]; };
// (hi, lo) split of two values with gfx950's packed converter: v_cvt_pk_bf16_f32 rounds to nearest even exactly as fbf() does, and the whole
// split is 5 instructions where the integer form needs ~23 -- the kernel packs ~30 of these quads per 16 samples and is VALU-bound
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float v0, float v1, uint32_t& h, uint32_t& l) {
    const f32x2_t v = {v0, v1};
    h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t back = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(v - back, bf16x2_t));
}
__device__ __forceinline__ Half pack4(float v0, float v1, float v2, float v3) {
    Half r;
    split2(v0, v1, r.h[0], r.l[0]);
    split2(v2, v3, r.h[1], r.l[1]);
    return r;
}
__device__ __forceinline__ Half zero_half() { Half r; r.h[0] = r.h[1] = r.l[0] = r.l[1] = 0u; return r; }
struct BFrag { bf16x8 hi, lo; };
__device__ __forceinline__ BFrag join(const Half& p0, const Half& p1) {
    BFrag f;
    f.hi = __builtin_bit_cast(bf16x8, u32x4{p0.h[0], p0.h[1], p1.h[0], p1.h[1]});
    f.lo = __builtin_bit_cast(bf16x8, u32x4{p0.l[0], p0.l[1], p1.l[0], p1.l[1]});
    return f;
}

// per-level constants in LDS: the level index differs from lane to lane, which kernel-argument arrays cannot serve
struct LevelTab { float scale[NLEV]; uint32_t resolution[NLEV], offset[NLEV], hashmap_size[NLEV], mask[NLEV]; };   // mask: hashmap_size - 1 if that is a power of two, else 0

// One 16-sample fragment per wave of the radiance field: samples s0 .. s0 + 15 (those below M) of a.xyzs / a.dirs -> a.sigmas, a.rgbs, a.amb_*, a.unc.
// `smem`: the NFRAG * NP KiB of weight fragments, `lt`: the level constants, both already in LDS.  A sample's result depends on nothing but that sample
// (a column of every MFMA), so any launch shape that calls this gets the same bits.
template <bool X3>
__device__ __forceinline__ void field_tile(const FusedArgs& a, const char* smem, const LevelTab& lt, const float eye_v, const int s0, const int M) {
    constexpr int NP = X3 ? 2 : 1;
    const int lane = threadIdx.x & 63, fr = lane & 15, g = lane >> 4;
    const float* const emb0 = a.emb[0];
    const float* const emb1 = a.emb[1];
    const float* const emb2 = a.emb[2];

    auto wfrag = [&](int f, int plane) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8*>(smem + ((size_t)(f * NP + plane) * 64 + lane) * 16);
    };
    // acc[blk][sf] += W(layer, blk, ks) * B[sf]
    auto mma = [&](int f, const BFrag (&b)[NSF], f32x4 (&acc)[NSF]) __attribute__((always_inline)) {
        const bf16x8 whi = wfrag(f, 0);
        if constexpr (X3) {
            const bf16x8 wlo = wfrag(f, 1);
#pragma unroll
            for (int sf = 0; sf < NSF; ++sf) {
                acc[sf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, b[sf].hi, acc[sf], 0, 0, 0);
                acc[sf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, b[sf].lo, acc[sf], 0, 0, 0);
            }
        }
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) acc[sf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, b[sf].hi, acc[sf], 0, 0, 0);
    };
    // between layers: stops the scheduler from hoisting the next layers' 1-KiB weight fragments into registers early
#define LAYER_FENCE() __builtin_amdgcn_sched_barrier(0)

    // The reference runs the network over every slot of the round's [n_alive x n_step] tensors, the zero-filled ones of rays that produced fewer samples included
    // (renderer.py:258-261); composite_rays stops at a slot with dt == 0 before it reads that slot's outputs (raymarching.cu:2180).  A fragment whose 16 slots are
    // all empty -- rays that miss the head in round 1 lie side by side, ended rays leave whole runs -- is skipped: same frame, fewer gathers and MFMAs.
    if (a.deltas) {
        const int mq = s0 + fr;
        const bool has = mq < M && a.deltas[2 * (size_t)mq] != 0.f;
        if (!__any(has)) return;
    }
    const float inv2b = 1.f / (2.f * a.bound);
    {
        // ---- inputs of this lane's two samples ------------------------------------------------------------------
        float px[NSF], py[NSF], pz[NSF], dx[NSF], dy[NSF], dz[NSF];
        bool live[NSF];
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) {
            int m = s0 + sf * 16 + fr;
            live[sf] = m < M;
            m = live[sf] ? m : M - 1;
            px[sf] = (a.xyzs[3 * m] + a.bound) * inv2b; py[sf] = (a.xyzs[3 * m + 1] + a.bound) * inv2b; pz[sf] = (a.xyzs[3 * m + 2] + a.bound) * inv2b;
            dx[sf] = a.dirs[3 * m]; dy[sf] = a.dirs[3 * m + 1]; dz[sf] = a.dirs[3 * m + 2];
        }
        // enc_x channel c = plane * 12 + level (network.py:204-219).  The contraction order of a layer is free (the weights are packed to it on the host), so the 36
        // channels are dealt to a sample's four lanes by LEVEL: lane group g gathers levels g, g + 4, g + 8 of all three planes -- nine lookups in every lane (the
        // block order 0..15 | 16..31 | 32..35 gave lane group 0 twelve and made the wave wait for them), the plane of every lookup a compile-time constant (its table a
        // scalar base, its coordinate pair fixed) and the level arithmetic -- constants from LDS, floor / fraction of x, y, z, dense-or-hashed -- done once per level
        // instead of once per lookup.  Values per channel are bit-identical to the per-channel form of rounds 3 - 4.  Slots: X0 = (l0 p0, l0 p1, l0 p2, l1 p0), X1 = (l1 p1, l1 p2, l2 p0,
        // l2 p1), X2 = (l2 p2, 0, 0, 0) of the lane's own three levels (mf_nerf_fused_pack: xblock()).
        Half x0[NSF], x1[NSF], x2[NSF];
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) {
            const bool okx = px[sf] >= 0.f && px[sf] <= 1.f, oky = py[sf] >= 0.f && py[sf] <= 1.f, okz = pz[sf] >= 0.f && pz[sf] <= 1.f;
            // outside [0, 1] a plane's feature is 0 (gridencoder.cu:96-104); the clamp only keeps the discarded lookup's addresses inside the table
            const float cx = __builtin_amdgcn_fmed3f(px[sf], 0.f, 1.f), cy = __builtin_amdgcn_fmed3f(py[sf], 0.f, 1.f), cz = __builtin_amdgcn_fmed3f(pz[sf], 0.f, 1.f);
            float v[3][3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int lv = g + 4 * t;
                const float scale = lt.scale[lv];
                const uint32_t res = lt.resolution[lv], hs = lt.hashmap_size[lv], msk = lt.mask[lv], off = lt.offset[lv];
                const uint32_t s1 = res + 1;
                const bool dense = s1 * s1 <= hs;
                float fx = cx * scale + 0.5f, fy = cy * scale + 0.5f, fz = cz * scale + 0.5f;
                const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
                const uint32_t ix = (uint32_t)flx, iy = (uint32_t)fly, iz = (uint32_t)flz;
                fx -= flx; fy -= fly; fz -= flz;
                // one plane of this level (gridencoder.cu:76-165 with D = 2, C = 1, hash grid): get_grid_index (gridencoder.cu:54-72) for the four corners at once.
                // With s = res + 1 the level is DENSE iff s * s <= hashmap_size (then index = x + y s, below s * s, so `% hashmap_size` is the identity); otherwise
                // index = x ^ (y * 2654435761) reduced mod hashmap_size -- a mask when the table is a power of two (2^log2_hashmap_size: grid.py:108-123), the
                // division for any other size.  One multiply per form instead of one of each per corner, and no 32-bit urem (~25 instructions) per lookup.
                auto plane = [&](const float* __restrict__ tab, uint32_t iu, uint32_t iv, float pu, float pv, bool ok) __attribute__((always_inline)) {
                    // (a dense level's table size is (res + 1)^2 rounded up to 8 -- not a power of two: taking the hashed form first and selecting afterwards, as
                    // rounds 3 - 4 did, ran the four divisions for every dense level and threw the results away)
                    uint32_t i00, i10, i01, i11;
                    if (dense) {
                        i00 = iu + iv * s1; i10 = i00 + 1; i01 = i00 + s1; i11 = i01 + 1;
                    } else {
                        const uint32_t h0 = iv * 2654435761u, h1 = h0 + 2654435761u;
                        i00 = iu ^ h0; i10 = (iu + 1) ^ h0; i01 = iu ^ h1; i11 = (iu + 1) ^ h1;
                        if (msk) { i00 &= msk; i10 &= msk; i01 &= msk; i11 &= msk; }
                        else { i00 %= hs; i10 %= hs; i01 %= hs; i11 %= hs; }
                    }
                    // scalar table base + 32-bit byte offset (the tables are a few MB): no 64-bit address arithmetic per corner
                    const char* base = reinterpret_cast<const char*>(tab);
                    const float g00 = *reinterpret_cast<const float*>(base + ((off + i00) << 2)), g10 = *reinterpret_cast<const float*>(base + ((off + i10) << 2));
                    const float g01 = *reinterpret_cast<const float*>(base + ((off + i01) << 2)), g11 = *reinterpret_cast<const float*>(base + ((off + i11) << 2));
                    const float qu = 1.f - pu, qv = 1.f - pv;
                    float r = 0.f;                     // corner order and arithmetic of the reference's loop: (0,0), (1,0), (0,1), (1,1)
                    r += (qu * qv) * g00;
                    r += (pu * qv) * g10;
                    r += (qu * pv) * g01;
                    r += (pu * pv) * g11;
                    return ok ? r : 0.f;
                };
                v[t][0] = plane(emb0, ix, iy, fx, fy, okx && oky);                // xy
                v[t][1] = plane(emb1, iy, iz, fy, fz, oky && okz);                // yz
                v[t][2] = plane(emb2, ix, iz, fx, fz, okx && okz);                // xz
                __builtin_amdgcn_sched_barrier(0);      // keep the 12 gathers of one level together, not all 36 of the lane in flight
            }
            x0[sf] = pack4(v[0][0], v[0][1], v[0][2], v[1][0]);
            x1[sf] = pack4(v[1][1], v[1][2], v[2][0], v[2][1]);
            x2[sf] = pack4(v[2][2], 0.f, 0.f, 0.f);
        }
        const Half Z = zero_half();
        BFrag bx0[NSF], bx1[NSF];
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) {
            bx0[sf] = join(x0[sf], x1[sf]);                                      // k-step (X0, X1)
            bx1[sf] = join(x2[sf], Z);                                           // k-step (X2, 0)
        }

        // ---- aud_ch_att_net: 36 -> 64 relu -> 32 (network.py:148, 284-285) ---------------------------------------
        f32x4 h1[4][NSF];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            for (int sf = 0; sf < NSF; ++sf) h1[blk][sf] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma(frag_base(L_AUD0) + blk * 2 + 0, bx0, h1[blk]);
            mma(frag_base(L_AUD0) + blk * 2 + 1, bx1, h1[blk]);
        }
        LAYER_FENCE();
        // ReLU as ONE integer instruction on the float's bits (v_max_i32 with 0: non-negative floats are non-negative integers, every negative one -- -0 included --
        // becomes +0); fmaxf() costs two (it first quiets its operand) and the kernel is VALU-bound
        auto relu1 = [](float x) __attribute__((always_inline)) { const int b = __float_as_int(x); return __int_as_float(b > 0 ? b : 0); };
        auto relu_half = [&](const f32x4& v) __attribute__((always_inline)) { return pack4(relu1(v[0]), relu1(v[1]), relu1(v[2]), relu1(v[3])); };
        auto lin_half = [&](const f32x4& v) __attribute__((always_inline)) { return pack4(v[0], v[1], v[2], v[3]); };
        BFrag t0[NSF], t1[NSF];
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) {
            t0[sf] = join(relu_half(h1[0][sf]), relu_half(h1[1][sf]));
            t1[sf] = join(relu_half(h1[2][sf]), relu_half(h1[3][sf]));
        }
        LAYER_FENCE();
        f32x4 aud[2][NSF];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            for (int sf = 0; sf < NSF; ++sf) aud[blk][sf] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma(frag_base(L_AUD1) + blk * 2 + 0, t0, aud[blk]);
            mma(frag_base(L_AUD1) + blk * 2 + 1, t1, aud[blk]);
        }
        LAYER_FENCE();
        // ambient_aud = ||aud_ch_att||_2 (network.py:306): 8 channels in this lane, the other 24 in lanes g' != g of the same sample
        float amb[NSF];
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) {
            float n2 = 0.f;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int e = 0; e < 4; ++e) n2 += aud[blk][sf][e] * aud[blk][sf][e];
            n2 += __shfl_xor(n2, 16);
            n2 += __shfl_xor(n2, 32);
            amb[sf] = sqrtf(n2);
        }
        // enc_w = enc_a * aud_ch_att (network.py:286): this lane's channels 16*blk + 4g + e
        Half ew[2][NSF];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const float4 ea = *reinterpret_cast<const float4*>(a.enc_a + blk * 16 + 4 * g);
#pragma unroll
            for (int sf = 0; sf < NSF; ++sf)
                ew[blk][sf] = pack4(ea.x * aud[blk][sf][0], ea.y * aud[blk][sf][1], ea.z * aud[blk][sf][2], ea.w * aud[blk][sf][3]);
        }

        LAYER_FENCE();
        // ---- eye_att_net: 36 -> 16 relu -> 1, sigmoid (network.py:137, 291-292) ----------------------------------
        float eye_att[NSF];
        Half eyeh[NSF];
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) { eye_att[sf] = 0.f; eyeh[sf] = Z; }
        if (a.has_eye) {
            f32x4 e1[NSF], e2[NSF];
            BFrag te[NSF];
#pragma unroll
            for (int sf = 0; sf < NSF; ++sf) e1[sf] = e2[sf] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma(frag_base(L_EYE0) + 0, bx0, e1);
            mma(frag_base(L_EYE0) + 1, bx1, e1);
#pragma unroll
            for (int sf = 0; sf < NSF; ++sf) te[sf] = join(relu_half(e1[sf]), Z);
            mma(frag_base(L_EYE1), te, e2);
#pragma unroll
            for (int sf = 0; sf < NSF; ++sf) {
                // row 0 of the block lives in lanes g == 0 (element 0); the other lanes hold zero-weight rows
                const float s = 1.f / (1.f + __expf(-e2[sf][0]));
                eye_att[sf] = s;
                eyeh[sf] = g == 0 ? pack4(eye_v * s, 0.f, 0.f, 0.f) : Z;
            }
        }

        LAYER_FENCE();
        // ---- sigma_net: [enc_x 36 | enc_w 32 | e 1] -> 64 -> 64 -> 65 (network.py:139, 294-302) -------------------
        BFrag bs1[NSF], bs2[NSF];
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) {
            bs1[sf] = join(x2[sf], ew[0][sf]);                                   // k-step (X2, W0)
            bs2[sf] = join(ew[1][sf], eyeh[sf]);                                 // k-step (W1, eye)
        }
        f32x4 s1[4][NSF];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            for (int sf = 0; sf < NSF; ++sf) s1[blk][sf] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma(frag_base(L_SIG0) + blk * 3 + 0, bx0, s1[blk]);
            mma(frag_base(L_SIG0) + blk * 3 + 1, bs1, s1[blk]);
            mma(frag_base(L_SIG0) + blk * 3 + 2, bs2, s1[blk]);
        }
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) {
            t0[sf] = join(relu_half(s1[0][sf]), relu_half(s1[1][sf]));
            t1[sf] = join(relu_half(s1[2][sf]), relu_half(s1[3][sf]));
        }
        LAYER_FENCE();
        f32x4 s2[4][NSF];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            for (int sf = 0; sf < NSF; ++sf) s2[blk][sf] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma(frag_base(L_SIG1) + blk * 2 + 0, t0, s2[blk]);
            mma(frag_base(L_SIG1) + blk * 2 + 1, t1, s2[blk]);
        }
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) {
            t0[sf] = join(relu_half(s2[0][sf]), relu_half(s2[1][sf]));
            t1[sf] = join(relu_half(s2[2][sf]), relu_half(s2[3][sf]));
        }
        LAYER_FENCE();
        f32x4 s3[5][NSF];
#pragma unroll
        for (int blk = 0; blk < 5; ++blk) {
            for (int sf = 0; sf < NSF; ++sf) s3[blk][sf] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma(frag_base(L_SIG2) + blk * 2 + 0, t0, s3[blk]);
            mma(frag_base(L_SIG2) + blk * 2 + 1, t1, s3[blk]);
        }
        // row 0 = log sigma (lanes g == 0, element 0); rows 1..64 = geo_feat (network.py:300-301)

        LAYER_FENCE();
        // ---- colour_net: [geo (blocks 0..4) | SH 16 | ind 4] -> 64 -> 3 (network.py:144, 262-272) -----------------
        Half shh[NSF], indh;
        {
            const float4 iv = g == 0 ? make_float4(a.n_ind > 0 ? a.ind[0] : 0.f, a.n_ind > 1 ? a.ind[1] : 0.f, a.n_ind > 2 ? a.ind[2] : 0.f,
                                                   a.n_ind > 3 ? a.ind[3] : 0.f)
                                     : (g == 1 ? make_float4(a.n_ind > 4 ? a.ind[4] : 0.f, a.n_ind > 5 ? a.ind[5] : 0.f, a.n_ind > 6 ? a.ind[6] : 0.f,
                                                             a.n_ind > 7 ? a.ind[7] : 0.f)
                                               : make_float4(0.f, 0.f, 0.f, 0.f));
            indh = pack4(iv.x, iv.y, iv.z, iv.w);
        }
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) {
            const float x = dx[sf], y = dy[sf], z = dz[sf];
            const float xy = x * y, xz = x * z, yz = y * z, x2_ = x * x, y2 = y * y, z2 = z * z;
            float sh[16];
            sh[0] = 0.28209479177387814f;
            sh[1] = -0.48860251190291987f * y; sh[2] = 0.48860251190291987f * z; sh[3] = -0.48860251190291987f * x;
            sh[4] = 1.0925484305920792f * xy; sh[5] = -1.0925484305920792f * yz; sh[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
            sh[7] = -1.0925484305920792f * xz; sh[8] = 0.54627421529603959f * x2_ - 0.54627421529603959f * y2;
            sh[9] = 0.59004358992664352f * y * (-3.0f * x2_ + y2); sh[10] = 2.8906114426405538f * xy * z;
            sh[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2); sh[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
            sh[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2); sh[14] = 1.4453057213202769f * z * (x2_ - y2);
            sh[15] = 0.59004358992664352f * x * (-x2_ + 3.0f * y2);
            float v0 = sh[0], v1 = sh[1], v2 = sh[2], v3 = sh[3];
            if (g == 1) { v0 = sh[4]; v1 = sh[5]; v2 = sh[6]; v3 = sh[7]; }
            if (g == 2) { v0 = sh[8]; v1 = sh[9]; v2 = sh[10]; v3 = sh[11]; }
            if (g == 3) { v0 = sh[12]; v1 = sh[13]; v2 = sh[14]; v3 = sh[15]; }
            shh[sf] = pack4(v0, v1, v2, v3);
        }
        BFrag c0[NSF], c1[NSF], c2[NSF], c3[NSF];
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) {
            c0[sf] = join(lin_half(s3[0][sf]), lin_half(s3[1][sf]));
            c1[sf] = join(lin_half(s3[2][sf]), lin_half(s3[3][sf]));
            c2[sf] = join(lin_half(s3[4][sf]), shh[sf]);
            c3[sf] = join(indh, Z);
        }
        LAYER_FENCE();
        f32x4 k1[4][NSF];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            for (int sf = 0; sf < NSF; ++sf) k1[blk][sf] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma(frag_base(L_COL0) + blk * 4 + 0, c0, k1[blk]);
            mma(frag_base(L_COL0) + blk * 4 + 1, c1, k1[blk]);
            mma(frag_base(L_COL0) + blk * 4 + 2, c2, k1[blk]);
            mma(frag_base(L_COL0) + blk * 4 + 3, c3, k1[blk]);
        }
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) {
            t0[sf] = join(relu_half(k1[0][sf]), relu_half(k1[1][sf]));
            t1[sf] = join(relu_half(k1[2][sf]), relu_half(k1[3][sf]));
        }
        LAYER_FENCE();
        f32x4 rgb[NSF];
#pragma unroll
        for (int sf = 0; sf < NSF; ++sf) rgb[sf] = f32x4{0.f, 0.f, 0.f, 0.f};
        mma(frag_base(L_COL1) + 0, t0, rgb);
        mma(frag_base(L_COL1) + 1, t1, rgb);

        LAYER_FENCE();
        // ---- outputs: lanes g == 0 hold row 0 of sigma_net (element 0) and rows 0..2 of colour_net ---------------
        if (g == 0) {
#pragma unroll
            for (int sf = 0; sf < NSF; ++sf) {
                if (!live[sf]) continue;
                const int m = s0 + sf * 16 + fr;
                a.sigmas[m] = a.sigma_scale * expf(s3[0][sf][0]);                                                   // network.py:300
#pragma unroll
                for (int k = 0; k < 3; ++k) a.rgbs[3 * m + k] = 1.f / (1.f + __expf(-rgb[sf][k])) * 1.002f - 0.001f;   // network.py:272
                a.amb_aud[m] = amb[sf];
                a.amb_eye[m] = eye_att[sf];
                if (a.unc) a.unc[m] = 0.69314718055994530942f;
            }
        }
    }
}

// weight fragments and level constants into LDS (every thread of the workgroup; ends with a barrier)
template <bool X3>
__device__ __forceinline__ void field_stage_weights(const FusedArgs& a, char* smem, LevelTab& lt, int nthreads) {
    constexpr int NP = X3 ? 2 : 1;
    const int tid = threadIdx.x;
    for (int i = tid; i < NFRAG * NP * 64; i += nthreads)
        reinterpret_cast<u32x4*>(smem)[i] = reinterpret_cast<const u32x4*>(a.w)[i];
    if (tid < NLEV) { lt.scale[tid] = a.scale[tid]; lt.resolution[tid] = a.resolution[tid]; lt.offset[tid] = a.offset[tid]; lt.hashmap_size[tid] = a.hashmap_size[tid]; lt.mask[tid] = (a.hashmap_size[tid] & (a.hashmap_size[tid] - 1)) == 0 ? a.hashmap_size[tid] - 1 : 0u; }
    __syncthreads();
}

template <bool X3>
__global__ __launch_bounds__(NWAVE * 64) void k_nerf_field_fused(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // NFRAG * NP KiB of weight fragments
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ LevelTab lt;
    const int M = a.M_dev ? *a.M_dev : a.M;
    const int ntiles = (M + TILE - 1) / TILE;
    if ((int)blockIdx.x >= ntiles) return;          // also the "round already finished" case of the device-controlled loop (M == 0)
    field_stage_weights<X3>(a, smem, lt, NWAVE * 64);
    const float eye_v = a.eye_dev ? *a.eye_dev : a.eye;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
        field_tile<X3>(a, smem, lt, eye_v, tile * TILE + wave * 16 * NSF, M);
}

// ---- the render loop's TAIL: every round the launch chain did not enqueue, in one launch ------------------------------------------------------
// mf_nerf_head_render enqueues R rounds as (march, field, composite) launches -- R follows the round counts of the frames before (a frame needs ~5 of the
// max_steps = 16 the reference allows: renderer.py:246-270) -- and then this kernel ONCE.  It finds the loop ended (ctl[1] == 0: the usual case, one empty launch
// instead of 3 x (16 - R)) or runs the remaining rounds itself:
//   * a round is cut into chunks of 512 alive rays; a workgroup takes a chunk by ticket and carries it through the WHOLE round -- march (one ray per lane, the
//     reference's loop: march_ray_ref), field (field_tile over the chunk's own samples), composite (composite_ray) and the append of its survivors.  Rays do not
//     interact inside a round, so the only grid-wide step is the head of the next round (survivor count -> n_step): the workgroup that finishes the round's last
//     chunk computes it and publishes `ready[j + 1]`; the others wait for that flag -- and only for that flag, which a RUNNING workgroup will set.  No workgroup
//     ever waits for one that has not started, so the kernel needs no co-residency (two sessions' tails on one GPU cannot block each other).
//   * same per-ray and per-sample arithmetic as the launches (the survivors' order differs; results do not depend on it), so a frame is the same bits wherever the
//     chain hands over (tests/test_ernerf.py: hand-over after 0, 1, 2, 3 rounds against the launch-only loop).
// Visibility between workgroups (other XCDs have their own L2): agent-scope release (fence + atomic) by the writer, acquire (atomic + fence) by the readers.
struct TailArgs {
    int* ctl;
    int N, max_steps, first_parity;      // first_parity: which of alive[0 / 1] the first tail round reads
    float T_thresh, dt_gamma;
    uint32_t C, H;
    int *alive0, *alive1;                // (two fields, selected -- a dynamically indexed array in the kernel arguments goes through scratch, and a kernel with scratch
                                         // costs more to dispatch: the EMPTY tail launch is what every frame pays)
    float* rays_t;
    const float *rays_o, *rays_d, *fars;
    const uint8_t* grid;
    float *xyzs, *dirs, *deltas;
    float *wsum, *depth, *image, *aasum, *aesum, *unsum;
};
constexpr int TAIL_NW = 8;               // waves per tail workgroup: two per SIMD, so the compiler has 256 registers per lane -- with the field's 16 waves (128) the loop
                                         // state around field_tile spilled, and a kernel with SCRATCH costs more to dispatch: 4.8 us for the empty tail launch every frame pays
constexpr int TAIL_RC = TAIL_NW * 64;    // rays per chunk: one per lane
constexpr int TAIL_TILE = TAIL_NW * 16 * NSF;
#ifndef MF_TAIL_SPINS
#define MF_TAIL_SPINS (1 << 20)
#endif
constexpr int TAIL_SPINS = MF_TAIL_SPINS;   // polls of ~2 us before a waiting workgroup gives up (seconds)

template <bool X3>
__global__ __launch_bounds__(TAIL_NW * 64) void k_loop_tail(const FusedArgs a, const TailArgs t) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ LevelTab lt;
    __shared__ int s_i[4], s_wave[TAIL_NW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the round about to run, as the last composite launch (or k_loop_init) left it; nothing writes ctl[0..2] while this kernel runs
    int n_alive = t.ctl[0], n_step = t.ctl[1], step_after = t.ctl[2];
    if (n_step <= 0) return;                                                       // the loop has ended: the usual case
    field_stage_weights<X3>(a, smem, lt, TAIL_NW * 64);
    const float eye_v = a.eye_dev ? *a.eye_dev : a.eye;
    const int TS = t.max_steps + 1;
    int* const take = t.ctl + LOOP_CTL_TAIL;
    int* const fin = take + TS;
    int* const surv = fin + TS;
    int* const ready = surv + TS;
    int* const p_alive = ready + TS;
    int* const p_step = p_alive + TS;
    int* const p_after = p_step + TS;
    // every loop condition below is a wave-uniform SCALAR (readfirstlane of a value all lanes read from LDS): the workgroup's 16 waves take every branch together
    // and meet at every barrier.  (Written with plain ints the compiler cannot prove the ticket loop's exit uniform, builds per-lane exit masks around the
    // barriers, and the waves of a workgroup left the loop at different times -- a hang, found with the stage markers of profiles/r06_tail_trace.patch.)
    auto uni = [](int x) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(x); };
    auto next_ticket = [&](int j) __attribute__((always_inline)) {
        if (tid == 0) s_i[0] = atomicAdd(&take[j], 1);
        __syncthreads();
        const int c = uni(s_i[0]);
        __syncthreads();
        return c;
    };
    n_alive = uni(n_alive); n_step = uni(n_step); step_after = uni(step_after);
    for (int j = 0; j < t.max_steps; ++j) {
        const bool odd = ((t.first_parity + j) & 1) != 0;
        const int* a_in = odd ? t.alive1 : t.alive0;
        int* a_out = odd ? t.alive0 : t.alive1;
        const int nchunks = (n_alive + TAIL_RC - 1) / TAIL_RC;
        for (int c = next_ticket(j); c < nchunks; c = next_ticket(j)) {
            const int base = c * TAIL_RC;
            const int rays_here = min(TAIL_RC, n_alive - base);
            const uint32_t n = (uint32_t)(base + tid);
            int v = -1;
            if (tid < rays_here) {
                v = a_in[n];
                march_ray_ref(n, (uint32_t)n_step, v, 0.f, t.rays_t, t.rays_o, t.rays_d, a.bound, t.dt_gamma, (uint32_t)t.max_steps, t.C, t.H, t.grid, t.fars,
                              t.xyzs, t.dirs, t.deltas, true);
            }
            __threadfence();
            __syncthreads();
            const int m0 = base * n_step, mend = m0 + rays_here * n_step;
            for (int s0 = m0; s0 < mend; s0 += TAIL_TILE) field_tile<X3>(a, smem, lt, eye_v, s0 + wave * 16 * NSF, mend);
            __threadfence();
            __syncthreads();
            bool keep = false;
            if (tid < rays_here)
                keep = !composite_ray(n, (uint32_t)n_step, t.T_thresh, v, t.rays_t, a.sigmas, a.rgbs, t.deltas, a.amb_aud, a.amb_eye, a.unc, t.wsum, t.depth, t.image,
                                      t.aasum, t.aesum, t.unsum);
            // `rays_alive = rays_alive[rays_alive >= 0]` (renderer.py:266): wave-aggregated append, one atomic per chunk
            const unsigned long long m = __ballot(keep);
            if (lane == 0) s_wave[wave] = __popcll(m);
            __syncthreads();
            int before = 0, mine = 0;
#pragma unroll
            for (int w = 0; w < TAIL_NW; ++w) { before += w < wave ? s_wave[w] : 0; mine += s_wave[w]; }
            if (tid == 0) s_i[1] = atomicAdd(&surv[j], mine);
            __syncthreads();
            if (keep) a_out[s_i[1] + before + __popcll(m & ((1ull << lane) - 1))] = v;
            __threadfence();
            __syncthreads();
            if (tid == 0) {
                const int done = atomicAdd(&fin[j], 1);
                if (done == nchunks - 1) {                                          // the round's last chunk: head of the next round
                    const int na = __hip_atomic_load(&surv[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int ns = j + 1 < t.max_steps ? loop_n_step(na, step_after, t.N, t.max_steps) : 0;
                    p_alive[j + 1] = ns ? na : 0; p_step[j + 1] = ns; p_after[j + 1] = step_after + ns;
                    if (ns) t.ctl[LOOP_CTL_ROUNDS] += 1;
                    else loop_post_feedback(t.ctl);
                    __hip_atomic_store(&ready[j + 1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (tid == 0) {
            // every chunk of round j is taken, each by a workgroup that is running it: the flag WILL be set.  The bound only turns a protocol bug into an error the
            // host sees (ctl[9] -> the feedback word) instead of a hung GPU.
            int spins = 0;
            while (__hip_atomic_load(&ready[j + 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0 && spins < TAIL_SPINS) { __builtin_amdgcn_s_sleep(32); ++spins; }
            if (spins >= TAIL_SPINS) {
                __hip_atomic_store(&t.ctl[LOOP_CTL_ERR], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                loop_post_feedback(t.ctl);
                s_i[1] = 0; s_i[2] = 0; s_i[3] = 0;
            } else {
                s_i[1] = p_alive[j + 1]; s_i[2] = p_step[j + 1]; s_i[3] = p_after[j + 1];
            }
        }
        __syncthreads();
        n_alive = uni(s_i[1]); n_step = uni(s_i[2]); step_after = uni(s_i[3]);
        __syncthreads();
        if (n_step <= 0) return;
        __threadfence();                                                            // every lane sees what the other workgroups wrote in round j
    }
}
}  // namespace

// ---- host: weight fragments in the kernel's K order ----------------------------------------------------------------------
// `blocks`: the 16-channel K blocks of a layer in k-step order (pairs), each entry a source column of `w` ([cout][cin]) or -1.
static void pack_layer(const float* w, int cout, int cin, const std::vector<std::vector<int>>& blocks, int nblk, int nks, bool x3,
                       std::vector<bf16_t>& dst, int frag0) {
    const int np = x3 ? 2 : 1;
    for (int blk = 0; blk < nblk; ++blk)
        for (int ks = 0; ks < nks; ++ks) {
            const int f = frag0 + blk * nks + ks;
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 15, g = lane >> 4;
                for (int j = 0; j < 8; ++j) {
                    const std::vector<int>& kb = blocks[2 * ks + (j >> 2)];
                    const int src = kb[4 * g + (j & 3)];
                    const int row = blk * 16 + m;
                    const float v = (src >= 0 && row < cout) ? w[(size_t)row * cin + src] : 0.f;
                    const bf16_t hi = mf_f2bf(v);
                    dst[((size_t)(f * np + 0) * 64 + lane) * 8 + j] = hi;
                    if (x3) dst[((size_t)(f * np + 1) * 64 + lane) * 8 + j] = mf_f2bf(v - mf_bf2f(hi));
                }
            }
        }
}

static std::vector<int> kblock(int first, int n = 16) {   // n real channels first..first+n-1, padded with -1
    std::vector<int> b(16, -1);
    for (int i = 0; i < n; ++i) b[i] = first + i;
    return b;
}

// weights: the nine [cout][cin] matrices in reference order (aud0, aud1, eye0, eye1, sig0, sig1, sig2, col0, col1; eye* may be null)
int mf_nerf_fused_pack(const float* const w[9], int n_ind, bool has_eye, bool x3, bf16_t** dev_out) {
    const int np = x3 ? 2 : 1;
    std::vector<bf16_t> buf((size_t)NFRAG * np * 64 * 8, 0);
    const std::vector<int> Z(16, -1);
    const int sig_in = 36 + 32 + (has_eye ? 1 : 0), col_in = 16 + 64 + n_ind;
    // enc_x blocks in the kernel's gather order: lane group g holds levels g, g + 4, g + 8; slot i of block b is (level t, plane p) of that lane, channel p * 12 + level
    auto xblock = [](int which) {
        static const int tp[3][4][2] = {{{0, 0}, {0, 1}, {0, 2}, {1, 0}}, {{1, 1}, {1, 2}, {2, 0}, {2, 1}}, {{2, 2}, {-1, -1}, {-1, -1}, {-1, -1}}};
        std::vector<int> b(16, -1);
        for (int g = 0; g < 4; ++g)
            for (int i = 0; i < 4; ++i)
                if (tp[which][i][0] >= 0) b[4 * g + i] = tp[which][i][1] * 12 + g + 4 * tp[which][i][0];
        return b;
    };
    const std::vector<std::vector<int>> X = {xblock(0), xblock(1), xblock(2), Z};
    const std::vector<std::vector<int>> H64 = {kblock(0), kblock(16), kblock(32), kblock(48)};
    pack_layer(w[0], 64, 36, X, 4, 2, x3, buf, frag_base(L_AUD0));
    pack_layer(w[1], 32, 64, H64, 2, 2, x3, buf, frag_base(L_AUD1));
    if (has_eye) {
        pack_layer(w[2], 16, 36, X, 1, 2, x3, buf, frag_base(L_EYE0));
        pack_layer(w[3], 1, 16, {kblock(0), Z}, 1, 1, x3, buf, frag_base(L_EYE1));
    }
    // sigma_net.0: reference columns [enc_x 0..35 | enc_w 36..67 | e 68]; K blocks (X0, X1), (X2, W0), (W1, eye)
    pack_layer(w[4], 64, sig_in, {xblock(0), xblock(1), xblock(2), kblock(36), kblock(52), has_eye ? kblock(68, 1) : Z}, 4, 3, x3, buf, frag_base(L_SIG0));
    pack_layer(w[5], 64, 64, H64, 4, 2, x3, buf, frag_base(L_SIG1));
    pack_layer(w[6], 65, 64, H64, 5, 2, x3, buf, frag_base(L_SIG2));
    {
        // colour_net.0: reference columns [SH 0..15 | geo 16..79 | ind 80..]; K blocks = sigma_net rows (row 0 = log sigma -> no column,
        // row r -> geo r-1 -> column 16 + r - 1), then SH, then the individual code
        std::vector<std::vector<int>> kb;
        for (int blk = 0; blk < 5; ++blk) {
            std::vector<int> b(16, -1);
            for (int i = 0; i < 16; ++i) {
                const int r = blk * 16 + i;
                if (r >= 1 && r <= 64) b[i] = 16 + r - 1;
            }
            kb.push_back(b);
        }
        kb.push_back(kblock(0));
        kb.push_back(kblock(80, n_ind));
        kb.push_back(Z);
        pack_layer(w[7], 64, col_in, kb, 4, 4, x3, buf, frag_base(L_COL0));
    }
    pack_layer(w[8], 3, 64, H64, 1, 2, x3, buf, frag_base(L_COL1));
    bf16_t* d = nullptr;
    MF_HIP(hipMalloc(&d, buf.size() * sizeof(bf16_t)));
    MF_HIP(hipMemcpy(d, buf.data(), buf.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
    *dev_out = d;
    return MF_OK;
}

static void fused_args(FusedArgs& a, const bf16_t* packed, const float* const emb[3], const int* offsets, float log2_pls, int base_res, float bound, const float* xyzs,
                       const float* dirs, const float* enc_a, const float* ind, int n_ind, float eye, int has_eye, int M, float* sigmas, float* rgbs, float* amb_aud,
                       float* amb_eye, float* unc, const int* M_dev, float sigma_scale, const float* eye_dev, const float* deltas) {
    a.M_dev = M_dev; a.sigma_scale = sigma_scale; a.eye_dev = eye_dev; a.deltas = deltas;
    a.xyzs = xyzs; a.dirs = dirs; a.enc_a = enc_a; a.ind = ind; a.w = packed;
    for (int p = 0; p < 3; ++p) a.emb[p] = emb[p];
    for (int l = 0; l < NLEV; ++l) {
        const float scale = exp2f((float)l * log2_pls) * (float)base_res - 1.0f;      // gridencoder.cu:123-124
        a.scale[l] = scale;
        a.resolution[l] = (uint32_t)std::ceil(scale) + 1;
        a.offset[l] = (uint32_t)offsets[l];
        a.hashmap_size[l] = (uint32_t)(offsets[l + 1] - offsets[l]);
    }
    a.bound = bound; a.eye = eye; a.n_ind = n_ind; a.has_eye = has_eye; a.M = M;
    a.ntiles = (M + TILE - 1) / TILE;
    a.sigmas = sigmas; a.rgbs = rgbs; a.amb_aud = amb_aud; a.amb_eye = amb_eye; a.unc = unc;
}

template <typename K>
static int fused_lds_attr(K kernel, bool& done, size_t lds) {
    if (!done) {
        MF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        done = true;
    }
    return MF_OK;
}

int mf_nerf_fused_launch(const bf16_t* packed, bool x3, const float* const emb[3], const int* offsets, float log2_pls, int base_res, float bound,
                         const float* xyzs, const float* dirs, const float* enc_a, const float* ind, int n_ind, float eye, int has_eye, int M,
                         float* sigmas, float* rgbs, float* amb_aud, float* amb_eye, float* unc, hipStream_t s, const int* M_dev, float sigma_scale, const float* eye_dev,
                         const float* deltas) {
    FusedArgs a{};
    fused_args(a, packed, emb, offsets, log2_pls, base_res, bound, xyzs, dirs, enc_a, ind, n_ind, eye, has_eye, M, sigmas, rgbs, amb_aud, amb_eye, unc, M_dev, sigma_scale, eye_dev,
               deltas);
    const size_t lds = (size_t)NFRAG * (x3 ? 2 : 1) * 1024;
    static bool attr_done[2] = {false, false};
    int rc;
    if ((rc = x3 ? fused_lds_attr(k_nerf_field_fused<true>, attr_done[1], lds) : fused_lds_attr(k_nerf_field_fused<false>, attr_done[0], lds))) return rc;
    const int grid = std::min(a.ntiles, 256);
    if (x3) hipLaunchKernelGGL(k_nerf_field_fused<true>, dim3(grid), dim3(NWAVE * 64), lds, s, a);
    else hipLaunchKernelGGL(k_nerf_field_fused<false>, dim3(grid), dim3(NWAVE * 64), lds, s, a);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

// the render loop's tail (k_loop_tail): the rounds after the `rounds_launched` that mf_nerf_head_render enqueued as launches, in one launch
int mf_nerf_tail_launch(const bf16_t* packed, bool x3, const float* const emb[3], const int* offsets, float log2_pls, int base_res, float bound, const float* enc_a,
                        const float* ind, int n_ind, float eye, int has_eye, float sigma_scale, const float* eye_dev, float* sigmas, float* rgbs, float* amb_aud,
                        float* amb_eye, float* unc, int* ctl, int N, int max_steps, int rounds_launched, float T_thresh, float dt_gamma, uint32_t cascades,
                        uint32_t grid_size, int* alive0, int* alive1, float* rays_t, const float* rays_o, const float* rays_d, const float* fars,
                        const uint8_t* bitfield, float* xyzs, float* dirs, float* deltas, float* wsum, float* depth, float* image, float* aasum, float* aesum,
                        float* unsum, hipStream_t s) {
    FusedArgs a{};
    const char* skip = getenv("MF_NERF_SKIP_EMPTY");
    fused_args(a, packed, emb, offsets, log2_pls, base_res, bound, xyzs, dirs, enc_a, ind, n_ind, eye, has_eye, N, sigmas, rgbs, amb_aud, amb_eye, unc, nullptr, sigma_scale, eye_dev,
               skip && skip[0] == '0' ? nullptr : deltas);
    TailArgs t{};
    t.ctl = ctl; t.N = N; t.max_steps = max_steps; t.first_parity = rounds_launched & 1; t.T_thresh = T_thresh; t.dt_gamma = dt_gamma; t.C = cascades; t.H = grid_size;
    t.alive0 = alive0; t.alive1 = alive1; t.rays_t = rays_t; t.rays_o = rays_o; t.rays_d = rays_d; t.fars = fars; t.grid = bitfield;
    t.xyzs = xyzs; t.dirs = dirs; t.deltas = deltas; t.wsum = wsum; t.depth = depth; t.image = image; t.aasum = aasum; t.aesum = aesum; t.unsum = unsum;
    const size_t lds = (size_t)NFRAG * (x3 ? 2 : 1) * 1024;
    static bool attr_done[2] = {false, false};
    int rc;
    if ((rc = x3 ? fused_lds_attr(k_loop_tail<true>, attr_done[1], lds) : fused_lds_attr(k_loop_tail<false>, attr_done[0], lds))) return rc;
    // workgroups: 64 (one per 512 rays if the frame has fewer).  When the loop has ended -- the usual case -- each of them reads three words and leaves, and the
    // launch costs what its waves cost to start: 4.9 us with 256 workgroups of 16 waves, a quarter of that with 64; when it has not, the late rounds the tail is
    // there for have a few ten thousand rays at most (a 512 x 512 frame's fifth round: 30 k)
    const char* e = getenv("MF_NERF_TAIL_WGS");
    const int grid = std::min(std::max(1, (N + TAIL_RC - 1) / TAIL_RC), e && atoi(e) > 0 ? std::min(atoi(e), 256) : 64);
    if (x3) hipLaunchKernelGGL(k_loop_tail<true>, dim3(grid), dim3(TAIL_NW * 64), lds, s, a, t);
    else hipLaunchKernelGGL(k_loop_tail<false>, dim3(grid), dim3(TAIL_NW * 64), lds, s, a, t);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

// ER-NeRF torso branch as ONE kernel on gfx950: `run_torso` (reference: ernerf/nerf_triplane/renderer.py:294-352) over
// `forward_torso` (network.py:166-201) -- frequency encoding of the pixel, the deform MLP, the warp, the 16-level tiled 2-D grid
// encoder, the colour / alpha MLP, the occupancy mask and the background mix -- one lane per pixel, nothing but the frame in HBM.
//
// Why not MFMA: the two MLPs are 32 wide (5 440 multiply-adds per pixel, 1.4 GMAC per 512 x 512 frame).  As six GEMM launches
// (mf_nerf_net.hip, kept behind MF_TORSO=gemm) every layer moved a [N, 32..72] (hi, lo) tensor out and back and the chain of
// twelve launches took ~0.42 ms of a 1.0 ms frame; the arithmetic itself is ~25 us of fp32 FMAs.  Here a lane keeps its pixel's
// 34 frequency features and 32 grid features in registers and parks each 32-wide activation in its own column of a 32 KB LDS slab (no
// barrier: a lane reads only what it wrote); the weights are wave-uniform, so they reach v_fmac as SGPR operands (s_load of one weight
// row at a time from the 21 KB table: scalar cache / L2).
// Arithmetic is plain fp32 (one fused multiply-add per weight; even and odd inputs of a row in two partial sums) in BOTH precision modes: closer to the reference's fp32
// nets than the bf16x3 GEMMs were.  Everything outside the dense layers rounds as written (contraction off), so the frequency
// features, the warp and the grid features are bit-identical to mf_freq_encode_forward / mf_grid_encode_forward.
#include "mf_common.h"
#include "mf_nerf_grid.h"
#include <cmath>

#pragma clang fp contract(off)

namespace {

constexpr int FQ = 34, TG = 32, HID = 32, NLEV = 16;
// row-major [out][in] fp32 tables, concatenated: deform net 34 -> 32 -> 32 -> 2, torso net (32 + 34) -> 32 -> 32 -> 4
constexpr int W_D1 = 0, W_D2 = W_D1 + HID * FQ, W_D3 = W_D2 + HID * HID, W_T1 = W_D3 + 2 * HID, W_T2 = W_T1 + HID * (TG + FQ),
              W_T3 = W_T2 + HID * HID, W_TOTAL = W_T3 + 4 * HID;

struct TorsoFusedArgs {
    const float *bg_coords, *w, *emb, *density, *bg;
    float bias_d[HID], bias_t[HID];               // per-frame: W[:, constant inputs] . [freq(wrapped anchors) | individual code]
    float scale[NLEV];
    uint32_t resolution[NLEV], offset[NLEV], hashmap_size[NLEV];
    float shrink, thresh, bg_const;
    int bg_per_ray, G, N;
    float *out, *alpha_out, *deform;
};

// the frequency features' sine is the reference's own: freqencoder.cu:56 calls __sinf -- the hardware sine of x / 2 pi -- and so does k_freq_encode (mf_nerf.hip);
// rounds 3 - 4 called the accurate sinf() here, 32 range reductions of ~50 instructions per pixel: a quarter of the kernel's VALU work for a feature the
// reference does not compute that way
__device__ __forceinline__ float torso_sin(float x) { return __sinf(x); }

// The weight table is read through the CONSTANT address space: a uniform load from it is always a scalar load (s_load -> SGPR operand of
// v_pk_fma), whatever stores the kernel has issued before -- through a global pointer the compiler falls back to per-lane
// global_load_dwordx4 as soon as it cannot prove the table unclobbered (the LDS slab writes below are enough).
typedef const __attribute__((address_space(4))) float* cw_ptr;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) f32x2* cw2_ptr;

// One output of a layer: the inputs are register PAIRS (x[2j], x[2j+1]), the weight row arrives as aligned SGPR pairs straight from
// s_load, and v_pk_fma_f32 accumulates the even and the odd inputs in the two halves of one register pair -- two multiply-adds per lane and
// instruction with no SGPR shuffling (packing two ROWS instead, which the vectoriser does on its own, costs 52 s_mov + 30 s_nop per 34
// packed instructions to build the pairs).  Every row of the table starts on an even index (all layer widths are even).
template <int IN2>
__device__ __forceinline__ float row_dot(cw_ptr wr, float bias, const f32x2 (&x)[IN2]) {
    cw2_ptr w2 = (cw2_ptr)wr;
    f32x2 acc = {bias, 0.f};
    // <= 17 weight pairs in flight: the 66-wide row goes in two halves (all 66 scalars at once overflow the ~100 SGPRs and spill into VGPR lanes)
#pragma unroll
    for (int c = 0; c < IN2; c += 17) {
#pragma unroll
        for (int j = c; j < (c + 17 < IN2 ? c + 17 : IN2); ++j) acc = __builtin_elementwise_fma(w2[j], x[j], acc);
        if (c + 17 < IN2) __builtin_amdgcn_sched_barrier(0);
    }
    return acc.x + acc.y;
}

// A layer: the inputs sit in registers, the OUTPUT ROW is a real loop -- one weight row is live at a time instead of the whole table (fully
// unrolled, the scheduler hoists every s_load to the top and spills 3 200 SGPRs; the same holds for the 2- and 4-output layers, whose
// loads otherwise migrate into the loop in front of them).  Output o goes to this lane's column of an LDS slab ([32][256] floats; a lane only
// ever touches its own column, so there is no barrier), from where the next stage reads it back with compile-time indices.
template <int IN2, int OUT, bool RELU, bool BIAS>
__device__ __forceinline__ void dense_to_lds(cw_ptr w, const float* __restrict__ bias, const f32x2 (&x)[IN2], float* col) {
    if constexpr (OUT % 2 == 0) {
        // two rows per iteration: their s_loads go out together, and two independent
        // accumulator chains interleave -- a lone chain of v_pk_fma_f32 pays a wait state after every instruction (the packed op's result is not forwarded)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int o = 0; o < OUT; o += 2) {
            cw2_ptr w0 = (cw2_ptr)(w + o * 2 * IN2), w1 = (cw2_ptr)(w + (o + 1) * 2 * IN2);
            f32x2 a0 = {BIAS ? bias[o] : 0.f, 0.f}, a1 = {BIAS ? bias[o + 1] : 0.f, 0.f};
            // <= 17 weight pairs of each row in flight (2 x 34 scalars): the 66-wide rows go in two halves -- all of a pair of them at once overflows the ~100 SGPRs
#pragma unroll
            for (int c = 0; c < IN2; c += 17) {
#pragma unroll
                for (int j = c; j < (c + 17 < IN2 ? c + 17 : IN2); ++j) {
                    a0 = __builtin_elementwise_fma(w0[j], x[j], a0);
                    a1 = __builtin_elementwise_fma(w1[j], x[j], a1);
                }
                if (c + 17 < IN2) __builtin_amdgcn_sched_barrier(0);
            }
            const float v0 = a0.x + a0.y, v1 = a1.x + a1.y;
            col[o * 256] = RELU ? fmaxf(v0, 0.f) : v0;
            col[(o + 1) * 256] = RELU ? fmaxf(v1, 0.f) : v1;
        }
    } else {
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int o = 0; o < OUT; ++o) {
            const float v = row_dot<IN2>(w + o * 2 * IN2, BIAS ? bias[o] : 0.f, x);
            col[o * 256] = RELU ? fmaxf(v, 0.f) : v;
        }
    }
}
__device__ __forceinline__ void from_lds(const float* col, f32x2 (&x)[HID / 2]) {
#pragma unroll
    for (int j = 0; j < HID / 2; ++j) x[j] = f32x2{col[2 * j * 256], col[(2 * j + 1) * 256]};
}

__global__ __launch_bounds__(256) void k_torso_fused(const TorsoFusedArgs a) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= a.N) return;
    const float2 bc = *reinterpret_cast<const float2*>(a.bg_coords + 2 * (size_t)n);
    // TH / TX: [grid 32 | x 2 | sin, cos of 2^f x, f = 0..7 (freqencoder.cu:30-58: sinf(scalbnf(x, f) + phase)) ]
    f32x2 in[(TG + FQ) / 2];                      // pairs 0..15: grid features, 16..32: frequency features
    float fq[FQ];
    fq[0] = bc.x * a.shrink;
    fq[1] = bc.y * a.shrink;
#pragma unroll
    for (int col = 0; col < 16; ++col)
#pragma unroll
        for (int d = 0; d < 2; ++d) fq[2 + 2 * col + d] = torso_sin(scalbnf(fq[d], col >> 1) + (float)(col & 1) * 1.57079632679489661923f);
#pragma unroll
    for (int j = 0; j < FQ / 2; ++j) in[TG / 2 + j] = f32x2{fq[2 * j], fq[2 * j + 1]};

    // deform net (network.py:177-185): [freq(x) | anchors | code] -> 32 -> 32 -> 2, the constant inputs folded into bias_d
    __shared__ float slab[HID * 256];
    float* col = slab + threadIdx.x;
    cw_ptr w = (cw_ptr)a.w;
    float dx[2];
    {
        f32x2 fqv[FQ / 2], h[HID / 2];
#pragma unroll
        for (int j = 0; j < FQ / 2; ++j) fqv[j] = in[TG / 2 + j];
        dense_to_lds<FQ / 2, HID, true, true>(w + W_D1, a.bias_d, fqv, col);
        from_lds(col, h);
        dense_to_lds<HID / 2, HID, true, false>(w + W_D2, nullptr, h, col);
        from_lds(col, h);
        dense_to_lds<HID / 2, 2, false, false>(w + W_D3, nullptr, h, col);
        dx[0] = col[0];
        dx[1] = col[256];
    }
    // x2 = clamp(x + dx, -1, 1) (network.py:187), mapped to [0, 1] for the tiled grid (grid.py:144, bound 1)
    float u[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float x2 = fminf(fmaxf(fq[k] + dx[k], -1.f), 1.f);
        u[k] = (x2 + 1.f) * 0.5f;
    }
    // torso_encoder: tiled grid, D = 2, C = 2, 16 levels, align_corners off (gridencoder.cu:76-165, as k_grid_encode<2, 2>)
    const bool oob = u[0] < 0 || u[0] > 1 || u[1] < 0 || u[1] > 1;
#pragma unroll
    for (int l = 0; l < NLEV; ++l) {
        float r0 = 0, r1 = 0;
        if (!oob) {
            const float* grid = a.emb + (size_t)a.offset[l] * 2;
            float pos[2];
            uint32_t pg[2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                pos[d] = u[d] * a.scale[l] + 0.5f;
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
            }
#pragma unroll
            for (uint32_t idx = 0; idx < 4; ++idx) {
                float w = 1;
                uint32_t pl[2];
#pragma unroll
                for (uint32_t d = 0; d < 2; ++d) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                // get_grid_index (gridencoder.cu:54-72) for D = 2, C = 2, the tiled type, with the level's constants -- wave-uniform kernel arguments -- deciding
                // on the SCALAR unit what the lanes must do: a level whose (res + 1)^2 cells fit the table is addressed directly (`% hashmap_size` is the
                // identity there: no instruction), a larger one wraps by a mask when the table is a power of two (2^log2_hashmap_size, grid.py:108-123) and by
                // the division only otherwise.  grid_index<2>() divided for all 64 corners of a pixel: ~1 600 of the kernel's VALU instructions.
                const uint32_t hs = a.hashmap_size[l], s1 = a.resolution[l] + 1;
                uint32_t index = pl[0];
                if (s1 <= hs) index += pl[1] * s1;
                if (!(s1 <= hs && s1 * s1 <= hs)) index = (hs & (hs - 1)) == 0 ? (index & (hs - 1)) : index % hs;
                index *= 2;
                const float2 g = *reinterpret_cast<const float2*>(grid + index);          // index is a multiple of C = 2: 8-byte aligned
                r0 += w * g.x;
                r1 += w * g.y;
            }
        }
        in[l] = f32x2{r0, r1};
    }
    // torso net (network.py:189-199): [grid | freq(x) | anchors | code] -> 32 -> 32 -> 4, sigmoid
    float o4[4];
    {
        f32x2 h[HID / 2];
        dense_to_lds<(TG + FQ) / 2, HID, true, true>(w + W_T1, a.bias_t, in, col);
        from_lds(col, h);
        dense_to_lds<HID / 2, HID, true, false>(w + W_T2, nullptr, h, col);
        from_lds(col, h);
        dense_to_lds<HID / 2, 4, false, false>(w + W_T3, nullptr, h, col);
#pragma unroll
        for (int k = 0; k < 4; ++k) o4[k] = col[k * 256];
    }
    // occupancy = grid_sample(density_grid_torso, bg_coords, align_corners=True) (renderer.py:326), mask = occupancy > thresh;
    // alpha / colour = sigmoid(.) * 1.002 - 0.001 (network.py:198-199); bg = colour * alpha + bg * (1 - alpha) (renderer.py:343)
    const int G = a.G;
    const float gx = (bc.x + 1.f) * 0.5f * (float)(G - 1), gy = (bc.y + 1.f) * 0.5f * (float)(G - 1);
    const float fx = floorf(gx), fy = floorf(gy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx = gx - fx, wy = gy - fy;
    auto at = [&](int y, int x) { return (x >= 0 && x < G && y >= 0 && y < G) ? a.density[y * G + x] : 0.f; };   // zeros padding
    const float occ = at(y0, x0) * (1 - wx) * (1 - wy) + at(y0, x0 + 1) * wx * (1 - wy) + at(y0 + 1, x0) * (1 - wx) * wy + at(y0 + 1, x0 + 1) * wx * wy;
    const bool m = occ > a.thresh;
    float sg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sg[k] = m ? (1.f / (1.f + __expf(-o4[k]))) * 1.002f - 0.001f : 0.f;
    // every store sits behind the last weight load: a load the compiler can prove unclobbered since kernel entry becomes a scalar load
    if (a.deform) *reinterpret_cast<float2*>(a.deform + 2 * (size_t)n) = make_float2(dx[0], dx[1]);
    if (a.alpha_out) a.alpha_out[n] = sg[0];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float b = a.bg ? (a.bg_per_ray ? a.bg[3 * (size_t)n + k] : a.bg[k]) : a.bg_const;
        a.out[3 * (size_t)n + k] = sg[1 + k] * sg[0] + b * (1.f - sg[0]);
    }
}

}  // namespace

int mf_nerf_torso_fused_weight_count() { return W_TOTAL; }

// w: device table laid out as W_D1 .. W_T3 above; bias_d / bias_t: host, 32 each; offsets_host: 17 level offsets
int mf_nerf_torso_fused_launch(const float* w, const float* bias_d, const float* bias_t, const float* emb, const int* offsets_host, float log2_pls,
                               int base_res, const float* density, int G, const float* bg_coords, float shrink, float thresh, const float* bg,
                               int bg_per_ray, float bg_const, int N, float* out, float* alpha_out, float* deform, hipStream_t s) {
    TorsoFusedArgs a{};
    a.bg_coords = bg_coords; a.w = w; a.emb = emb; a.density = density; a.bg = bg;
    for (int i = 0; i < HID; ++i) { a.bias_d[i] = bias_d[i]; a.bias_t[i] = bias_t[i]; }
    for (int l = 0; l < NLEV; ++l) {
        const float scale = exp2f((float)l * log2_pls) * (float)base_res - 1.0f;              // gridencoder.cu:123
        a.scale[l] = scale;
        a.resolution[l] = (uint32_t)std::ceil(scale) + 1;                                     // gridencoder.cu:124
        a.offset[l] = (uint32_t)offsets_host[l];
        a.hashmap_size[l] = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
    }
    a.shrink = shrink; a.thresh = thresh; a.bg_const = bg_const; a.bg_per_ray = bg_per_ray; a.G = G; a.N = N;
    a.out = out; a.alpha_out = alpha_out; a.deform = deform;
    hipLaunchKernelGGL(k_torso_fused, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, a);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

// ER-NeRF inference kernels for gfx950: ray/box intersection, occupancy-grid ray marching, tri-plane hash-grid encoder,
// spherical-harmonics and frequency encoders, front-to-back compositing.
//
// These replace the functions the reference's torch extensions export to its Python wrappers
// (`_raymarching_face`, `_gridencoder`, `_shencoder`, `_freqencoder`; reference: ernerf/raymarching/raymarching.py:44,393,666,
// ernerf/gridencoder/grid.py:49, ernerf/shencoder/sphere_harmonics.py:32, ernerf/freqencoder/freq.py:29), with the same
// argument order and the same in-place output conventions, so those wrappers run unchanged on top of them.
//
// All of this is HBM / latency-bound integer and fp32 work: one lane per ray or per sample, coalesced where the
// reference's layouts allow it, no MFMA.  The file is compiled with floating-point contraction OFF so that +,-,*,/ round
// exactly as written (the CPU restatement the parity tests compare against is built the same way); divisions are the
// correctly rounded default of hipcc.
#include "mf_common.h"
#include "mf_nerf_grid.h"
#include "mf_nerf_march.h"
#include <cstdlib>
#include <cstring>
#include <cfloat>
#include <cmath>

#pragma clang fp contract(off)

namespace {

constexpr int NT = 256;

// kernel_near_far_from_aabb, raymarching.cu:92-145
__global__ __launch_bounds__(NT) void k_near_far_from_aabb(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                           const float* __restrict__ aabb, uint32_t N, float min_near,
                                                           float* nears, float* fars) {
    const uint32_t n = threadIdx.x + blockIdx.x * NT;
    if (n >= N) return;
    float near, far;
    near_far_ray(rays_o, rays_d, aabb, n, min_near, near, far);
    nears[n] = near;
    fars[n] = far;
}

// kernel_march_rays, raymarching.cu:828-929.  One lane per alive ray; the 256 KB occupancy bitfield stays in L2 / the
// vector L1, the per-ray DDA diverges exactly as in the reference.
__global__ __launch_bounds__(NT) void k_march_rays(uint32_t n_alive, uint32_t n_step, const int* __restrict__ rays_alive,
                                                   const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                                   const float* __restrict__ rays_d, float bound, float dt_gamma,
                                                   uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                                                   const float* __restrict__ fars, float* xyzs, float* dirs, float* deltas,
                                                   const float* __restrict__ noises) {
    const uint32_t n = threadIdx.x + blockIdx.x * NT;
    if (n >= n_alive || n_step == 0) return;
    march_ray_ref(n, n_step, rays_alive[n], noises ? noises[n] : 0.f, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas, false);
}

// k_march_rays for the device-controlled loop: the same per-ray DDA (same float operations in the same order), but a lane only
// records the (t, dt) of its samples in LDS while it marches -- on gfx9 stores share vmcnt with the occupancy loads, so writing
// samples inside the loop puts a store round trip on every iteration of the dependent chain.  The block then writes its contiguous
// [256 rays x n_step] slab of xyzs / dirs / deltas cooperatively (coalesced; empty slots are zeroed as raymarching.py:383-385 does).
constexpr int LOOP_MAX_STEP = 8;                 // renderer.py:256 caps n_step at 8
// FAST: one cascade and a power-of-two grid.  Then level == 0 for every position (both mip_from_* clamp to [0, C - 1]), so the mip bound and
// its reciprocal leave the loop, and 0.5 * v * H is a power-of-two scaling -- exact in float as in the reference's double -- so the
// double round trip goes too.  Same bits, about a third fewer instructions on the dependent chain.
template <bool FAST>
__global__ __launch_bounds__(NT) void k_loop_march(const int* __restrict__ ctl, const int* __restrict__ rays_alive, const float* __restrict__ rays_t,
                                                   const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bound, float dt_gamma,
                                                   uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                                                   const float* __restrict__ fars, float* xyzs, float* dirs, float* deltas) {
    const uint32_t n_alive = (uint32_t)ctl[0], n_step = (uint32_t)ctl[1];
    const uint32_t base = blockIdx.x * NT;
    if (base >= n_alive || n_step == 0) return;
    __shared__ float s_t[LOOP_MAX_STEP][NT], s_dt[LOOP_MAX_STEP][NT], s_o[3][NT], s_d[3][NT];
    __shared__ int s_cnt[NT];
    // FAST: the Morton code's bit spreading as a 128-entry table (H <= 128: mf_march_rays' bound) -- the march is VALU-bound (a ray crossing the
    // volume evaluates ~100 positions of ~80 instructions), and 3 LDS reads replace 24 of them
    __shared__ uint32_t s_spread[128];
    if (FAST) {
        if (threadIdx.x < 128) s_spread[threadIdx.x] = expand_bits(threadIdx.x);
        __syncthreads();
    }
    const uint32_t n = threadIdx.x + base;
    uint32_t step = 0;
    if (n < n_alive) {
        const float SQRT3 = 1.7320508075688772f;
        const int index = rays_alive[n];
        const float* ro = rays_o + (size_t)index * 3;
        const float* rd = rays_d + (size_t)index * 3;
        const float ox = ro[0], oy = ro[1], oz = ro[2];
        const float dx = rd[0], dy = rd[1], dz = rd[2];
        s_o[0][threadIdx.x] = ox; s_o[1][threadIdx.x] = oy; s_o[2][threadIdx.x] = oz;
        s_d[0][threadIdx.x] = dx; s_d[1][threadIdx.x] = dy; s_d[2][threadIdx.x] = dz;
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        const float rH = 1 / (float)H;
        const float H3 = (float)(H * H * H);
        float t = rays_t[index];
        const float far = fars[index];
        const float dt_max = 2 * SQRT3 * (float)(1 << (C - 1)) / (float)H;
        const float dt_min = fminf(dt_max, 2 * SQRT3 / (float)max_steps);
        const float mip_bound0 = fminf(1.f, bound), mip_rbound0 = 1 / mip_bound0, halfH = 0.5f * (float)H;
        // voxel of the position at parameter tj: clamped position, cell, occupancy bit index, mip bound -- the reference's operations in its order
        auto cell = [&](float tj, float dtj, float& x, float& y, float& z, int& nx, int& ny, int& nz, uint32_t& gi, float& mip_bound) __attribute__((always_inline)) {
            x = clampf_(fmaf(tj, dx, ox), -bound, bound);
            y = clampf_(fmaf(tj, dy, oy), -bound, bound);
            z = clampf_(fmaf(tj, dz, oz), -bound, bound);
            if constexpr (FAST) {
                mip_bound = mip_bound0;
                nx = (int)clampf_(fmaf(x, mip_rbound0, 1.0f) * halfH, 0.0f, (float)(H - 1));
                ny = (int)clampf_(fmaf(y, mip_rbound0, 1.0f) * halfH, 0.0f, (float)(H - 1));
                nz = (int)clampf_(fmaf(z, mip_rbound0, 1.0f) * halfH, 0.0f, (float)(H - 1));
                gi = s_spread[nx] | (s_spread[ny] << 1) | (s_spread[nz] << 2);       // = morton3d(nx, ny, nz)
            } else {
                const int la = mip_from_pos(x, y, z, (float)C), lb = mip_from_dt(dtj, (float)H, (float)C);
                const int level = la > lb ? la : lb;
                mip_bound = fminf(scalbnf(1.f, level), bound);
                const float mip_rbound = 1 / mip_bound;
                nx = (int)clampf_((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
                ny = (int)clampf_((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
                nz = (int)clampf_((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
                gi = (uint32_t)fmaf((float)level, H3, (float)morton3d((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
            }
        };
        // The reference's loop (raymarching.cu:862-925) reads one occupancy bit per iteration and the next position depends on it: a ray
        // crossing empty space is a chain of ~100 dependent loads.  But every parameter it ever visits is a member of ONE sequence that does
        // not depend on occupancy -- t_{k+1} = t_k + clamp(t_k dt_gamma, dt_min, dt_max): an occupied cell advances by exactly that step, an
        // empty one by that step repeated until the cell's exit (`do t += ...; while (t < tt)`).  So the bits of the next MB members are fetched
        // together (MB independent loads, one round trip) and the serial walk over them is pure arithmetic: same float operations in the same
        // order, same samples, an eighth of the round trips.
        constexpr int MB = 8;
        float tt = -FLT_MAX;                         // exit parameter of the empty cell being skipped (the do-while above); none yet
        bool done = !(t < far && step < n_step);
        while (!done) {
            float ts[MB + 1], ds[MB], xs[MB], ys[MB], zs[MB], mbs[MB];
            int nxs[MB], nys[MB], nzs[MB];
            uint32_t occ = 0;
            ts[0] = t;
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                ds[j] = clampf_(ts[j] * dt_gamma, dt_min, dt_max);
                ts[j + 1] = ts[j] + ds[j];
                uint32_t gi;
                cell(ts[j], ds[j], xs[j], ys[j], zs[j], nxs[j], nys[j], nzs[j], gi, mbs[j]);
                occ |= (uint32_t)((grid[gi / 8] >> (gi % 8)) & 1) << j;
            }
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                if (done || ts[j] < tt) continue;                             // finished, or still inside the skipped cell
                if (!(ts[j] < far && step < n_step)) { done = true; continue; }   // the while condition, tested where the reference tests it
                if ((occ >> j) & 1) {
                    s_t[step][threadIdx.x] = ts[j]; s_dt[step][threadIdx.x] = ds[j];
                    step++;
                    tt = -FLT_MAX;
                } else {
                    const float tx = fmaf(mbs[j], fmaf(((float)nxs[j] + 0.5f + 0.5f * signf_(dx)) * rH, 2.0f, -1.0f), -xs[j]) * rdx;
                    const float ty = fmaf(mbs[j], fmaf(((float)nys[j] + 0.5f + 0.5f * signf_(dy)) * rH, 2.0f, -1.0f), -ys[j]) * rdy;
                    const float tz = fmaf(mbs[j], fmaf(((float)nzs[j] + 0.5f + 0.5f * signf_(dz)) * rH, 2.0f, -1.0f), -zs[j]) * rdz;
                    tt = ts[j] + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
                }
            }
            t = ts[MB];
        }
    }
    s_cnt[threadIdx.x] = (int)step;
    __syncthreads();
    const uint32_t rays_here = min((uint32_t)NT, n_alive - base);
    const uint32_t total = rays_here * n_step;
    const size_t slab = (size_t)base * n_step;
    for (uint32_t i = threadIdx.x; i < total; i += NT) {
        // (n_step is 1 ... 8 and wave-uniform: a division by a CONSTANT is a multiply and a shift, by a register ~25 instructions)
        uint32_t r;
        switch (n_step) {
            case 1: r = i; break;
            case 2: r = i >> 1; break;
            case 3: r = i / 3u; break;
            case 4: r = i >> 2; break;
            case 5: r = i / 5u; break;
            case 6: r = i / 6u; break;
            case 7: r = i / 7u; break;
            default: r = n_step == 8 ? i >> 3 : i / n_step; break;
        }
        const uint32_t k = i - r * n_step;
        float x = 0.f, y = 0.f, z = 0.f, dx = 0.f, dy = 0.f, dz = 0.f, dt = 0.f, t1 = 0.f;
        if ((int)k < s_cnt[r]) {
            const float t = s_t[k][r];
            dt = s_dt[k][r];
            dx = s_d[0][r]; dy = s_d[1][r]; dz = s_d[2][r];
            x = clampf_(fmaf(t, dx, s_o[0][r]), -bound, bound);
            y = clampf_(fmaf(t, dy, s_o[1][r]), -bound, bound);
            z = clampf_(fmaf(t, dz, s_o[2][r]), -bound, bound);
            t1 = t + dt;
        }
        float* px = xyzs + (slab + i) * 3;
        float* pd = dirs + (slab + i) * 3;
        float* pt = deltas + (slab + i) * 2;
        px[0] = x; px[1] = y; px[2] = z;
        pd[0] = dx; pd[1] = dy; pd[2] = dz;
        pt[0] = dt; pt[1] = t1;
    }
}

// kernel_composite_rays_triplane, raymarching.cu:2142-2249 (composite_ray: mf_nerf_march.h)
__global__ __launch_bounds__(NT) void k_composite_rays_triplane(uint32_t n_alive, uint32_t n_step, float T_thresh, int* rays_alive,
                                                                float* rays_t, const float* __restrict__ sigmas,
                                                                const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                                                const float* __restrict__ ambs_aud, const float* __restrict__ ambs_eye,
                                                                const float* __restrict__ uncertainties, float* weights_sum,
                                                                float* depth, float* image, float* amb_aud_sum, float* amb_eye_sum,
                                                                float* uncertainty_sum) {
    const uint32_t n = threadIdx.x + blockIdx.x * NT;
    if (n >= n_alive || n_step == 0) return;
    if (composite_ray(n, n_step, T_thresh, rays_alive[n], rays_t, sigmas, rgbs, deltas, ambs_aud, ambs_eye, uncertainties, weights_sum, depth, image,
                      amb_aud_sum, amb_eye_sum, uncertainty_sum))
        rays_alive[n] = -1;
}

// ---- grid encoder -----------------------------------------------------------------------------------------------------
// GridLevels, grid_index<D>: mf_nerf_grid.h

// kernel_grid forward, gridencoder.cu:76-165.  grid (ceil(B/256), L); out_lbc: [L][B][C] as the reference extension writes it
// (grid.py:42) -- out_blc != 0 writes [B][L*C] directly, the layout grid.py:52 permutes to.
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(NT) void k_grid_encode(const float* __restrict__ inputs, const float* __restrict__ embeddings,
                                                    float* __restrict__ outputs, uint32_t B, uint32_t L, GridLevels lv,
                                                    uint32_t gridtype, int align_corners, int out_blc) {
    const uint32_t b = blockIdx.x * NT + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const float* grid = embeddings + (size_t)lv.offset[level] * C;
    const float* in = inputs + (size_t)b * D;
    float* out = out_blc ? outputs + ((size_t)b * L + level) * C : outputs + ((size_t)level * B + b) * C;
    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = in[d];
        if (x[d] < 0 || x[d] > 1) oob = true;
    }
    if (oob) {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) out[ch] = 0;
        return;
    }
    const uint32_t hashmap_size = lv.hashmap_size[level], resolution = lv.resolution[level];
    const float scale = lv.scale[level];
    float pos[D];
    uint32_t pos_grid[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = x[d] * scale + (align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
    }
    float results[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) results[ch] = 0;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pos_grid[d]; }
            else { w *= pos[d]; pl[d] = pos_grid[d] + 1; }
        }
        const uint32_t index = grid_index<D>(C, gridtype, align_corners != 0, hashmap_size, resolution, pl);
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) results[ch] += w * grid[index + ch];
    }
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) out[ch] = results[ch];
}

// kernel_sh, shencoder.cu:28-110 (degree <= 4: the first 16 basis functions; the renderer uses degree 4)
__global__ __launch_bounds__(NT) void k_sh_encode(const float* __restrict__ inputs, float* outputs, uint32_t B, uint32_t degree) {
    const uint32_t b = threadIdx.x + blockIdx.x * NT;
    if (b >= B) return;
    const float x = inputs[3 * b], y = inputs[3 * b + 1], z = inputs[3 * b + 2];
    float* o = outputs + (size_t)b * degree * degree;
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = 0.28209479177387814f;
    if (degree <= 1) return;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    if (degree <= 2) return;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    if (degree <= 3) return;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

// kernel_freq, freqencoder.cu:30-58
__global__ __launch_bounds__(NT) void k_freq_encode(const float* __restrict__ inputs, uint32_t B, uint32_t D, uint32_t C, float* outputs) {
    const uint32_t t = threadIdx.x + blockIdx.x * NT;
    if (t >= B * C) return;
    const uint32_t b = t / C, c = t - b * C;
    const float* in = inputs + (size_t)b * D;
    if (c < D) {
        outputs[t] = in[c];
    } else {
        const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
        const float phase_shift = (float)(col % 2) * (3.14159265358979323846f / 2);
        outputs[t] = __sinf(scalbnf(in[d], (int)freq) + phase_shift);
    }
}

// tail of run_cuda (renderer.py:275-280) + the uint8 conversion of nerfreal.py:111: image = clamp(image + (1 - w) * bg, 0, 1),
// depth = clamp(depth - near, 0) / (far - near); frame = uint8(image * 255) (numpy astype truncates)
__global__ __launch_bounds__(NT) void k_nerf_finish(float* image, float* depth, const float* __restrict__ weights_sum,
                                                    const float* __restrict__ nears, const float* __restrict__ fars,
                                                    const float* __restrict__ bg, int bg_per_ray, float bg_const, uint32_t N, uint8_t* frame) {
    const uint32_t n = threadIdx.x + blockIdx.x * NT;
    if (n >= N) return;
    const float t = 1 - weights_sum[n];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float b = bg ? (bg_per_ray ? bg[3 * n + k] : bg[k]) : bg_const;
        const float v = fminf(fmaxf(image[3 * n + k] + t * b, 0.f), 1.f);
        image[3 * n + k] = v;
        if (frame) frame[3 * n + k] = (uint8_t)(v * 255.f);
    }
    depth[n] = fmaxf(depth[n] - nears[n], 0.f) / (fars[n] - nears[n]);
}

// a24, utils.py:1208-1216 + nerfreal.py:111: the rendered [h, w] frame resized to the GUI's [H, W] -- image bilinear with half-pixel centres
// (F.interpolate(mode='bilinear'), align_corners=False: src = max((dst + 0.5) * in / out - 0.5, 0), weights (1 - l, l), the row blend of
// two column blends as aten's upsample_bilinear2d forms it), depth nearest (src = min(floor(dst * in / out), in - 1)), frame =
// uint8(image * 255).  One lane per output pixel.
__global__ __launch_bounds__(NT) void k_nerf_resize(const float* __restrict__ image, const float* __restrict__ depth, int h, int w, int H, int W,
                                                    float* out_image, float* out_depth, uint8_t* frame) {
    const int i = threadIdx.x + blockIdx.x * NT;
    if (i >= H * W) return;
    const int oy = i / W, ox = i - oy * W;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    float fy = fmaf(sy, (float)oy + 0.5f, -0.5f), fx = fmaf(sx, (float)ox + 0.5f, -0.5f);     // fused, as aten's builds contract it
    fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float p00 = image[((size_t)y0 * w + x0) * 3 + k], p01 = image[((size_t)y0 * w + x1) * 3 + k];
        const float p10 = image[((size_t)y1 * w + x0) * 3 + k], p11 = image[((size_t)y1 * w + x1) * 3 + k];
        const float v = ly0 * (lx0 * p00 + lx1 * p01) + ly1 * (lx0 * p10 + lx1 * p11);
        if (out_image) out_image[(size_t)i * 3 + k] = v;
        if (frame) frame[(size_t)i * 3 + k] = (uint8_t)(v * 255.f);
    }
    if (out_depth) {
        int ny = (int)floorf((float)oy * sy), nx = (int)floorf((float)ox * sx);
        ny = ny < h - 1 ? ny : h - 1; nx = nx < w - 1 ? nx : w - 1;
        out_depth[i] = depth[(size_t)ny * w + nx];
    }
}

// ---- device-controlled render loop (no host sync between rounds) ---------------------------------------------------------
// control block `ctl` and the head of a round (loop_next_round): mf_nerf_march.h
// rays_o != null: near / far of every ray (k_near_far_from_aabb's arithmetic) are computed and stored here too -- one launch less at the head of a frame
__global__ void k_loop_init(int* ctl, int N, int max_steps, int* alive, float* rays_t, float* nears, float* weights_sum, float* depth,
                            float* image, float* amb_aud_sum, float* amb_eye_sum, float* unc_sum, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                            const float* __restrict__ aabb, float min_near, float* fars) {
    const int n = blockIdx.x * NT + threadIdx.x;
    if (n == 0) { ctl[4] = 0; ctl[5] = 0; ctl[LOOP_CTL_ROUNDS] = 0; ctl[LOOP_CTL_ERR] = 0; loop_next_round(ctl, N, 0, N, max_steps); }
    // the tail kernel's tickets (take | finished | survivors | ready, one entry per round: mf_nerf_fused.hip k_loop_tail)
    for (int i = n; i < 4 * (max_steps + 1); i += (int)gridDim.x * NT) ctl[LOOP_CTL_TAIL + i] = 0;
    if (n >= N) return;
    alive[n] = n;                                   // renderer.py:242
    float near = 0.f;
    if (rays_o) {
        float far;
        near_far_ray(rays_o, rays_d, aabb, (uint32_t)n, min_near, near, far);
        nears[n] = near;
        fars[n] = far;
    } else near = nears[n];
    rays_t[n] = near;                               // renderer.py:243
    weights_sum[n] = depth[n] = amb_aud_sum[n] = amb_eye_sum[n] = unc_sum[n] = 0.f;
    image[3 * n] = image[3 * n + 1] = image[3 * n + 2] = 0.f;
}
// tail of a round in one launch: composite (raymarching.cu:2142-2249), `rays_alive = rays_alive[rays_alive >= 0]` (renderer.py:266) as a
// wave-aggregated append (the order of the survivors is not kept: rays are independent, only their slot changes), and -- by the last
// block to finish -- the head of the next round.
#ifndef MF_NERF_CT
#define MF_NERF_CT 512
#endif
constexpr int CT = MF_NERF_CT;           // composite block: one contended 64-bit atomic per 512 rays (with the fence in front of that atomic 1 024 measured best; without it
                                         // 512: 0.487 -> 0.481 ms per frame, 256: 0.485)
__global__ __launch_bounds__(CT) void k_loop_composite(int* ctl, int N, int max_steps, float T_thresh, const int* __restrict__ alive_in, int* alive_out,
                                                       float* rays_t, const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                       const float* __restrict__ deltas, const float* __restrict__ ambs_aud,
                                                       const float* __restrict__ ambs_eye, const float* __restrict__ uncertainties, float* weights_sum,
                                                       float* depth, float* image, float* amb_aud_sum, float* amb_eye_sum, float* uncertainty_sum) {
    const uint32_t n_alive = (uint32_t)ctl[0], n_step = (uint32_t)ctl[1];
    const uint32_t base = blockIdx.x * CT;
    if (n_step == 0 || base >= n_alive) return;     // ended loop: ctl stays as it is; blocks past the alive rays take no part
    const uint32_t n = threadIdx.x + base;
    int v = -1;
    bool keep = false;
    if (n < n_alive) {
        v = alive_in[n];
        // all of the ray's samples are fetched before the serial blend (the loop's early exits would otherwise put one load round trip
        // per sample on the dependent chain); the blend itself is composite_ray's, operation for operation
        float sg[LOOP_MAX_STEP], d0[LOOP_MAX_STEP], d1[LOOP_MAX_STEP], cr[LOOP_MAX_STEP], cg[LOOP_MAX_STEP], cb[LOOP_MAX_STEP], aa[LOOP_MAX_STEP],
            ae[LOOP_MAX_STEP], un[LOOP_MAX_STEP];
        const size_t o = (size_t)n * n_step;
        // (n_step is uniform: a scalar branch per k, so a round of n_step = 1 -- every ray alive -- issues 9 loads per ray, not 72 of which 63 re-read sample 0)
#pragma unroll
        for (uint32_t k = 0; k < LOOP_MAX_STEP; ++k) {
            if (k < n_step) {
                const size_t q = o + k;
                sg[k] = sigmas[q]; d0[k] = deltas[2 * q]; d1[k] = deltas[2 * q + 1];
                cr[k] = rgbs[3 * q]; cg[k] = rgbs[3 * q + 1]; cb[k] = rgbs[3 * q + 2];
                aa[k] = ambs_aud[q]; ae[k] = ambs_eye[q]; un[k] = uncertainties[q];
            } else {
                sg[k] = d0[k] = d1[k] = cr[k] = cg[k] = cb[k] = aa[k] = ae[k] = un[k] = 0.f;
            }
        }
        float t = rays_t[v];
        float weight_sum = weights_sum[v], d = depth[v];
        float r = image[3 * v], g = image[3 * v + 1], b = image[3 * v + 2];
        float a_aud = amb_aud_sum[v], a_eye = amb_eye_sum[v], u = uncertainty_sum[v];
        bool live = true;
#pragma unroll
        for (uint32_t k = 0; k < LOOP_MAX_STEP; ++k) {
            if (live && k < n_step) {
                if (d0[k] == 0) live = false;
                else {
                    const float alpha = 1.0f - __expf(-sg[k] * d0[k]);
                    const float T = 1 - weight_sum;
                    const float weight = alpha * mf_opaque(T);          // (alpha, T) share a register pair: see mf_opaque
                    weight_sum += weight;
                    t = d1[k];
                    d += weight * t;
                    r += weight * cr[k];
                    g += weight * cg[k];
                    b += weight * cb[k];
                    a_aud += aa[k];
                    a_eye += ae[k];
                    u += weight * un[k];
                    if (T < T_thresh) live = false;
                }
            }
        }
        keep = live;
        if (keep) rays_t[v] = t;
        // a ray whose first slot is empty (it missed, or ran out of the volume: seven of ten rays in round 1) blended nothing: its sums are what they were
        if (d0[0] != 0) {
            weights_sum[v] = weight_sum;
            depth[v] = d;
            image[3 * v] = r; image[3 * v + 1] = g; image[3 * v + 2] = b;
            amb_aud_sum[v] = a_aud;
            amb_eye_sum[v] = a_eye;
            uncertainty_sum[v] = u;
        }
    }
    // one atomic per block: ctl[5] counts survivors, ctl[6] blocks; packed in one 64-bit add so the last block also learns the total
    __shared__ int s_wave[CT / 64];
    __shared__ unsigned long long s_old;
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int before = 0, mine = 0;
#pragma unroll
    for (int w = 0; w < CT / 64; ++w) { before += w < wave ? s_wave[w] : 0; mine += s_wave[w]; }
    if (threadIdx.x == 0) {
        // (no fence in front of it: the block that arrives last reads nothing the others wrote but this counter -- an atomic at the device's coherence point -- and the
        // survivors and sums are for the NEXT launch.  An agent-scope fence here writes back the XCD's dirty L2 lines from every block: 0.483 -> 0.473 ms per frame without)
        s_old = atomicAdd(reinterpret_cast<unsigned long long*>(ctl + 6), ((unsigned long long)1 << 32) | (unsigned long long)mine);
    }
    __syncthreads();
    const unsigned long long old = s_old;
    const int slot = (int)(old & 0xffffffffu);
    if (keep) alive_out[slot + before + __popcll(m & ((1ull << lane) - 1))] = v;
    const int nblocks = (int)((n_alive + CT - 1) / CT);
    if (threadIdx.x == 0 && (int)(old >> 32) == nblocks - 1) loop_next_round(ctl, slot + mine, ctl[2], N, max_steps);
}

inline unsigned blocks(uint64_t n) { return (unsigned)((n + NT - 1) / NT); }

}  // namespace

// ---- C ABI -----------------------------------------------------------------------------------------------------------
extern "C" int mf_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t n_rays, float min_near,
                                     float* nears, float* fars, void* stream) {
    MF_REQUIRE(rays_o && rays_d && aabb && nears && fars, "near_far_from_aabb: null argument");
    if (n_rays == 0) return MF_OK;
    hipLaunchKernelGGL(k_near_far_from_aabb, dim3(blocks(n_rays)), dim3(NT), 0, (hipStream_t)stream, rays_o, rays_d, aabb, n_rays, min_near, nears, fars);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

extern "C" int mf_march_rays(uint32_t n_alive, uint32_t n_step, const int* rays_alive, const float* rays_t, const float* rays_o,
                             const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t cascades, uint32_t grid_size,
                             const uint8_t* density_bitfield, const float* nears, const float* fars, float* xyzs, float* dirs,
                             float* deltas, const float* noises, void* stream) {
    MF_REQUIRE(rays_alive && rays_t && rays_o && rays_d && density_bitfield && nears && fars && xyzs && dirs && deltas && noises,
               "march_rays: null argument");
    MF_REQUIRE(n_step > 0 && max_steps > 0 && cascades >= 1 && cascades <= 8 && grid_size >= 2 && grid_size <= 128,
               "march_rays: n_step=%u max_steps=%u cascades=%u grid=%u", n_step, max_steps, cascades, grid_size);
    if (n_alive == 0) return MF_OK;
    hipLaunchKernelGGL(k_march_rays, dim3(blocks(n_alive)), dim3(NT), 0, (hipStream_t)stream, n_alive, n_step, rays_alive, rays_t, rays_o,
                       rays_d, bound, dt_gamma, max_steps, cascades, grid_size, density_bitfield, fars, xyzs, dirs, deltas, noises);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

extern "C" int mf_composite_rays_triplane(uint32_t n_alive, uint32_t n_step, float T_thresh, int* rays_alive, float* rays_t,
                                          const float* sigmas, const float* rgbs, const float* deltas, const float* ambs_aud,
                                          const float* ambs_eye, const float* uncertainties, float* weights_sum, float* depth,
                                          float* image, float* amb_aud_sum, float* amb_eye_sum, float* uncertainty_sum, void* stream) {
    MF_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && ambs_aud && ambs_eye && uncertainties && weights_sum && depth && image &&
               amb_aud_sum && amb_eye_sum && uncertainty_sum, "composite_rays_triplane: null argument");
    MF_REQUIRE(n_step > 0, "composite_rays_triplane: n_step must be positive");
    if (n_alive == 0) return MF_OK;
    hipLaunchKernelGGL(k_composite_rays_triplane, dim3(blocks(n_alive)), dim3(NT), 0, (hipStream_t)stream, n_alive, n_step, T_thresh,
                       rays_alive, rays_t, sigmas, rgbs, deltas, ambs_aud, ambs_eye, uncertainties, weights_sum, depth, image, amb_aud_sum,
                       amb_eye_sum, uncertainty_sum);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

extern "C" int mf_grid_encode_forward(const float* inputs, const float* embeddings, const int* offsets_host, float* outputs, uint32_t B,
                                      uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                      int out_blc, void* stream) {
    MF_REQUIRE(inputs && embeddings && offsets_host && outputs, "grid_encode_forward: null argument");
    MF_REQUIRE(L >= 1 && L <= GRID_MAX_L, "grid_encode_forward: %u levels (1..%d supported)", L, GRID_MAX_L);
    MF_REQUIRE(gridtype <= 1, "grid_encode_forward: gridtype %u (0 hash, 1 tiled)", gridtype);
    if (B == 0) return MF_OK;
    GridLevels lv{};
    for (uint32_t l = 0; l < L; ++l) {
        MF_REQUIRE(offsets_host[l + 1] > offsets_host[l] && offsets_host[l] >= 0, "grid_encode_forward: offsets must increase");
        const float scale = exp2f((float)l * S) * (float)H - 1.0f;              // gridencoder.cu:123
        lv.scale[l] = scale;
        lv.resolution[l] = (uint32_t)std::ceil(scale) + 1;                      // gridencoder.cu:124
        lv.offset[l] = (uint32_t)offsets_host[l];
        lv.hashmap_size[l] = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
    }
    const dim3 grid(blocks(B), L), block(NT);
    hipStream_t s = (hipStream_t)stream;
#define MF_GCASE(DD, CC)                                                                                                   \
    if (D == DD && C == CC) {                                                                                              \
        hipLaunchKernelGGL((k_grid_encode<DD, CC>), grid, block, 0, s, inputs, embeddings, outputs, B, L, lv, gridtype,    \
                           align_corners, out_blc);                                                                        \
        MF_HIP(hipGetLastError());                                                                                         \
        return MF_OK;                                                                                                      \
    }
    MF_GCASE(2, 1) MF_GCASE(2, 2) MF_GCASE(2, 4) MF_GCASE(2, 8)
    MF_GCASE(3, 1) MF_GCASE(3, 2) MF_GCASE(3, 4) MF_GCASE(3, 8)
#undef MF_GCASE
    mf_set_error("grid_encode_forward: D=%u C=%u not built (D in {2,3}, C in {1,2,4,8}: gridencoder.cu:283-306)", D, C);
    return MF_ERR_INVALID;
}

extern "C" int mf_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t degree, void* stream) {
    MF_REQUIRE(inputs && outputs, "sh_encode_forward: null argument");
    MF_REQUIRE(degree >= 1 && degree <= 4, "sh_encode_forward: degree %u (1..4 built; the renderer uses 4)", degree);
    if (B == 0) return MF_OK;
    hipLaunchKernelGGL(k_sh_encode, dim3(blocks(B)), dim3(NT), 0, (hipStream_t)stream, inputs, outputs, B, degree);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

extern "C" int mf_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t degree, uint32_t C, float* outputs, void* stream) {
    MF_REQUIRE(inputs && outputs, "freq_encode_forward: null argument");
    MF_REQUIRE(D >= 1 && C == D + D * degree * 2, "freq_encode_forward: output_dim %u != D + 2*D*degree (freq.py:62)", C);
    if ((uint64_t)B * C == 0) return MF_OK;
    hipLaunchKernelGGL(k_freq_encode, dim3(blocks((uint64_t)B * C)), dim3(NT), 0, (hipStream_t)stream, inputs, B, D, C, outputs);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

extern "C" int mf_nerf_finish(float* image, float* depth, const float* weights_sum, const float* nears, const float* fars, const float* bg_color,
                              int bg_per_ray, float bg_const, uint32_t n_rays, uint8_t* frame_u8, void* stream) {
    MF_REQUIRE(image && depth && weights_sum && nears && fars, "nerf_finish: null argument");
    if (n_rays == 0) return MF_OK;
    hipLaunchKernelGGL(k_nerf_finish, dim3(blocks(n_rays)), dim3(NT), 0, (hipStream_t)stream, image, depth, weights_sum, nears, fars, bg_color,
                       bg_per_ray, bg_const, n_rays, frame_u8);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

extern "C" int mf_nerf_resize_frame(const float* image, const float* depth, int h, int w, int H, int W, float* out_image, float* out_depth,
                                    uint8_t* frame_u8, void* stream) {
    MF_REQUIRE(image && h > 0 && w > 0 && H > 0 && W > 0 && (out_image || frame_u8 || out_depth), "nerf_resize_frame: bad argument");
    MF_REQUIRE(!out_depth || depth, "nerf_resize_frame: out_depth without depth");
    MF_REQUIRE((int64_t)H * W < (1ll << 31) && (int64_t)h * w < (1ll << 31), "nerf_resize_frame: frame too large");
    hipLaunchKernelGGL(k_nerf_resize, dim3(blocks((uint64_t)H * W)), dim3(NT), 0, (hipStream_t)stream, image, depth, h, w, H, W, out_image, out_depth, frame_u8);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

// ---- launchers of the device-controlled loop pieces (used by mf_nerf_head_render, mf_nerf_net.hip) -----------------------------
int mf_nerf_loop_init(int* ctl, int N, int max_steps, int* alive, float* rays_t, float* nears, float* weights_sum, float* depth, float* image,
                      float* amb_aud_sum, float* amb_eye_sum, float* unc_sum, hipStream_t s, const float* rays_o, const float* rays_d, const float* aabb, float min_near,
                      float* fars) {
    hipLaunchKernelGGL(k_loop_init, dim3(blocks(N)), dim3(NT), 0, s, ctl, N, max_steps, alive, rays_t, nears, weights_sum, depth, image, amb_aud_sum, amb_eye_sum,
                       unc_sum, rays_o, rays_d, aabb, min_near, fars);
    MF_HIP(hipGetLastError());
    return MF_OK;
}
int mf_nerf_loop_round(int* ctl, int N, int max_steps, const int* alive_in, int* alive_out, float* rays_t, const float* rays_o, const float* rays_d,
                       float bound, float dt_gamma, uint32_t cascades, uint32_t grid_size, const uint8_t* bitfield, const float* fars,
                       float* xyzs, float* dirs, float* deltas, int phase, float T_thresh, const float* sigmas, const float* rgbs, const float* amb_aud,
                       const float* amb_eye, const float* unc, float* weights_sum, float* depth, float* image, float* amb_aud_sum, float* amb_eye_sum,
                       float* unc_sum, hipStream_t s) {
    if (phase == 0)            // march (the round's n_alive / n_step were set by k_loop_init or the previous round's tail)
    {
        const char* e = getenv("MF_NERF_MARCH");                                 // "generic" forces the reference-shaped index arithmetic (tests)
        const bool fast = cascades == 1 && (grid_size & (grid_size - 1)) == 0 && grid_size <= 128 && !(e && !strcmp(e, "generic"));   // 128: k_loop_march's spread table
        if (fast)
            hipLaunchKernelGGL(k_loop_march<true>, dim3(blocks(N)), dim3(NT), 0, s, (const int*)ctl, alive_in, rays_t, rays_o, rays_d, bound, dt_gamma,
                               (uint32_t)max_steps, cascades, grid_size, bitfield, fars, xyzs, dirs, deltas);
        else
            hipLaunchKernelGGL(k_loop_march<false>, dim3(blocks(N)), dim3(NT), 0, s, (const int*)ctl, alive_in, rays_t, rays_o, rays_d, bound, dt_gamma,
                               (uint32_t)max_steps, cascades, grid_size, bitfield, fars, xyzs, dirs, deltas);
    }
    else                       // composite + compaction + head of the next round
        hipLaunchKernelGGL(k_loop_composite, dim3((unsigned)((N + CT - 1) / CT)), dim3(CT), 0, s, ctl, N, max_steps, T_thresh, alive_in, alive_out, rays_t, sigmas, rgbs, deltas,
                           amb_aud, amb_eye, unc, weights_sum, depth, image, amb_aud_sum, amb_eye_sum, unc_sum);
    MF_HIP(hipGetLastError());
    return MF_OK;
}

// Per-ray pieces of the ER-NeRF render loop shared by the seam kernels (mf_nerf.hip) and the loop's tail kernel (mf_nerf_fused.hip):
// the occupancy-grid march of one ray (kernel_march_rays, raymarching.cu:828-929), the front-to-back blend of one ray's samples
// (kernel_composite_rays_triplane, raymarching.cu:2142-2249) and the head of a round (renderer.py:246-256).
//
// Every function body switches floating-point contraction OFF (the including file may compile with it on): +, -, *, / round exactly as
// written, and the places where the reference's build contracts are explicit fmaf()s (the C restatement the tests check against states the same
// operations; tests/test_ernerf_reference_kernels.py holds both to the reference's own kernels bit for bit).
#pragma once
#include "mf_common.h"
#include <cfloat>
#include <cmath>
#include <cstdint>

namespace {

__device__ __forceinline__ float signf_(float x) { return copysignf(1.0f, x); }
__device__ __forceinline__ float clampf_(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// raymarching.cu:42-54
__device__ __forceinline__ int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0.f, (float)exponent));
}
__device__ __forceinline__ int mip_from_dt(float dt, float H, float max_cascade) {
#pragma clang fp contract(off)
    const float mx = dt * H * 0.5f;
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0.f, (float)exponent));
}

// raymarching.cu:56-71
__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}

// kernel_near_far_from_aabb for one ray (raymarching.cu:92-145)
__device__ __forceinline__ void near_far_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ aabb, uint32_t n,
                                             float min_near, float& near_out, float& far_out) {
#pragma clang fp contract(off)
    const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
    const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx, t;
    if (near > far) { t = near; near = far; far = t; }
    float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
    if (near_y > far_y) { t = near_y; near_y = far_y; far_y = t; }
    if (near > far_y || near_y > far) { near_out = far_out = FLT_MAX; return; }
    if (near_y > near) near = near_y;
    if (far_y < far) far = far_y;
    float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
    if (near_z > far_z) { t = near_z; near_z = far_z; far_z = t; }
    if (near > far_z || near_z > far) { near_out = far_out = FLT_MAX; return; }
    if (near_z > near) near = near_z;
    if (far_z < far) far = far_z;
    if (near < min_near) near = min_near;
    near_out = near;
    far_out = far;
}

// kernel_march_rays for alive slot n (raymarching.cu:828-929): up to n_step samples into xyzs / dirs / deltas at slot n.  `zero_rest`: the
// slots the ray does not reach are zeroed here (the reference's caller hands over zero-filled tensors, raymarching.py:383-385).
__device__ __forceinline__ void march_ray_ref(uint32_t n, uint32_t n_step, int index, float noise, const float* __restrict__ rays_t,
                                              const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bound, float dt_gamma,
                                              uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                                              const float* __restrict__ fars, float* xyzs, float* dirs, float* deltas, bool zero_rest) {
#pragma clang fp contract(off)
    const float SQRT3 = 1.7320508075688772f;
    const float* ro = rays_o + (size_t)index * 3;
    const float* rd = rays_d + (size_t)index * 3;
    float* px = xyzs + (size_t)n * n_step * 3;
    float* pd = dirs + (size_t)n * n_step * 3;
    float* pt = deltas + (size_t)n * n_step * 2;
    const float ox = ro[0], oy = ro[1], oz = ro[2];
    const float dx = rd[0], dy = rd[1], dz = rd[2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    const float rH = 1 / (float)H;
    const float H3 = (float)(H * H * H);
    float t = rays_t[index];
    const float far = fars[index];
    const float dt_max = 2 * SQRT3 * (float)(1 << (C - 1)) / (float)H;
    const float dt_min = fminf(dt_max, 2 * SQRT3 / (float)max_steps);
    uint32_t step = 0;
    t = fmaf(clampf_(t * dt_gamma, dt_min, dt_max), noise, t);                    /* the reference build contracts this (and the lines marked fmaf below) */
    while (t < far && step < n_step) {
        const float x = clampf_(fmaf(t, dx, ox), -bound, bound);
        const float y = clampf_(fmaf(t, dy, oy), -bound, bound);
        const float z = clampf_(fmaf(t, dz, oz), -bound, bound);
        const float dt = clampf_(t * dt_gamma, dt_min, dt_max);
        const int la = mip_from_pos(x, y, z, (float)C), lb = mip_from_dt(dt, (float)H, (float)C);
        const int level = la > lb ? la : lb;
        const float mip_bound = fminf(scalbnf(1.f, level), bound);
        const float mip_rbound = 1 / mip_bound;
        // the reference forms this product in double (`0.5 * ...`), narrows to float in clamp() and truncates
        const int nx = (int)clampf_((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
        const int ny = (int)clampf_((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
        const int nz = (int)clampf_((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
        const uint32_t gi = (uint32_t)fmaf((float)level, H3, (float)morton3d((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
        const bool occ = grid[gi / 8] & (1 << (gi % 8));
        if (occ) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = dx; pd[1] = dy; pd[2] = dz;
            t += dt;
            pt[0] = dt; pt[1] = t;
            px += 3; pd += 3; pt += 2;
            step++;
        } else {
            const float tx = fmaf(mip_bound, fmaf(((float)nx + 0.5f + 0.5f * signf_(dx)) * rH, 2.0f, -1.0f), -x) * rdx;
            const float ty = fmaf(mip_bound, fmaf(((float)ny + 0.5f + 0.5f * signf_(dy)) * rH, 2.0f, -1.0f), -y) * rdy;
            const float tz = fmaf(mip_bound, fmaf(((float)nz + 0.5f + 0.5f * signf_(dz)) * rH, 2.0f, -1.0f), -z) * rdz;
            const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
            do {
                t += clampf_(t * dt_gamma, dt_min, dt_max);
            } while (t < tt);
        }
    }
    if (zero_rest)
        for (; step < n_step; ++step) {
            px[0] = px[1] = px[2] = 0.f; pd[0] = pd[1] = pd[2] = 0.f; pt[0] = pt[1] = 0.f;
            px += 3; pd += 3; pt += 2;
        }
}

// kernel_composite_rays_triplane for alive slot n (raymarching.cu:2142-2249); returns true when the ray ended inside this round
// (rays_alive[n] = -1 in the reference)
__device__ __forceinline__ bool composite_ray(uint32_t n, uint32_t n_step, float T_thresh, int index, float* rays_t, const float* __restrict__ sigmas,
                                              const float* __restrict__ rgbs, const float* __restrict__ deltas, const float* __restrict__ ambs_aud,
                                              const float* __restrict__ ambs_eye, const float* __restrict__ uncertainties, float* weights_sum, float* depth,
                                              float* image, float* amb_aud_sum, float* amb_eye_sum, float* uncertainty_sum) {
#pragma clang fp contract(off)
    const float* sg = sigmas + (size_t)n * n_step;
    const float* rg = rgbs + (size_t)n * n_step * 3;
    const float* dl = deltas + (size_t)n * n_step * 2;
    const float* aa = ambs_aud + (size_t)n * n_step;
    const float* ae = ambs_eye + (size_t)n * n_step;
    const float* un = uncertainties + (size_t)n * n_step;
    float t = rays_t[index];
    float weight_sum = weights_sum[index], d = depth[index];
    float r = image[3 * index], g = image[3 * index + 1], b = image[3 * index + 2];
    float a_aud = amb_aud_sum[index], a_eye = amb_eye_sum[index], u = uncertainty_sum[index];
    uint32_t step = 0;
    while (step < n_step) {
        if (dl[0] == 0) break;
        const float alpha = 1.0f - __expf(-sg[0] * dl[0]);
        const float T = 1 - weight_sum;
        const float weight = alpha * mf_opaque(T);          // (alpha, T) share a register pair: see mf_opaque
        weight_sum += weight;
        t = dl[1];
        d += weight * t;
        r += weight * rg[0];
        g += weight * rg[1];
        b += weight * rg[2];
        a_aud += aa[0];
        a_eye += ae[0];
        u += weight * un[0];
        if (T < T_thresh) break;
        sg++; rg += 3; dl += 2; step++; aa++; ae++; un++;
    }
    if (step >= n_step) rays_t[index] = t;
    weights_sum[index] = weight_sum;
    depth[index] = d;
    image[3 * index] = r; image[3 * index + 1] = g; image[3 * index + 2] = b;
    amb_aud_sum[index] = a_aud;
    amb_eye_sum[index] = a_eye;
    uncertainty_sum[index] = u;
    return step < n_step;
}

// ---- control block of the device-controlled render loop -------------------------------------------------------------------------------
// ctl (ints): [0] n_alive, [1] n_step, [2] step after this round, [3] M = n_alive * n_step of the round about to run (or running); [6..7] one 64-bit counter of
// the round's composite launch (survivors appended so far | blocks done); [8] rounds started in this frame, [9] error flag of the tail kernel;
// [10..11] a device-visible HOST pointer (or null): when the loop ends, the frame's round count is posted there -- the host reads it whenever it likes,
// without a sync, to decide how many rounds the NEXT frames enqueue as launches (mf_nerf_head_render); [16 ...] the tail kernel's per-round tickets.
constexpr int LOOP_CTL_ROUNDS = 8, LOOP_CTL_ERR = 9, LOOP_CTL_FB = 10, LOOP_CTL_TAIL = 16, LOOP_TAIL_ARRAYS = 7;
inline size_t loop_ctl_ints(int max_rounds) { return (size_t)LOOP_CTL_TAIL + (size_t)LOOP_TAIL_ARRAYS * (size_t)(max_rounds + 1); }

// head of a round: `while step < max_steps`, `n_alive <= 0 -> break`, n_step = max(min(N // n_alive, 8), 1) (renderer.py:246-256)
__device__ __forceinline__ int loop_n_step(int n_alive, int step, int N, int max_steps) {
    int n_step = 0;
    if (n_alive > 0 && step < max_steps) { n_step = N / n_alive; n_step = n_step < 8 ? n_step : 8; n_step = n_step > 1 ? n_step : 1; }
    return n_step;
}
// the loop has ended: post the frame's round count (and the tail's error flag) where the host can see it
__device__ __forceinline__ void loop_post_feedback(int* ctl) {
    int* fb = *reinterpret_cast<int**>(ctl + LOOP_CTL_FB);
    if (fb) {
        fb[1] = ctl[LOOP_CTL_ERR];
        fb[0] = ctl[LOOP_CTL_ROUNDS];
        // (no fence: the two words are pinned host memory -- the stores go out over the bus as they are -- and the host only ever wants a recent value; a
        // system-scope fence here would write back the L2's dirty lines once per frame, on the loop's critical path)
    }
}
__device__ __forceinline__ void loop_next_round(int* ctl, int n_alive, int step, int N, int max_steps) {
    const int n_step = loop_n_step(n_alive, step, N, max_steps);
    ctl[0] = n_step ? n_alive : 0; ctl[1] = n_step; ctl[2] = step + n_step; ctl[3] = n_step ? n_alive * n_step : 0;
    ctl[6] = 0; ctl[7] = 0;
    if (n_step) ctl[LOOP_CTL_ROUNDS] += 1;
    else loop_post_feedback(ctl);
}

}  // namespace

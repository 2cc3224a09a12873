/* ORACLE (test infrastructure, not product): plain-C restatement of the ER-NeRF inference kernels.
 *
 * PARITY UNPINNED: the reference kernels are CUDA (`ernerf/raymarching/src/raymarching.cu`, `gridencoder.cu`,
 * `shencoder.cu`, `freqencoder.cu`); nvcc is not in this image and there is no NVIDIA GPU, so they cannot be run to produce
 * golden vectors, and the reference ships no fixtures for them.  Each function below follows the cited
 * kernel statement by statement, one loop iteration per CUDA thread; what pins it are the hand-derived known-answer tests
 * in tests/test_ernerf.py (slab intersections, Morton codes, SH constants, grid interpolation of affine tables,
 * closed-form compositing) and, for march_rays, expectations derived from the DDA GEOMETRY alone -- which samples an
 * axis-aligned ray must emit through an occupied voxel slab, and where a two-cascade grid must switch levels (|x| = 1) --
 * which the HIP kernel is ALSO held to directly, without this file in the loop (test_hip_march_dda_geometry_kats): that
 * breaks the common mode of two transcriptions by one author.
 *
 * Built with -ffp-contract=off: the HIP kernels are built the same way, so float results agree bit for bit wherever
 * only +,-,*,/ and exact libm functions (frexpf, scalbnf, floorf, ceil) are involved; expf / sinf differ by an ulp or two
 * between libm and the device and are compared with a stated tolerance.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the shared object built from this file.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

/* raymarching.cu:30-36 */
static float signf_(float x) { return copysignf(1.0f, x); }
static float clampf_(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

/* raymarching.cu:42-54 */
static int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}
static int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = dt * H * 0.5f;
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

/* raymarching.cu:56-71 */
static uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
uint32_t ref_morton3d(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }

/* kernel_near_far_from_aabb, raymarching.cu:92-145 */
void ref_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                            float* nears, float* fars) {
    for (uint32_t n = 0; n < N; ++n) {
        const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
        const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx, t;
        if (near > far) { t = near; near = far; far = t; }
        float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) { t = near_y; near_y = far_y; far_y = t; }
        if (near > far_y || near_y > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { t = near_z; near_z = far_z; far_z = t; }
        if (near > far_z || near_z > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;
        if (near < min_near) near = min_near;
        nears[n] = near;
        fars[n] = far;
    }
}

/* kernel_march_rays, raymarching.cu:828-929.  xyzs / dirs / deltas must be zero-filled by the caller (raymarching.py:383-385). */
void ref_march_rays(uint32_t n_alive, uint32_t n_step, const int* rays_alive, const float* rays_t, const float* rays_o,
                    const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                    const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                    const float* noises) {
    const float SQRT3 = 1.7320508075688772f;
    for (uint32_t n = 0; n < n_alive; ++n) {
        const int index = rays_alive[n];
        const float noise = noises[n];
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
            const int a = mip_from_pos(x, y, z, (float)C), b = mip_from_dt(dt, (float)H, (float)C);
            const int level = a > b ? a : b;
            const float mip_bound = fminf(scalbnf(1, level), bound);
            const float mip_rbound = 1 / mip_bound;
            /* `0.5 * (x * mip_rbound + 1) * H` is a DOUBLE product in the reference (0.5 is a double literal), narrowed to
             * float by clamp()'s parameter and truncated to int */
            const int nx = (int)clampf_((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
            const int ny = (int)clampf_((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
            const int nz = (int)clampf_((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
            const uint32_t gi = (uint32_t)fmaf((float)level, H3, (float)ref_morton3d((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
            const int occ = grid[gi / 8] & (1 << (gi % 8));
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
    }
}

/* kernel_composite_rays_triplane, raymarching.cu:2142-2249 */
void ref_composite_rays_triplane(uint32_t n_alive, uint32_t n_step, float T_thresh, int* rays_alive, float* rays_t,
                                 const float* sigmas, const float* rgbs, const float* deltas, const float* ambs_aud,
                                 const float* ambs_eye, const float* uncertainties, float* weights_sum, float* depth,
                                 float* image, float* amb_aud_sum, float* amb_eye_sum, float* uncertainty_sum) {
    for (uint32_t n = 0; n < n_alive; ++n) {
        const int index = rays_alive[n];
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
            const float alpha = 1.0f - expf(-sg[0] * dl[0]);
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
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
        if (step < n_step) rays_alive[n] = -1;
        else rays_t[index] = t;
        weights_sum[index] = weight_sum;
        depth[index] = d;
        image[3 * index] = r; image[3 * index + 1] = g; image[3 * index + 2] = b;
        amb_aud_sum[index] = a_aud;
        amb_eye_sum[index] = a_eye;
        uncertainty_sum[index] = u;
    }
}

/* fast_hash / get_grid_index, gridencoder.cu:35-72 */
static uint32_t grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners, uint32_t hashmap_size,
                           uint32_t resolution, const uint32_t* pos_grid) {
    static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pos_grid[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) {
        index = 0;
        for (uint32_t d = 0; d < D; ++d) index ^= pos_grid[d] * primes[d];
    }
    return (index % hashmap_size) * C;
}

/* kernel_grid forward (dy_dx == NULL), gridencoder.cu:76-165.  outputs is [L][B][C] like the reference (grid.py:42). */
void ref_grid_encode_forward(const float* inputs, const float* embeddings, const int* offsets, float* outputs, uint32_t B,
                             uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners) {
    for (uint32_t level = 0; level < L; ++level) {
        const float* grid = embeddings + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float scale = exp2f((float)level * S) * (float)H - 1.0f;
        const uint32_t resolution = (uint32_t)ceil(scale) + 1;
        for (uint32_t b = 0; b < B; ++b) {
            const float* in = inputs + (size_t)b * D;
            float* out = outputs + ((size_t)level * B + b) * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) {
                for (uint32_t ch = 0; ch < C; ch++) out[ch] = 0;
                continue;
            }
            float pos[3];
            uint32_t pos_grid[3];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = in[d] * scale + (align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
            }
            float results[8] = {0};
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pl[3];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pl[d] = pos_grid[d] + 1; }
                }
                const uint32_t index = grid_index(D, C, gridtype, align_corners, hashmap_size, resolution, pl);
                for (uint32_t ch = 0; ch < C; ch++) results[ch] += w * grid[index + ch];
            }
            for (uint32_t ch = 0; ch < C; ch++) out[ch] = results[ch];
        }
    }
}

/* kernel_sh, shencoder.cu:28-110, degrees 1..4 (the renderer uses 4: network.py encoder_dir) */
void ref_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t degree) {
    const uint32_t C2 = degree * degree;
    for (uint32_t b = 0; b < B; ++b) {
        const float x = inputs[3 * b], y = inputs[3 * b + 1], z = inputs[3 * b + 2];
        float* o = outputs + (size_t)b * C2;
        const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
        o[0] = 0.28209479177387814f;
        if (degree <= 1) continue;
        o[1] = -0.48860251190291987f * y;
        o[2] = 0.48860251190291987f * z;
        o[3] = -0.48860251190291987f * x;
        if (degree <= 2) continue;
        o[4] = 1.0925484305920792f * xy;
        o[5] = -1.0925484305920792f * yz;
        o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
        o[7] = -1.0925484305920792f * xz;
        o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
        if (degree <= 3) continue;
        o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
        o[10] = 2.8906114426405538f * xy * z;
        o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
        o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
        o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
        o[14] = 1.4453057213202769f * z * (x2 - y2);
        o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    }
}

/* kernel_freq, freqencoder.cu:30-58 */
void ref_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs) {
    const float PI_2 = 3.14159265358979323846f / 2;
    for (uint32_t t = 0; t < B * C; ++t) {
        const uint32_t b = t / C, c = t - b * C;
        const float* in = inputs + (size_t)b * D;
        if (c < D) {
            outputs[t] = in[c];
        } else {
            const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
            const float phase_shift = (float)(col % 2) * PI_2;
            outputs[t] = sinf(scalbnf(in[d], (int)freq) + phase_shift);
        }
    }
    (void)deg;
}

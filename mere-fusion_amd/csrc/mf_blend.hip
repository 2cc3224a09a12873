// Paste-back of generated faces into the cached full frames, on the device: the uint8 frames never leave HBM
// between the generator and the final copy to the host.
//
//   Wav2Lip   lipreal.py:207-214     res_frame.astype(np.uint8) -> cv2.resize(bbox size) -> combine_frame[y1:y2, x1:x2] = res_frame
//   MuseTalk  musereal.py:238-247    cv2.resize(res_frame.astype(np.uint8), bbox size) -> get_image_blending(...)
//             musetalk/utils/blending.py:103-125   mask = BGR2GRAY(mask_array) / 255;  crop = blendLinear(face_large, crop, mask, 1 - mask)
//
// This is byte / integer work and HBM-bound (one read of the cached frame, one write of the composed frame: 2 x H x W x 3 bytes per
// frame); it is kept bit-exact with OpenCV's published 8-bit algorithms, restated integer for integer:
//   cv::resize INTER_LINEAR 8UC3  : 11-bit fixed-point coefficients, int32 horizontal pass, `>> 4 ... >> 16 ... + 2 >> 2` vertical pass; an exact
//                                   2 x 2 decimation takes INTER_AREA's fast path (resize.cpp)
//   cv::cvtColor BGR2GRAY 8U      : (B * 1868 + G * 9617 + R * 4899 + 8192) >> 14
//   cv::blendLinear 8UC3          : fp32 (s1 * w1 + s2 * w2) / (w1 + w2 + 1e-5f), round half to even, every product / sum rounded on its own
// Floating-point contraction is OFF in this file: an FMA would change the last bit of a coefficient or a blend.
#include "mf_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int MAX_JOBS = 32;

struct PasteArgs {
    const void* res;            // [n][Sh][Sw][3] uint8, or fp32 (Wav2Lip: pred * 255, truncated like astype(np.uint8))
    int res_is_f32, Sh, Sw;
    const uint8_t* frames;      // [n_frames][H][W][3]
    int H, W;
    uint8_t* out;               // [n_jobs][H][W][3]
    int job0;                   // index of job[0] within the call (res / out slot)
    mf_paste_job job[MAX_JOBS];
};

__device__ __forceinline__ int src_u8(const void* res, int is_f32, int64_t i) {
    if (!is_f32) return reinterpret_cast<const uint8_t*>(res)[i];
    float v = reinterpret_cast<const float*>(res)[i];
    v = fminf(fmaxf(v, 0.f), 255.f);
    return (int)v;                                       // astype(np.uint8): truncation (lipreal.py:211)
}

// One axis of cv::resize's linear tables (resize.cpp): source offset and the two 11-bit coefficients of destination index d.
// `clamp_offset` reproduces the horizontal pass (offset clamped, fraction zeroed); the vertical pass keeps the fraction and clamps rows.
__device__ __forceinline__ void axis(int d, int ssize, int dsize, bool clamp_offset, int& s0, int& s1, int& a0, int& a1) {
    const double inv_scale = (double)dsize / (double)ssize;
    const double scale = 1.0 / inv_scale;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (clamp_offset) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    }
    a0 = (int)rintf((1.f - f) * 2048.f);                 // saturate_cast<short>(cbuf * INTER_RESIZE_COEF_SCALE): cvRound = round half to even
    a1 = (int)rintf(f * 2048.f);
    s0 = min(max(s, 0), ssize - 1);
    s1 = min(max(s + 1, 0), ssize - 1);
}

// pixel (dy, dx), channels 0..2 of cv2.resize(res[slot], (dw, dh))
__device__ __forceinline__ void resized_px(const PasteArgs& a, int slot, int dw, int dh, int dx, int dy, int (&px)[3]) {
    const int64_t base = (int64_t)slot * a.Sh * a.Sw * 3;
    if (a.Sh == dh && a.Sw == dw) {                      // same size: cv::resize copies
#pragma unroll
        for (int c = 0; c < 3; ++c) px[c] = src_u8(a.res, a.res_is_f32, base + ((int64_t)dy * a.Sw + dx) * 3 + c);
        return;
    }
    if (a.Sw == 2 * dw && a.Sh == 2 * dh) {              // exact 2 x 2 decimation: INTER_AREA fast path
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t p = base + ((int64_t)(2 * dy) * a.Sw + 2 * dx) * 3 + c;
            px[c] = (src_u8(a.res, a.res_is_f32, p) + src_u8(a.res, a.res_is_f32, p + 3) + src_u8(a.res, a.res_is_f32, p + (int64_t)a.Sw * 3) +
                     src_u8(a.res, a.res_is_f32, p + (int64_t)a.Sw * 3 + 3) + 2) >> 2;
        }
        return;
    }
    int x0, x1, ax0, ax1, y0, y1, by0, by1;
    axis(dx, a.Sw, dw, true, x0, x1, ax0, ax1);
    axis(dy, a.Sh, dh, false, y0, y1, by0, by1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int S0 = src_u8(a.res, a.res_is_f32, base + ((int64_t)y0 * a.Sw + x0) * 3 + c) * ax0 + src_u8(a.res, a.res_is_f32, base + ((int64_t)y0 * a.Sw + x1) * 3 + c) * ax1;
        const int S1 = src_u8(a.res, a.res_is_f32, base + ((int64_t)y1 * a.Sw + x0) * 3 + c) * ax0 + src_u8(a.res, a.res_is_f32, base + ((int64_t)y1 * a.Sw + x1) * 3 + c) * ax1;
        const int v = (((by0 * (S0 >> 4)) >> 16) + ((by1 * (S1 >> 4)) >> 16) + 2) >> 2;
        px[c] = min(max(v, 0), 255);
    }
}

__device__ __forceinline__ int blend_px(int s1, int s2, float w1, float w2) {
    const float den = (w1 + w2) + 1e-5f;
    const float num = (float)s1 * w1 + (float)s2 * w2;   // (contraction off: two rounded products, one rounded sum)
    const float r = rintf(num / den);
    return (int)fminf(fmaxf(r, 0.f), 255.f);
}

// grid (pixel groups, jobs): a thread composes PXT consecutive pixels of one row of one output frame
constexpr int PXT = 4;
__global__ __launch_bounds__(256) void k_paste_frames(const PasteArgs a) {
    const mf_paste_job& j = a.job[blockIdx.y];
    const int slot = a.job0 + blockIdx.y;
    const int groups_x = (a.W + PXT - 1) / PXT;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= groups_x * a.H) return;
    const int y = g / groups_x, xb = (g - y * groups_x) * PXT;
    const uint8_t* ori = a.frames + ((int64_t)j.frame_index * a.H + y) * a.W * 3;
    uint8_t* dst = a.out + ((int64_t)slot * a.H + y) * a.W * 3;
    const int dw = j.x2 - j.x1, dh = j.y2 - j.y1;
    const bool blend = j.mask != nullptr;
    const int cw = j.cx2 - j.cx1;
    uint8_t o[PXT * 3];
    const int n = min(PXT, a.W - xb);
    if (n == PXT && (a.W & 3) == 0) {                   // 12 aligned bytes
        const uint32_t* p = reinterpret_cast<const uint32_t*>(ori + xb * 3);
        uint32_t w[3] = {p[0], p[1], p[2]};
        __builtin_memcpy(o, w, 12);
    } else {
        for (int i = 0; i < n * 3; ++i) o[i] = ori[xb * 3 + i];
    }
    const bool row_in_face = y >= j.y1 && y < j.y2;
    const bool row_in_crop = blend && y >= j.cy1 && y < j.cy2;
    if (row_in_face || row_in_crop) {
        for (int i = 0; i < n; ++i) {
            const int x = xb + i;
            const bool in_face = row_in_face && x >= j.x1 && x < j.x2;
            int s1[3] = {o[3 * i], o[3 * i + 1], o[3 * i + 2]};
            if (in_face) resized_px(a, slot, dw, dh, x - j.x1, y - j.y1, s1);
            if (!blend) {
                if (in_face) { o[3 * i] = (uint8_t)s1[0]; o[3 * i + 1] = (uint8_t)s1[1]; o[3 * i + 2] = (uint8_t)s1[2]; }
                continue;
            }
            if (!(row_in_crop && x >= j.cx1 && x < j.cx2)) continue;       // outside the crop box nothing changes (blending.py:121)
            const uint8_t* m = j.mask + ((int64_t)(y - j.cy1) * cw + (x - j.cx1)) * 3;
            const int gray = (m[0] * 1868 + m[1] * 9617 + m[2] * 4899 + (1 << 13)) >> 14;
            const float w1 = (float)((double)gray / 255.0);                   // (mask_image / 255).astype(np.float32), blending.py:111
            const float w2 = 1.f - w1;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[3 * i + c] = (uint8_t)blend_px(s1[c], o[3 * i + c], w1, w2);
        }
    }
    if (n == PXT && (a.W & 3) == 0) {
        uint32_t w[3];
        __builtin_memcpy(w, o, 12);
        uint32_t* p = reinterpret_cast<uint32_t*>(dst + xb * 3);
        p[0] = w[0]; p[1] = w[1]; p[2] = w[2];
    } else {
        for (int i = 0; i < n * 3; ++i) dst[xb * 3 + i] = o[i];
    }
}

}  // namespace

extern "C" int mf_paste_frames(const void* res, int res_is_f32, int res_h, int res_w, const uint8_t* frames, int n_frames, int H, int W,
                               const mf_paste_job* jobs, int n_jobs, uint8_t* out, void* stream) {
    MF_REQUIRE(res && frames && jobs && out, "paste_frames: null argument");
    MF_REQUIRE(res_h > 0 && res_w > 0 && H > 0 && W > 0 && n_frames > 0 && n_jobs > 0, "paste_frames: bad size");
    for (int i = 0; i < n_jobs; ++i) {
        const mf_paste_job& j = jobs[i];
        MF_REQUIRE(j.frame_index >= 0 && j.frame_index < n_frames, "paste_frames: job %d: frame index %d out of range (%d frames)", i, j.frame_index, n_frames);
        // an empty or out-of-frame bbox makes cv2.resize / the slice assignment raise in the reference (lipreal.py:210-213 skips the frame)
        MF_REQUIRE(j.x1 >= 0 && j.y1 >= 0 && j.x2 <= W && j.y2 <= H && j.x2 > j.x1 && j.y2 > j.y1,
                   "paste_frames: job %d: bbox (%d, %d, %d, %d) is empty or outside the %d x %d frame", i, j.x1, j.y1, j.x2, j.y2, W, H);
        if (j.mask)
            MF_REQUIRE(j.cx1 >= 0 && j.cy1 >= 0 && j.cx2 <= W && j.cy2 <= H && j.cx1 <= j.x1 && j.cy1 <= j.y1 && j.cx2 >= j.x2 && j.cy2 >= j.y2,
                       "paste_frames: job %d: crop box (%d, %d, %d, %d) must lie inside the frame and contain the bbox", i, j.cx1, j.cy1, j.cx2, j.cy2);
    }
    hipStream_t s = (hipStream_t)stream;
    const int groups = ((W + PXT - 1) / PXT) * H;
    for (int j0 = 0; j0 < n_jobs; j0 += MAX_JOBS) {
        PasteArgs a{};
        a.res = res; a.res_is_f32 = res_is_f32; a.Sh = res_h; a.Sw = res_w;
        a.frames = frames; a.H = H; a.W = W; a.out = out; a.job0 = j0;
        const int nj = n_jobs - j0 < MAX_JOBS ? n_jobs - j0 : MAX_JOBS;
        for (int i = 0; i < nj; ++i) a.job[i] = jobs[j0 + i];
        hipLaunchKernelGGL(k_paste_frames, dim3((groups + 255) / 256, nj), dim3(256), 0, s, a);
        MF_HIP(hipGetLastError());
    }
    return MF_OK;
}

// cv2.resize(src, (dw, dh)) alone: a single job over a frame that is exactly the destination (the bbox is the whole frame)
extern "C" int mf_resize_linear_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, void* stream) {
    MF_REQUIRE(src && dst && sh > 0 && sw > 0 && dh > 0 && dw > 0, "resize_linear_u8: bad argument");
    mf_paste_job j{};
    j.frame_index = 0; j.x1 = 0; j.y1 = 0; j.x2 = dw; j.y2 = dh; j.mask = nullptr;
    // `frames` is only read where the bbox does not cover: nowhere.  The destination doubles as the (never used) frame.
    return mf_paste_frames(src, 0, sh, sw, dst, 1, dh, dw, &j, 1, dst, stream);
}

"""ORACLE (test infrastructure, not product): numpy restatement of the per-frame paste-back that follows the generators
(SURVEY 8f rank 2).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

  Wav2Lip   lipreal.py:207-214        res_frame.astype(uint8) -> cv2.resize to the bbox -> rectangle copy into the full frame
  MuseTalk  musereal.py:238-247       cv2.resize(res_frame.astype(uint8), bbox size) -> get_image_blending
            musetalk/utils/blending.py:103-125   face_large = crop with the face pasted in; mask = BGR2GRAY / 255;
                                                 crop = cv2.blendLinear(face_large, crop, mask, 1 - mask)

PARITY UNPINNED.  The byte arithmetic lives in OpenCV (`opencv-python`, requirements.txt, un-vendored, unpinned, absent here: `import cv2`
fails).  What is restated is OpenCV 4.x's published algorithm for 8-bit images, integer for integer:
  * cv::resize INTER_LINEAR, 8UC3 (imgproc/src/resize.cpp): source position fx = (float)((dx + 0.5) * scale - 0.5) with
    scale = 1 / (dsize / ssize) in double; sx = floor(fx); clamped at both borders with fx = 0; coefficients
    cvRound((1 - fx) * 2048), cvRound(fx * 2048) as int16 (INTER_RESIZE_COEF_BITS = 11); horizontal pass in int32 without a shift;
    vertical pass `(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2` (VResizeLinear<uchar, int, short, FixedPtCast<..., 22>>);
    rows clamped to the image.  An exact 2 x 2 decimation is rerouted to INTER_AREA's fast path ((a + b + c + d + 2) >> 2).  The IPP linear
    path is skipped for 8-bit data unless `useIPP_NotExact`, so this IS the result of a stock build.
  * cv::cvtColor BGR2GRAY, 8U (color_rgb.simd.hpp RGB2Gray<uchar>): (B * 1868 + G * 9617 + R * 4899 + (1 << 13)) >> 14.
  * cv::blendLinear, 8UC3 (imgproc/src/blend.cpp): per channel, fp32: (s1 * w1 + s2 * w2) / (w1 + w2 + 1e-5f), saturate_cast<uchar> =
    round half to even; products and sums individually rounded (no FMA in the SSE2 / scalar baseline).
Pinned by hand-derived known-answer vectors (tests/test_blend.py): pure B / G / R -> 29 / 150 / 76, the [0, 100] -> [0, 25, 75, 100]
upscale, the 3 -> 2 downscale, the 2 x 2 decimation, blend weights 0 / 128 / 255."""
import numpy as np

COEF_BITS = 11
ONE = 1 << COEF_BITS


def _axis_tables(ssize, dsize):
    """(offset[dsize], a0[dsize], a1[dsize]) of cv::resize's linear interpolation along one axis (resize.cpp, `resize` -> `xofs / ialpha`).
    The vertical pass keeps an un-clamped offset and clamps the two ROWS instead; horizontally the offset itself is clamped.  Both give
    the same taps for the linear kernel, so one table serves both (offset in [0, ssize - 1], second tap = min(offset + 1, ssize - 1))."""
    inv_scale = np.float64(dsize) / np.float64(ssize)
    scale = np.float64(1.0) / inv_scale
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _coefs(f):
    a0 = np.rint((np.float32(1.0) - f).astype(np.float32) * np.float32(ONE)).astype(np.int64)    # cvRound: round half to even
    a1 = np.rint(f * np.float32(ONE)).astype(np.int64)
    return a0, a1


def resize_linear_u8(src, dw, dh):
    """cv2.resize(src, (dw, dh)) for uint8 [h, w, c] with the default INTER_LINEAR."""
    src = np.asarray(src)
    assert src.dtype == np.uint8 and src.ndim == 3
    sh, sw, _ = src.shape
    if sw == 2 * dw and sh == 2 * dh:                    # is_area_fast with iscale 2: INTER_LINEAR becomes INTER_AREA (resize.cpp)
        s = src.astype(np.int64)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, fx = _axis_tables(sw, dw)
    # horizontal clamps (resize.cpp: `if (sx < 0) fx = 0, sx = 0;  if (sx >= ssize.width - 1) fx = 0, sx = ssize.width - 1`)
    lo, hi = sx < 0, sx >= sw - 1
    fx = np.where(lo | hi, np.float32(0), fx).astype(np.float32)
    sx = np.where(lo, 0, np.where(hi, sw - 1, sx))
    ax0, ax1 = _coefs(fx)
    sy, fy = _axis_tables(sh, dh)                        # vertical: coefficients from the UN-clamped position, rows clamped
    by0, by1 = _coefs(fy)
    r0 = np.clip(sy, 0, sh - 1)
    r1 = np.clip(sy + 1, 0, sh - 1)
    s = src.astype(np.int64)
    x1 = np.minimum(sx + 1, sw - 1)
    rows = s[:, sx] * ax0[None, :, None] + s[:, x1] * ax1[None, :, None]                 # [sh, dw, c] int32 range
    S0, S1 = rows[r0], rows[r1]
    out = (((by0[:, None, None] * (S0 >> 4)) >> 16) + ((by1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def bgr2gray_u8(img):
    img = np.asarray(img).astype(np.int64)
    return ((img[..., 0] * 1868 + img[..., 1] * 9617 + img[..., 2] * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def blend_linear_u8(src1, src2, w1, w2):
    """cv2.blendLinear for uint8 [h, w, c] images and float32 [h, w] weights."""
    f = np.float32
    s1, s2 = np.asarray(src1).astype(f), np.asarray(src2).astype(f)
    w1, w2 = np.asarray(w1, f)[..., None], np.asarray(w2, f)[..., None]
    den = ((w1 + w2).astype(f) + f(1e-5)).astype(f)
    num = ((s1 * w1).astype(f) + (s2 * w2).astype(f)).astype(f)
    return np.clip(np.rint((num / den).astype(f)), 0, 255).astype(np.uint8)


def get_image_blending(image, face, face_box, mask_array, crop_box):
    """musetalk/utils/blending.py:103-125 (modifies and returns `image`, like the reference)."""
    body = image
    x, y, x1, y1 = face_box
    x_s, y_s, x_e, y_e = crop_box
    face_large = body[y_s:y_e, x_s:x_e].copy()
    face_large[y - y_s:y1 - y_s, x - x_s:x1 - x_s] = face
    mask_image = bgr2gray_u8(mask_array)
    mask_image = (mask_image / 255).astype(np.float32)
    body[y_s:y_e, x_s:x_e] = blend_linear_u8(face_large, body[y_s:y_e, x_s:x_e], mask_image, 1 - mask_image)
    return body


def muse_paste(ori_frame, res_frame_u8, bbox, mask, mask_crop_box):
    """musereal.py:238-247 for one frame: deepcopy, resize to the bbox, blend."""
    x1, y1, x2, y2 = bbox
    frame = np.array(ori_frame, copy=True)
    res = resize_linear_u8(np.asarray(res_frame_u8).astype(np.uint8), x2 - x1, y2 - y1)
    return get_image_blending(frame, res, bbox, mask, mask_crop_box)


def lip_paste(ori_frame, res_frame, bbox):
    """lipreal.py:207-214 for one frame; NOTE the bbox order of the Wav2Lip avatars: (y1, y2, x1, x2)."""
    y1, y2, x1, x2 = bbox
    frame = np.array(ori_frame, copy=True)
    res = resize_linear_u8(np.asarray(res_frame).astype(np.uint8), x2 - x1, y2 - y1)        # astype(uint8): truncation (lipreal.py:211)
    frame[y1:y2, x1:x2] = res
    return frame

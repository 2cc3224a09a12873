"""`musetalk.utils.blending` drop-in, hot-path subset (musetalk/utils/blending.py:8-14, 103-125).

`get_image_blending(image, face, face_box, mask_array, crop_box)` keeps the reference's signature and its in-place contract: numpy
arrays in, the modified `image` back.  The arithmetic (BGR2GRAY, / 255, cv2.blendLinear) runs in `mf_paste_frames` on the GPU, bit-exact
with OpenCV's 8-bit algorithms.  Called this way every frame crosses PCIe twice, like any other use of the GPU from process_frames; the
route that pays is `mere_fusion_amd.muse_driver.MuseBatcher(paste=...)`, which composes the frames before they ever leave HBM."""
import numpy as np
import torch

from ...paste import AvatarFrames


def get_crop_box(box, expand):
    """blending.py:8-14."""
    x, y, x1, y1 = box
    x_c, y_c = (x + x1) // 2, (y + y1) // 2
    w, h = x1 - x, y1 - y
    s = int(max(w, h) // 2 * expand)
    return [x_c - s, y_c - s, x_c + s, y_c + s], s


def get_image_blending(image, face, face_box, mask_array, crop_box):
    """blending.py:103-125.  `face` is the generator's frame already resized to the bbox (musereal.py:241); a face of another size is
    resized exactly as cv2.resize would."""
    if not torch.cuda.is_available():
        raise RuntimeError("get_image_blending needs a HIP device; no CPU path exists here")
    mask = np.asarray(mask_array)
    if mask.ndim == 2:
        raise RuntimeError("mask_array must be the 3-channel image cv2.imread returns (blending.py:110 converts it with COLOR_BGR2GRAY)")
    av = AvatarFrames(np.asarray(image)[None], [face_box], [mask], [crop_box])
    out = av.paste(torch.from_numpy(np.ascontiguousarray(face))[None].cuda(), [0])
    image[...] = out[0].cpu().numpy()
    return image

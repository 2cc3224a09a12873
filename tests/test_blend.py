"""Paste-back on the GPU: lipreal.py:207-214, musereal.py:238-247, musetalk/utils/blending.py:103-125.

Byte work: the bar is BIT-EXACT against oracle/blend_ref.py.  PARITY UNPINNED: OpenCV is absent here, the oracle restates its published
8-bit algorithms and is pinned by the hand-derived known-answer vectors below."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from oracle import blend_ref as R

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


# ---- known answers, derived by hand from the algorithm statements in the oracle's header ----------------------------------------
def _gray3(v):
    return np.repeat(np.asarray(v, np.uint8)[..., None], 3, axis=-1)


def test_resize_kat_upscale_2_to_4():
    # scale 0.5: fx = -0.25 (clamped to pixel 0), 0.25, 0.75, 1.25 (clamped to pixel 1) -> 0, 100/4, 300/4, 100
    assert R.resize_linear_u8(_gray3([[0, 100]]), 4, 1)[0, :, 0].tolist() == [0, 25, 75, 100]


def test_resize_kat_downscale_3_to_2():
    # scale 1.5: fx = 0.25 -> (0 * 1536 + 100 * 512) / 2048 = 25;  fx = 1.75 -> (100 * 512 + 200 * 1536) / 2048 = 175
    assert R.resize_linear_u8(_gray3([[0, 100, 200]]), 2, 1)[0, :, 0].tolist() == [25, 175]


def test_resize_kat_exact_decimation_takes_the_area_path():
    # (10 + 20 + 30 + 41 + 2) >> 2 = 25; the bilinear formula would sample the centre with weights 1/4 each: same value here, so also check
    # a case where they differ: bilinear of a 4 -> 2 row samples pixels (0,1) at fx = 0.5 -> (0 + 255) / 2 -> 128 (rounded), area gives 127
    assert R.resize_linear_u8(_gray3([[10, 20], [30, 41]]), 1, 1)[0, 0, 0] == 25
    assert R.resize_linear_u8(_gray3([[0, 255, 0, 0], [0, 254, 0, 0]]), 2, 1)[0, :, 0].tolist() == [(0 + 255 + 0 + 254 + 2) >> 2, 0]


def test_resize_identity_and_vertical_border():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    assert np.array_equal(R.resize_linear_u8(img, 7, 5), img)
    # 1 -> 3 rows: every output row can only see the single source row
    row = rng.integers(0, 256, (1, 7, 3), dtype=np.uint8)
    assert np.array_equal(R.resize_linear_u8(row, 7, 3), np.repeat(row, 3, axis=0))


def test_bgr2gray_kat():
    # the well-known OpenCV values of pure blue / green / red, and a grey pixel maps to itself
    assert R.bgr2gray_u8(np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [7, 7, 7], [255, 255, 255]]], np.uint8)).tolist() == [[29, 150, 76, 7, 255]]


def test_blend_linear_kat():
    m = np.array([[255, 0, 128]], np.uint8)
    w = (m / 255).astype(np.float32)
    s1, s2 = np.full((1, 3, 3), 200, np.uint8), np.full((1, 3, 3), 100, np.uint8)
    # w = 1: 200 / 1.00001 = 199.998 -> 200;  w = 0: 100;  w = 128 / 255: (200 * .50196 + 100 * .49804) / 1.00001 = 150.19 -> 150
    assert R.blend_linear_u8(s1, s2, w, 1 - w)[0, :, 0].tolist() == [200, 100, 150]
    # round half to even on an exact tie is not reachable through the 1e-5 epsilon; saturation is: weights > 1 cannot occur for mask / 255


def test_get_image_blending_touches_only_the_crop_box():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    face = rng.integers(0, 256, (10, 12, 3), dtype=np.uint8)
    mask = _gray3(rng.integers(0, 256, (20, 24), dtype=np.uint8))
    before = img.copy()
    out = R.get_image_blending(img, face, (14, 12, 26, 22), mask, (8, 6, 32, 26))
    assert out is img
    changed = np.any(out != before, axis=2)
    assert changed[6:26, 8:32].any() and not changed[:6].any() and not changed[26:].any() and not changed[:, :8].any() and not changed[:, 32:].any()
    # mask 255 inside the face box -> the face itself; mask 0 -> the original
    img2 = before.copy()
    R.get_image_blending(img2, face, (14, 12, 26, 22), np.full((20, 24, 3), 255, np.uint8), (8, 6, 32, 26))
    assert np.array_equal(img2[12:22, 14:26], face)
    img3 = before.copy()
    R.get_image_blending(img3, face, (14, 12, 26, 22), np.zeros((20, 24, 3), np.uint8), (8, 6, 32, 26))
    assert np.array_equal(img3, before)


def test_dropin_packages_fall_through_to_the_reference(tmp_path):
    """with mere-fusion_amd/dropin ahead of the reference on sys.path, musereal.py:21-24's import block must still
    resolve -- hot-path modules here, everything else in the reference's own `musetalk` / `wav2lip` trees (a stand-in tree is built here:
    the reference itself does not travel to the GPU box and needs cv2 / diffusers to import)."""
    ref = tmp_path / "ref"
    (ref / "musetalk" / "utils" / "face_parsing").mkdir(parents=True)
    (ref / "musetalk" / "whisper" / "whisper").mkdir(parents=True)
    (ref / "musetalk" / "models").mkdir(parents=True)
    (ref / "wav2lip" / "models").mkdir(parents=True)
    (ref / "musetalk" / "utils" / "__init__.py").write_text("")
    (ref / "musetalk" / "utils" / "preprocessing.py").write_text("MARK = 'reference preprocessing'\n")
    (ref / "musetalk" / "utils" / "blending.py").write_text("def get_image(*a): return 'reference get_image'\n")
    (ref / "musetalk" / "utils" / "utils.py").write_text("raise ImportError('the reference utils must be shadowed')\n")
    (ref / "musetalk" / "utils" / "face_parsing" / "__init__.py").write_text("class FaceParsing: pass\n")
    (ref / "musetalk" / "whisper" / "whisper" / "__init__.py").write_text("MARK = 'vendored whisper'\n")
    (ref / "musetalk" / "mere_musetalk.py").write_text("MARK = 'avatar builder'\n")
    (ref / "wav2lip" / "hparams.py").write_text("MARK = 'hparams'\n")
    (ref / "wav2lip" / "audio.py").write_text("raise ImportError('needs librosa: must be shadowed')\n")
    (ref / "wav2lip" / "models" / "syncnet.py").write_text("MARK = 'syncnet'\n")
    code = textwrap.dedent(f"""
        import sys
        sys.path[:0] = [{str(os.path.join(ROOT, 'mere-fusion_amd', 'dropin'))!r}, {ROOT!r}, {str(ref)!r}]
        from musetalk.utils.utils import get_file_type, get_video_fps, datagen                       # musereal.py:21
        from musetalk.utils.blending import get_image, get_image_prepare_material, get_image_blending   # musereal.py:23
        from musetalk.utils.utils import load_all_model, load_diffusion_model, load_audio_model      # musereal.py:24
        from musetalk.whisper.audio2feature import Audio2Feature                                     # museasr.py:8
        from musetalk.models.unet import UNet, PositionalEncoding
        from musetalk.models.vae import VAE
        import mere_fusion_amd.musetalk.models.unet as U, mere_fusion_amd.musetalk.utils.blending as Bl
        assert UNet is U.UNet and get_image_blending is Bl.get_image_blending
        assert get_file_type('a.PNG') == 'image' and get_file_type('b.mp4') == 'video' and get_file_type('c.txt') == 'unsupported'
        import musetalk.utils.preprocessing as P, musetalk.whisper.whisper as WW, musetalk.mere_musetalk as MM
        assert P.MARK == 'reference preprocessing' and WW.MARK == 'vendored whisper' and MM.MARK == 'avatar builder'
        assert get_image() == 'reference get_image'                      # offline helper forwarded to the reference's module
        from face_parsing import FaceParsing                             # the sys.path side effect of musetalk/utils/__init__.py:1-5
        from wav2lip.models import Wav2Lip
        from wav2lip import audio
        import wav2lip.hparams as HP, wav2lip.models.syncnet as SN, wav2lip.audio as A2
        assert HP.MARK == 'hparams' and SN.MARK == 'syncnet' and A2 is audio and hasattr(audio, 'melspectrogram')
        print('ok')
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_legacy_vae_attention_keys_are_remapped():
    """the published sd-vae-ft-mse file stores the mid-block attention as query / key / value / proj_attn."""
    from mere_fusion_amd.musetalk.models.vae import remap_legacy_attention_keys
    a = "decoder.mid_block.attentions.0."
    sd = {a + "query.weight": 1, a + "key.bias": 2, a + "value.weight": 3, a + "proj_attn.weight": 4, a + "proj_attn.bias": 5, a + "group_norm.weight": 6,
          "decoder.conv_in.weight": 7, a + "to_q.bias": 8}
    got = remap_legacy_attention_keys(sd)
    assert got == {a + "to_q.weight": 1, a + "to_k.bias": 2, a + "to_v.weight": 3, a + "to_out.0.weight": 4, a + "to_out.0.bias": 5,
                   a + "group_norm.weight": 6, "decoder.conv_in.weight": 7, a + "to_q.bias": 8}


# ---- GPU: bit-exact against the oracle -----------------------------------------------------------------------------------------
RESIZE_CASES = [(256, 256, 171, 203), (256, 256, 300, 340), (256, 256, 128, 128), (256, 256, 256, 256), (96, 96, 131, 77), (96, 96, 48, 48),
                (96, 96, 333, 402), (5, 7, 13, 3), (256, 256, 1, 1), (3, 2, 2, 3)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", RESIZE_CASES, ids=lambda c: "%dx%d_to_%dx%d" % c)
def test_hip_resize_bit_exact(lib_built, case):
    from mere_fusion_amd.paste import resize_linear_u8
    sh, sw, dw, dh = case
    src = np.random.default_rng(sh * 1000 + dw).integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    got = resize_linear_u8(torch.from_numpy(src).cuda(), dw, dh).cpu().numpy()
    assert np.array_equal(got, R.resize_linear_u8(src, dw, dh))


def _avatar(rng, n, H, W, lip):
    frames = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    boxes, masks, crops = [], [], []
    for i in range(n):
        w, h = int(rng.integers(20, W // 2)), int(rng.integers(20, H // 2))
        x1, y1 = int(rng.integers(8, W - w - 8)), int(rng.integers(8, H - h - 8))
        boxes.append((y1, y1 + h, x1, x1 + w) if lip else (x1, y1, x1 + w, y1 + h))
        ex, ey = int(rng.integers(0, 8)), int(rng.integers(0, 8))
        crops.append((x1 - ex, y1 - ey, x1 + w + ex, y1 + h + ey))
        m = rng.integers(0, 256, (h + 2 * ey, w + 2 * ex, 3), dtype=np.uint8)
        if i % 2 == 0:
            m = np.repeat(m[..., :1], 3, axis=2)                      # what cv2.imread of a grey png gives; odd frames keep a colour mask
        m[:2] = 255; m[-2:] = 0
        masks.append(m)
    return frames, boxes, masks, crops


@pytest.mark.gpu
@pytest.mark.parametrize("W", [180, 183])
def test_hip_lip_paste_bit_exact(lib_built, W):
    """lipreal.py:207-214 on fp32 `pred * 255` frames, (y1, y2, x1, x2) coords; 183 exercises the unaligned-row path."""
    from mere_fusion_amd.paste import AvatarFrames
    rng = np.random.default_rng(3)
    frames, boxes, _, _ = _avatar(rng, 5, 150, W, lip=True)
    res = (rng.random((4, 96, 96, 3)) * 255).astype(np.float32)
    res[0, 0, 0] = [255.0, 0.0, 254.999]
    idx = [4, 0, 2, 2]
    av = AvatarFrames(frames, boxes, lip_order=True)
    got = av.paste(torch.from_numpy(res).cuda(), idx).cpu().numpy()
    for k, i in enumerate(idx):
        assert np.array_equal(got[k], R.lip_paste(frames[i], res[k], boxes[i])), k


@pytest.mark.gpu
def test_hip_muse_blend_bit_exact(lib_built):
    """musereal.py:238-247 + blending.py:103-125 on uint8 256 x 256 frames; more jobs than one launch carries (32)."""
    from mere_fusion_amd.paste import AvatarFrames
    rng = np.random.default_rng(4)
    frames, boxes, masks, crops = _avatar(rng, 7, 220, 260, lip=False)
    n = 37
    res = rng.integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)
    idx = [int(v) for v in rng.integers(0, 7, n)]
    av = AvatarFrames(frames, boxes, masks, crops)
    got = av.paste(torch.from_numpy(res).cuda(), idx).cpu().numpy()
    for k, i in enumerate(idx):
        want = R.muse_paste(frames[i], res[k], boxes[i], masks[i], crops[i])
        assert np.array_equal(got[k], want), (k, int(np.abs(got[k].astype(int) - want.astype(int)).max()))
    assert np.array_equal(av.frames.cpu().numpy(), frames)            # the cache is never written (copy.deepcopy, musereal.py:237)


@pytest.mark.gpu
def test_hip_get_image_blending_dropin(lib_built):
    from mere_fusion_amd.musetalk.utils.blending import get_crop_box, get_image_blending
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    box = (60, 40, 110, 95)
    crop, s = get_crop_box(box, 1.2)
    assert crop == [85 - 32, 67 - 32, 85 + 32, 67 + 32] and s == 32      # blending.py:8-14 by hand: centre (85, 67), int(max(50, 55) // 2 * 1.2) = int(32.4)
    face = rng.integers(0, 256, (55, 50, 3), dtype=np.uint8)
    mask = np.repeat(rng.integers(0, 256, (crop[3] - crop[1], crop[2] - crop[0], 1), dtype=np.uint8), 3, axis=2)
    want = R.get_image_blending(img.copy(), face, box, mask, crop)
    got = get_image_blending(img, face, box, mask, crop)
    assert got is img and np.array_equal(got, want)


@pytest.mark.gpu
def test_hip_paste_rejects_bad_geometry(lib_built):
    from mere_fusion_amd.paste import AvatarFrames
    fr = np.zeros((1, 50, 60, 3), np.uint8)
    res = torch.zeros((1, 96, 96, 3), dtype=torch.uint8).cuda()
    with pytest.raises(RuntimeError, match="bbox"):
        AvatarFrames(fr, [(10, 10, 70, 40)]).paste(res, [0])            # x2 > W: cv2 / numpy would raise in the reference too
    with pytest.raises(RuntimeError, match="bbox"):
        AvatarFrames(fr, [(10, 10, 10, 40)]).paste(res, [0])            # empty
    with pytest.raises(RuntimeError, match="no CPU path"):
        AvatarFrames(fr, [(10, 10, 20, 40)]).paste(res.cpu(), [0])

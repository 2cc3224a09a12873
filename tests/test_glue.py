"""CPU: host-side glue -- oracle and product against the
hand-derived vectors of the survey, and against each other."""
import numpy as np
import pytest

from mere_fusion_amd import lip_driver as D
from oracle import glue_ref as G


def test_mel_chunk_starts_kat():
    # lipasr.py:24-35 at fps=50, l=r=10: B=16 -> 2B+20 = 52 chunks, T = 84
    want = [16, 19, 22, 25, 28, 32, 35, 38, 41, 44, 48, 51, 54, 57, 60, 64]
    assert D.mel_chunk_starts(52, 10, 10, 50, 84) == want
    mel = np.arange(80 * 84, dtype=np.float64).reshape(80, 84)
    chunks, starts = G.mel_chunks(mel, 52, 10, 10, 50)
    assert starts == want and len(chunks) == 16 and all(c.shape == (80, 16) for c in chunks)
    assert D.mel_chunk_starts(22, 10, 10, 50, 36) == [16]          # B=1
    assert D.mel_chunk_starts(22, 10, 10, 50, 30) == [14]          # tail clamp (lipasr.py:31-32)
    assert G.mel_chunks(np.zeros((80, 30)), 22, 10, 10, 50)[1] == [14]


@pytest.mark.parametrize("B", [1, 2, 8, 16, 32])
def test_mel_chunk_starts_product_equals_oracle(B):
    n = 2 * B + 20
    T = 1 + n * 320 // 200
    assert D.mel_chunk_starts(n, 10, 10, 50, T) == G.mel_chunks(np.zeros((80, T)), n, 10, 10, 50)[1]


def test_mirror_index_kat():
    want = [0, 1, 2, 3, 4, 4, 3, 2, 1, 0, 0, 1]
    assert [D.mirror_index(5, i) for i in range(12)] == want
    assert [G.mirror_index(5, i) for i in range(12)] == want
    assert D.mirror_index(1, 7) == 0


def test_face_batch_kat():
    # constant-255 face -> ch0-2 = 1 for rows < 48, 0 for rows >= 48; ch3-5 = 1 (lipreal.py:115-122)
    faces = np.full((2, 96, 96, 3), 255, np.uint8)
    img, mel = G.face_batch(faces, [np.zeros((80, 16))] * 2)
    assert img.shape == (2, 6, 96, 96) and img.dtype == np.float32 and mel.shape == (2, 1, 80, 16)
    assert (img[:, :3, :48] == 1).all() and (img[:, :3, 48:] == 0).all() and (img[:, 3:] == 1).all()


def test_frame_scaling_kat():
    # lipreal.py:126 then lipreal.py:211: truncation, not rounding: 0.999 -> 254
    f = G.frames_from_pred(np.full((1, 3, 2, 2), 0.999, np.float32))
    assert f.shape == (1, 2, 2, 3)
    assert G.to_uint8(f).max() == 254

"""ORACLE (test infrastructure, not product): numpy restatement of the per-batch glue around the
Wav2Lip generator.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it.

  mel chunking      lipasr.py:24-35
  face batch prep   lipreal.py:109-122
  frame scaling     lipreal.py:126, consumer truncation lipreal.py:211
  mirror index      lipreal.py:65-72

None of lipasr.py / lipreal.py can be imported in the build container (they pull cv2, av, librosa,
soundfile), so these functions are pinned by the hand-derived vectors of SURVEY 8a
(tests/test_glue.py), not by running the reference.
"""
import numpy as np


def mirror_index(size, index):
    turn = index // size
    res = index % size
    if turn % 2 == 0:
        return res
    return size - res - 1


def mel_chunks(mel, n_frames, stride_left, stride_right, fps):
    """mel: (80, T).  Returns the list of (80,16) windows one run_step puts on feat_queue."""
    left = max(0, stride_left * 80 / 50)
    mel_idx_multiplier = 80. * 2 / fps
    mel_step_size = 16
    i = 0
    out, starts = [], []
    while i < (n_frames - stride_left - stride_right) / 2:
        start_idx = int(left + i * mel_idx_multiplier)
        if start_idx + mel_step_size > len(mel[0]):
            out.append(mel[:, len(mel[0]) - mel_step_size:])
            starts.append(len(mel[0]) - mel_step_size)
        else:
            out.append(mel[:, start_idx: start_idx + mel_step_size])
            starts.append(start_idx)
        i += 1
    return out, starts


def face_batch(faces_u8, mel_list):
    """faces_u8: (B,96,96,3) uint8 BGR; mel_list: B x (80,16).  -> img (B,6,96,96) f32, mel (B,1,80,16) f32."""
    img_batch = np.asarray(faces_u8)
    mel_batch = np.asarray(mel_list)
    img_masked = img_batch.copy()
    img_masked[:, img_batch.shape[1] // 2:] = 0
    img_batch = np.concatenate((img_masked, img_batch), axis=3) / 255.
    mel_batch = np.reshape(mel_batch, [len(mel_batch), mel_batch.shape[1], mel_batch.shape[2], 1])
    img = np.transpose(img_batch, (0, 3, 1, 2)).astype(np.float32)
    mel = np.transpose(mel_batch, (0, 3, 1, 2)).astype(np.float32)
    return img, mel


def frames_from_pred(pred_nchw):
    """pred (B,3,96,96) in [0,1] -> (B,96,96,3) float * 255 (lipreal.py:126)."""
    return np.asarray(pred_nchw).transpose(0, 2, 3, 1) * 255.


def to_uint8(frame):
    """process_frames: res_frame.astype(np.uint8) -- truncation, not rounding (lipreal.py:211)."""
    return np.asarray(frame).astype(np.uint8)

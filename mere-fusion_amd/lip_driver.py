"""Per-batch glue of the Wav2Lip render loop, restated for the GPU path.

The reference runs this logic inside `LipASR.run_step` (lipasr.py:14-37) and `inference()`
(lipreal.py:75-141) around mp.Queues.  The drop-in keeps those files untouched; this module is
the queue-free equivalent that bench.py and the multi-session harness drive directly:
mel chunking, ping-pong face selection, and the fused uint8 -> generator -> frame*255 step.
"""
import numpy as np
import torch

from .wav2lip import audio


def mirror_index(size, index):
    """Ping-pong walk over the cached face crops (lipreal.py:65-72, basereal.py:133-139)."""
    turn, res = divmod(index, size)
    return res if turn % 2 == 0 else size - res - 1


def mel_chunk_starts(n_frames, stride_left, stride_right, fps, mel_len, mel_step=16):
    """Start column of every 16-wide mel window of one run_step (lipasr.py:24-35).

    n_frames = number of 20 ms audio chunks in the window (2B + l + r in steady state);
    one video frame consumes two of them, hence the /2 and the 80*2/fps columns per frame.
    Windows that would run past the spectrogram are clamped to its tail (lipasr.py:31-32)."""
    left = max(0, stride_left * 80 / 50)
    mult = 80.0 * 2 / fps
    starts = []
    i = 0
    while i < (n_frames - stride_left - stride_right) / 2:
        s = int(left + i * mult)
        if s + mel_step > mel_len:
            s = mel_len - mel_step
        starts.append(s)
        i += 1
    return starts


class LipASRFrontend:
    """LipASR.run_step without the queues: 2B new 20 ms chunks in, B mel windows [B,1,80,16] out."""

    def __init__(self, batch_size, fps=50, stride_left=10, stride_right=10, device="cuda"):
        self.batch_size, self.fps = batch_size, fps
        self.l, self.r = stride_left, stride_right
        self.device = device
        self.frames = []

    def warm_up(self, chunk=320):
        """baseasr.py:53-59: prime the context with l+r silent chunks."""
        self.frames = [np.zeros(chunk, dtype=np.float32) for _ in range(self.l + self.r)]

    def run_step(self, new_chunks):
        self.frames.extend(new_chunks)
        if len(self.frames) <= self.l + self.r:
            return None
        wav = torch.from_numpy(np.concatenate(self.frames)).to(self.device)
        mel = audio.melspectrogram_device(wav)                       # [80, T] on device
        starts = mel_chunk_starts(len(self.frames), self.l, self.r, self.fps, mel.shape[1])
        idx = torch.tensor(starts, device=mel.device)[:, None] + torch.arange(16, device=mel.device)[None, :]
        chunks = mel[:, idx].permute(1, 0, 2).unsqueeze(1).contiguous()   # [B,1,80,16]
        self.frames = self.frames[-(self.l + self.r):]
        return chunks


class LipSession:
    """One talking-head session: cached uint8 face crops on the device + the generator.  With `avatar_frames`
    (mere_fusion_amd.paste.AvatarFrames built from frame_list_cycle / coord_list_cycle with lip_order=True) `step_pasted` also does
    process_frames' paste-back (lipreal.py:207-214) on the device."""

    def __init__(self, model, faces_u8, avatar_frames=None):
        self.model = model
        self.faces = faces_u8 if torch.is_tensor(faces_u8) else torch.from_numpy(np.asarray(faces_u8))
        self.faces = self.faces.to(next(model.parameters()).device)
        self.avatar_frames = avatar_frames
        self.index = 0

    def step(self, mel_batch):
        """lipreal.py:109-137 for one non-silent batch: returns fp32 frames [B,96,96,3] (pred*255) and
        the face indices they belong to; process_frames truncates with astype(uint8) (lipreal.py:211)."""
        B = mel_batch.shape[0]
        n = self.faces.shape[0]
        idx = [mirror_index(n, self.index + i) for i in range(B)]
        self.index += B
        sel = self.faces[torch.tensor(idx, device=self.faces.device)]
        return self.model.forward_u8(mel_batch, sel), idx

    def step_pasted(self, mel_batch):
        """step() + lipreal.py:207-214: the full uint8 BGR frames [B, H, W, 3] with the generated mouth region resized into the bbox,
        still on the device."""
        if self.avatar_frames is None:
            raise RuntimeError("LipSession.step_pasted needs AvatarFrames (full frames + coords)")
        frames, idx = self.step(mel_batch)
        return self.avatar_frames.paste(frames, idx), idx

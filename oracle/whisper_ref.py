"""ORACLE (test infrastructure, not product): CPU fp32 restatement of MuseTalk's Whisper feature path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Restated (paths relative to the reference checkout):
  * log_mel_spectrogram       musetalk/whisper/whisper/audio.py:92-125 (constants :13-19)
  * mel filterbank            audio.py:77-89 loads assets/mel_filters.npz, which its docstring defines as
                              librosa.filters.mel(sr=16000, n_fft=400, n_mels=80); rebuilt here from that
                              definition (Slaney scale + Slaney norm) and checked against the asset's values
                              recorded in tests/golden/whisper_golden.npz
  * AudioEncoder.forward      whisper/model.py:143-171 with ResidualAttentionBlock :103-128,
                              MultiHeadAttention.qkv_attention :89-100, sinusoids :48-54
  * transcribe segment loop   whisper/transcribe.py:85-128 (pad every <=30 s segment to 3000 frames)
  * Audio2Feature             musetalk/whisper/audio2feature.py:16-45 (get_sliced_feature), :82-97
                              (feature2chunks), :99-112 (audio2feat)

Pinned by tests/golden/whisper_golden.npz, recorded by tests/golden/make_whisper_golden.py from the real
`musetalk.whisper.whisper` modules (imported with ffmpeg / soundfile stubs; only load_audio(file) uses them)
on the seeded encoder weights of mere-fusion_amd/weights.py.  Model dims are the public "tiny" ones
[upstream-knowledge]; the reference takes them from the absent checkpoint (whisper/__init__.py:112).
"""
import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE, N_FFT, N_MELS, HOP_LENGTH, N_FRAMES = 16000, 400, 80, 160, 3000


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filters():
    """librosa.filters.mel(sr=16000, n_fft=400, n_mels=80): fmin 0, fmax sr/2, Slaney, float32 (80, 201)."""
    n_bins = 1 + N_FFT // 2
    weights = np.zeros((N_MELS, n_bins), dtype=np.float32)
    fftfreqs = np.linspace(0, SAMPLE_RATE / 2.0, n_bins)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(SAMPLE_RATE / 2.0), N_MELS + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(N_MELS):
        weights[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    enorm = 2.0 / (mel_f[2:N_MELS + 2] - mel_f[:N_MELS])
    weights *= enorm[:, np.newaxis]
    return weights


def log_mel_spectrogram(audio):
    audio = torch.as_tensor(np.asarray(audio, dtype=np.float32))
    window = torch.hann_window(N_FFT)
    stft = torch.stft(audio, N_FFT, HOP_LENGTH, window=window, return_complex=True)
    magnitudes = stft[:, :-1].abs() ** 2
    mel_spec = torch.from_numpy(mel_filters()) @ magnitudes
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0


def sinusoids(length, channels, max_timescale=10000):
    inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    st = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(st), torch.cos(st)], dim=1)


def _attention(sd, p, x, n_head):
    q = F.linear(x, sd[p + "query.weight"], sd[p + "query.bias"])
    k = F.linear(x, sd[p + "key.weight"])
    v = F.linear(x, sd[p + "value.weight"], sd[p + "value.bias"])
    B, T, C = q.shape
    scale = (C // n_head) ** -0.25
    q = q.view(B, T, n_head, -1).permute(0, 2, 1, 3) * scale
    k = k.view(B, T, n_head, -1).permute(0, 2, 3, 1) * scale
    v = v.view(B, T, n_head, -1).permute(0, 2, 1, 3)
    w = F.softmax((q @ k).float(), dim=-1)
    o = (w @ v).permute(0, 2, 1, 3).flatten(start_dim=2)
    return F.linear(o, sd[p + "out.weight"], sd[p + "out.bias"])


@torch.no_grad()
def encoder_forward(sd, mel, n_head=6, n_layer=4):
    """mel (B, 80, 3000) -> (ln_post(x), embeddings (B, n_layer+1, 1500, C)) as model.py:143-171."""
    x = F.gelu(F.conv1d(mel, sd["conv1.weight"], sd["conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, sd["conv2.weight"], sd["conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1)
    C = x.shape[-1]
    x = x + sinusoids(x.shape[1], C)
    embs = [x]
    for i in range(n_layer):
        p = f"blocks.{i}."
        h = F.layer_norm(x, (C,), sd[p + "attn_ln.weight"], sd[p + "attn_ln.bias"])
        x = x + _attention(sd, p + "attn.", h, n_head)
        h = F.layer_norm(x, (C,), sd[p + "mlp_ln.weight"], sd[p + "mlp_ln.bias"])
        h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"])), sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])
        x = x + h
        embs.append(x)
    out = F.layer_norm(x, (C,), sd["ln_post.weight"], sd["ln_post.bias"])
    return out, torch.stack(embs, dim=1)


def pad_or_trim(mel, length=N_FRAMES):
    if mel.shape[-1] > length:
        mel = mel[..., :length]
    if mel.shape[-1] < length:
        mel = F.pad(mel, (0, length - mel.shape[-1]))
    return mel


def audio2feat(sd, audio, n_head=6, n_layer=4):
    """transcribe (transcribe.py:85-128) + Audio2Feature.audio2feat (audio2feature.py:99-112):
    float32 waveform -> (T50, n_layer+1, C) numpy, T50 = frames // 2 per <=3000-frame segment."""
    mel = log_mel_spectrogram(audio)
    num_frames = mel.shape[-1]
    out = []
    seek = 0
    while seek < num_frames:
        end_seek = min(seek + 3000, num_frames)
        seg = pad_or_trim(mel[:, seek:seek + 3000])[None]
        _, emb = encoder_forward(sd, seg, n_head, n_layer)          # (1, L+1, 1500, C)
        e = emb.numpy().transpose(0, 2, 1, 3).squeeze(0)            # (1500, L+1, C)
        out.append(e[: int((end_seek - seek) / 2)])
        seek += 3000
    return np.concatenate(out, axis=0)


def get_sliced_feature(feature_array, vid_idx, audio_feat_length=(2, 2), fps=25):
    length = len(feature_array)
    center_idx = int(vid_idx * 50 / fps)
    left_idx = center_idx - audio_feat_length[0] * 2
    right_idx = center_idx + (audio_feat_length[1] + 1) * 2
    sel, idxs = [], []
    for idx in range(left_idx, right_idx):
        idx = min(length - 1, max(0, idx))
        sel.append(feature_array[idx])
        idxs.append(idx)
    return np.concatenate(sel, axis=0).reshape(-1, 384), idxs


def feature2chunks(feature_array, fps, batch_size, audio_feat_length=(2, 2), start=0):
    chunks, idx_lists = [], []
    for i in range(batch_size):
        f, idxs = get_sliced_feature(feature_array, i + start, audio_feat_length, fps)
        chunks.append(f)
        idx_lists.append(idxs)
    return chunks, idx_lists

"""MuseTalk per-batch glue on the device (musereal.py:91-108, museasr.py:15-29) and the cross-session batcher."""
import numpy as np
import pytest
import torch

from mere_fusion_amd import muse_driver as D
from mere_fusion_amd import weights as W
from oracle import glue_ref


def test_chunk_rows_match_survey_8c():
    # SURVEY 8c (captured from the reference by import-with-stubs): B = 16 -> row lists [6..15] ... [36..45]; B = 8 -> [6..15] ... [20..29]
    assert D.chunk_left_rows(16, 50 / 2, 10 / 2) == list(range(6, 37, 2))
    assert D.chunk_left_rows(8, 50 / 2, 10 / 2) == list(range(6, 21, 2))
    assert [D.mirror_index(5, i) for i in range(12)] == [0, 1, 2, 3, 4, 4, 3, 2, 1, 0, 0, 1]
    assert [D.mirror_index(5, i) for i in range(12)] == [glue_ref.mirror_index(5, i) for i in range(12)]


def test_session_walk_and_errors():
    s = D.MuseSession([torch.zeros(1, 8, 32, 32) for _ in range(3)])
    assert s.next_indices(4) == [0, 1, 2, 2] and s.next_indices(3) == [1, 0, 0]
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU path"):
            D.MuseBatcher(None, None, [s])
        with pytest.raises(RuntimeError, match="no CPU path"):
            D.feature_chunks_device(torch.zeros(4, 5, 384), [0])


@pytest.mark.gpu
def test_hip_feature_chunks_vs_oracle(lib_built):
    from oracle import whisper_ref
    rng = np.random.default_rng(0)
    for T, B in ((52, 16), (36, 8), (7, 8)):                      # the last window is shorter than the chunks reach: rows clamp to T - 1
        feat = rng.standard_normal((T, 5, 384)).astype(np.float32)
        want, _ = whisper_ref.feature2chunks(feat, fps=25.0, batch_size=B, start=5.0)
        got = D.feature_chunks_device(torch.from_numpy(feat).cuda(), D.chunk_left_rows(B, 25.0, 5.0)).cpu().numpy()
        assert got.shape == (B, 50, 384) and np.array_equal(got, np.stack(want))
    want0, _ = whisper_ref.feature2chunks(feat, fps=25.0, batch_size=2, start=0)               # left rows negative: clamp to 0
    got0 = D.feature_chunks_device(torch.from_numpy(feat).cuda(), D.chunk_left_rows(2, 25.0, 0)).cpu().numpy()
    assert np.array_equal(got0, np.stack(want0))


@pytest.mark.gpu
def test_hip_frontend_matches_oracle_run_step(lib_built):
    """MuseASRFrontend.run_step == museasr.py:15-29 on the oracle: features of the sliding window, then the chunks."""
    from mere_fusion_amd.musetalk.whisper.audio2feature import Audio2Feature
    from oracle import whisper_ref
    wsd = W.make_whisper_encoder_state_dict(0)
    a2f = Audio2Feature(state_dict=wsd, n_head=6)
    fe = D.MuseASRFrontend(a2f, batch_size=8)
    fe.warm_up()
    wav = W.make_speech_like_wav(16 * 320, 3)
    new = [wav[i * 320:(i + 1) * 320] for i in range(16)]
    got = fe.run_step(new)
    full = np.concatenate([np.zeros(20 * 320, np.float32), wav])
    want, _ = whisper_ref.feature2chunks(whisper_ref.audio2feat(wsd, full), fps=25.0, batch_size=8, start=5.0)
    assert got.shape == (8, 50, 384) and len(fe.frames) == 20
    assert np.abs(got.cpu().numpy() - np.stack(want)).max() <= 2e-3


@pytest.mark.gpu
def test_hip_batcher_three_sessions_vs_oracle(lib_built):
    """Three sessions (one silent) through one UNet / VAE pair, with the GPU paste-back: every session's frames against the oracle step on
    ITS latents / chunks (musereal.py:91-108) and against the oracle paste (musereal.py:238-247) of those frames."""
    from mere_fusion_amd.musetalk.models.unet import UNet
    from mere_fusion_amd.musetalk.models.vae import VAE
    from mere_fusion_amd.paste import AvatarFrames
    from mere_fusion_amd.musetalk.config import unet_config_json, vae_config_json
    from oracle import blend_ref, musetalk_ref as R
    cfg = R.MUSETALK_SMALL
    usd, vsd = W.make_musetalk_unet_state_dict(cfg, 0), W.make_musetalk_vae_state_dict(cfg, 0)
    B, S = 2, 3
    unet = UNet(unet_config_json(cfg["unet"]), usd, max_batch=B * S)
    vae = VAE(config=vae_config_json(cfg["vae"]), state_dict=vsd, max_batch=B * S)
    rng = np.random.default_rng(9)
    sessions, lat_lists, avatars = [], [], []
    for s in range(S):
        n = 3 + s
        lats = [W.make_musetalk_inputs(1, 40 + 10 * s + i)[0] for i in range(n)]
        H_, W_ = 300, 320
        frames = rng.integers(0, 256, (n, H_, W_, 3), dtype=np.uint8)
        boxes = [(40 + 3 * i, 30 + 2 * i, 40 + 3 * i + 200 + i, 30 + 2 * i + 220 - i) for i in range(n)]
        crops = [(b[0] - 10, b[1] - 12, b[2] + 10, b[3] + 12) for b in boxes]
        masks = [np.repeat(rng.integers(0, 256, (c[3] - c[1], c[2] - c[0], 1), dtype=np.uint8), 3, axis=2) for c in crops]
        av = AvatarFrames(frames, boxes, masks, crops)
        sessions.append(D.MuseSession(lats, avatar_frames=av))
        lat_lists.append(lats)
        avatars.append((frames, boxes, masks, crops))
    bat = D.MuseBatcher(unet, vae, sessions, batch_size=B, paste=True)
    plain = D.MuseBatcher(unet, vae, [D.MuseSession(l) for l in lat_lists], batch_size=B, paste=False)
    index = [0] * S
    for step in range(3):                                                     # step 2 walks past the end of session 0's 3 latents: mirrored
        chunks = [W.make_musetalk_inputs(B, 100 * step + s)[1] for s in range(S)]
        silent = step % S                                                     # a different session is silent each step
        dev = [None if s == silent else chunks[s].cuda() for s in range(S)]
        out = bat.step(dev)
        raw = plain.step(dev)
        for s in range(S):
            want_idx = [glue_ref.mirror_index(len(lat_lists[s]), index[s] + i) for i in range(B)]
            index[s] += B
            assert out[s][1] == want_idx and raw[s][1] == want_idx
            if s == silent:
                assert out[s][0] is None and raw[s][0] is None
                continue
            lat = torch.cat([lat_lists[s][i] for i in want_idx], dim=0)                    # musereal.py:92-97
            want_u8, _ = R.musetalk_step(usd, vsd, cfg, lat, chunks[s])
            got_u8 = raw[s][0].cpu().numpy()
            d = np.abs(got_u8.astype(int) - want_u8.astype(int))
            assert d.max() <= 2 and (d > 0).mean() < 0.05, (step, s, d.max())
            frames, boxes, masks, crops = avatars[s]
            pasted = out[s][0].cpu().numpy()
            for k, i in enumerate(want_idx):                                                # byte work: bit-exact on the SAME generated frame
                assert np.array_equal(pasted[k], blend_ref.muse_paste(frames[i], got_u8[k], boxes[i], masks[i], crops[i])), (step, s, k)

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


def test_pick_sessions_policy():
    """Which sessions share a step (host logic of SessionScheduler): oldest first, a full step at once, a partial one after the hold."""
    assert D.pick_sessions({}, 1.0, 8, 0.08) == []
    assert D.pick_sessions({3: 1.00}, 1.05, 8, 0.08) == []                               # alone and young: wait for company
    assert D.pick_sessions({3: 1.00}, 1.08, 8, 0.08) == [3]                              # ... but never longer than the hold
    assert D.pick_sessions({0: 1.00, 1: 0.50, 2: 0.70}, 1.01, 2, 0.08) == [1, 2]         # capacity reached: no hold, oldest first
    assert D.pick_sessions({5: 2.0, 1: 2.0, 4: 1.0}, 2.0, 3, 0.5) == [4, 1, 5]           # ties broken by session number
    assert D.pick_sessions({0: 1.0}, 9.0, 0, 0.0) == []


def test_session_scheduler_queues_on_a_fake_clock():
    """SessionScheduler over a stand-in batcher (no device): per-session FIFO, one batch of a session per step, sessions outside a step
    untouched, latency = frames-ready time - arrival."""
    class FakeBatcher:
        def __init__(self, n, cap):
            self.sessions, self.batch_size, self.max_sessions_per_step, self.device = [D.MuseSession([torch.zeros(1, 8, 32, 32)] * 5) for _ in range(n)], 8, cap, "cpu"
            self.calls = []

        def step(self, chunks, only=None):
            self.calls.append((sorted(only), [c for c in chunks if c is not None]))
            return [None if k not in only else ("frames%d" % k, self.sessions[k].next_indices(self.batch_size)) for k in range(len(chunks))]

    now = [0.0]
    fb = FakeBatcher(4, cap=2)

    def sync():
        now[0] += 0.100                                                                  # a step "takes" 100 ms
    sch = D.SessionScheduler(fb, clock=lambda: now[0], sync=sync)
    assert sch.period == pytest.approx(0.320) and sch.hold == pytest.approx(0.080) and sch.capacity == 2
    assert sch.run_once() == [] and sch.next_due() is None
    sch.submit(2, "a2", 0.00)
    sch.submit(2, "b2", 0.01)                                                            # a second batch of the same session queues behind the first
    assert sch.next_due() == pytest.approx(0.080) and sch.run_once(0.05) == []           # one session waiting, hold not over
    sch.submit(0, "a0", 0.06)
    now[0] = 0.06
    done = sch.run_once()                                                                # two sessions = capacity: goes at once
    assert [(k, f, round(lat, 3)) for k, f, _, lat in done] == [(2, "frames2", 0.16), (0, "frames0", 0.10)]
    assert fb.calls[-1] == ([0, 2], ["a0", "a2"]) and done[0][2] == [0, 1, 2, 3, 4, 4, 3, 2]
    assert sch.backlog() == 1 and sch.pending() == {2: 0.01}
    sch.submit(1, None, 0.17)                                                            # a silent batch is a batch (indices advance, no frames)
    sch.submit(3, "a3", 0.18)
    done = sch.run_once()                                                                # 3 waiting, capacity 2: the two oldest
    assert [k for k, *_ in done] == [2, 1] and fb.calls[-1] == ([1, 2], ["b2"])
    assert done[0][2] == [1, 0, 0, 1, 2, 3, 4, 4]                                        # session 2 went on where its first batch stopped
    assert fb.sessions[3].index == 0                                                     # session 3 was not touched
    done = sch.run_once(now[0] + 1.0)
    assert [k for k, *_ in done] == [3] and sch.steps == 3 and sch.sessions_served == 5 and sch.pending() == {}


def test_frontend_window_is_the_host_side_of_run_step():
    """museasr.py:17-29 without the encoder call: the window is context + new chunks, the context kept for the next step is the last l + r chunks."""
    fe = D.MuseASRFrontend(None, batch_size=8)
    new = [np.full(320, i, np.float32) for i in range(16)]
    assert fe.window(new) is None and len(fe.frames) == 16                      # 16 <= l + r = 20: museasr.py:22-23 returns before the encoder
    win = fe.window([np.full(320, 16 + i, np.float32) for i in range(16)])
    assert win.shape == (32 * 320,) and win[0] == 0 and win[-1] == 31 and [float(f[0]) for f in fe.frames] == list(range(12, 32))
    fe2 = D.MuseASRFrontend(None, batch_size=8)
    fe2.warm_up()                                                               # baseasr.py:53-59: l + r silent chunks
    win = fe2.window(new)
    assert win.shape == ((20 + 16) * 320,) and not win[:20 * 320].any() and win[20 * 320] == 0 and win[-1] == 15
    assert len(fe2.frames) == 20 and float(fe2.frames[0][0]) == 0.0 and float(fe2.frames[4][0]) == 0.0 and float(fe2.frames[-1][0]) == 15.0


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
            print(f"batcher step {step} session {s}: uint8 max diff {d.max()}, differing pixels {100 * (d > 0).mean():.3f} %")
            assert d.max() <= 1 and (d > 0).mean() < 0.01, (step, s, d.max(), (d > 0).mean())
            frames, boxes, masks, crops = avatars[s]
            pasted = out[s][0].cpu().numpy()
            for k, i in enumerate(want_idx):                                                # byte work: bit-exact on the SAME generated frame
                assert np.array_equal(pasted[k], blend_ref.muse_paste(frames[i], got_u8[k], boxes[i], masks[i], crops[i])), (step, s, k)


@pytest.mark.gpu
def test_hip_scheduler_serves_paced_sessions_vs_oracle(lib_built):
    """SessionScheduler over the real batcher: three sessions on their own clocks, two per step at most (handles built for 2 x B frames).
    Whatever company a session's batch had in its step, its frames are the oracle step on ITS latents / chunks, its indices the mirror walk,
    and a session without a batch in a step does not move."""
    from mere_fusion_amd.musetalk.models.unet import UNet
    from mere_fusion_amd.musetalk.models.vae import VAE
    from mere_fusion_amd.musetalk.config import unet_config_json, vae_config_json
    from oracle import musetalk_ref as R
    cfg = R.MUSETALK_SMALL
    usd, vsd = W.make_musetalk_unet_state_dict(cfg, 0), W.make_musetalk_vae_state_dict(cfg, 0)
    B, S, CAP = 2, 3, 2
    unet = UNet(unet_config_json(cfg["unet"]), usd, max_batch=B * CAP)
    vae = VAE(config=vae_config_json(cfg["vae"]), state_dict=vsd, max_batch=B * CAP)
    lat_lists = [[W.make_musetalk_inputs(1, 300 + 10 * s + i)[0] for i in range(3 + s)] for s in range(S)]
    with pytest.raises(RuntimeError, match="max_batch"):
        D.MuseBatcher(unet, vae, [D.MuseSession(l) for l in lat_lists], batch_size=B)                      # 3 sessions do not fit one step ...
    bat = D.MuseBatcher(unet, vae, [D.MuseSession(l) for l in lat_lists], batch_size=B, max_sessions_per_step=CAP)   # ... but two at a time do
    with pytest.raises(RuntimeError, match="active sessions"):
        bat.step([torch.zeros(B, 50, 384, device="cuda")] * S)
    assert [s.index for s in bat.sessions] == [0, 0, 0]                                                 # a refused step moves nothing
    now = [0.0]
    sch = D.SessionScheduler(bat, clock=lambda: now[0])
    chunks = {(s, j): W.make_musetalk_inputs(B, 900 + 10 * s + j)[1] for s in range(S) for j in range(2)}
    # arrivals: session 1 at 0.00 and 0.05, session 0 at 0.01, session 2 at 0.02 and 0.06
    for s, j, t in ((1, 0, 0.00), (0, 0, 0.01), (2, 0, 0.02), (1, 1, 0.05), (2, 1, 0.06)):
        sch.submit(s, chunks[(s, j)].cuda(), t)
    served, index, seen = [], [0] * S, {s: 0 for s in range(S)}
    now[0] = 0.03
    while sch.pending():
        done = sch.run_once()
        assert done, "three sessions wait and capacity is two: every call must serve"
        served.append([k for k, *_ in done])
        for k, frames, idx, lat in done:
            want_idx = [glue_ref.mirror_index(len(lat_lists[k]), index[k] + i) for i in range(B)]
            index[k] += B
            assert idx == want_idx and lat > 0
            lat_in = torch.cat([lat_lists[k][i] for i in want_idx], dim=0)
            want_u8, _ = R.musetalk_step(usd, vsd, cfg, lat_in, chunks[(k, seen[k])])
            seen[k] += 1
            d = np.abs(frames.cpu().numpy().astype(int) - want_u8.astype(int))
            print(f"scheduler session {k}: uint8 max diff {d.max()}, differing pixels {100 * (d > 0).mean():.3f} %")
            assert d.max() <= 1 and (d > 0).mean() < 0.01, (k, d.max(), (d > 0).mean())
        now[0] += 0.05
    assert served == [[1, 0], [2, 1], [2]] and sch.steps == 3 and sch.sessions_served == 5


@pytest.mark.gpu
def test_hip_end_to_end_scheduler_delivers_the_session_loop_through_rings(lib_built):
    """EndToEndScheduler: PCM chunks in, (res_frame, idx, audio_frames) tuples out of each session's FrameRing.  Every delivered frame equals what the
    step-by-step path of this library (MuseASRFrontend.run_step -> MuseBatcher.step with paste) produces for that session alone -- up to the fp32
    summation order of a Whisper call that holds two windows instead of one -- the indices are the mirror walk, the audio pairs are the session's own
    chunks, and a session's batches come out in order whatever company they had."""
    from mere_fusion_amd.musetalk.models.unet import UNet
    from mere_fusion_amd.musetalk.models.vae import VAE
    from mere_fusion_amd.musetalk.whisper.audio2feature import Audio2Feature
    from mere_fusion_amd.musetalk.config import unet_config_json, vae_config_json
    from mere_fusion_amd.paste import AvatarFrames
    from mere_fusion_amd.transport import FrameRing
    from oracle import musetalk_ref as R
    cfg = R.MUSETALK_SMALL
    usd, vsd = W.make_musetalk_unet_state_dict(cfg, 0), W.make_musetalk_vae_state_dict(cfg, 0)
    B, S, CAP, NB = 2, 3, 2, 3
    unet = UNet(unet_config_json(cfg["unet"]), usd, max_batch=B * CAP)
    vae = VAE(config=vae_config_json(cfg["vae"]), state_dict=vsd, max_batch=B * CAP)
    a2f = Audio2Feature(state_dict=W.make_whisper_encoder_state_dict(0), n_head=6)
    rng = np.random.default_rng(3)
    H_, W_ = 300, 320
    lat_lists, avatars = [], []
    for s in range(S):
        n = 3 + s
        lat_lists.append([W.make_musetalk_inputs(1, 700 + 10 * s + i)[0] for i in range(n)])
        boxes = [(40 + 3 * i, 30 + 2 * i, 40 + 3 * i + 200, 30 + 2 * i + 220) for i in range(n)]
        crops = [(b[0] - 10, b[1] - 12, b[2] + 10, b[3] + 12) for b in boxes]
        masks = [np.repeat(rng.integers(0, 256, (c[3] - c[1], c[2] - c[0], 1), dtype=np.uint8), 3, axis=2) for c in crops]
        avatars.append(AvatarFrames(rng.integers(0, 256, (n, H_, W_, 3), dtype=np.uint8), boxes, masks, crops))
    pcm = [[[W.make_speech_like_wav(320, 50 * s + 7 * j + i) for i in range(2 * B)] for j in range(NB)] for s in range(S)]

    # the step-by-step path, one session at a time
    want = []
    for s in range(S):
        fe = D.MuseASRFrontend(a2f, B)
        fe.warm_up()
        bat1 = D.MuseBatcher(unet, vae, [D.MuseSession(lat_lists[s], avatar_frames=avatars[s])], batch_size=B, paste=True, max_sessions_per_step=1)
        rows = []
        for j in range(NB):
            fr, idx = bat1.step([fe.run_step(pcm[s][j])])[0]
            rows.append((fr.cpu().numpy(), idx))
        want.append(rows)

    fes = [D.MuseASRFrontend(a2f, B) for _ in range(S)]
    for fe in fes:
        fe.warm_up()
    rings = [FrameRing(2 * B, (H_, W_, 3)) for _ in range(S)]
    bat = D.MuseBatcher(unet, vae, [D.MuseSession(lat_lists[s], avatar_frames=avatars[s]) for s in range(S)], batch_size=B, paste=True, max_sessions_per_step=CAP)
    now = [0.0]
    sch = D.EndToEndScheduler(bat, fes, a2f, rings=rings, clock=lambda: now[0], hold_s=0.0)
    for j in range(NB):
        for s in range(S):
            sch.submit(s, pcm[s][j], 0.01 * (S * j + s))
    got = [[] for _ in range(S)]
    served = []
    for _ in range(200):
        now[0] += 0.05
        done = sch.run_once()
        done += sch.drain()                                                                   # (a test has no other work to overlap the copy with)
        served.extend(k for k, *_ in done)
        for k, fr, idx, lat in done:
            assert lat > 0
            for i in range(B):                                                                # the consumer side: process_frames' get()
                f, fidx, audio = rings[k].get(timeout=5)
                j = len(got[k]) // B
                assert fidx == idx[i] and len(audio) == 2 and audio[0][1] == 0 and np.array_equal(audio[0][0], pcm[k][j][2 * i]) and np.array_equal(audio[1][0], pcm[k][j][2 * i + 1])
                got[k].append(f)
        if not sch.pending() and not sch.inflight:
            break
    assert sorted(served) == sorted(list(range(S)) * NB) and sch.steps >= -(-S * NB // CAP)
    for s in range(S):
        assert len(got[s]) == NB * B
        for j in range(NB):
            wfr, widx = want[s][j]
            for i in range(B):
                d = np.abs(got[s][j * B + i].astype(int) - wfr[i].astype(int))
                assert d.max() <= 1 and (d > 0).mean() < 0.01, (s, j, i, d.max(), (d > 0).mean())
    for r in rings:
        r.close()


@pytest.mark.gpu
def test_hip_end_to_end_scheduler_stalled_consumer_and_silent_batches(lib_built):
    """ADVICE r03.  A session whose consumer is behind (its ring is full) is DEFERRED: its batch stays queued, nothing of it is lost, the other sessions of the
    step are delivered, and it is served once the consumer has caught up.  An all-silent batch (type-1 chunks, baseasr.py:33-45) skips the networks and reaches the
    ring as B (None, idx, audio_frames) tuples (musereal.py:82-86)."""
    from mere_fusion_amd.musetalk.models.unet import UNet
    from mere_fusion_amd.musetalk.models.vae import VAE
    from mere_fusion_amd.musetalk.whisper.audio2feature import Audio2Feature
    from mere_fusion_amd.musetalk.config import unet_config_json, vae_config_json
    from mere_fusion_amd.transport import FrameRing
    from oracle import musetalk_ref as R
    cfg = R.MUSETALK_SMALL
    usd, vsd = W.make_musetalk_unet_state_dict(cfg, 0), W.make_musetalk_vae_state_dict(cfg, 0)
    B, S = 2, 2
    unet = UNet(unet_config_json(cfg["unet"]), usd, max_batch=B * S)
    vae = VAE(config=vae_config_json(cfg["vae"]), state_dict=vsd, max_batch=B * S)
    a2f = Audio2Feature(state_dict=W.make_whisper_encoder_state_dict(0), n_head=6)
    lat = [[W.make_musetalk_inputs(1, 900 + 10 * s + i)[0] for i in range(4)] for s in range(S)]
    fes = [D.MuseASRFrontend(a2f, B) for _ in range(S)]
    for fe in fes:
        fe.warm_up()
    rings = [FrameRing(2 * B, (256, 256, 3)) for _ in range(S)]
    bat = D.MuseBatcher(unet, vae, [D.MuseSession(lat[s]) for s in range(S)], batch_size=B, max_sessions_per_step=S)
    now = [0.0]
    sch = D.EndToEndScheduler(bat, fes, a2f, rings=rings, clock=lambda: now[0], hold_s=0.0)
    pcm = lambda s, j: [W.make_speech_like_wav(320, 31 * s + 5 * j + i) for i in range(2 * B)]
    for j in range(3):                                               # three batches each; session 0's consumer is stalled, session 1's drains
        sch.submit(0, pcm(0, j), 0.01 * j)
        sch.submit(1, pcm(1, j) if j != 1 else [(c, 1) for c in pcm(1, j)], 0.01 * j + 0.001)     # session 1's second batch is silence
    got1, served = [], []
    for _ in range(40):
        now[0] += 0.05
        done = sch.run_once() + sch.drain()
        served += [k for k, *_ in done]
        for k, fr, idx, _ in done:
            if k == 1:
                for i in range(B):
                    got1.append(rings[1].get(timeout=5))
    # session 1 got everything, in order, the silent batch as None frames with its type-1 audio; session 0 two batches (its ring holds 2B frames), the third deferred
    assert served.count(1) == 3 and served.count(0) == 2 and sch.ring_full > 0 and len(sch.queues[0]) == 1 and 0 in sch._deferred   # (deferred: left out of pending() until its ring has room)
    assert [g[1] for g in got1] == [D.mirror_index(4, i) for i in range(3 * B)]
    assert all(g[0] is not None for g in got1[:B] + got1[2 * B:]) and all(g[0] is None and g[2][0][1] == 1 and len(g[2]) == 2 for g in got1[B:2 * B])
    first = [rings[0].get(timeout=5) for _ in range(2 * B)]         # the consumer catches up ...
    assert [g[1] for g in first] == [D.mirror_index(4, i) for i in range(2 * B)]
    for _ in range(10):
        now[0] += 0.05
        done = sch.run_once() + sch.drain()
        served += [k for k, *_ in done]
    assert served.count(0) == 3 and not sch.pending()                # ... and the deferred batch is served, with the indices that follow
    last = [rings[0].get(timeout=5) for _ in range(B)]
    assert [g[1] for g in last] == [D.mirror_index(4, 2 * B + i) for i in range(B)] and all(g[0] is not None for g in last)
    for r in rings:
        r.close()


def test_end_to_end_scheduler_leaves_a_deferred_session_out_until_its_ring_has_room():
    """ADVICE r04 (host logic only, no device): a session deferred because its ring was full must not sit at the head of the order with an arrival time in the
    past -- next_due() would lie in the past and the serving loop would spin on run_once -- it is offered again when its ring reports B free slots or after a
    quarter period."""
    from collections import deque
    from types import SimpleNamespace

    class Ring:
        def __init__(self, free):
            self.free = free

        def free_slots(self):
            return self.free

    now = [1.0]
    sch = D.EndToEndScheduler.__new__(D.EndToEndScheduler)          # (the constructor needs a device for its copy stream; the policy under test does not)
    sch.queues = [deque([(0.2, ("win", []))]), deque([(0.9, ("win", []))])]
    sch.capacity, sch.hold, sch.period, sch.clock = 2, 0.08, 0.32, (lambda: now[0])
    sch.batcher = SimpleNamespace(batch_size=8)
    sch.rings = [Ring(0), Ring(16)]
    sch.inflight, sch._deferred = deque(), {0: 1.05}                # session 0 was deferred at t = 0.97 (+ period / 4)
    assert sch.pending() == {1: 0.9}                                # the stalled session does not age the order ...
    assert abs(sch.next_due() - 0.98) < 1e-9 or sch.next_due() == 0.98   # ... session 1's own hold decides (0.9 + 0.08), not a time in the past
    sch.queues[1].clear()
    assert sch.pending() == {} and sch.next_due() == 1.05           # nothing else queued: wake up when the back-off ends, do not spin
    now[0] = 1.06
    assert sch.pending() == {0: 0.2} and 0 not in sch._deferred     # back-off over: offered again (and counted as a new episode if still full)
    sch._deferred = {0: 9.0}
    sch.rings[0].free = 8
    assert sch.pending() == {0: 0.2}                                # ... or as soon as the consumer has freed a batch's worth of slots


@pytest.mark.gpu
def test_hip_end_to_end_scheduler_silent_batches_with_a_stalled_consumer_never_block(lib_built):
    """silent batches used to take no slot and B descriptor messages each, so with a stalled consumer two of them filled the bounded descriptor
    queue and the NEXT publish blocked the single scheduler thread for ever -- every session of the GPU with it.  They now reserve B of the ring's places
    before the step and travel as one message: the third silent batch of the stalled session is deferred, the other session is served throughout."""
    from mere_fusion_amd.musetalk.models.unet import UNet
    from mere_fusion_amd.musetalk.models.vae import VAE
    from mere_fusion_amd.musetalk.whisper.audio2feature import Audio2Feature
    from mere_fusion_amd.musetalk.config import unet_config_json, vae_config_json
    from mere_fusion_amd.transport import FrameRing
    from oracle import musetalk_ref as R
    cfg = R.MUSETALK_SMALL
    usd, vsd = W.make_musetalk_unet_state_dict(cfg, 0), W.make_musetalk_vae_state_dict(cfg, 0)
    B, S = 2, 2
    unet = UNet(unet_config_json(cfg["unet"]), usd, max_batch=B * S)
    vae = VAE(config=vae_config_json(cfg["vae"]), state_dict=vsd, max_batch=B * S)
    a2f = Audio2Feature(state_dict=W.make_whisper_encoder_state_dict(0), n_head=6)
    lat = [[W.make_musetalk_inputs(1, 900 + 10 * s + i)[0] for i in range(4)] for s in range(S)]
    fes = [D.MuseASRFrontend(a2f, B) for _ in range(S)]
    for fe in fes:
        fe.warm_up()
    rings = [FrameRing(2 * B, (256, 256, 3)) for _ in range(S)]
    bat = D.MuseBatcher(unet, vae, [D.MuseSession(lat[s]) for s in range(S)], batch_size=B, max_sessions_per_step=S)
    now = [0.0]
    sch = D.EndToEndScheduler(bat, fes, a2f, rings=rings, clock=lambda: now[0], hold_s=0.0)
    pcm = lambda s, j: [W.make_speech_like_wav(320, 31 * s + 5 * j + i) for i in range(2 * B)]
    for j in range(4):
        sch.submit(0, [(c, 1) for c in pcm(0, j)], 0.01 * j)          # session 0: silence only, and nobody reads its ring
        sch.submit(1, pcm(1, j), 0.01 * j + 0.001)
    served, got1 = [], 0
    for _ in range(60):                                              # (a blocked publish would hang here: the test's time-out is the failure mode of the old code)
        now[0] += 0.005                                              # (period B x 40 ms = 80 ms: a deferral backs off 20 ms = 4 polls)
        done = sch.run_once() + sch.drain()
        served += [k for k, *_ in done]
        for k, fr, idx, _ in done:
            if k == 1:
                for i in range(B):
                    assert rings[1].get(timeout=5)[0] is not None
                    got1 += 1
    assert served.count(1) == 4 and got1 == 4 * B                    # the healthy session got everything
    assert served.count(0) == 2 and sch.ring_full >= 1 and len(sch.queues[0]) == 2    # two silent batches fill session 0's ring (2B places), the rest wait
    assert sch.ring_full <= 20                                       # deferral episodes (one per back-off), not polls (60 polls happened)
    first = [rings[0].get(timeout=5) for _ in range(2 * B)]          # the consumer wakes up: (None, idx, audio) tuples with the type-1 audio, in order
    assert [g[1] for g in first] == [D.mirror_index(4, i) for i in range(2 * B)] and all(g[0] is None and g[2][0][1] == 1 for g in first)
    for _ in range(20):
        now[0] += 0.01
        served += [k for k, *_ in sch.run_once() + sch.drain()]
    assert served.count(0) == 4 and not sch.pending()
    for r in rings:
        r.close()


def test_end_to_end_scheduler_returns_reservations_and_batches_when_a_step_fails():
    """ADVICE r05 (host logic only): the rings publish in begin order, so a reservation leaked by an exception between try_reserve and begin_batch -- the queue
    pop, the Whisper stage, the step -- would wedge that ring for good.  Whatever raises, every reservation goes back and the popped batches return to the head
    of their queues."""
    from collections import deque
    from types import SimpleNamespace
    import queue as _q
    import threading

    class Ring:
        def __init__(self):
            self.open, self.begun, self.aborted = [], [], []

        def free_slots(self):
            return 16

        def try_reserve(self, n):
            tok = object()
            self.open.append(tok)
            return tok

        def unreserve(self, tok):
            self.open.remove(tok)

        def begin_batch(self, fr, idx, stream=None, reserved=None):
            self.open.remove(reserved)
            self.begun.append(reserved)
            return reserved

        def abort_batch(self, tok):
            self.aborted.append(tok)

    def make(step, fixed_chunks):
        sch = D.EndToEndScheduler.__new__(D.EndToEndScheduler)
        sch.queues = [deque([(0.1, ("win0", ["a0"])), (0.5, ("win0b", ["a0b"]))]), deque([(0.2, ("win1", ["a1"]))])]
        sch.capacity, sch.hold, sch.period, sch.clock, sch.depth = 2, 0.0, 0.32, (lambda: 1.0), 2
        sch.batcher = SimpleNamespace(batch_size=8, device="cpu", step=step)
        sch.rings = [Ring(), Ring()]
        sch.inflight, sch._deferred, sch.ring_full, sch.copy_stream, sch.asr_stream = deque(), {}, 0, None, None
        sch.fixed_chunks, sch.frontends, sch.audio_processor = fixed_chunks, [None, None], None
        sch._evq, sch._wake, sch._waiter, sch.waiter_errors = _q.SimpleQueue(), threading.Event(), None, []
        sch.steps = sch.sessions_served = 0
        sch.busy_s, sch._busy_until = 0.0, 0.0
        return sch

    def boom(chunks, only=None):
        raise RuntimeError("step failed")

    # (a) the step itself raises
    sch = make(boom, fixed_chunks="chunks")
    with pytest.raises(RuntimeError, match="step failed"):
        sch.run_once(now=1.0)
    assert all(not r.open and not r.begun for r in sch.rings)                     # no reservation left open
    assert [q[0][0] for q in sch.queues] == [0.1, 0.2] and len(sch.queues[0]) == 2   # both batches back at the head of their queues, order kept
    # (b) the Whisper stage raises (no device here: torch.cuda.current_stream is the first thing it touches)
    sch = make(boom, fixed_chunks=None)
    with pytest.raises(Exception):
        sch.run_once(now=1.0)
    assert all(not r.open for r in sch.rings) and [q[0][0] for q in sch.queues] == [0.1, 0.2]
    # (c) the second ring's begin_batch raises after the first one's succeeded: the begun token is aborted, the other reservation returned
    def ok_step(chunks, only=None):
        return {k: (None, [0]) for k in only}
    sch = make(ok_step, fixed_chunks="chunks")
    def bad_begin(fr, idx, stream=None, reserved=None):
        sch.rings[1].unreserve(reserved)                                             # (FrameRing.begin_batch returns its reservation itself when it fails)
        raise RuntimeError("ring torn down")
    sch.rings[1].begin_batch = bad_begin
    import torch
    real_event, real_cur = torch.cuda.Event, torch.cuda.current_stream
    torch.cuda.Event = lambda *a, **k: SimpleNamespace(record=lambda *_: None, query=lambda: True, synchronize=lambda: None)
    torch.cuda.current_stream = lambda *a, **k: None
    try:
        with pytest.raises(RuntimeError, match="ring torn down"):
            sch.run_once(now=1.0)
    finally:
        torch.cuda.Event, torch.cuda.current_stream = real_event, real_cur
    assert sch.rings[0].aborted == sch.rings[0].begun and len(sch.rings[0].aborted) == 1 and not sch.rings[1].open
    assert sch._waiter is None                                                       # no step succeeded: no waiter thread was ever started

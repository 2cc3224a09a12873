"""Per-batch glue of the MuseTalk render loop, restated for the GPU path (8f rank 2).

The reference runs this logic inside `MuseASR.run_step` (museasr.py:15-29) and `inference()` (musereal.py:53-122) around mp.Queues, one
process per session, each with its own model copy.  The drop-in keeps those files untouched; this module is the queue-free equivalent
that bench.py, the tests and a multi-session harness drive directly:

  MuseASRFrontend   MuseASR.run_step: sliding audio window -> Whisper features -> one (50, 384) chunk per video frame, never leaving HBM
  MuseSession       one avatar: cached latents (and, optionally, full frames + paste geometry) resident on the device, ping-pong index
  MuseBatcher       ONE UNet + VAE pair serving N sessions: every step gathers each active session's latents by mirror index
                    (mf_gather_rows_f32) and its audio chunks into one N * B-frame batch, runs musereal.py:100-108 once, and hands every
                    session its own uint8 frames (optionally already pasted into the full frame, mf_paste_frames)
  SessionScheduler  the per-GPU serving loop over a MuseBatcher for sessions that arrive on their own clocks (one batch of B frames per
                    B * 40 ms of audio each, musereal.py:53-58 / basereal's 25 fps pacing): queues every session's batches, picks which
                    sessions share the next step (oldest first, up to the handle's capacity; a short hold lets a step fill), and stamps
                    every batch's arrival -> frames-ready latency.  bench.py's `paced_sessions` leg measures BASELINE.json's "max concurrent
                    >= 25 fps sessions" with it (p99 batch latency <= B * 40 ms) instead of dividing a free-running rate by 25

Cross-session batching is what fills an MI355X: the UNet at 8 frames per step is launch-latency bound, at 64 it is not."""
import ctypes as C
import queue
import threading
import weakref
import time
from collections import deque

import numpy as np
import torch

from . import _lib


def mirror_index(size, index):
    """`__mirror_index` (musereal.py:44-50, basereal.py:133-139): ping-pong walk over the cached frames."""
    turn, res = divmod(index, size)
    return res if turn % 2 == 0 else size - res - 1


def chunk_left_rows(batch_size, fps, start, audio_feat_length=(2, 2)):
    """First (unclamped) feature row of every chunk of one run_step: `center_idx = int(vid_idx * 50 / fps)`, `left_idx = center_idx -
    audio_feat_length[0] * 2` with vid_idx = i + start (audio2feature.py:29-31, 93-96)."""
    return [int((i + start) * 50 / fps) - audio_feat_length[0] * 2 for i in range(batch_size)]


def feature_chunks_device(feat, left_rows, rows_per_chunk=10, out=None):
    """Audio2Feature.feature2chunks on the device: feat [T, L+1, C] fp32 -> [B, rows_per_chunk * (L+1), C] (= [B, 50, 384])."""
    if not feat.is_cuda:
        raise RuntimeError("feature_chunks_device needs the Whisper features on the HIP device; no CPU path exists here")
    feat = feat.float().contiguous()
    T, L1, Cc = feat.shape
    B = len(left_rows)
    if out is None:
        out = torch.empty((B, rows_per_chunk * L1, Cc), dtype=torch.float32, device=feat.device)
    rows = (C.c_int * B)(*[int(r) for r in left_rows])
    with torch.cuda.device(feat.device):
        _lib.check(_lib.lib().mf_whisper_feature_chunks(feat.data_ptr(), T, L1 * Cc, rows, rows_per_chunk, B, out.data_ptr(),
                                                        C.c_void_p(torch.cuda.current_stream(feat.device).cuda_stream)), "whisper_feature_chunks")
    return out


class MuseASRFrontend:
    """MuseASR.run_step without the queues: 2B new 20 ms chunks in, B Whisper chunks [B, 50, 384] out (on the device)."""

    def __init__(self, audio_processor, batch_size, fps=50, stride_left=10, stride_right=10):
        self.audio_processor, self.batch_size, self.fps = audio_processor, batch_size, fps
        self.l, self.r = stride_left, stride_right
        self.frames = []

    def warm_up(self, chunk=320):
        """baseasr.py:53-59: prime the context with l + r silent chunks."""
        self.frames = [np.zeros(chunk, dtype=np.float32) for _ in range(self.l + self.r)]

    def window(self, new_chunks):
        """museasr.py:17-29 up to the encoder call: take the 2B new 20 ms chunks, return the (l + 2B + r) x 320-sample window the encoder sees
        (None while the context is still filling, museasr.py:22-23) and keep the last l + r chunks as the next window's context."""
        self.frames.extend(new_chunks)
        if len(self.frames) <= self.l + self.r:                       # museasr.py:22-23
            return None
        win = np.concatenate(self.frames)                             # museasr.py:25
        self.frames = self.frames[-(self.l + self.r):]                # museasr.py:29
        return win

    def chunks_from_features(self, feat, out=None):
        """museasr.py:27: the B (50, 384) chunks of this step from the window's features [T, 5, 384] (device)."""
        return feature_chunks_device(feat, chunk_left_rows(self.batch_size, self.fps / 2, self.l / 2), out=out)

    def run_step(self, new_chunks, out=None):
        win = self.window(new_chunks)
        if win is None:
            return None
        return self.chunks_from_features(self.audio_processor.audio2feat_device(win), out=out)   # museasr.py:25-27


class MuseSession:
    """One talking-head session: the avatar's cached latents (musereal.py:64 `torch.load(latents.pt)`: a list of [1, 8, 32, 32] tensors)
    and its position in the ping-pong walk.  `avatar_frames` (mere_fusion_amd.paste.AvatarFrames) enables the GPU paste-back."""

    def __init__(self, latents, avatar_frames=None):
        lat = latents if torch.is_tensor(latents) else torch.cat([torch.as_tensor(x).reshape(1, *torch.as_tensor(x).shape[-3:]) for x in latents], dim=0)
        self.latents = lat.float().contiguous()
        self.length = self.latents.shape[0]
        self.avatar_frames = avatar_frames
        if avatar_frames is not None and avatar_frames.n != self.length:
            raise RuntimeError("one cached full frame per cached latent is required (frame_list_cycle / input_latent_list_cycle)")
        self.index = 0
        self.pool_offset = None

    def next_indices(self, n):
        idx = [mirror_index(self.length, self.index + i) for i in range(n)]
        self.index += n
        return idx


class MuseBatcher:
    """N sessions through one UNet / VAE handle per step (BASELINE.json north star: 8 sessions per GPU)."""

    def __init__(self, unet, vae, sessions, batch_size=8, paste=False, device="cuda", max_sessions_per_step=None):
        if not torch.cuda.is_available():
            raise RuntimeError("MuseBatcher needs a HIP device; no CPU path exists here")
        self.unet, self.vae, self.sessions, self.batch_size, self.paste = unet, vae, list(sessions), batch_size, paste
        self.device = torch.device(device)
        # more sessions than one step holds: step(..., only=[...]) serves a subset (SessionScheduler)
        self.max_sessions_per_step = len(self.sessions) if max_sessions_per_step is None else int(max_sessions_per_step)
        need = self.max_sessions_per_step * batch_size
        for h, what in ((unet.model.max_batch, "UNet"), (getattr(vae, "max_batch", need), "VAE")):
            if h < need:
                raise RuntimeError(f"{what} handle was created with max_batch {h}; {self.max_sessions_per_step} sessions x {batch_size} frames need {need}")
        off = 0
        for s in self.sessions:
            s.pool_offset = off
            off += s.length
        # every session's cached latents in one pool: a batch is one gather
        self.pool = torch.cat([s.latents.reshape(s.length, -1) for s in self.sessions], dim=0).to(self.device).contiguous()
        self.row_elems = self.pool.shape[1]
        self.lat_shape = tuple(self.sessions[0].latents.shape[1:])
        self.t0 = torch.tensor([0], device=self.device)

    @torch.no_grad()
    def prewarm(self, tune=False):
        """Everything a step size costs the FIRST time -- the eager forward that sizes the split-K workspaces, the hipGraph capture, and (tune=True)
        the explicit launch-configuration measurement (mf_unet_tune / mf_vae_tune, seconds per size) -- for every number of sessions a step can hold,
        so that the serving loop never meets a new batch size.  Session state (frame indices) is left untouched.  Call once at start-up."""
        B = self.batch_size
        zeros = torch.zeros((self.max_sessions_per_step * B, 50, self.unet.model.config["cross_attention_dim"]), dtype=torch.float32, device=self.device)
        lat = torch.zeros((self.max_sessions_per_step * B,) + self.lat_shape, dtype=torch.float32, device=self.device)
        lat.copy_(self.pool[:1].reshape((1,) + self.lat_shape).expand_as(lat))
        for k in range(1, self.max_sessions_per_step + 1):
            n = k * B
            for it in range(3 if tune else 2):                         # eager (+ table lookup), [tune + eager], capture
                pred = self.unet.model(lat[:n], self.t0, encoder_hidden_states=self.unet.pe(zeros[:n])).sample
                self.vae.decode_latents_device(pred)
                if tune and it == 0:
                    self.unet.model.tune(n)
                    self.vae.tune(n)
        torch.cuda.synchronize(self.device)

    @torch.no_grad()
    def step(self, whisper_chunks, only=None):
        """whisper_chunks: one entry per session -- a device tensor [B, 50, 384] (MuseASRFrontend.run_step) or None for an all-silent batch
        (musereal.py:82-86: the net is skipped, only the frame indices advance).  Returns one (frames, indices) per session:
        frames = uint8 [B, 256, 256, 3] BGR on the device (`recon`, musereal.py:108), or, with paste=True, the composed full frames
        [B, H, W, 3] (musereal.py:238-247); None for a silent session.
        only: session numbers that take part in this step; every other session is left untouched (its frame index does not move, its entry
        of the result is None) -- sessions on their own clocks do not all have a batch at every step."""
        B = self.batch_size
        take = None if only is None else set(int(k) for k in only)
        if take is not None and (min(take, default=0) < 0 or max(take, default=0) >= len(self.sessions)):
            raise RuntimeError(f"only={sorted(take)}: session numbers run from 0 to {len(self.sessions) - 1}")
        n_act = sum(1 for k, ch in enumerate(whisper_chunks) if ch is not None and (take is None or k in take))
        if n_act > self.max_sessions_per_step:
            raise RuntimeError(f"{n_act} active sessions in one step; the handles hold {self.max_sessions_per_step} x {B} frames")
        for k, ch in enumerate(whisper_chunks):                          # every input is checked BEFORE any session's frame index moves
            if ch is not None and (take is None or k in take) and (ch.shape[0] != B or not ch.is_cuda):
                raise RuntimeError(f"session {k}: expected a device tensor of {B} whisper chunks, got {tuple(ch.shape)} on {ch.device}")
        rows, idx_per, active = [], [], []
        for k, (s, ch) in enumerate(zip(self.sessions, whisper_chunks)):
            if take is not None and k not in take:
                idx_per.append(None)
                continue
            idx = s.next_indices(B)
            idx_per.append(idx)
            if ch is None:
                continue
            active.append(k)
            rows.extend(s.pool_offset + i for i in idx)
        out = [None if idx is None else (None, idx) for idx in idx_per]
        if not active:
            return out
        n = len(rows)
        lat = torch.empty((n,) + self.lat_shape, dtype=torch.float32, device=self.device)
        crows = (C.c_int * n)(*rows)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mf_gather_rows_f32(self.pool.data_ptr(), self.pool.shape[0], self.row_elems, crows, n, lat.data_ptr(),
                                                     C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "gather_rows_f32")
        aud = whisper_chunks[active[0]] if len(active) == 1 else torch.cat([whisper_chunks[k] for k in active], dim=0)
        pred = self.unet.model(lat, self.t0, encoder_hidden_states=self.unet.pe(aud)).sample              # musereal.py:102-107
        frames = self.vae.decode_latents_device(pred)                                                   # musereal.py:108, frames stay in HBM
        for j, k in enumerate(active):
            fr = frames[j * B:(j + 1) * B]
            s = self.sessions[k]
            if self.paste:
                if s.avatar_frames is None:
                    raise RuntimeError(f"session {k} has no AvatarFrames to paste into")
                fr = s.avatar_frames.paste(fr, idx_per[k])
            out[k] = (fr, idx_per[k])
        return out


def pick_sessions(pending, now, capacity, hold):
    """Which sessions share the next step.  pending: {session number: arrival time of its OLDEST queued batch}.  Oldest first (ties: lower
    session number); a step goes out as soon as `capacity` sessions wait, or once the oldest batch has waited `hold` seconds -- a partly
    filled step costs almost what a full one does (the UNet is latency-bound at 8 frames), so a short hold buys throughput and
    its cost is bounded by `hold`.  Returns [] while it is better to wait.  Pure host logic (tests/test_muse_driver.py runs it on the CPU)."""
    if not pending or capacity < 1:
        return []
    order = sorted(pending, key=lambda k: (pending[k], k))
    if len(order) >= capacity or now - pending[order[0]] >= hold:
        return order[:capacity]
    return []


class SessionScheduler:
    """Serving loop of one GPU: N paced sessions over one MuseBatcher.

    submit(k, chunks, t_arrival)   queue one batch of session k (chunks: device tensor [B, 50, 384], or None = silent batch)
    run_once(now)                  if pick_sessions says so, run ONE batcher step for the picked sessions, wait for its frames and
                                   return [(k, frames, indices, latency_s)]; [] when nothing is due
    next_due()                     the time at which run_once would act on what is queued now (None: nothing queued)
    A session's batches are served in arrival order, one per step (its frame indices are consecutive: musereal.py:92-97)."""

    def __init__(self, batcher, period_s=None, hold_s=None, clock=time.perf_counter, sync=None):
        self.batcher = batcher
        self.capacity = batcher.max_sessions_per_step
        self.period = batcher.batch_size * 0.040 if period_s is None else float(period_s)          # B frames at 25 fps
        self.hold = self.period / 4 if hold_s is None else float(hold_s)
        self.clock = clock
        self.sync = sync if sync is not None else (lambda: torch.cuda.synchronize(batcher.device))
        self.queues = [deque() for _ in batcher.sessions]
        self.steps = 0
        self.sessions_served = 0
        self.busy_s = 0.0

    def submit(self, k, whisper_chunks, t_arrival=None):
        self.queues[k].append((self.clock() if t_arrival is None else t_arrival, whisper_chunks))

    def pending(self):
        return {k: q[0][0] for k, q in enumerate(self.queues) if q}

    def backlog(self):
        return max((len(q) for q in self.queues), default=0)

    def next_due(self):
        p = self.pending()
        if not p:
            return None
        t = sorted(p.values())
        return t[self.capacity - 1] if len(t) >= self.capacity else t[0] + self.hold

    def run_once(self, now=None):
        now = self.clock() if now is None else now
        ks = pick_sessions(self.pending(), now, self.capacity, self.hold)
        if not ks:
            return []
        chunks = [None] * len(self.queues)
        arrival = {}
        for k in ks:
            arrival[k], chunks[k] = self.queues[k].popleft()
        t0 = self.clock()
        out = self.batcher.step(chunks, only=ks)
        self.sync()
        t1 = self.clock()
        self.steps += 1
        self.sessions_served += len(ks)
        self.busy_s += t1 - t0
        return [(k, out[k][0], out[k][1], t1 - arrival[k]) for k in ks]


class EndToEndScheduler(SessionScheduler):
    """The whole per-GPU session loop in one place: what reaches a session's `process_frames` thread, from what its ASR thread saw.

      museasr.py:15-29    every session's 2B new 20 ms PCM chunks -> its sliding window; the windows of ALL sessions picked for a step go through the
                          Whisper encoder in ONE call (mf_whisper_encode_windows), then `feature2chunks` per session on the device
      musereal.py:91-108  MuseBatcher.step for the picked sessions (gather -> UNet -> VAE -> uint8 frames)
      musereal.py:238-247 paste-back into the cached full frames on the device (`paste` stage; the batcher must have been built with paste=True)
      musereal.py:116,153 each session's frames leave through ITS FrameRing as (res_frame, idx, audio_frames) tuples: the D2H runs on a copy stream
                          behind the step (the next step's kernels do not wait for it) and the descriptors are published when its event is done

    submit(k, pcm_chunks, t_arrival)  the 2B chunks of session k that completed a batch at t_arrival (audio_frames = [(chunk, 0)] * 2B)
    run_once(now) -> finished batches [(k, frames_or_None, indices, latency_s)]: latency = arrival -> descriptors published (ring stage on) or frames
    complete in HBM (ring stage off).  Stages can be switched off individually to price them (bench.py `paced_sessions.stages`)."""

    def __init__(self, batcher, frontends, audio_processor, rings=None, period_s=None, hold_s=None, clock=time.perf_counter, depth=2, fixed_chunks=None,
                 asr_stream=True, single_stream=False):
        super().__init__(batcher, period_s=period_s, hold_s=hold_s, clock=clock)
        self.fixed_chunks = fixed_chunks                             # (measurement only: this [B, 50, 384] tensor instead of the Whisper stage)
        if len(frontends) != len(batcher.sessions):
            raise RuntimeError("one MuseASRFrontend per session is required")
        self.frontends, self.audio_processor, self.rings = list(frontends), audio_processor, rings
        self.depth = max(int(depth), 1)                              # steps in flight: the one computing + the one whose frames are being copied
        # single_stream: the Whisper call, the step and the D2H of its frames all on the caller's stream -- no cross-stream event wait anywhere (with handles
        # created under MF_NO_GRAPH=2 the whole rank is one launch chain: no runtime thread spins, see mf_musetalk.hip); costs the copy / Whisper overlap
        self.single_stream = bool(single_stream)
        if self.single_stream:
            asr_stream = False
        self.copy_stream = torch.cuda.Stream(device=batcher.device) if rings is not None and not self.single_stream else None
        # The Whisper front-end of a step on its OWN stream: with two steps in flight it runs beside the previous step's VAE instead of between that VAE and
        # this step's UNet.  Its ~25 launches are latency-bound (1.5 ms for seven windows with the GPU mostly idle); beside the VAE's full-chip kernels they cost
        # the step almost nothing.  asr_stream=False: on the step's stream, as before (A/B, bench `stages`).
        self.asr_stream = torch.cuda.Stream(device=batcher.device) if asr_stream else None
        self.inflight = deque()
        # The serving loop SLEEPS between steps (polling run_once at 2 - 5 kHz was part of the 1.3 - 1.9 host cores a rank cost at
        # capacity): every step's event goes to one waiter thread that does nothing but hipEventQuery it every 0.5 ms and then sets `_wake`; the loop sleeps
        # in idle_wait() until then or until the next arrival is due.  (Not hipEventSynchronize: on ROCm 7.2 it spins a full core for as long as the GPU is
        # busy, with or without the blocking-sync event flag, and hipDeviceScheduleBlockingSync hangs on this driver: tools/host_wait_probe.py.)
        self._wake = threading.Event()
        self._evq = queue.SimpleQueue()
        self._waiter = None                                          # started by the first step: a scheduler that never steps owns no thread
        self.waiter_errors = []                                      # exceptions hipEventQuery raised in the waiter (a device error is not a completion)
        self.ring_full = 0                                           # deferral EPISODES (a session found its ring full), not polls
        self._deferred = {}                                          # session -> time at which it is offered again even if its ring still looks full
        self._busy_until = 0.0

    @staticmethod
    def _wait_loop(evq, wake, errors):
        """The waiter holds the queue, the wake flag and the error list -- NOT the scheduler: an abandoned scheduler (no close()) is collected with its batcher,
        rings and GPU handles, and its finaliser below stops this thread."""
        while True:
            ev = evq.get()
            if ev is None:
                return
            try:
                while not ev.query():
                    time.sleep(5e-4)
            except Exception as e:                                   # recorded, then the loop is woken: _retire's own query raises the same error to the caller
                errors.append(e)
            wake.set()

    def _start_waiter(self):
        if self._waiter is None:
            self._waiter = threading.Thread(target=EndToEndScheduler._wait_loop, args=(self._evq, self._wake, self.waiter_errors), daemon=True)
            self._waiter.start()
            weakref.finalize(self, self._evq.put, None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def idle_wait(self, timeout):
        """Sleeps until a step in flight completes or `timeout` seconds have passed, whichever is first (nothing in flight: a plain sleep).  What a serving
        loop calls between run_once()s instead of polling."""
        if timeout is None or timeout <= 0:
            return
        if self.inflight:
            self._wake.wait(timeout)
            self._wake.clear()
        else:
            time.sleep(timeout)

    def close(self):
        """stops the waiter thread (after drain(); `with EndToEndScheduler(...) as sch:` calls it)"""
        if self._waiter is not None:
            self._evq.put(None)
            self._waiter.join(timeout=2.0)
            self._waiter = None

    def submit(self, k, pcm_chunks, t_arrival=None):
        """pcm_chunks: the batch's 2B 20 ms chunks -- bare arrays (all speech, type 0) or (chunk, type) pairs exactly as `get_audio_frame` hands them out
        (baseasr.py:33-45; type 1 = silence).  An all-silent batch skips the networks, as musereal.py:82-86 does: its B (None, idx, audio_frames) tuples still
        reach the session's ring so that `process_frames` keeps audio and idle video in step."""
        t = self.clock() if t_arrival is None else t_arrival
        pairs = [(c if isinstance(c, tuple) else (c, 0)) for c in pcm_chunks]
        win = self.frontends[k].window([c for c, _ in pairs])         # host side of museasr.py:17-29, at arrival time (the window slides for silent batches too)
        if all(ty != 0 for _, ty in pairs):
            win = None
        self.queues[k].append((t, (win, pairs)))

    def _retire(self, block=False):
        done = []
        while self.inflight:
            item = self.inflight[0]
            if block:
                item["event"].synchronize()
            elif not item["event"].query():
                break
            self.inflight.popleft()
            t1 = self.clock()
            for k in item["ks"]:
                fr, idx = item["out"][k]
                tok = item["tokens"].get(k)
                if tok is not None:
                    # speech: the B frames' descriptors; a silent batch or a session whose context is still filling: B (None, idx, audio_frames[2i:2i+2])
                    # tuples as the reference puts them (musereal.py:82-86, lipreal.py:104) -- ONE message either way, on slots reserved before the step
                    # started, so nothing here can block the scheduler thread
                    self.rings[k].commit_batch(tok, item["audio"][k])
                    t1 = self.clock()
                done.append((k, fr, idx, t1 - item["arrival"][k]))
            self.busy_s += max(t1 - max(item["t0"], self._busy_until), 0.0)      # union of the steps' [launch, done] intervals
            self._busy_until = max(self._busy_until, t1)
        return done

    def pending(self):
        """Sessions whose oldest batch can be picked now.  A session deferred because its ring was full is left out until the ring reports B free slots or a
        back-off of a quarter period has passed (it otherwise sits at the head of the order with an arrival time in the past, next_due() lies in the
        past, the serving loop spins a host core on run_once, and its age makes every other session's step launch partly filled)."""
        p = super().pending()
        if self._deferred:
            now, B = self.clock(), self.batcher.batch_size
            for k in list(self._deferred):
                if k not in p:
                    del self._deferred[k]
                elif self.rings[k].free_slots() >= B or now >= self._deferred[k]:
                    del self._deferred[k]                            # (a new episode is counted if it is deferred again)
                else:
                    del p[k]
        return p

    def next_due(self):
        """when run_once would next act on what is QUEUED (None: nothing queued).  Steps in flight do not enter: idle_wait() wakes the loop when one completes."""
        t = super().next_due()
        if self._deferred:
            t_def = min(self._deferred.values())
            t = t_def if t is None else min(t, t_def)
        return t

    def run_once(self, now=None):
        now = self.clock() if now is None else now
        done = self._retire()
        if len(self.inflight) >= self.depth:
            return done
        ks = pick_sessions(self.pending(), now, self.capacity, self.hold)
        if not ks:
            return done
        dev = self.batcher.device
        # Ring slots for every picked session are taken BEFORE anything irreversible happens (queue entries popped, frame indices and ASR state advanced): a
        # session whose consumer is behind is DEFERRED -- its batch stays at the head of its queue and is picked again later -- and the other sessions go ahead;
        # the reference's loop likewise back-pressures only the one session's queue.
        reserved = {}
        if self.rings is not None:
            B = self.batcher.batch_size
            pend = self.pending()
            ok = []
            try:
                for k in sorted(pend, key=lambda k_: (pend[k_], k_)):     # oldest first; a waiting session takes the place of one that has to be deferred
                    if len(ok) == len(ks):
                        break
                    sl = self.rings[k].try_reserve(B)                     # (silent batches too: their B (None, idx, audio) tuples take B of the ring's places)
                    if sl is None:
                        self.ring_full += 1
                        self._deferred[k] = now + self.period / 4
                        continue
                    reserved[k] = sl
                    ok.append(k)
            except BaseException:
                for k in list(reserved):                                  # a ring that raised (closed, torn down) must not leave the others' places open
                    self.rings[k].unreserve(reserved[k])
                raise
            ks = ok
            if not ks:
                return done
        # Everything from here to the last begin_batch sits in ONE try: the rings publish in begin order, so a reservation left open by an exception anywhere
        # on the way (the queue pop, the H2D copy, the Whisper call, the step, a begin_batch) would wedge its ring for good -- every later batch queued behind
        # an entry that never completes.  On any exception the reservations go back and the popped batches return to the HEAD of their queues.
        arrival, audio, wins, tokens, popped = {}, {}, {}, {}, []
        try:
            for k in ks:
                item = self.queues[k].popleft()
                popped.append((k, item))
                arrival[k], (wins[k], audio[k]) = item
            t0 = self.clock()
            chunks = [None] * len(self.queues)
            speaking = [k for k in ks if wins[k] is not None]
            if speaking and self.fixed_chunks is not None:
                for k in speaking:
                    chunks[k] = self.fixed_chunks
            elif speaking:
                cur = torch.cuda.current_stream(dev)
                side = self.asr_stream if self.asr_stream is not None else cur
                with torch.cuda.stream(side):
                    wav = torch.from_numpy(np.stack([wins[k] for k in speaking])).to(dev, non_blocking=True)
                    feats = self.audio_processor.audio2feat_windows_device(wav)        # every picked session's window in one encoder call
                    for i, k in enumerate(speaking):
                        chunks[k] = self.frontends[k].chunks_from_features(feats[i])
                if side is not cur:
                    cur.wait_stream(side)                                              # the UNet below reads the chunks
                    for t in [wav, feats] + [chunks[k] for k in speaking]:
                        t.record_stream(cur)
            out = self.batcher.step(chunks, only=ks)
            ev = torch.cuda.Event()
            if self.rings is not None:
                cur = torch.cuda.current_stream(dev)
                if self.copy_stream is not None:
                    self.copy_stream.wait_stream(cur)
                for k in ks:
                    fr, idx = out[k]
                    tokens[k] = self.rings[k].begin_batch(fr, idx, stream=self.copy_stream, reserved=reserved.pop(k))   # (fr None: B silent frames)
                    if fr is not None and self.copy_stream is not None:
                        fr.record_stream(self.copy_stream)
                ev.record(self.copy_stream if self.copy_stream is not None else cur)
        except BaseException:
            # nothing of this step is published: tokens already begun and reservations not yet used go back (the rings publish in begin order whatever order
            # this happens in)
            for k in list(tokens):
                self.rings[k].abort_batch(tokens[k])
            for k in list(reserved):
                self.rings[k].unreserve(reserved[k])
            for k, item in reversed(popped):                       # the batches of this step are not lost: they are picked again
                self.queues[k].appendleft(item)
            raise
        if self.rings is None:
            ev.record(torch.cuda.current_stream(dev))
        self.inflight.append({"ks": ks, "out": out, "tokens": tokens, "audio": audio, "arrival": arrival, "event": ev, "t0": t0})
        self._start_waiter()
        self._evq.put(ev)
        self.steps += 1
        self.sessions_served += len(ks)
        return done

    def drain(self):
        """Waits for everything in flight (end of a run)."""
        return self._retire(block=True)

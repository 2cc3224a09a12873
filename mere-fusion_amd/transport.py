"""Frame transport between the inference process and the render thread.

The reference hands every generated frame to `process_frames` as a pickled tuple through `mp.Queue`:
`res_frame_queue.put((res_frame, idx, audio_frames))` (lipreal.py:136, musereal.py:116) on a queue of `batch_size * 2` items
(lipreal.py:161, musereal.py:153), read back with `.get(block=True, timeout=1)` (lipreal.py:195, musereal.py:226).  A 96 x 96 x 3 float32
frame is 110 KB and a 256 x 256 x 3 uint8 frame 196 KB -- pickled, written to a pipe, read, unpickled: four copies and two syscalls per
frame, 64 sessions x 25 fps of them per node.

`FrameRing` keeps the same contract -- `put((res_frame | None, idx, audio_frames))`, `get(block, timeout)` returning the same tuple,
`queue.Empty` / `queue.Full` on time-out, `qsize()` -- but the frame BYTES live in one `multiprocessing.shared_memory` block of fixed
slots and only a small descriptor `(slot, shape, dtype, idx, audio_frames)` goes through an `mp.Queue`.  On the GPU side the producer
page-locks the block once (`mf_host_register`), so a device tensor is copied by ONE asynchronous DMA straight into its slot
(`mf_copy_d2h_async` on the caller's stream; the slot is published after `mf_stream_synchronize`).  `put_batch` moves a whole batch of frames
with a single copy when the slots are contiguous and publishes its descriptors as ONE queue message.  Single producer, single consumer --
the shape of the reference's loop.

There is no GPU requirement for host frames (the ring then simply replaces the pickling); device tensors need the HIP library."""
import ctypes as C
import multiprocessing as mp
import queue
import time
from collections import deque
from multiprocessing import shared_memory

import numpy as np


class FrameRing:
    def __init__(self, slots, frame_shape, dtype=np.uint8, ctx=None):
        """slots: ring capacity (the reference uses batch_size * 2); frame_shape / dtype: the largest frame a slot must hold."""
        ctx = ctx or mp.get_context("spawn")                       # app.py:549 sets the spawn start method
        self.slots, self.frame_shape, self.dtype = int(slots), tuple(int(v) for v in frame_shape), np.dtype(dtype)
        self.slot_bytes = int(np.prod(self.frame_shape)) * self.dtype.itemsize
        self.slot_stride = (self.slot_bytes + 4095) // 4096 * 4096       # page-aligned slots: registrable, no false sharing
        self._shm = shared_memory.SharedMemory(create=True, size=self.slot_stride * self.slots)
        self._name = self._shm.name
        self._owner = True
        self._desc = ctx.Queue(self.slots)                         # descriptors only: the bound of the reference's queue
        self._free = ctx.Semaphore(self.slots)                     # free slots
        self._head = 0                                             # producer-side cursor (single producer)
        self._registered = False
        self._inbox = deque()                                      # consumer side: descriptors of the message being unpacked

    # ---- pickling: a child process re-attaches to the same block ---------------------------------------------------------------
    def __getstate__(self):
        d = self.__dict__.copy()
        d["_shm"] = None
        d["_owner"] = False
        d["_registered"] = False
        d["_inbox"] = deque()
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._shm = shared_memory.SharedMemory(name=self._name)

    def _slot_view(self, slot, shape=None, dtype=None):
        shape = self.frame_shape if shape is None else tuple(shape)
        dtype = self.dtype if dtype is None else np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        if n > self.slot_bytes:
            raise ValueError(f"frame of {n} bytes does not fit a {self.slot_bytes}-byte slot")
        return np.ndarray(shape, dtype=dtype, buffer=self._shm.buf, offset=slot * self.slot_stride)

    def _slot_ptr(self, slot):
        return C.addressof(C.c_char.from_buffer(self._shm.buf, slot * self.slot_stride))

    # ---- producer -----------------------------------------------------------------------------------------------------------
    def register_pinned(self):
        """Page-locks the block for DMA (producer process, once).  Needs the HIP library and a device."""
        if not self._registered:
            from . import _lib
            _lib.check(_lib.lib().mf_host_register(C.c_void_p(self._slot_ptr(0)), self.slot_stride * self.slots), "host_register")
            self._registered = True

    def _acquire(self, n, block, timeout):
        """n consecutive ring slots, all or nothing: a time-out part-way hands back what it took and leaves the cursor where it was (a
        failed put must neither leak slots nor let a later put overwrite an unread one)."""
        if n > self.slots:
            raise ValueError(f"a batch of {n} frames can never fit a ring of {self.slots} slots")
        deadline = None if (timeout is None or not block) else time.monotonic() + timeout
        got = 0
        while got < n:
            left = None if deadline is None else max(deadline - time.monotonic(), 0.0)
            if not self._free.acquire(block, left):
                for _ in range(got):
                    self._free.release()
                raise queue.Full
            got += 1
        first = self._head
        self._head = (self._head + n) % self.slots
        return [(first + i) % self.slots for i in range(n)]

    def try_reserve(self, n):
        """n consecutive slots if they are free RIGHT NOW, else None -- for a producer that must know every destination has room BEFORE it does work it
        cannot undo (EndToEndScheduler.run_once reserves for every picked session before the step advances frame indices and ASR state).  Pass the result
        to begin_batch(..., reserved=slots), or hand it back with unreserve(slots)."""
        try:
            return self._acquire(n, False, None)
        except queue.Full:
            return None

    def unreserve(self, slots):
        """hands back slots from try_reserve / a token's slots that were never published (see abort_batch for the ordering rule)"""
        self.abort_batch({"slots": list(slots)})

    def _check_fits(self, nbytes):
        if nbytes > self.slot_bytes:
            raise ValueError(f"frame of {nbytes} bytes does not fit a {self.slot_bytes}-byte slot")

    @staticmethod
    def _is_device(x):
        try:
            import torch
        except ImportError:
            return False
        return torch.is_tensor(x) and x.is_cuda

    def _dma(self, t, first_frame, slots, stream, lib):
        """frames t[first_frame ...] -> `slots` (consecutive): one linear copy when the slots abut (a page-multiple frame), else one pitched copy."""
        per = t[0].numel() * t.element_size()
        src, dst, n = C.c_void_p(t.data_ptr() + first_frame * per), C.c_void_p(self._slot_ptr(slots[0])), len(slots)
        if n == 1 or per == self.slot_stride:
            return lib.mf_copy_d2h_async(src, dst, per * n, stream)
        return lib.mf_copy_d2h_2d_async(src, per, dst, self.slot_stride, per, n, stream)

    def put(self, item, block=True, timeout=None):
        """item = (res_frame, idx, audio_frames) exactly as lipreal.py:136 / musereal.py:116 put it; res_frame is None for an all-silent chunk
        (lipreal.py:104), a numpy array, or a HIP device tensor (copied by DMA into the slot)."""
        frame, idx, audio = item
        if frame is None:
            self._desc.put([(None, None, None, idx, audio)], block, timeout)
            return
        if self._is_device(frame):
            self.put_batch(frame[None], [idx], audio, block, timeout, _single_audio=(audio,))     # (wrapped: `audio` itself may be None)
            return
        a = np.asarray(frame)
        self._check_fits(a.nbytes)                                  # everything that can fail, before a slot is taken
        slot = self._acquire(1, block, timeout)[0]
        self._slot_view(slot, a.shape, a.dtype)[...] = a
        self._desc.put([(slot, a.shape, a.dtype.str, idx, audio)])

    def put_batch(self, frames, idxs, audio_frames, block=True, timeout=None, _single_audio=None):
        """A whole batch (`for i, res_frame in enumerate(recon): res_frame_queue.put(...)`, musereal.py:116-119): frames [B, ...] on the device
        or the host, idxs the B frame indices, audio_frames the 2B (pcm, type) pairs (two per frame).  A device batch travels as ONE DMA per run
        of consecutive slots (the ring wraps at most once per batch) behind ONE stream fence, and its B descriptors as ONE queue message --
        the per-frame pickle + pipe round trip was what the ring's device path cost in round 2 (0.39 ms per batch of 8 against 0.05 ms of DMA)."""
        tok = self.begin_batch(frames, idxs, block=block, timeout=timeout)
        if tok is None:
            return
        if tok["stream"] is not None:
            from . import _lib
            try:
                _lib.check(_lib.lib().mf_stream_synchronize(tok["stream"]), "stream_synchronize")   # one fence per batch, then the slots are published
            except BaseException:
                self.abort_batch(tok)
                raise
        self.commit_batch(tok, audio_frames, _single_audio=_single_audio)

    def begin_batch(self, frames, idxs, stream=None, block=True, timeout=None, reserved=None):
        """First half of put_batch, for a producer that overlaps the copy with its next step: takes the slots and ENQUEUES the DMA on `stream`
        (a torch stream; default: the current one) without waiting.  Once the caller knows the copy is complete (an event recorded behind it on
        that stream) it calls commit_batch(token, audio_frames); abort_batch(token) hands the slots back instead.  Returns None for an empty batch.
        reserved: slots from try_reserve(len(idxs)) (then nothing here can block).  Tokens are COMMITTED in the order they were begun (the consumer reads
        slots in ring order); see abort_batch for aborting."""
        B = len(idxs)
        if B == 0:
            return None
        is_dev = self._is_device(frames)
        if is_dev:
            import torch
            t = frames.contiguous()
            if stream is not None and t is not frames:
                # the compaction above ran on the CURRENT stream; the DMA below is enqueued on `stream`: order it behind (a silent race otherwise)
                stream.wait_stream(torch.cuda.current_stream(t.device))
            shape, dtype = tuple(t.shape[1:]), np.dtype(str(t.dtype).replace("torch.", ""))
            self._check_fits(t[0].numel() * t.element_size())
        else:
            a = np.asarray(frames)
            shape, dtype = a.shape[1:], a.dtype
            self._check_fits(int(np.prod(shape)) * dtype.itemsize)
        if len(frames) != B:
            raise ValueError(f"{len(frames)} frames for {B} indices")
        if reserved is not None and len(reserved) != B:
            raise ValueError(f"{len(reserved)} reserved slots for {B} frames")
        slots = list(reserved) if reserved is not None else self._acquire(B, block, timeout)
        tok = {"slots": slots, "shape": shape, "dtype": dtype.str, "idxs": list(idxs), "stream": None, "keep": None}
        try:
            if is_dev:
                from . import _lib
                lib = _lib.lib()
                self.register_pinned()
                st = torch.cuda.current_stream(t.device) if stream is None else stream
                cs = C.c_void_p(st.cuda_stream)
                with torch.cuda.device(t.device):
                    i = 0
                    while i < B:
                        j = i + 1
                        while j < B and slots[j] == slots[j - 1] + 1:
                            j += 1
                        _lib.check(self._dma(t, i, slots[i:j], cs, lib), "copy_d2h_async")
                        i = j
                tok["stream"], tok["keep"] = cs, t                   # (the source tensor must outlive the copy)
            else:
                for i, sl in enumerate(slots):
                    self._slot_view(sl, shape, dtype)[...] = a[i]
        except BaseException:
            self.abort_batch(tok)
            raise
        return tok

    def abort_batch(self, tok):
        """Nothing of this batch was published: give the slots back.  If the token is the NEWEST one outstanding (its last slot is the one before the cursor)
        the cursor is rewound and the slots are simply free again.  With newer tokens outstanding (begin_batch / commit_batch allow several in flight) a rewind
        would hand out the newer tokens' slots a second time: the slots are then published as a SKIP descriptor instead -- the consumer releases them in ring
        order without surfacing a frame."""
        sl = tok["slots"]
        if not sl:
            return
        if (sl[-1] + 1) % self.slots == self._head:
            self._head = sl[0]
            for _ in sl:
                self._free.release()
        else:
            self._desc.put([("skip", len(sl), None, None, None)])

    def commit_batch(self, tok, audio_frames, _single_audio=None):
        """Second half of put_batch: publishes the batch's descriptors as one message.  The copy must be complete."""
        sl, shape, dt, idxs = tok["slots"], tok["shape"], tok["dtype"], tok["idxs"]
        tok["keep"] = None
        if _single_audio is not None:
            self._desc.put([(sl[0], shape, dt, idxs[0], _single_audio[0])])
        else:
            self._desc.put([(s_, shape, dt, idxs[i], audio_frames[2 * i:2 * i + 2]) for i, s_ in enumerate(sl)])

    # ---- consumer -----------------------------------------------------------------------------------------------------------
    def get(self, block=True, timeout=None, copy=True):
        """-> (res_frame, idx, audio_frames), the tuple `process_frames` unpacks (lipreal.py:195).  copy=True returns an ndarray the caller
        owns (the slot is free again immediately); copy=False returns a view into the ring and the caller must `release(view)` it."""
        while True:
            if not self._inbox:
                self._inbox.extend(self._desc.get(block, timeout))   # one message = the descriptors of one put / put_batch
            slot, shape, dtype, idx, audio = self._inbox.popleft()
            if isinstance(slot, str):                                # "skip": slots of an aborted batch, released in ring order (abort_batch)
                for _ in range(shape):
                    self._free.release()
                continue
            break
        if slot is None:
            return None, idx, audio
        view = self._slot_view(slot, shape, dtype)
        if copy:
            out = view.copy()
            self._free.release()
            return out, idx, audio
        return view, idx, audio

    def release(self, view=None):
        """Hands a slot obtained with get(copy=False) back to the producer (slots are consumed in order)."""
        self._free.release()

    def qsize(self):
        """frames waiting on the consumer's side (an estimate, like mp.Queue.qsize): unpacked descriptors + whole messages still in the pipe"""
        return len(self._inbox) + self._desc.qsize()

    def empty(self):
        return not self._inbox and self._desc.empty()

    def close(self):
        if self._shm is None:
            return
        if self._registered:
            try:
                from . import _lib
                _lib.lib().mf_host_unregister(C.c_void_p(self._slot_ptr(0)))
            except Exception:
                pass
            self._registered = False
        try:
            self._shm.close()
            if self._owner:
                self._shm.unlink()
        except FileNotFoundError:
            pass
        self._shm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

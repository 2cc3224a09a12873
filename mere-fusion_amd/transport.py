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


class _Slots(list):
    """The slot numbers of one reservation / batch; `.entry` is the ring's own record of it (state, pending message)."""
    entry = None


class FrameRing:
    """Every frame -- a silent one (`res_frame` None) included -- holds one ring slot from its put until the consumer has taken it, so the ring bounds what is
    queued exactly as the reference's `Queue(batch_size * 2)` of per-frame tuples does, every descriptor message carries at least one slot (the bounded descriptor
    queue can therefore never be the thing a producer blocks on: ADVICE r04), and a reservation (`try_reserve`) is all a producer needs to know that nothing
    it does later for that batch can block.  Batches are PUBLISHED in the order they were begun, whatever order they are committed or aborted in: the ring keeps
    its outstanding batches in a deque and releases messages from its head only (a skip descriptor published ahead of an older, still uncommitted
    batch let the producer wrap onto that batch's DMA target)."""

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
        self._views_out, self._held = 0, 0                         # consumer side: get(copy=False) views not yet released / releases held behind them
        self._open = deque()                                       # producer side: outstanding reservations / batches, oldest first

    # ---- pickling: a child process re-attaches to the same block ---------------------------------------------------------------
    def __getstate__(self):
        d = self.__dict__.copy()
        d["_shm"] = None
        d["_owner"] = False
        d["_registered"] = False
        d["_inbox"] = deque()
        d["_open"] = deque()
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
        failed put must neither leak slots nor let a later put overwrite an unread one).  Returns the ring's record of the new batch."""
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
        sl = _Slots((first + i) % self.slots for i in range(n))
        sl.entry = {"slots": sl, "state": "open", "msg": None, "shape": None, "dtype": None, "idxs": None, "stream": None, "keep": None}
        self._open.append(sl.entry)
        return sl.entry

    def _flush(self):
        """Publishes what can be published, oldest batch first: a committed batch's message; an aborted batch's slots -- as a plain rewind when nothing newer
        is outstanding (they are then the slots just before the cursor), else as a skip descriptor the consumer releases in ring order.  Stops at the first
        batch that is still open: nothing newer may overtake it."""
        while self._open and self._open[0]["state"] != "open":
            e = self._open.popleft()
            if e["state"] == "committed":
                self._desc.put(e["msg"])                            # (cannot block: every message holds >= 1 of `slots` slots, the queue holds `slots` messages)
            elif not self._open and e["slots"] and (e["slots"][-1] + 1) % self.slots == self._head:
                self._head = e["slots"][0]
                for _ in e["slots"]:
                    self._free.release()
            elif e["slots"]:
                self._desc.put([("skip", len(e["slots"]), None, None, None)])
        # aborted batches at the NEWEST end rewind at once (their slots are the ones just before the cursor), older ones wait their turn above
        while self._open and self._open[-1]["state"] == "aborted":
            e = self._open.pop()
            if e["slots"] and (e["slots"][-1] + 1) % self.slots == self._head:
                self._head = e["slots"][0]
                for _ in e["slots"]:
                    self._free.release()
            elif e["slots"]:                                        # (cannot happen with a single producer; keep the slots accounted for)
                self._open.append(e)
                break

    def try_reserve(self, n):
        """n consecutive slots if they are free RIGHT NOW, else None -- for a producer that must know every destination has room BEFORE it does work it
        cannot undo (EndToEndScheduler.run_once reserves for every picked session, silent batches included, before the step advances frame indices and ASR
        state).  Pass the result to begin_batch(..., reserved=slots), or hand it back with unreserve(slots)."""
        try:
            return self._acquire(n, False, None)["slots"]
        except queue.Full:
            return None

    def free_slots(self):
        """free slots right now (an estimate on the consumer's side of the fence, exact for the single producer's admission decisions)"""
        try:
            return self._free.get_value()
        except NotImplementedError:                                 # (macOS)
            return self.slots

    def unreserve(self, slots):
        """hands back slots from try_reserve that were never used (in whatever order: the ring publishes in begin order, see _flush)"""
        self.abort_batch(slots.entry if isinstance(slots, _Slots) else {"slots": list(slots)})

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
        (lipreal.py:104: it still takes one of the queue's places, here one slot without a payload), a numpy array, or a HIP device tensor (copied by DMA
        into the slot)."""
        frame, idx, audio = item
        if frame is None:
            tok = self.begin_batch(None, [idx], block=block, timeout=timeout)
            self.commit_batch(tok, None, _single_audio=(audio,))
            return
        if self._is_device(frame):
            self.put_batch(frame[None], [idx], audio, block, timeout, _single_audio=(audio,))     # (wrapped: `audio` itself may be None)
            return
        a = np.asarray(frame)
        self._check_fits(a.nbytes)                                  # everything that can fail, before a slot is taken
        tok = self.begin_batch(a[None], [idx], block=block, timeout=timeout)
        self.commit_batch(tok, None, _single_audio=(audio,))

    def put_batch(self, frames, idxs, audio_frames, block=True, timeout=None, _single_audio=None):
        """A whole batch (`for i, res_frame in enumerate(recon): res_frame_queue.put(...)`, musereal.py:116-119): frames [B, ...] on the device
        or the host (None: B silent frames), idxs the B frame indices, audio_frames the 2B (pcm, type) pairs (two per frame).  A device batch travels as ONE DMA
        per run of consecutive slots (the ring wraps at most once per batch) behind ONE stream fence, and its B descriptors as ONE queue message --
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
        frames None: B silent frames (slots without a payload).  reserved: slots from try_reserve(len(idxs)) (then nothing here can block); they are adopted
        FIRST, so that whatever fails afterwards hands them back.  Commit / abort in any order: the ring publishes in begin order."""
        B = len(idxs)
        if reserved is not None:
            tok = reserved.entry if isinstance(reserved, _Slots) else None
            if tok is None or tok["state"] != "open" or tok["idxs"] is not None:
                raise ValueError("reserved: not an unused reservation of this ring (use the list try_reserve returned)")
        elif B == 0:
            return None
        else:
            tok = None
        try:
            is_dev = self._is_device(frames)
            if reserved is not None and len(reserved) != B:
                raise ValueError(f"{len(reserved)} reserved slots for {B} frames")
            if frames is None:
                shape = dtype = None
            elif is_dev:
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
            if frames is not None and len(frames) != B:
                raise ValueError(f"{len(frames)} frames for {B} indices")
            if tok is None:
                tok = self._acquire(B, block, timeout)
            slots = tok["slots"]
            tok.update(shape=shape, dtype=None if dtype is None else dtype.str, idxs=list(idxs))
            if frames is None:
                pass
            elif is_dev:
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
            if tok is not None:
                self.abort_batch(tok)
            raise
        return tok

    def abort_batch(self, tok):
        """Nothing of this batch is to be published: its slots go back.  The newest outstanding batch is a plain rewind of the cursor; with newer batches
        outstanding a rewind would hand out THEIR slots a second time, so the slots travel as a SKIP descriptor the consumer releases in ring order without
        surfacing a frame -- and, like every message, only once every older batch has been published or aborted (_flush)."""
        if not tok.get("slots"):
            return
        if "state" not in tok:                                       # (a bare {"slots": [...]}: find the ring's record of it)
            tok = next((e for e in self._open if list(e["slots"]) == list(tok["slots"])), None)
            if tok is None:
                raise ValueError("abort_batch: these slots are not an outstanding batch of this ring")
        if tok["state"] != "open":
            return
        tok["state"], tok["keep"] = "aborted", None
        self._flush()

    def commit_batch(self, tok, audio_frames, _single_audio=None):
        """Second half of put_batch: the batch's descriptors become one message, published as soon as every older batch is (normally: at once).  The copy
        must be complete."""
        sl, shape, dt, idxs = tok["slots"], tok["shape"], tok["dtype"], tok["idxs"]
        if tok["state"] != "open" or idxs is None:
            raise ValueError("commit_batch: not a begun, uncommitted batch")
        tok["keep"] = None
        if _single_audio is not None:
            tok["msg"] = [(sl[0], shape, dt, idxs[0], _single_audio[0])]
        else:
            tok["msg"] = [(s_, shape, dt, idxs[i], audio_frames[2 * i:2 * i + 2]) for i, s_ in enumerate(sl)]
        tok["state"] = "committed"
        self._flush()

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
        if shape is None:                                            # a silent frame: a slot without a payload, free again at once
            if slot is not None:
                self._release_in_order()
            return None, idx, audio
        view = self._slot_view(slot, shape, dtype)
        if copy:
            out = view.copy()
            self._release_in_order()
            return out, idx, audio
        self._views_out += 1
        return view, idx, audio

    def _release_in_order(self):
        """The free count is a counter and the producer hands slots out by cursor order, so "one more free slot" always means the OLDEST unreleased one.  While
        the consumer still holds a copy=False view of an earlier frame, a later frame's slot (a silent frame's, or one taken with copy=True) may therefore not be
        released yet -- the producer could wrap onto the slot still being read.  Its release is held until the outstanding views are back."""
        if self._views_out > 0:
            self._held += 1
        else:
            self._free.release()

    def release(self, view=None):
        """Hands a slot obtained with get(copy=False) back to the producer (slots are consumed in order: views are released oldest first)."""
        if self._views_out <= 0:
            raise RuntimeError("FrameRing.release: no get(copy=False) view is outstanding")
        self._views_out -= 1
        self._free.release()
        if self._views_out == 0:
            while self._held:
                self._held -= 1
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

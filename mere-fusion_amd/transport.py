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
with a single copy when the slots are contiguous.  Single producer, single consumer -- the shape of the reference's loop.

There is no GPU requirement for host frames (the ring then simply replaces the pickling); device tensors need the HIP library."""
import ctypes as C
import multiprocessing as mp
import queue
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

    # ---- pickling: a child process re-attaches to the same block ---------------------------------------------------------------
    def __getstate__(self):
        d = self.__dict__.copy()
        d["_shm"] = None
        d["_owner"] = False
        d["_registered"] = False
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

    def _acquire(self, block, timeout):
        if not self._free.acquire(block, timeout):
            raise queue.Full
        slot = self._head
        self._head = (self._head + 1) % self.slots
        return slot

    def put(self, item, block=True, timeout=None):
        """item = (res_frame, idx, audio_frames) exactly as lipreal.py:136 / musereal.py:116 put it; res_frame is None for an all-silent chunk
        (lipreal.py:104), a numpy array, or a HIP device tensor (copied by DMA into the slot)."""
        frame, idx, audio = item
        if frame is None:
            self._desc.put((None, None, None, idx, audio), block, timeout)
            return
        slot = self._acquire(block, timeout)
        try:
            import torch
            is_dev = torch.is_tensor(frame) and frame.is_cuda
        except ImportError:
            is_dev = False
        if is_dev:
            from . import _lib
            self.register_pinned()
            t = frame.contiguous()
            shape, dtype = tuple(t.shape), np.dtype(str(t.dtype).replace("torch.", ""))
            nbytes = t.numel() * t.element_size()
            if nbytes > self.slot_bytes:
                self._free.release()
                raise ValueError(f"frame of {nbytes} bytes does not fit a {self.slot_bytes}-byte slot")
            stream = C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
            with torch.cuda.device(t.device):
                _lib.check(_lib.lib().mf_copy_d2h_async(C.c_void_p(t.data_ptr()), C.c_void_p(self._slot_ptr(slot)), nbytes, stream), "copy_d2h_async")
                _lib.check(_lib.lib().mf_stream_synchronize(stream), "stream_synchronize")
        else:
            a = np.asarray(frame)
            shape, dtype = a.shape, a.dtype
            self._slot_view(slot, shape, dtype)[...] = a
        self._desc.put((slot, shape, dtype.str, idx, audio))

    def put_batch(self, frames, idxs, audio_frames, block=True, timeout=None):
        """A whole batch (`for i, res_frame in enumerate(recon): res_frame_queue.put(...)`, musereal.py:116-119): frames [B, ...] on the device
        or the host, idxs the B frame indices, audio_frames the 2B (pcm, type) pairs (two per frame).  Device batches whose slots are
        consecutive in the ring travel as ONE pitched DMA per run of slots (the ring wraps at most once per batch)."""
        B = len(idxs)
        slots = [self._acquire(block, timeout) for _ in range(B)]
        try:
            import torch
            is_dev = torch.is_tensor(frames) and frames.is_cuda
        except ImportError:
            is_dev = False
        if is_dev:
            from . import _lib
            self.register_pinned()
            t = frames.contiguous()
            per = t[0].numel() * t.element_size()
            if per > self.slot_bytes:
                raise ValueError(f"frame of {per} bytes does not fit a {self.slot_bytes}-byte slot")
            shape, dtype = tuple(t.shape[1:]), np.dtype(str(t.dtype).replace("torch.", ""))
            stream = C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
            with torch.cuda.device(t.device):
                i = 0
                while i < B:                                   # runs of consecutive slots (the ring wraps at most once per batch): one pitched DMA each
                    j = i + 1
                    while j < B and slots[j] == slots[j - 1] + 1:
                        j += 1
                    _lib.check(_lib.lib().mf_copy_d2h_2d_async(C.c_void_p(t.data_ptr() + i * per), per, C.c_void_p(self._slot_ptr(slots[i])),
                                                               self.slot_stride, per, j - i, stream), "copy_d2h_2d_async")
                    i = j
                _lib.check(_lib.lib().mf_stream_synchronize(stream), "stream_synchronize")   # one fence per batch, then the slots are published
        else:
            a = np.asarray(frames)
            shape, dtype = a.shape[1:], a.dtype
            for i, s in enumerate(slots):
                self._slot_view(s, shape, dtype)[...] = a[i]
        for i, s in enumerate(slots):
            self._desc.put((s, shape, np.dtype(dtype).str, idxs[i], audio_frames[2 * i:2 * i + 2]))

    # ---- consumer -----------------------------------------------------------------------------------------------------------
    def get(self, block=True, timeout=None, copy=True):
        """-> (res_frame, idx, audio_frames), the tuple `process_frames` unpacks (lipreal.py:195).  copy=True returns an ndarray the caller
        owns (the slot is free again immediately); copy=False returns a view into the ring and the caller must `release(view)` it."""
        slot, shape, dtype, idx, audio = self._desc.get(block, timeout)
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
        return self._desc.qsize()

    def empty(self):
        return self._desc.empty()

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

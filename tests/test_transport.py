"""Frame transport: the shared-memory ring that stands where `res_frame_queue = mp.Queue(batch_size * 2)` stood
(lipreal.py:161, musereal.py:153), behind the same `(res_frame | None, idx, audio_frames)` tuple contract."""
import multiprocessing as mp
import queue
import time

import numpy as np
import pytest
import torch

from mere_fusion_amd.transport import FrameRing


def _audio(i):
    return [((np.arange(320, dtype=np.float32) + i) / 1000, 0), (np.zeros(320, np.float32), 1)]     # two (pcm, type) pairs per frame (lipreal.py:136)


def _producer(ring, n, shape, seed):
    """The inference process of lipreal.py:85-141 in miniature: frames, a silent chunk (None frame), frames."""
    rng = np.random.default_rng(seed)
    for i in range(n):
        frame = None if i % 5 == 3 else rng.integers(0, 256, shape, dtype=np.uint8)
        ring.put((frame, i, _audio(i)))
    ring.put((None, -1, []))                                        # end marker


def test_ring_keeps_the_tuple_contract_across_processes():
    shape = (256, 256, 3)
    ring = FrameRing(slots=4, frame_shape=shape)                    # far fewer slots than frames: the producer must block and resume
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_producer, args=(ring, 23, shape, 7))
    p.start()
    rng = np.random.default_rng(7)
    got = 0
    while True:
        frame, idx, audio = ring.get(block=True, timeout=30)        # process_frames: res_frame, idx, audio_frames = queue.get(block=True, timeout=1)
        if idx == -1:
            break
        assert idx == got
        want = None if idx % 5 == 3 else rng.integers(0, 256, shape, dtype=np.uint8)
        if want is None:
            assert frame is None
        else:
            assert frame.dtype == np.uint8 and frame.shape == shape and np.array_equal(frame, want)
        assert len(audio) == 2 and audio[0][1] == 0 and audio[1][1] == 1 and np.array_equal(audio[0][0], _audio(idx)[0][0])
        got += 1
    p.join(30)
    assert got == 23 and p.exitcode == 0
    with pytest.raises(queue.Empty):
        ring.get(block=True, timeout=0.05)
    ring.close()


def test_ring_bounds_and_views():
    ring = FrameRing(slots=2, frame_shape=(96, 96, 3), dtype=np.float32)      # Wav2Lip: float32 `pred * 255` frames (lipreal.py:126)
    a, b = np.full((96, 96, 3), 1.5, np.float32), np.full((96, 96, 3), 2.5, np.float32)
    ring.put((a, 0, []))
    ring.put((b, 1, []))
    with pytest.raises(queue.Full):
        ring.put((a, 2, []), block=True, timeout=0.05)              # the reference's bounded queue blocks the producer the same way
    time.sleep(0.05)                                                # (mp.Queue hands items to its feeder thread asynchronously)
    view, idx, _ = ring.get(timeout=5, copy=False)                  # zero-copy read: a view into the ring
    assert idx == 0 and view.dtype == np.float32 and float(view[3, 4, 1]) == 1.5 and not view.flags.owndata
    with pytest.raises(queue.Full):
        ring.put((a, 2, []), block=True, timeout=0.05)              # still held by the consumer
    del view
    ring.release()
    ring.put((a * 2, 2, []), timeout=5)                              # slot 0 again
    f1, i1, _ = ring.get(timeout=5)
    f2, i2, _ = ring.get(timeout=5)
    assert (i1, i2) == (1, 2) and float(f1[0, 0, 0]) == 2.5 and float(f2[0, 0, 0]) == 3.0
    with pytest.raises(ValueError, match="does not fit"):
        ring.put((np.zeros((97, 96, 3), np.float32), 3, []))
    ring.close()


def test_ring_put_batch_host_frames():
    ring = FrameRing(slots=16, frame_shape=(256, 256, 3))
    frames = np.random.default_rng(1).integers(0, 256, (8, 256, 256, 3), dtype=np.uint8)
    audio = [(np.zeros(320, np.float32), 0)] * 16
    for rep in range(3):                                            # the third batch wraps around the ring
        ring.put_batch(frames, list(range(8 * rep, 8 * rep + 8)), audio)
        for i in range(8):
            f, idx, au = ring.get(timeout=5)
            assert idx == 8 * rep + i and np.array_equal(f, frames[i]) and len(au) == 2
    ring.close()


def test_ring_failed_puts_leave_the_ring_intact():
    """an oversized frame, a batch larger than the ring, or a time-out part-way through a batch must neither leak slots nor move the
    cursor onto a slot whose frame is still unread."""
    ring = FrameRing(slots=4, frame_shape=(8, 8, 3))
    ok = np.arange(8 * 8 * 3, dtype=np.uint8).reshape(8, 8, 3)
    big = np.zeros((9, 8, 3), np.uint8)
    audio = [(np.zeros(320, np.float32), 0)] * 16
    ring.put((ok, 0, []))
    with pytest.raises(ValueError, match="does not fit"):
        ring.put((big, 1, []))
    with pytest.raises(ValueError, match="does not fit"):
        ring.put_batch(np.stack([big, big]), [1, 2], audio[:4])
    with pytest.raises(ValueError, match="never fit"):
        ring.put_batch(np.stack([ok] * 5), list(range(5)), audio[:10])           # B > slots would block for ever
    with pytest.raises(queue.Full):
        ring.put_batch(np.stack([ok] * 4), [1, 2, 3, 4], audio[:8], timeout=0.05)  # 3 free slots, 4 wanted: all-or-nothing
    ring.put_batch(np.stack([ok + 1, ok + 2, ok + 3]), [1, 2, 3], audio[:6], timeout=5)   # ... and the 3 are all still there, in order behind frame 0
    for want in range(4):
        f, idx, _ = ring.get(timeout=5)
        assert idx == want and np.array_equal(f, ok + want)
    with pytest.raises(queue.Empty):
        ring.get(block=True, timeout=0.05)
    ring.put_batch(np.stack([ok] * 4), [4, 5, 6, 7], audio[:8], timeout=5)       # the full ring is usable again
    assert [ring.get(timeout=5)[1] for _ in range(4)] == [4, 5, 6, 7]
    ring.close()


def test_ring_reservations_and_aborting_an_older_token():
    """(1) try_reserve is all-or-nothing and non-blocking, and a reservation can be used or handed back; (2) aborting a token while a NEWER one is
    outstanding must not rewind the cursor onto the newer token's slots: the aborted slots travel as a skip descriptor and come free in ring order."""
    ring = FrameRing(slots=6, frame_shape=(4, 4, 3))
    f = lambda v: np.full((2, 4, 4, 3), v, np.uint8)
    audio = [(np.zeros(320, np.float32), 0)] * 4
    r = ring.try_reserve(2)
    assert r == [0, 1]
    assert ring.try_reserve(5) is None                              # 4 free: nothing taken, cursor unmoved
    ring.unreserve(r)                                               # newest reservation: plain rewind
    r = ring.try_reserve(2)
    assert r == [0, 1]
    t_old = ring.begin_batch(f(1), [10, 11], reserved=r)
    t_new = ring.begin_batch(f(2), [20, 21])
    assert t_new["slots"] == [2, 3]
    ring.abort_batch(t_old)                                         # an OLDER token: must not hand slots 2, 3 out again
    t3 = ring.begin_batch(f(3), [30, 31])
    assert t3["slots"] == [4, 5]
    assert ring.try_reserve(1) is None                              # slots 0, 1 are not free until the consumer has passed the skip marker
    ring.commit_batch(t_new, audio)
    ring.commit_batch(t3, audio)
    got = [ring.get(timeout=5) for _ in range(4)]
    assert [g[1] for g in got] == [20, 21, 30, 31] and [int(g[0][0, 0, 0]) for g in got] == [2, 2, 3, 3]
    assert ring.try_reserve(6) == [0, 1, 2, 3, 4, 5]                # everything came back, in order
    ring.close()


@pytest.mark.gpu
def test_ring_device_frame_without_audio(lib_built):
    """put() of a DEVICE frame whose audio_frames is None used to fall into the batch path's slicing (TypeError after the slot was taken)."""
    ring = FrameRing(slots=2, frame_shape=(16, 16, 3))
    fr = torch.randint(0, 256, (16, 16, 3), dtype=torch.uint8, device="cuda")
    ring.put((fr, 5, None))
    g, idx, au = ring.get(timeout=5)
    assert idx == 5 and au is None and np.array_equal(g, fr.cpu().numpy())
    nc = torch.randint(0, 256, (3, 16, 32, 3), dtype=torch.uint8, device="cuda")[:, :, ::2]      # non-contiguous batch + an explicit copy stream
    st = torch.cuda.Stream()
    tok = ring.begin_batch(nc[:2], [1, 2], stream=st)
    st.synchronize()
    ring.commit_batch(tok, [(np.zeros(320, np.float32), 0)] * 4)
    for i in range(2):
        g, idx, _ = ring.get(timeout=5)
        assert idx == i + 1 and np.array_equal(g, nc[i].cpu().numpy())
    ring.close()


@pytest.mark.gpu
def test_ring_takes_device_frames_by_dma(lib_built):
    """uint8 frames straight from HBM into the page-locked ring (single put and put_batch incl. the wrap-around split into two DMAs)."""
    ring = FrameRing(slots=12, frame_shape=(256, 256, 3))
    frames = torch.randint(0, 256, (8, 256, 256, 3), dtype=torch.uint8, device="cuda")
    host = frames.cpu().numpy()
    ring.put((frames[3], 42, _audio(1)))
    f, idx, _ = ring.get(timeout=5)
    assert idx == 42 and np.array_equal(f, host[3])
    audio = [(np.zeros(320, np.float32), 0)] * 16
    for rep in range(4):                                            # 1 + 8 * rep slots in: reps 1.. cross the end of the 12-slot ring
        ring.put_batch(frames, list(range(8)), audio)
        for i in range(8):
            f, idx, _ = ring.get(timeout=5)
            assert idx == i and np.array_equal(f, host[i]), (rep, i)
    f32 = torch.rand(2, 96, 96, 3, device="cuda") * 255
    ring2 = FrameRing(slots=4, frame_shape=(96, 96, 3), dtype=np.float32)
    ring2.put_batch(f32, [0, 1], audio[:4])
    for i in range(2):
        f, idx, _ = ring2.get(timeout=5)
        assert f.dtype == np.float32 and np.array_equal(f, f32[i].cpu().numpy())
    ring.close(); ring2.close()


def test_ring_publishes_in_begin_order_when_the_middle_of_three_batches_is_aborted():
    """a skip descriptor published ahead of an older, still uncommitted batch let the producer wrap onto that batch's slots (its in-flight DMA
    target).  The ring now publishes strictly in begin order: the aborted middle batch's slots stay taken until the oldest batch is committed."""
    ring = FrameRing(slots=16, frame_shape=(4, 4, 3))
    f = lambda v: np.full((4, 4, 4, 3), v, np.uint8)
    audio = [(np.zeros(320, np.float32), 0)] * 8
    z = ring.begin_batch(f(1), [0, 1, 2, 3])                        # slots 0-3: "its DMA is still in flight"
    a = ring.begin_batch(f(2), [4, 5, 6, 7])                        # slots 4-7
    b = ring.begin_batch(f(3), [8, 9, 10, 11])                      # slots 8-11
    ring.abort_batch(a)                                             # the MIDDLE one
    assert ring.empty()                                             # nothing may overtake z: no skip message yet
    r = ring.try_reserve(8)                                         # 4 free slots (12-15): a wrap onto 0-3 must be impossible
    assert r is None
    r4 = ring.try_reserve(4)
    assert r4 == [12, 13, 14, 15] and ring.try_reserve(1) is None
    ring.commit_batch(b, audio)                                     # committed out of order: held back behind z
    assert ring.empty()
    ring.commit_batch(z, audio)                                     # now z, the skip for a, and b go out, in that order
    got = [ring.get(timeout=5) for _ in range(8)]
    assert [g[1] for g in got] == [0, 1, 2, 3, 8, 9, 10, 11] and [int(g[0][0, 0, 0]) for g in got] == [1] * 4 + [3] * 4
    ring.unreserve(r4)                                              # newest: a plain rewind
    assert ring.try_reserve(16) == list(range(12, 16)) + list(range(0, 12))
    ring.close()


def test_ring_silent_frames_take_places_and_a_failed_begin_returns_its_reservation():
    """(1) a silent frame holds one of the ring's places like any other frame (the reference's Queue(2B) counts per-frame tuples), so a stalled
    consumer stops silent batches at the ring's size instead of filling the descriptor queue, and every message holds >= 1 slot: publishing never blocks;
    (2) a begin_batch that fails AFTER adopting a reservation hands the slots back."""
    ring = FrameRing(slots=4, frame_shape=(4, 4, 3))
    audio = [(np.zeros(320, np.float32), 1)] * 4
    for j in range(2):                                              # two silent batches of 2 fill the ring ...
        r = ring.try_reserve(2)
        assert r is not None
        tok = ring.begin_batch(None, [2 * j, 2 * j + 1], reserved=r)
        ring.commit_batch(tok, audio)
    assert ring.try_reserve(1) is None and ring.free_slots() == 0   # ... and the producer KNOWS (it would have queue.Full'ed four puts later before)
    with pytest.raises(queue.Full):
        ring.put((None, 9, []), block=True, timeout=0.05)
    got = [ring.get(timeout=5) for _ in range(4)]                   # silent frames come out as (None, idx, audio) and free their place at once
    assert [g[1] for g in got] == [0, 1, 2, 3] and all(g[0] is None and g[2][0][1] == 1 for g in got)
    assert ring.free_slots() == 4
    r = ring.try_reserve(2)
    with pytest.raises(ValueError, match="does not fit"):
        ring.begin_batch(np.zeros((2, 5, 4, 3), np.uint8), [0, 1], reserved=r)     # fails after adoption: the two slots come back
    with pytest.raises(ValueError, match="3 frames for 2"):
        ring.begin_batch(np.zeros((3, 4, 4, 3), np.uint8), [0, 1], reserved=ring.try_reserve(2))
    assert ring.free_slots() == 4 and ring.try_reserve(4) == [0, 1, 2, 3]
    ring.close()


def test_a_later_slot_is_not_released_while_an_earlier_view_is_held():
    """the free count is a counter and the producer takes slots in cursor order, so releasing a silent frame's slot (or a copied frame's) while a
    copy=False view of an EARLIER frame is still held would let the producer wrap onto the slot being read.  The release is held until the view is back."""
    ring = FrameRing(slots=2, frame_shape=(4, 4, 3))
    a = np.full((4, 4, 3), 7, np.uint8)
    ring.put((a, 0, []))
    ring.put((None, 1, []))                                          # a silent frame holds a ring place too
    time.sleep(0.05)
    view, idx, _ = ring.get(timeout=5, copy=False)
    assert idx == 0 and int(view[0, 0, 0]) == 7
    f, idx, _ = ring.get(timeout=5)                                  # the silent frame, taken while the view is still out
    assert f is None and idx == 1
    with pytest.raises(queue.Full):
        ring.put((a * 2, 2, []), block=True, timeout=0.05)           # its place is NOT free yet: the producer's next slot is the one under the view
    assert int(view[0, 0, 0]) == 7
    ring.release()                                                   # the view goes back, and the held release with it
    ring.put((a * 2, 2, []), timeout=5)
    ring.put((a * 3, 3, []), timeout=5)
    f2, i2, _ = ring.get(timeout=5)
    f3, i3, _ = ring.get(timeout=5)
    assert (i2, i3) == (2, 3) and int(f2[0, 0, 0]) == 14 and int(f3[0, 0, 0]) == 21
    with pytest.raises(RuntimeError, match="no get"):
        ring.release()                                               # nothing outstanding: a stray release would corrupt the count
    ring.close()

#!/usr/bin/env python3
"""device batch -> FrameRing -> consumer in one process, profiled (GPU box)"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np, torch
from mere_fusion_amd.transport import FrameRing
ring = FrameRing(16, (256, 256, 3))
frames = torch.randint(0, 256, (8, 256, 256, 3), dtype=torch.uint8, device="cuda")
audio = [(np.zeros(320, np.float32), 0)] * 16
def one():
    ring.put_batch(frames, list(range(8)), audio)
    for _ in range(8):
        ring.get(timeout=5, copy=False); ring.release()
for _ in range(3): one()
t = time.perf_counter()
for _ in range(50): one()
print((time.perf_counter() - t) / 50 * 1e3, "ms per batch")
cProfile.run("for _ in range(50): one()", "/tmp/p.prof")
pstats.Stats("/tmp/p.prof").sort_stats("tottime").print_stats(10)
ring.close()

#!/usr/bin/env python3
"""Which thread of a process burns host CPU while the GPU is busy, by how the host waits (GPU box).  python tools/host_wait_probe.py <mode> [seconds]
   modes: query (poll event.query() with 0.5 ms sleeps), sync (event.synchronize()), blocksync (Event(blocking=True).synchronize())"""
import os, sys, time, threading
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import torch
import bench
mode, secs = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
print({k: v for k, v in os.environ.items() if any(t in k for t in ("HSA", "HIP", "ROC", "AMD", "GPU"))}, flush=True)
x = torch.randn(8192, 8192, device="cuda")
torch.cuda.synchronize()
t0c = bench.thread_cpu_times(); t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20):
        y = x @ x                                       # ~1 ms each
    ev = torch.cuda.Event(blocking=(mode == "blocksync"))
    ev.record()
    if mode == "query":
        while not ev.query():
            time.sleep(0.0005)
    else:
        ev.synchronize()
    n += 1
wall = time.perf_counter() - t0
t1c = bench.thread_cpu_times()
rows = sorted(((sec - t0c.get(tid, (c, 0))[1]) / wall, tid, c) for tid, (c, sec) in t1c.items())
print(mode, "iters", n, "main tid", threading.get_native_id(), [(round(a, 3), tid, c) for a, tid, c in rows if a > 0.01], flush=True)

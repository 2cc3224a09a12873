#!/usr/bin/env python3
"""Which runtime thread burns host CPU while the GPU is busy, by WHAT is launched (the host itself only polls event.query() with 0.5 ms sleeps).
   python tools/host_wait_probe2.py <what> [seconds]     what: eager | graph | unet | unet_nograph | streams"""
import os, sys, time, threading
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import torch
import bench
what, secs = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
x = torch.randn(4096, 4096, device="cuda")
step = None
if what == "eager":
    def step():
        for _ in range(40):
            x @ x
elif what == "graph":
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            y = x @ x
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(40):
                y = x @ x
    step = g.replay
elif what == "streams":
    s2 = torch.cuda.Stream()
    def step():
        for _ in range(20):
            x @ x
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            for _ in range(20):
                x @ x
        torch.cuda.current_stream().wait_stream(s2)
else:
    if what == "unet_nograph":
        os.environ["MF_NO_GRAPH"] = "1"
    run = bench.MuseTalkRunner("bf16x3", 8, "cuda:0")
    step = run.step
for _ in range(3):
    step()
torch.cuda.synchronize()
t0c = bench.thread_cpu_times(); t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    step()
    ev = torch.cuda.Event()
    ev.record()
    while not ev.query():
        time.sleep(0.0005)
    n += 1
wall = time.perf_counter() - t0
t1c = bench.thread_cpu_times()
rows = sorted(((sec - t0c.get(tid, (c, 0))[1]) / wall, tid, c) for tid, (c, sec) in t1c.items())
print(what, "iters", n, "main tid", threading.get_native_id(), [(round(a, 3), tid, c) for a, tid, c in rows if a > 0.01], flush=True)

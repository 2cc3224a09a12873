"""MuseTalk: S concurrent sessions, each batch B on its own stream (one hipStream per session), aggregate frames/s (GPU box).
    python tools/mt_streams.py 8 1 2 3 4"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]); counts = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
dev = torch.device("cuda:0")
runners, streams = [], []
for S in counts:
    while len(runners) < S:
        runners.append(bench.MuseTalkRunner("bf16x3", B, dev, seed=len(runners)))
        streams.append(torch.cuda.Stream())

    def sweep(n):
        for _ in range(n):
            for r, s in zip(runners[:S], streams[:S]):
                with torch.cuda.stream(s):
                    r.step()
    sweep(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sweep(10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"S={S} B={B}: {dt * 1e3:.2f} ms per sweep -> {S * B / dt:.1f} frames/s aggregate, {dt * 1e3:.1f} ms latency per session step", flush=True)

"""MuseTalk B=8 frames with / without GroupNorm folded into the halo convs (MF_GN_FUSE): run twice, second run compares (GPU box).
    MF_GN_FUSE=0 python tools/mt_gnfuse_check.py save; MF_GN_FUSE=1 python tools/mt_gnfuse_check.py cmp"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
r = bench.MuseTalkRunner("bf16x3", 8, torch.device("cuda:0"))
f = r.step().clone()
torch.cuda.synchronize()
for _ in range(3):
    r.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    r.step()
torch.cuda.synchronize()
print(f"MF_GN_FUSE={os.environ.get('MF_GN_FUSE')}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step")
if sys.argv[1] == "save":
    torch.save(f.cpu(), "/tmp/mt_frames.pt")
else:
    g = torch.load("/tmp/mt_frames.pt")
    d = (f.cpu().int() - g.int()).abs()
    print(f"frames vs unfused: max u8 diff {int(d.max())}, differing {float((d > 0).float().mean()):.3e}, mean |frame| {float(g.float().mean()):.1f}")

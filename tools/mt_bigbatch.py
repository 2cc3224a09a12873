"""MuseTalk at a cross-session batch (default 64 = 8 sessions x 8 frames): frames/s, UNet conv-block MFMA issue, and a check that frames
[0:8] and [56:64] equal a batch-8 run of the same inputs (GPU box).  python tools/mt_bigbatch.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
small = bench.MuseTalkRunner("bf16x3", 8, dev)
f8 = small.step().clone()
del small.unet, small.vae
torch.cuda.empty_cache()
big = bench.MuseTalkRunner("bf16x3", B, dev)
big.lat = f8.new_zeros(0) if False else big.lat
reps = B // 8
lat8, aud8 = bench.W.make_musetalk_inputs(8, 0)
big.lat, big.aud = lat8.repeat(reps, 1, 1, 1).to(dev), aud8.repeat(reps, 1, 1).to(dev)
fb = big.step()
torch.cuda.synchronize()
print("mem GiB", round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 2**30, 1))
for lo in (0, B - 8):
    d = (fb[lo:lo + 8].int() - f8.int()).abs()
    print(f"frames [{lo}:{lo + 8}] vs batch-8 run: max u8 diff {int(d.max())}, differing {float((d > 0).float().mean()):.2e}")
for _ in range(2):
    big.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    big.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"B={B}: {dt * 1e3:.1f} ms per step -> {B / dt:.1f} frames/s = {B / dt / 25:.1f} sessions at 25 fps")
rows = big.profile(2)
cb = [r for r in rows if r["layer"].startswith("unet:") and r["flops"] > 0 and "attention" not in r["layer"]]
tb, fl = sum(r["ms"] for r in cb), sum(r["flops"] for r in cb)
print(f"UNet conv blocks: {tb:.2f} ms, {fl / tb / 1e9:.1f} TF algorithmic, MFMA issue {3 * fl / tb / 1e9 / 2500:.3f} of bf16 dense peak")

"""Per-op times of the UNet's 320-channel 3x3 convs (32 x 32 level) at batch B, GPU box; MF_LIB_PATH selects the build."""
import os, sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import bench
r = bench.MuseTalkRunner("bf16x3", int(os.environ.get("B", "8")), "cuda:0")
for _ in range(3):
    r.step()
rows = r.profile(30)
tot = 0.0
for row in rows:
    L = row["layer"]
    if L.startswith("unet:") and ("down_blocks.0.resnets" in L or "up_blocks.3.resnets" in L) and (L.endswith("conv1") or L.endswith("conv2")):
        print(f"{L:50s} {row['kernel'][:44]:44s} {row['ms'] * 1e3:8.1f} us")
        tot += row["ms"]
print("these convs: %.3f ms; unet ops: %.3f ms" % (tot, sum(x["ms"] for x in rows if x["layer"].startswith("unet:"))))

"""Per-op times of the VAE decoder's resnet convs (conv1: no residual, conv2: residual) at batch B, GPU box."""
import os, sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import bench
r = bench.MuseTalkRunner("bf16x3", int(os.environ.get("B", "8")), "cuda:0")
for _ in range(3):
    r.step()
rows = r.profile(20)
for row in rows:
    if row["layer"].startswith("vae:") and ("conv1" in row["layer"] or "conv2" in row["layer"] or "conv_out" in row["layer"]):
        print(f"{row['layer']:60s} {row['ms'] * 1e3:8.1f} us")
print("sum of vae ops: %.3f ms; all: %.3f ms" % (sum(x["ms"] for x in rows if x["layer"].startswith("vae:")), sum(x["ms"] for x in rows)))

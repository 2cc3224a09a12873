"""Per-op times of the tails of the UNet / VAE schedules (fused GroupNorm + SiLU + conv_out vs MF_TAIL_FUSE=0), GPU box."""
import os, sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import bench
r = bench.MuseTalkRunner("bf16x3", int(os.environ.get("B", "8")), "cuda:0")
for _ in range(3):
    r.step()
rows = r.profile(20)
for row in rows:
    if "conv_out" in row["layer"] or "conv_norm_out" in row["layer"]:
        print(f"{row['layer']:60s} {row['kernel'][:60]:60s} {row['ms'] * 1e3:8.1f} us")
print("sum of ops: %.3f ms" % sum(x["ms"] for x in rows))

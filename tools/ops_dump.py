"""Per-op profile of the MuseTalk step at batch B (default 64) -> gpurun_out/ops_b<B>.json (GPU box)."""
import os, sys, json
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import bench
B = int(os.environ.get("B", "64"))
r = bench.MuseTalkRunner("bf16x3", B, "cuda:0")
for _ in range(3):
    r.step()
rows = r.profile(10)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open(f"gpurun_out/ops_b{B}.json", "w"))
print("sum of ops: %.3f ms" % sum(x["ms"] for x in rows))

"""Per-op profile of one MuseTalk step grouped by (kernel, layer family) (GPU box).  python tools/mt_ops.py [B]"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = bench.MuseTalkRunner("bf16x3", B, torch.device("cuda:0"))
r.step(); torch.cuda.synchronize()
rows = r.profile(10)
tot = sum(x["ms"] for x in rows)
print(f"total {tot:.2f} ms over {len(rows)} ops")
fam = {}
for x in rows:
    name = re.sub(r"\.\d+\.", ".N.", x["layer"])
    name = re.sub(r"(down_blocks|up_blocks)\.N", r"\1", name)
    key = (name, x["kernel"])
    a = fam.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += x["ms"]; a[2] += x["flops"]
for (name, k), (n, ms, fl) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{ms:7.3f} ms  n={n:3d}  {fl / max(ms, 1e-9) / 1e9:7.1f} TF  {name[:70]:70s} {k[:60]}")

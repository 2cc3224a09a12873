"""Every op of one MuseTalk step in schedule order: ms (hipEvents, graph off), algorithmic GFLOP, TF, kernel (GPU box).
python tools/mt_oplist.py [B] [unet|vae|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
which = sys.argv[2] if len(sys.argv) > 2 else "all"
r = bench.MuseTalkRunner("bf16x3", B, torch.device("cuda:0"))
r.step(); torch.cuda.synchronize()
rows = r.profile(10)
tot = {"unet": 0.0, "vae": 0.0}
for i, x in enumerate(rows):
    part = "unet" if x["layer"].startswith("unet:") else "vae"
    tot[part] += x["ms"]
    if which != "all" and part != which:
        continue
    tf = x["flops"] / max(x["ms"], 1e-9) / 1e9
    print(f"{i:4d} {x['ms'] * 1e3:8.1f} us {x['flops'] / 1e9:8.2f} GF {tf:7.1f} TF  {x['layer'][:64]:64s} {x['kernel'][:70]}")
print(f"unet {tot['unet']:.3f} ms, vae {tot['vae']:.3f} ms (sum of per-op event times)")

#!/usr/bin/env python3
"""Tile x split-K sweep of the MuseTalk UNet GEMM shapes through tools/conv_probe.py (GPU box).
    python tools/unet_shape_sweep.py > gpurun_out/unet_sweep.txt"""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
# (cin, cout, k, hw) at batch 8
SHAPES = [(320, 320, 3, 32), (640, 320, 3, 32), (960, 320, 3, 32), (640, 640, 3, 16), (1280, 640, 3, 16), (1920, 640, 3, 16),
          (1280, 1280, 3, 8), (2560, 1280, 3, 8), (1280, 1280, 3, 4), (2560, 1280, 3, 4),
          (320, 2560, 1, 32), (1280, 320, 1, 32), (640, 5120, 1, 16), (2560, 640, 1, 16), (1280, 10240, 1, 8), (5120, 1280, 1, 8),
          (320, 960, 1, 32), (640, 1920, 1, 16), (320, 320, 1, 32), (640, 640, 1, 16), (1280, 1280, 1, 8)]
CONFIGS = [("default", {})] + [(f"{t}/s{s}", {"MF_FORCE_TILE": t, "MF_FORCE_SPLIT": str(s)})
                               for t in ("64x64", "128x64", "128x128") for s in (1, 2, 4, 8)]
only = sys.argv[1:]
for cin, cout, k, hw in SHAPES:
    best = None
    line = []
    for name, env in CONFIGS:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, os.path.join(HERE, "conv_probe.py"), "--cin", str(cin), "--cout", str(cout), "--k", str(k),
                              "--pad", str(k // 2), "--hw", str(hw), "--batch", "8", "--residual", "0", "--iters", "30"],
                             env=e, capture_output=True, text=True).stdout
        m = re.search(r"conv launch alone: ([\d.]+) us -> ([\d.]+) TFLOP", out)
        if not m:
            line.append(f"{name}: FAIL"); continue
        us = float(m.group(1))
        line.append(f"{name}: {us:.1f}")
        if name != "default" and (best is None or us < best[1]): best = (name, us)
    print(f"{cin}->{cout} k{k} @{hw}: " + "  ".join(line) + f"   BEST {best}", flush=True)

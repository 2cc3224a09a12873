#!/usr/bin/env python3
"""One or more end-to-end paced trials at fixed session counts (GPU box):  python tools/paced_trial.py 22:40 23:20   (N:seconds ...)"""
import json, os, sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import types
import torch
import bench
args = types.SimpleNamespace(sessions=8, batch=8, precision="bf16x3")
LEAN = os.environ.get("LEAN") == "1"       # host-lean rank: eager single-stream handles (MF_NO_GRAPH=2) + single-stream scheduler
if LEAN:
    os.environ["MF_NO_GRAPH"] = "2"
dev = "cuda:0"
big = bench.MuseTalkRunner("bf16x3", 64, dev)
specs = [(int(a.split(":")[0]), float(a.split(":")[1])) for a in sys.argv[1:]] or [(22, 40.0)]
rig = bench.PacedRig(big, args, dev, n_max=max(n for n, _ in specs), lean=LEAN)
try:
    for n, sec in specs:
        r = rig.trial(n, sec)
        print(json.dumps({k: r[k] for k in ("sessions", "seconds", "p50_ms", "p99_ms", "max_ms", "sustained", "sessions_per_step_mean", "gpu_busy_frac", "frames_per_s", "host_cpu_s_per_wall_s", "host_cpu_by_thread", "first_third_mean_ms", "last_third_mean_ms")}), flush=True)
finally:
    rig.close()

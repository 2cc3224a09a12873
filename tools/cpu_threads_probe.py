import os, sys, time, torch
sys.path.insert(0, '.')
from mere_fusion_amd import weights as W
from oracle import wav2lip_ref
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cgroup", e)
sd = W.make_wav2lip_state_dict(0); mel, face, _ = W.make_lip_inputs(16, 0)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    wav2lip_ref.wav2lip_forward(sd, mel, face)
    t0=time.perf_counter(); wav2lip_ref.wav2lip_forward(sd, mel, face); dt=time.perf_counter()-t0
    print(nt, "threads:", round(dt,3), "s ->", round(16/dt,1), "fps", flush=True)
    if dt > 20: break

"""Where an ER-NeRF frame's time goes: audio / torso / head timed apart and together (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
run = bench.ErNeRFRunner("bf16x3", 512, dev)
r = run.r


def t(f, n=100):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3


enc_a = r.audio.encode_audio(run.auds)
bg = r.torso.run_torso(run.bg_coords, run.pose, 1.0)["bg_color"]
print("audio  ms (gpu-inclusive, host-enqueue): %.3f %.3f" % t(lambda: r.audio.encode_audio(run.auds)))
print("torso  %.3f %.3f" % t(lambda: r.torso.run_torso(run.bg_coords, run.pose, 1.0)))
print("head   %.3f %.3f" % t(lambda: r.run_cuda_device(run.ro, run.rd, enc_a, run.d_ind, run.eye, bg_color=bg, want_u8=True)))
print("head-g %.3f %.3f" % t(lambda: r.run_cuda_device(run.ro, run.rd, enc_a, run.d_ind, run.eye, bg_color=bg, want_u8=True, graph=True)))
print("head-h %.3f %.3f" % t(lambda: r.run_cuda(run.ro, run.rd, enc_a, run.d_ind, run.d_eye, bg_color=bg, want_u8=True)))
print("frame  %.3f %.3f" % t(run.step))

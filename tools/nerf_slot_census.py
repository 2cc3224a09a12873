"""GPU box: how many of the sample slots a 512 x 512 ER-NeRF frame hands to the field hold a sample, per round -- the reference evaluates every slot of its zero-filled
[n_alive x n_step] tensors; k_nerf_field_fused skips 16-slot fragments without a sample.  Printed: slots, samples, slots in fragments that are evaluated."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mere_fusion_amd.ernerf import _raymarching_face as rm
r = bench.ErNeRFRunner("bf16x3", int(sys.argv[1]) if len(sys.argv) > 1 else 512, torch.device("cuda:0"), seed=0)
rows = []
orig = rm.march_rays
def spy(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas, noises):
    out = orig(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas, noises)
    has = deltas[:, 0] != 0
    M = has.numel()
    pad = (-M) % 16
    frag = torch.cat([has, has.new_zeros(pad)]).view(-1, 16).any(1)
    rows.append((n_alive, n_step, M, int(has.sum()), int(frag.sum()) * 16))
    return out
rm.march_rays = spy
r.r.run_cuda(r.ro, r.rd, r.d_enc_a, r.d_ind, r.eye, bg_color=1.0)
rm.march_rays = orig
print("| round | alive rays | n_step | slots | samples | slots in evaluated fragments |\n|---:|---:|---:|---:|---:|---:|")
for i, (a, s, M, n, f) in enumerate(rows):
    print(f"| {i + 1} | {a} | {s} | {M} | {n} | {f} |")
t = [sum(x[k] for x in rows) for k in (2, 3, 4)]
print(f"| all | | | {t[0]} | {t[1]} ({100 * t[1] / t[0]:.1f} %) | {t[2]} ({100 * t[2] / t[0]:.1f} %) |")

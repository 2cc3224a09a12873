#!/usr/bin/env python3
"""Per-op floor table of the MuseTalk UNet at the bench's batch: for every op of the schedule

    floor = max(algorithmic FLOP / 833 TF (bf16x3: 2.5 PF / 3 passes),  operand bytes / 6 TB/s,  1.5 us (a dependent launch boundary))

beside the measured per-op time of `bench.py --dump-layers` (hipEvents around every op, graph off), ranked by (measured - floor).  Operand bytes: the
(hi, lo) bf16 planes of the input, the output, the residual where one is added, and the weights (4 B per element each); attention: q, k, v, o.
Shapes come from the state dict's shapes (mere_fusion_amd.weights, shapes_only) and the op's own FLOP count (map size = FLOP / (2 B cin cout k^2)).

    python tools/unet_floor_table.py gpurun_out/r05_layers.json [batch] > profiles/r05_unet_b8_floor.md
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.config import MUSETALK_V1

PEAK_TF, HBM_TBS, BOUNDARY_US = 833.3, 6.0, 1.5
rows = [r for r in json.load(open(sys.argv[1]))["musetalk_rows"] if r["layer"].startswith("unet:")]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
shapes = W.make_musetalk_unet_state_dict(MUSETALK_V1, 0, shapes_only=True)
shapes = {k: tuple(v) if not hasattr(v, "shape") else tuple(v.shape) for k, v in shapes.items()}


def weight_of(name):
    w = shapes.get(name + ".weight")
    return w


out = []
last_tensor_bytes = 0.0
for i, r in enumerate(rows):
    name, flops, us = r["layer"][5:], r["flops"], r["ms"] * 1e3
    by, kind = 0.0, "other"
    w = weight_of(name)
    m = re.match(r"fused linear (\d+)->(\d+)", name)
    if name.startswith("attention "):
        mm = re.match(r"attention (\d+)x(\d+) heads (\d+) dh (\d+)", name)
        tq, tk, h, dh = (int(x) for x in mm.groups())
        by = B * (2 * tq + 2 * tk) * h * dh * 4.0
        kind = "attention"
    elif m and flops > 0:
        ci, co = int(m.group(1)), int(m.group(2))
        tokens = flops / (2.0 * ci * co)
        by = ci * co * 4.0 + tokens * (ci + co) * 4.0
        kind = "linear"
    elif w is not None and len(w) == 4 and flops > 0:
        co, ci, kh, kw = w
        px = flops / (2.0 * ci * co * kh * kw)                      # output pixels over the batch
        stride2 = "downsamplers" in name
        by = ci * co * kh * kw * 4.0 + px * (4 if stride2 else 1) * ci * 4.0 + px * co * 4.0
        if name.endswith("conv2") or "proj_out" in name:
            by += px * co * 4.0                                     # residual / skip added in the epilogue
        kind = "conv3x3" if kh == 3 else "conv1x1"
    elif w is not None and len(w) == 2 and flops > 0:
        co, ci = w
        tokens = flops / (2.0 * ci * co)
        geglu = name.endswith("ff.net.0.proj")
        by = ci * co * 4.0 + tokens * ci * 4.0 + tokens * (co // 2 if geglu else co) * 4.0
        if name.endswith("to_out.0") or name.endswith("ff.net.2"):
            by += tokens * co * 4.0                                 # residual
        kind = "linear"
    elif flops == 0:
        kind = "norm / layout"
        # a normalisation reads and writes the tensor its consumer reads: take the next conv / linear's input
        for r2 in rows[i + 1:i + 4]:
            w2 = weight_of(r2["layer"][5:])
            m2 = re.match(r"fused linear (\d+)->(\d+)", r2["layer"][5:])
            if r2["flops"] > 0 and (w2 is not None or m2):
                if m2:
                    ci2, co2 = int(m2.group(1)), int(m2.group(2)); k2 = 1
                else:
                    co2, ci2 = w2[0], w2[1]; k2 = (w2[2] * w2[3]) if len(w2) == 4 else 1
                px2 = r2["flops"] / (2.0 * ci2 * co2 * k2) * (4 if "downsamplers" in r2["layer"] else 1)
                by = 2.0 * px2 * ci2 * 4.0
                break
    t_flop = flops / (PEAK_TF * 1e12) * 1e6
    t_byte = by / (HBM_TBS * 1e12) * 1e6
    floor = max(t_flop, t_byte, BOUNDARY_US)
    bound = "MFMA" if floor == t_flop else ("HBM" if floor == t_byte else "launch")
    out.append(dict(i=i, name=name, kind=kind, kernel=r["kernel"], gf=flops / 1e9, mb=by / 1e6, floor=floor, bound=bound, us=us, gap=us - floor))

tot_us, tot_floor = sum(o["us"] for o in out), sum(o["floor"] for o in out)
print(f"# MuseTalk UNet at batch {B}: per-op floor against measured ({os.path.basename(sys.argv[1])})\n")
print(f"floor = max(FLOP / {PEAK_TF:.0f} TF, operand bytes / {HBM_TBS:.0f} TB/s, {BOUNDARY_US} us); measured = hipEvents around every op, graph off.  "
      f"{len(out)} ops: measured **{tot_us / 1e3:.2f} ms**, sum of floors **{tot_floor / 1e3:.2f} ms** ({tot_us / tot_floor:.1f} x).\n")
print("## by kind\n\n| kind | ops | measured ms | floor ms | x | share of the gap |\n|---|---:|---:|---:|---:|---:|")
kinds = {}
for o in out:
    k = kinds.setdefault(o["kind"], [0, 0.0, 0.0])
    k[0] += 1; k[1] += o["us"]; k[2] += o["floor"]
gap_all = tot_us - tot_floor
for k, (n, u, f) in sorted(kinds.items(), key=lambda kv: -(kv[1][1] - kv[1][2])):
    print(f"| {k} | {n} | {u / 1e3:.3f} | {f / 1e3:.3f} | {u / f:.1f} | {100 * (u - f) / gap_all:.0f} % |")
print("\n## by what bounds the floor\n\n| floor set by | ops | measured ms | floor ms |\n|---|---:|---:|---:|")
for bnd in ("MFMA", "HBM", "launch"):
    sel = [o for o in out if o["bound"] == bnd]
    print(f"| {bnd} | {len(sel)} | {sum(o['us'] for o in sel) / 1e3:.3f} | {sum(o['floor'] for o in sel) / 1e3:.3f} |")
print("\n## every op, ranked by (measured - floor)\n\n| # | op | kernel | GFLOP | MB | floor us (by) | measured us | gap us |\n|---:|---|---|---:|---:|---:|---:|---:|")
for o in sorted(out, key=lambda o: -o["gap"]):
    print(f"| {o['i']} | {o['name'][:70]} | `{o['kernel'][:48]}` | {o['gf']:.2f} | {o['mb']:.1f} | {o['floor']:.1f} ({o['bound']}) | {o['us']:.1f} | {o['gap']:.1f} |")

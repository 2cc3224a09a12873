#!/usr/bin/env python3
"""Top UNet ops and time by kernel from a bench.py --dump-layers file:  python tools/unet_ops_from_dump.py gpurun_out/layers_b64.json [prefix]"""
import collections, json, sys
d = json.load(open(sys.argv[1]))
pre = sys.argv[2] if len(sys.argv) > 2 else "unet:"
rows = d["musetalk_rows"]
u = [r for r in rows if r["layer"].startswith(pre)]
print("sum of all ops", round(sum(r["ms"] for r in rows), 2), "ms;", pre, round(sum(r["ms"] for r in u), 2), "ms")
for r in sorted(u, key=lambda r: -r["ms"])[:40]:
    print(f"{r['layer'][len(pre):len(pre) + 62]:62s} {r['kernel'][:50]:50s} {r['ms'] * 1e3:8.1f} us {r['flops'] / max(r['ms'], 1e-9) / 1e9:6.0f} TF")
grp = collections.Counter()
for r in u:
    grp[r["kernel"].split(" grid")[0][:60]] += r["ms"]
for k, v in grp.most_common(14):
    print(f"{k:62s} {v:7.2f} ms")

#!/usr/bin/env python3
"""Pretty-prints the per-launch table bench.py --dump-layers wrote."""
import json
import sys

d = json.load(open(sys.argv[1]))
tot = sum(r["ms"] for r in d["rows"])
for r in d["rows"]:
    tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0
    print(f'{r["layer"]:26s} {r["kernel"]:40s} {r["ms"] * 1e3:8.1f} us {r["flops"] / 1e9:8.3f} GF {tf:7.1f} TF')
print(f"sum {tot * 1e3:.1f} us")

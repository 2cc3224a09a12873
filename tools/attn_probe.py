"""Error pattern + timing of the fused attention kernel per case (GPU box): python tools/attn_probe.py [time]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_attention import _qkv, _hip_attention
from oracle import musetalk_ref as R

cases = [(2, 1024, 1024, 8, 40), (2, 256, 256, 8, 80), (1, 100, 37, 2, 40)]
for prec in ("bf16x3", "bf16"):
    for c in cases:
        b, tq, tk, heads, dh = c
        q, k, v = _qkv(b, tq, tk, heads, dh, tq * 7 + tk)
        want = R.attention_core(q, k, v, heads)
        got = _hip_attention(q, k, v, heads, prec)
        e = (got - want).abs()
        print(prec, c, "L_inf %.3e" % e.max().item())
        if e.max() > (2e-4 if prec == "bf16x3" else 8e-2):
            eh = e.view(b, tq, heads, dh)
            print("  per batch", eh.amax((1, 2, 3)).numpy().round(3))
            print("  per head ", eh.amax((0, 1, 3)).numpy().round(3))
            print("  per chan ", eh.amax((0, 1, 2)).numpy().round(2))
            qe = eh.amax((0, 2, 3))
            print("  per query (first 80)", qe[:80].numpy().round(2))
            print("  frac queries bad", (qe > 0.05).float().mean().item())

#!/usr/bin/env python3
"""Reads a MF_DEBUG=copies MF_DEBUG_DUMP=<prefix> dump (mf_conv.hip: the first layer whose copies of one frame disagree although their inputs agree) and says WHAT is wrong
in the disagreeing item -- the decisive question of the packed-FMA study: is a wrong value an FMA result, a column sum, a bias, a row statistic?

For a LayerNorm-folded layer  y = rs_m * (acc_mn - mu_m * cs_n) + b_n.  Item 0 is taken as right, so acc_mn follows from it; for every wrong element of item k the
script then tests which single substitution explains the value:  the high-half operand of a broadcast pair used instead of the low one, a missing term, a neighbour's
column sum / bias, a stale statistic ...        python tools/pkfma_dump_analyze.py gpurun_out/pkdump"""
import sys
import numpy as np

pre = sys.argv[1]
meta = dict(l.split(" ", 1) for l in open(pre + "_meta.txt").read().strip().splitlines())
C, Wp, H, W, halo, coff, vC, tok = (int(meta[k]) for k in ("C", "Wp", "H", "W", "halo", "coff", "vC", "tokens"))
print({k: meta[k] for k in ("kernel", "cin", "cout", "Npad", "item", "act")})
y0 = np.fromfile(pre + "_y0.f32", np.float32).reshape(H + 2 * halo, Wp, C)[halo:halo + H, halo:halo + W, coff:coff + vC].reshape(H * W, vC)
yk = np.fromfile(pre + "_yk.f32", np.float32).reshape(H + 2 * halo, Wp, C)[halo:halo + H, halo:halo + W, coff:coff + vC].reshape(H * W, vC)
bad = np.argwhere(y0 != yk)
print(f"{len(bad)} of {y0.size} elements differ; rows (pixels) {np.unique(bad[:, 0]).size}, channels {np.unique(bad[:, 1]).size}")
# structure: group by (16-pixel fragment, channel)
frag = {}
for m, n in bad:
    frag.setdefault((m // 16, n), []).append(m % 16)
sizes = np.bincount([len(v) for v in frag.values()], minlength=17)
print("wrong elements per (16-row fragment, channel):", {i: int(c) for i, c in enumerate(sizes) if c})
chan_mod4 = np.bincount(bad[:, 1] % 4, minlength=4)
print("channel index mod 4 of wrong elements:", chan_mod4.tolist(), " mod 16:", np.bincount(bad[:, 1] % 16, minlength=16).tolist())
print("row index mod 16:", np.bincount(bad[:, 0] % 16, minlength=16).tolist(), " (row // 16) mod 8:", np.bincount((bad[:, 0] // 16) % 8, minlength=8).tolist())
try:
    st = np.fromfile(pre + "_stats.f64", np.float64).reshape(tok, 2)
    cs = np.fromfile(pre + "_cs.f32", np.float32)
except FileNotFoundError:
    st = None
b = np.fromfile(pre + "_bias.f32", np.float32)
if st is not None and int(meta["act"]) == 0:
    cin, eps = int(meta["cin"]), float(meta["ln_eps"])
    inv_c = np.float32(1.0 / cin)
    mean = st[:, 0] * np.float64(inv_c)
    var = st[:, 1] * np.float64(inv_c) - mean * mean
    mu = mean.astype(np.float32)
    rs = (1.0 / np.sqrt(np.maximum(var, 0.0) + np.float64(np.float32(eps)))).astype(np.float32)
    acc = (y0.astype(np.float64) - b[None, :vC]) / rs[:, None] + mu[:, None].astype(np.float64) * cs[None, :vC]        # what item 0 says the accumulators were
    hyp = {
        "no mean term: rs*acc + b": lambda m, n: rs[m] * acc[m, n] + b[n],
        "no rstd: (acc - mu*cs) + b": lambda m, n: (acc[m, n] - mu[m] * cs[n]) + b[n],
        "raw accumulator": lambda m, n: acc[m, n],
        "t only: acc - mu*cs": lambda m, n: acc[m, n] - mu[m] * cs[n],
        "no bias": lambda m, n: rs[m] * (acc[m, n] - mu[m] * cs[n]),
        "cs of channel n^1": lambda m, n: rs[m] * (acc[m, n] - mu[m] * cs[n ^ 1]) + b[n],
        "cs of channel n^2": lambda m, n: rs[m] * (acc[m, n] - mu[m] * cs[n ^ 2]) + b[n],
        "bias of channel n^1": lambda m, n: rs[m] * (acc[m, n] - mu[m] * cs[n]) + b[n ^ 1],
        "bias of channel n^2": lambda m, n: rs[m] * (acc[m, n] - mu[m] * cs[n]) + b[n ^ 2],
        "acc of channel n^1": lambda m, n: rs[m] * (acc[m, n ^ 1] - mu[m] * cs[n]) + b[n],
        "acc of channel n^2": lambda m, n: rs[m] * (acc[m, n ^ 2] - mu[m] * cs[n]) + b[n],
        "whole value of channel n^1": lambda m, n: y0[m, n ^ 1],
        "whole value of channel n^2": lambda m, n: y0[m, n ^ 2],
        "mu of row m^1": lambda m, n: rs[m] * (acc[m, n] - mu[m ^ 1] * cs[n]) + b[n],
        "rs used as mu": lambda m, n: rs[m] * (acc[m, n] - rs[m] * cs[n]) + b[n],
        "mu used as rs": lambda m, n: mu[m] * (acc[m, n] - mu[m] * cs[n]) + b[n],
    }
    score = {k: 0 for k in hyp}
    unexplained = []
    for m, n in bad:
        got = yk[m, n]
        hit = False
        for k, f in hyp.items():
            try:
                v = f(m, n)
            except IndexError:
                continue
            if abs(v - got) <= 2e-3 * max(1.0, abs(got)):
                score[k] += 1
                hit = True
        if not hit:
            unexplained.append((int(m), int(n), float(y0[m, n]), float(got), float(acc[m, n]), float(mu[m]), float(rs[m]), float(cs[n]), float(b[n])))
    print("hypotheses that reproduce a wrong element (within 2e-3 relative):", {k: v for k, v in score.items() if v})
    print(f"unexplained: {len(unexplained)}")
    for row in unexplained[:24]:
        print("  m %5d n %4d  right %+.5f  got %+.5f   acc %+.5f mu %+.5f rs %.5f cs %+.5f b %+.5f" % row)
    # is the error of a fragment's channel constant over its rows (a wrong bias), proportional to mu (a wrong column sum), or to rs?
    print("per wrong (fragment, channel) with >= 8 rows: error vs mu / rs correlation")
    shown = 0
    for (f, n), rows in frag.items():
        if len(rows) < 8 or shown >= 12:
            continue
        m = np.array([f * 16 + r for r in rows])
        e = (yk[m, n] - y0[m, n]).astype(np.float64)
        x1 = rs[m].astype(np.float64) * mu[m]                      # a wrong cs:   e = -rs*mu*dcs
        x2 = rs[m].astype(np.float64)                              # a wrong acc:  e = rs*dacc
        def fit(x):
            k = (x @ e) / (x @ x)
            return k, float(np.abs(e - k * x).max())
        k1, r1 = fit(x1); k2, r2 = fit(x2)
        print(f"  fragment {f} channel {n}: {len(rows)} rows; |e| max {np.abs(e).max():.3e}; e = -rs*mu*({-k1:+.4f}) residual {r1:.2e} | e = rs*({k2:+.4f}) residual {r2:.2e} | e constant? spread {e.max() - e.min():.2e}")
        shown += 1

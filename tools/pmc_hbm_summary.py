#!/usr/bin/env python3
"""HBM traffic and achieved GB/s per (kernel, grid) from two `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` result directories of the same
command (tools/pmc_hbm_step.sh; the two counters do not fit one pass -- MI355X_MICROARCH.md, PMC table).

    python tools/pmc_hbm_summary.py /tmp/fetch_dir /tmp/write_dir [min share of kernel time, default 0.005]

bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, counters in KiB: on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced
reads at 64 bytes, so the read side is doubled as the guide prescribes (an over-estimate for kernels whose reads are narrow); WRITE_SIZE is
used as reported.  GB/s = those bytes / the kernel-trace duration of the launch (mean of the two passes), against HBM3E's ~8000 GB/s.
Traffic well above a kernel's algorithmic bytes would be wasted re-reads; traffic below it is L2 / MALL reuse.
"""
import collections
import csv
import glob
import os
import sys


def load(d, ctr):
    tot, n, ns = collections.defaultdict(float), collections.defaultdict(int), collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            key = (r.get("Kernel_Name", "?"), r.get("Grid_Size", "?"))
            tot[key] += float(r["Counter_Value"]) * 1024.0
            n[key] += 1
            ns[key] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return tot, n, ns


def main(dfetch, dwrite, min_share=0.005):
    ft, fn, fns = load(dfetch, "FETCH_SIZE")
    wt, wn, wns = load(dwrite, "WRITE_SIZE")
    if not ft or not wt:
        print("missing counter rows:", len(ft), "FETCH_SIZE keys,", len(wt), "WRITE_SIZE keys")
        return
    total_ns = sum(fns.values()) or 1.0
    print("| kernel | grid | launches | avg us (under PMC) | % of kernel time | read MB / launch (2 x FETCH_SIZE) | written MB / launch | GB/s | of 8 TB/s |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    all_bytes = 0.0
    for key in sorted(ft, key=lambda k: -fns[k]):
        if key not in wt or not fn[key] or not wn[key]:
            continue
        rd = 2.0 * ft[key] / fn[key]
        wr = wt[key] / wn[key]
        us = 0.5 * (fns[key] / fn[key] + wns[key] / wn[key]) / 1e3
        all_bytes += (rd + wr) * fn[key]
        if fns[key] / total_ns < min_share:
            continue
        gbs = (rd + wr) / (us * 1e3)
        print(f"| `{key[0][:72]}` | {key[1]} | {fn[key]} | {us:.2f} | {100 * fns[key] / total_ns:.1f} | {rd / 1e6:.2f} | {wr / 1e6:.2f} | {gbs:.0f} | {gbs / 8000.0:.3f} |")
    print(f"\nall kernels: {all_bytes / 1e9:.2f} GB over {total_ns / 1e6:.2f} ms of kernel time under PMC = {all_bytes / total_ns:.0f} GB/s")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.005)

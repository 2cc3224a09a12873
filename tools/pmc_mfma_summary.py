#!/usr/bin/env python3
"""MFMA utilisation per (kernel, grid) from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv` directory (tools/pmc_mfma_step.sh).

    python tools/pmc_mfma_summary.py /tmp/pmc_dir [min share of kernel time, default 0.005]

Units (MI355X_MICROARCH.md, PMC table): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles with the matrix pipe busy, summed over the chip's
256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE counts busy cycles per XCD, summed over the 8 XCDs.  So
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
is the fraction of SIMD-cycles of the dispatch in which an MFMA was executing (at the clock the dispatch really ran at, which
`clock GHz` = GRBM_GUI_ACTIVE / 8 / wall ns reports).  The GRBM window of a dispatch is a little longer than its kernel-trace timestamps, so for
launches of a few tens of microseconds `clock GHz` reads above the 2.4 GHz maximum and mfma_util is a lower bound there; `util @2.4` =
SQ_VALU_MFMA_BUSY_CYCLES / (wall ns x 2.4 x 1024) is the same ratio against the kernel-trace duration at the nominal clock (a lower bound for
the long, power-capped launches instead).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY count quad-cycles per wave:
their ratios split a wave's life into parked (s_waitcnt / barrier), issue-stalled and issuing.
"""
import collections
import csv
import glob
import os
import sys


def main(d, min_share=0.005):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no *counter_collection.csv under", d)
        return
    rows = collections.defaultdict(lambda: collections.defaultdict(float))      # (kernel, grid) -> counter -> sum
    ndisp = collections.defaultdict(set)
    ns = collections.defaultdict(float)
    for f in files:
        for r in csv.DictReader(open(f)):
            key = (r.get("Kernel_Name", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")))
            did = r.get("Dispatch_Id", "0")
            rows[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if did not in ndisp[key]:
                ndisp[key].add(did)
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    ns[key] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    total_ns = sum(ns.values()) or 1.0
    print("| kernel | grid | launches | avg us (under PMC) | % of kernel time | clock GHz | mfma_util | util @2.4 | parked | issue-stalled | issuing |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    tot_busy = tot_simd_cycles = 0.0
    for key in sorted(rows, key=lambda k: -ns[k]):
        c, n = rows[key], len(ndisp[key])
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        per_xcd = gui / 8.0 if ns[key] and gui / ns[key] > 4.0 else gui
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        tot_busy += busy
        tot_simd_cycles += per_xcd * 1024.0
        if ns[key] / total_ns < min_share:
            continue
        util = busy / (per_xcd * 1024.0) if per_xcd else float("nan")
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or float("nan")
        print(f"| `{key[0][:72]}` | {key[1]} | {n} | {ns[key] / n / 1e3:.2f} | {100 * ns[key] / total_ns:.1f} | "
              f"{(per_xcd / ns[key]) if ns[key] else float('nan'):.2f} | {util:.3f} | {busy / (ns[key] * 2.4 * 1024.0):.3f} | {c.get('SQ_WAIT_ANY', 0.0) / wc:.2f} | "
              f"{c.get('SQ_WAIT_INST_ANY', 0.0) / wc:.2f} | {c.get('SQ_ACTIVE_INST_ANY', 0.0) / wc:.2f} |")
    if tot_simd_cycles:
        print(f"\nall kernels: mfma_util = {tot_busy / tot_simd_cycles:.3f} of the SIMD-cycles the GPU was busy ({total_ns / 1e6:.2f} ms of kernel time under PMC)")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.005)

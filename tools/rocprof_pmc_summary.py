#!/usr/bin/env python3
"""Per-kernel averages of the counters in a `rocprofv3 --pmc ... --output-format csv` result directory.

    python tools/rocprof_pmc_summary.py /tmp/pmc_dir [kernel-substring]
"""
import collections
import csv
import glob
import os
import sys


def main(d, needle=""):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no *counter_collection.csv under", d)
        for f in glob.glob(os.path.join(d, "**", "*"), recursive=True)[:20]:
            print("  ", f)
        return
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "?")
            if needle and needle not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r.get("Dispatch_Id", "0"))
    for k in sorted(acc, key=lambda k: -len(disp[k])):
        n = len(disp[k])
        print(f"{k[:110]}  dispatches={n}")
        for c, v in sorted(acc[k].items()):
            print(f"    {c:32s} {v / n:16.1f} per dispatch")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")

#!/usr/bin/env python3
"""Prints the kernel timeline of ONE forward from a rocprofv3 rocpd .db: start offset, duration and gap
to the previous dispatch on the same queue.  Usage: rocprof_timeline.py results.db [occurrence]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
occ = int(sys.argv[2]) if len(sys.argv) > 2 else -3
rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
heads = [i for i, r in enumerate(rows) if "k_head" in r[0]]
if len(heads) < abs(occ) + 1:
    sys.exit("not enough forwards in the trace")
lo, hi = heads[occ - 1] + 1, heads[occ] + 1
t0 = rows[lo][1]
last_end = {}
busy = 0
for name, s, e, q, st in rows[lo:hi]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    busy += e - s
    short = name.replace("void ", "").replace("(anonymous namespace)::", "")[:44]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  q{q} {short}")
print(f"span {(rows[hi - 1][2] - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us, {hi - lo} dispatches")

#!/usr/bin/env python3
"""The last N dispatches of a rocprofv3 rocpd .db in launch order: start offset, duration, gap to the previous kernel.

    python tools/rocprof_timeline.py results.db 80
"""
import sqlite3
import sys


def main(path, n):
    c = sqlite3.connect(path)
    rows = c.execute(f"select name, start, end from kernels order by start desc limit {int(n)}").fetchall()[::-1]
    t0, prev_end = rows[0][1], rows[0][1]
    for name, s, e in rows:
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:6.2f}  {name[:70]}")
        prev_end = e


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 80)

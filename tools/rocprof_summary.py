#!/usr/bin/env python3
"""Turns a rocprofv3 result (the rocpd sqlite .db that `rocprofv3 --kernel-trace --stats` writes on
ROCm 7.2, or its *_kernel_stats.csv) into the per-kernel summary committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r1/r1_results.db > profiles/r01_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary ({path.split('/')[-1]})\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, avg, mn, mx in rows:
        print(f"| `{name[:90]}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")
    print(f"\ntotal kernel time: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])

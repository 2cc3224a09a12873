#!/usr/bin/env python3
"""Turns a rocprofv3 result (the rocpd sqlite .db that `rocprofv3 --kernel-trace --stats` writes on ROCm 7.2) into the per-kernel summary
committed under profiles/.  Two tables: by kernel NAME (what `--stats` prints), and by (kernel, grid, workgroup) -- one template
instantiation serves layers of very different sizes (the 256-channel halo tile runs the VAE's 128 x 128 maps and the split 32 x 32
ones), and the roofline in bench.py is quoted per launch of the dominant (kernel, shape), so its average must be readable off a row.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/r02_kernel_stats_musetalk.md
"""
import sqlite3
import sys


def main(path, top=60):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary ({path.split('/')[-1]})\n")
    print("## by kernel name\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, avg, mn, mx in rows:
        print(f"| `{name[:90]}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")
    print(f"\ntotal kernel time: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    grid = [k for k in ("grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z") if k in cols][:3]
    wg = [k for k in ("workgroup_x", "workgroup_y", "workgroup_z", "workgroup_size_x", "workgroup_size_y", "workgroup_size_z") if k in cols][:3]
    if not grid:
        print(f"\n(no grid columns in the kernels view: {cols})")
        return
    sel = ", ".join(grid + wg)
    rows = c.execute(f"select name, {sel}, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                     f"group by name, {sel} order by sum(duration) desc limit {top}").fetchall()
    print(f"\n## by (kernel, grid x * y * z in work-items, workgroup) -- top {top} by total time\n")
    print("| kernel | grid | workgroup | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---:|---:|---:|---:|---:|---:|")
    for r in rows:
        name, g, w = r[0], r[1:1 + len(grid)], r[1 + len(grid):1 + len(grid) + len(wg)]
        n, tot, avg, mn, mx = r[1 + len(grid) + len(wg):]
        print(f"| `{name[:70]}` | {'x'.join(str(v) for v in g)} | {'x'.join(str(v) for v in w)} | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | "
              f"{mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])

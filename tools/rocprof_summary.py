#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share,
and per-(kernel, grid) rows so one shape of a templated kernel can be read off.
    python tools/rocprof_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys


def main(path, top=45):
    c = sqlite3.connect(path)
    tot = c.execute("select sum(end-start) from kernels").fetchone()[0]
    n = c.execute("select count(*) from kernels").fetchone()[0]
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("# %d dispatches, total kernel time %.3f ms" % (n, tot / 1e6))
    print("%-78s %7s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "share"))
    for name, cnt, t, avg in c.execute(
            "select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc limit ?", (top,)):
        print("%-78s %7d %12.1f %10.2f %6.2f%%" % (name[:78], cnt, t / 1e3, avg / 1e3, 100.0 * t / tot))
    print("\n# per (kernel, grid) for the conv family")
    print("%-58s %-16s %7s %12s %10s" % ("kernel", "grid(x,y,z)", "calls", "total_us", "avg_us"))
    for name, gx, gy, gz, cnt, t, avg in c.execute(
            "select name, grid_x, grid_y, grid_z, count(*), sum(end-start), avg(end-start) from kernels "
            "where name like '%conv_%' or name like '%dcn_%' group by name, grid_x, grid_y, grid_z order by 6 desc limit 40"):
        print("%-58s %-16s %7d %12.1f %10.2f" % (name[:58], "%d,%d,%d" % (gx, gy, gz), cnt, t / 1e3, avg / 1e3))


if __name__ == '__main__':
    main(sys.argv[1])

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
    timeline(c)


def timeline(c):
    """Occupancy of the device over the last two steps of the trace: time with 0 / 1 / 2 / >= 3 kernels in flight (stream
    lanes overlap kernels), and the kernels that run alone for the longest total time (the serial sections)."""
    rows = c.execute("select start, end, name from kernels order by start").fetchall()
    if len(rows) < 100:
        return
    t_end = max(r[1] for r in rows)
    # the eager profile run has no step markers: take the last 2/7 of the dispatches (two of seven steps)
    rows = rows[len(rows) * 5 // 7:]
    t0 = rows[0][0]
    ev = []
    for s, e, n in rows:
        ev.append((s, 1, n))
        ev.append((e, -1, n))
    ev.sort(key=lambda x: (x[0], x[1]))
    hist = {}
    alone = {}
    live = {}
    last = t0
    for t, d, n in ev:
        k = sum(live.values())
        dt = t - last
        if dt > 0:
            hist[min(k, 3)] = hist.get(min(k, 3), 0) + dt
            if k == 1:
                nm = next(iter(x for x, v in live.items() if v > 0))
                alone[nm] = alone.get(nm, 0) + dt
        live[n] = live.get(n, 0) + d
        last = t
    span = t_end - t0
    print("\n# timeline of the last two steps (%.2f ms): kernels in flight -> share of wall time" % (span / 1e6))
    for k in sorted(hist):
        print("#   %s kernels: %6.2f ms  %5.1f%%" % (("%d" % k) if k < 3 else ">=3", hist[k] / 1e6, 100.0 * hist[k] / span))
    print("# kernels running alone, by total time")
    for nm, t in sorted(alone.items(), key=lambda x: -x[1])[:12]:
        print("#   %-70s %8.2f ms" % (nm[:70], t / 1e6))


if __name__ == '__main__':
    main(sys.argv[1])

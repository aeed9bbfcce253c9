"""GPU occupancy of a traced run: fraction of wall time with >= 1 kernel running, average concurrency, and the largest
idle gaps with the kernels around them (rocprofv3 --kernel-trace sqlite).  usage: python tools/busy.py <db> [t0_frac t1_frac]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select start, end, name from kernels order by start"))
T0, T1 = rows[0][0], max(r[1] for r in rows)
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1.0)
a, b = T0 + (T1 - T0) * f0, T0 + (T1 - T0) * f1
rows = [r for r in rows if r[0] >= a and r[1] <= b]
ev = sorted([(r[0], 1) for r in rows] + [(r[1], -1) for r in rows])
busy = 0; conc_area = 0; cur = 0; last = ev[0][0]
for t, d in ev:
    if cur > 0: busy += t - last
    conc_area += cur * (t - last)
    cur += d; last = t
wall = ev[-1][0] - ev[0][0]
print('window %.2f ms, %d kernels, busy %.1f %%, mean concurrency while busy %.2f' % (wall / 1e6, len(rows), 100.0 * busy / wall, conc_area / max(busy, 1)))
# idle gaps
gaps = []
end = rows[0][1]; prev = rows[0]
for r in rows[1:]:
    if r[0] > end: gaps.append((r[0] - end, prev[2][:50], r[2][:50]))
    if r[1] > end: end = r[1]; prev = r
gaps.sort(reverse=True)
print('idle total %.2f ms in %d gaps; largest:' % (sum(g[0] for g in gaps) / 1e6, len(gaps)))
for g in gaps[:12]: print('  %.1f us  after %-50s before %s' % (g[0] / 1e3, g[1], g[2]))
import collections
hist = collections.Counter()
for g in gaps: hist[min(int(g[0] / 1e3), 20)] += 1
print('gap histogram (us: count):', sorted(hist.items()))

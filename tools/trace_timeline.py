"""Offline timeline analysis of a kernel trace pickled by the GPU box ((cols, rows) of the rocpd `kernels` view):
last step only -- wall span, kernels in flight histogram, idle gaps, per-queue busy time, kernels that run alone."""
import gzip, pickle, sys, collections
cols, rows = pickle.load(gzip.open(sys.argv[1]))
ix = {c: i for i, c in enumerate(cols)}
S, E, NM, Q = ix['start'], ix['end'], ix['name'], ix['queue_id']
# step boundaries: the adam kernel ends a step
ends = [r[S] for r in rows if 'adam_prep' in r[NM]] or [r[E] for r in rows if 'adam_kernel' in r[NM]]      # one per step
t1 = ends[-1]; t0 = ends[-2]
step = [r for r in rows if r[S] >= t0 and r[E] <= t1 + 1]
print('last step: %.3f ms, %d kernels, sum of kernel time %.3f ms' % ((t1 - t0) / 1e6, len(step), sum(r[E] - r[S] for r in step) / 1e6))
ev = []
for r in step:
    ev.append((r[S], 1, r)); ev.append((r[E], -1, r))
ev.sort(key=lambda x: (x[0], x[1]))
live = 0; last = t0; hist = collections.Counter(); alone = collections.Counter(); cur = {}
gaps = []
for t, d, r in ev:
    dt = t - last
    if dt > 0:
        hist[min(live, 4)] += dt
        if live == 1:
            alone[next(iter(cur.values()))[NM][:60]] += dt
        if live == 0:
            gaps.append(dt)
    if d > 0: cur[id(r)] = r
    else: cur.pop(id(r), None)
    live += d; last = t
span = t1 - t0
for k in sorted(hist): print('  %s in flight: %6.2f ms %5.1f%%' % (k if k < 4 else '>=4', hist[k] / 1e6, 100 * hist[k] / span))
print('  idle gaps: %d, mean %.2f us, total %.2f ms' % (len(gaps), sum(gaps) / max(len(gaps), 1) / 1e3, sum(gaps) / 1e6))
print('kernels running alone (top 15):')
for n, t in alone.most_common(15): print('   %6.2f ms  %s' % (t / 1e6, n))
qs = collections.Counter()
for r in step: qs[r[Q]] += r[E] - r[S]
print('per-queue busy ms:', {q: round(t / 1e6, 2) for q, t in qs.items()})
# launch-to-launch gaps on the same queue (dependent boundaries)
byq = collections.defaultdict(list)
for r in step: byq[r[Q]].append(r)
for q, rs in byq.items():
    rs.sort(key=lambda r: r[S])
    g = [b[S] - a[E] for a, b in zip(rs, rs[1:]) if b[S] >= a[E]]
    small = [x for x in g if x < 20000]
    print('queue %s: %d kernels, %d back-to-back gaps < 20 us: mean %.2f us, total %.2f ms' % (q, len(rs), len(small), sum(small) / max(len(small), 1) / 1e3, sum(small) / 1e6))

"""Where is the step a serial chain?  From a kernel trace pickle (tools/trace_dump.sh): the last replayed step, cut into
windows; per window the wall time, the union of kernel-busy time, the share with exactly one kernel in flight, and the kernels
that ran alone there (count, total us).  usage: trace_serial.py <trace.pkl.gz> [window_ms=1.0]"""
import gzip, pickle, re, sys, collections
sel, rows = pickle.load(gzip.open(sys.argv[1], 'rb'))
W = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 1e6
ad = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
step = rows[ad[-2] + 1:ad[-1] + 1]
t0, t1 = step[0][0], step[-1][1]
def short(n):
    n = re.sub(r'^void ', '', n)
    if n.startswith('_Z'):
        m = re.match(r'_Z(\d+)', n); L = int(m.group(1)); n = n[2 + len(m.group(1)):][:L]
    return re.sub(r'[<(].*', '', n)[:28]
print('step %.2f ms, %d launches' % ((t1 - t0) / 1e6, len(step)))
# sweep events; attribute time with concurrency 1 to the running kernel
ev = []
for i, r in enumerate(step):
    ev.append((r[0], 1, i)); ev.append((r[1], -1, i))
ev.sort()
live = set(); last = t0
alone = collections.defaultdict(float); alone_cnt = collections.Counter(); conc = collections.Counter()
win_alone = collections.defaultdict(lambda: collections.defaultdict(float)); win_idle = collections.Counter()
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        w = int((last - t0) / W)
        conc[len(live)] += dt
        if len(live) == 1:
            k = short(step[next(iter(live))][2]); alone[k] += dt; win_alone[w][k] += dt
        elif len(live) == 0:
            win_idle[w] += dt
    last = t
    if d == 1: live.add(i)
    else: live.discard(i)
tot = t1 - t0
print('concurrency share: ' + ', '.join('%d: %.1f%%' % (k, 100 * v / tot) for k, v in sorted(conc.items())))
cnt = collections.Counter(short(r[2]) for r in step)
print('kernels running ALONE (top 25 by time):')
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:25]:
    print('  %-30s %7.1f us alone   (%d launches in the step)' % (k, v / 1e3, cnt[k]))
print('per %.1f ms window: idle us | top alone kernels' % (W / 1e6))
for w in range(int(tot / W) + 1):
    tops = sorted(win_alone[w].items(), key=lambda kv: -kv[1])[:4]
    print('  %2d idle %5.0f | %s' % (w, win_idle[w] / 1e3, ', '.join('%s %.0f' % (k, v / 1e3) for k, v in tops)))

"""profiles/in_step.json from the kernel traces of a graph-mode bench run (one pickle per dtype, written on the GPU box by
tools/round_profile.sh: (cols, rows) of the rocpd `kernels` view).  Per dtype: launches per step, summed kernel time per
step, share of the step's wall time with no kernel in flight, and the dominant convolution's average duration inside the
step -- the numbers bench.py quotes next to its isolated-launch roofline.
usage: python tools/in_step_summary.py <source tag> f32=trace_f32.pkl.gz bf16=trace_bf16.pkl.gz > profiles/in_step.json"""
import gzip, json, pickle, sys

# (kernel-name substring, grid_x in threads) of the 3x3 convolutions on the 96x72 map with a 48-channel output block (f32: the
# split-product instance with three tiles per wave, 18 bands of 24 tiles x 20 frames x 512 threads; bf16: 36 bands of 12 tiles)
DOMINANT = {'f32': ('conv3x3_t4_kernel<float, 3, 7, true, 8, 3', 184320), 'bf16': ('conv3x3_t4_kernel', 368640)}
out = {}
tag = sys.argv[1]
for arg in sys.argv[2:]:
    dt, path = arg.split('=')
    cols, rows = pickle.load(gzip.open(path))
    ix = {c: i for i, c in enumerate(cols)}
    S, E, NM, GX = ix['start'], ix['end'], ix['name'], ix['grid_x']
    ends = [r[S] for r in rows if 'adam_prep' in r[NM]] or [r[E] for r in rows if 'adam_kernel' in r[NM]]      # one per step
    steps = []
    for t0, t1 in zip(ends[-3:-1], ends[-2:]):          # the last two steps of the trace
        st = [r for r in rows if r[S] >= t0 and r[E] <= t1 + 1]
        ev = sorted([(r[S], 1) for r in st] + [(r[E], -1) for r in st])
        live, last, idle = 0, t0, 0
        for t, d in ev:
            if live == 0:
                idle += t - last
            live += d
            last = t
        key, gx = DOMINANT[dt]
        dom = [r[E] - r[S] for r in st if key in r[NM] and r[GX] == gx]
        steps.append((len(st), sum(r[E] - r[S] for r in st), idle / (t1 - t0), (t1 - t0), sum(dom) / max(len(dom), 1), len(dom)))
    n = len(steps)
    out[dt] = {'source': tag, 'launches_per_step': round(sum(s[0] for s in steps) / n),
               'kernel_time_ms_per_step': round(sum(s[1] for s in steps) / n / 1e6, 2),
               'idle_share': round(sum(s[2] for s in steps) / n, 3),
               'profiled_step_ms': round(sum(s[3] for s in steps) / n / 1e6, 2),
               'dominant_kernel': DOMINANT[dt][0], 'dominant_avg_us': round(sum(s[4] for s in steps) / n / 1e3, 2),
               'dominant_launches_per_step': round(sum(s[5] for s in steps) / n)}
print(json.dumps(out, indent=1))

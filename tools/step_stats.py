"""ONE aggregation window for both in-step artefacts: from the kernel-trace pickles of a graph-mode
bench run (tools/round_profile.sh) take the replayed steps only (the spans between the last `adam_prep` launches, one per step: no
warm-up, no eager capture steps) and write
  * <out>_kernel_stats_<dtype>.txt : the per-kernel table (calls, total, average, share) and the per-(kernel, grid) rows
  * in_step.json                  : launches / kernel time / idle share per step and the dominant convolution's in-step average
so the dominant kernel's microseconds agree between the two by construction.
usage: python tools/step_stats.py <tag> <outprefix> f32=trace_f32.pkl.gz bf16=trace_bf16.pkl.gz   (in_step.json -> stdout)"""
import gzip, json, pickle, sys, collections

NSTEPS = 3      # replayed steps aggregated
# the 3x3 convolution of the 96x72 map with a 48-channel block, N = 20 frames (name substring; grid threads)
DOMINANT = {'f32': [('conv3x3_t5_kernel<float', 122880), ('conv3x3_t4_kernel<float, 3, 7, true, 8, 3', 184320)],
            'bf16': [('conv3x3_t6_kernel', 122880), ('conv3x3_t4_kernelIDF16b', 368640), ('conv3x3_t4_kernel', 368640)]}
tag, prefix = sys.argv[1], sys.argv[2]
out = {}
for arg in sys.argv[3:]:
    dt, path = arg.split('=')
    cols, rows = pickle.load(gzip.open(path))
    ix = {c: i for i, c in enumerate(cols)}
    S, E, NM, GX, GY, GZ = ix['start'], ix['end'], ix['name'], ix['grid_x'], ix['grid_y'], ix['grid_z']
    # one adam_prep launch per step (at its top since the optimizer update runs in two parts, at its end before): the spans
    # between consecutive ones are whole steps
    ends = [r[S] for r in rows if 'adam_prep' in r[NM]] or [r[E] for r in rows if 'adam_kernel' in r[NM]]
    spans = list(zip(ends[-NSTEPS - 1:-1], ends[-NSTEPS:]))
    win = [r for r in rows if any(r[S] >= t0 and r[S] < t1 for t0, t1 in spans)]
    wall = sum(t1 - t0 for t0, t1 in spans)
    tot = sum(r[E] - r[S] for r in win)
    by = collections.defaultdict(lambda: [0, 0])
    byg = collections.defaultdict(lambda: [0, 0])
    for r in win:
        by[r[NM]][0] += 1; by[r[NM]][1] += r[E] - r[S]
        if 'conv' in r[NM] or 'dcn' in r[NM]:
            k = (r[NM], r[GX], r[GY], r[GZ]); byg[k][0] += 1; byg[k][1] += r[E] - r[S]
    with open('%s_kernel_stats_%s.txt' % (prefix, dt), 'w') as f:
        f.write('# kernel trace of `rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-frozen '
                '--dtype %s --also none` (MI355X, hipGraph replay), restricted to the last %d REPLAYED steps (between adam_kernel '
                'launches): the window profiles/in_step.json is computed from.  Stream lanes overlap kernels, so durations include contention.\n' % (dt, NSTEPS))
        f.write('# %d dispatches in %.2f ms of wall time (%.2f ms per step), total kernel time %.3f ms\n' % (len(win), wall / 1e6, wall / 1e6 / NSTEPS, tot / 1e6))
        f.write('%-78s %7s %12s %10s %7s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'share'))
        for name, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:48]:
            f.write('%-78s %7d %12.1f %10.2f %6.2f%%\n' % (name[:78], c, t / 1e3, t / 1e3 / c, 100.0 * t / tot))
        f.write('\n# per (kernel, grid) for the conv / dcn family\n%-58s %-16s %7s %12s %10s\n' % ('kernel', 'grid(x,y,z)', 'calls', 'total_us', 'avg_us'))
        for (name, gx, gy, gz), (c, t) in sorted(byg.items(), key=lambda kv: -kv[1][1])[:40]:
            f.write('%-58s %-16s %7d %12.1f %10.2f\n' % (name[:58], '%d,%d,%d' % (gx, gy, gz), c, t / 1e3, t / 1e3 / c))
    # idle share and concurrency
    idle = 0
    for t0, t1 in spans:
        st = [r for r in rows if r[S] >= t0 and r[S] < t1]
        ev = sorted([(r[S], 1) for r in st] + [(r[E], -1) for r in st])
        live, last = 0, t0
        for t, d in ev:
            if live == 0:
                idle += t - last
            live += d
            last = t
    dom_name, dom = None, []
    for key, gx in DOMINANT[dt]:
        # (the launches that also carry a BatchNorm backward statistics epilogue -- last template argument 2 -- read one tensor more: not the kernel the bench line's roofline is about)
        # (... and neither are the round-6 instances that run the input BatchNorm's apply pass inside the launch: last template argument true)
        dom = [r[E] - r[S] for r in win if key in r[NM] and r[GX] == gx and ', 2>(' not in r[NM] and not ('conv3x3_t6' in r[NM] and ('true>(' in r[NM] or 'Lb1EEv' in r[NM]))]
        if dom:
            dom_name = key
            break
    out[dt] = {'source': tag, 'window': 'last %d replayed steps' % NSTEPS, 'launches_per_step': round(len(win) / NSTEPS),
               'kernel_time_ms_per_step': round(tot / NSTEPS / 1e6, 2), 'idle_share': round(idle / wall, 3),
               'profiled_step_ms': round(wall / NSTEPS / 1e6, 2), 'dominant_kernel': dom_name,
               'dominant_avg_us': round(sum(dom) / max(len(dom), 1) / 1e3, 2), 'dominant_launches_per_step': round(len(dom) / NSTEPS)}
print(json.dumps(out, indent=1))

"""Convolution with / without the BatchNorm statistics in its epilogue, per launch, on the four branch shapes (GPU box).
Chains timed as they run in the step: [conv, stats, apply] vs [conv+stats, apply]; [dgrad, bwd-stats, bwd-apply] vs
[dgrad+stats, bwd-apply]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N = int(os.environ.get('FB_N', 20))
def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
p = lambda t: None if t is None else t.data_ptr()
for dt in ('f32', 'bf16'):
    tdt = torch.bfloat16 if dt == 'bf16' else torch.float32
    for (H, W, C) in ((96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)):
        P = N * H * W
        x = torch.randn(N, H, W, C, device=dev).to(tdt); z = torch.empty_like(x); y = torch.empty_like(x)
        dy = torch.randn(N, H, W, C, device=dev).to(tdt); dx = torch.empty_like(x); dz = torch.empty_like(x)
        w = torch.randn(C, C, 3, 3, device=dev) * 0.05
        if dt == 'f32':
            wp0 = torch.empty(L.cdll.fami_packed_weight_elems(C, C, 3, 3, 0), device=dev); wp1 = torch.empty(L.cdll.fami_packed_weight_elems(C, C, 3, 3, 1), device=dev)
            L.call('fami_pack_conv_weight_f32', p(w), p(wp0), C, C, 3, 3, 0, st); L.call('fami_pack_conv_weight_f32', p(w), p(wp1), C, C, 3, 3, 1, st)
        else:
            wp0 = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, 3, 3, 0), device=dev, dtype=tdt); wp1 = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, 3, 3, 1), device=dev, dtype=tdt)
            L.call('fami_pack_conv_weight_bf16', p(w), p(wp0), C, C, 3, 3, 0, st); L.call('fami_pack_conv_weight_bf16', p(w), p(wp1), C, C, 3, 3, 1, st)
        mean, inv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        slots = torch.zeros(L.cdll.fami_bn_slots_bytes(C) // 4, device=dev)     # never re-zeroed here: timing only
        geo = (N, H, W, C, C, 3, 3, 1, 1, 1)
        def conv():
            if dt == 'f32': L.call('fami_conv2d_fwd_f32', p(x), p(wp0), None, None, p(z), *geo, 0, 0, st)
            else: L.call('fami_conv2d_fwd_bf16', p(x), p(wp0), None, p(z), *geo, 0, 0, 0, st)
        def conv_stats(): L.call('fami_conv2d_fwd_stats_' + dt, p(x), p(wp0), None, p(z), *geo, p(slots), p(rm), st)
        def bn2(): L.call('fami_bn_train_fwd2_' + dt, p(z), None, p(y), p(g), p(b), p(mean), p(inv), p(rm), p(rv), P, C, 1, 0.1, 1e-5, p(slots), st)
        def apply_slots(): L.call('fami_bn_apply_slots_' + dt, p(z), None, p(y), p(g), p(b), p(mean), p(inv), p(rm), p(rv), P, C, 1, 0.1, 1e-5, p(slots), st)
        def dgrad():
            if dt == 'f32': L.call('fami_conv2d_dgrad_f32', p(dy), p(wp1), None, p(dz), *geo, 0, st)
            else: L.call('fami_conv2d_dgrad_bf16', p(dy), p(wp1), p(dz), *geo, 0, st)
        def dgrad_stats(): L.call('fami_conv2d_dgrad_bnstats_' + dt, p(dy), p(wp1), p(dz), *geo, 0, p(z), None, p(mean), p(inv), p(g), p(b), 2, p(slots), st)
        def bwd2(): L.call('fami_bn_bwd2_' + dt, p(dz), p(z), None, p(mean), p(inv), p(g), p(b), p(dx), p(dg), p(db), None, P, C, 2, 0, 0, 0, p(slots), st)
        def bwd_slots(): L.call('fami_bn_bwd_apply_slots_' + dt, p(dz), p(z), p(mean), p(inv), p(g), p(b), p(dx), p(dg), p(db), None, P, C, 0, 0, 0, p(slots), st)
        conv(); torch.cuda.synchronize()
        t = {k: timeit(f) for k, f in (('conv', conv), ('conv+st', conv_stats), ('bn2', bn2), ('apply', apply_slots), ('dgrad', dgrad), ('dgrad+st', dgrad_stats),
                                        ('bwd2', bwd2), ('bwdapply', bwd_slots))}
        c0 = timeit(lambda: (conv(), bn2())); c1 = timeit(lambda: (conv_stats(), apply_slots()))
        d0 = timeit(lambda: (dgrad(), bwd2())); d1 = timeit(lambda: (dgrad_stats(), bwd_slots()))
        print('%s %3dx%-3d C=%-3d | conv %5.1f +stats %5.1f | bn2 %5.1f apply %5.1f | chain %5.1f -> %5.1f || dgrad %5.1f +stats %5.1f | bwd2 %5.1f apply %5.1f | chain %5.1f -> %5.1f us' %
              (dt, H, W, C, t['conv'], t['conv+st'], t['bn2'], t['apply'], c0, c1, t['dgrad'], t['dgrad+st'], t['bwd2'], t['bwdapply'], d0, d1), flush=True)

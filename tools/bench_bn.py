"""BatchNorm kernels: achieved bandwidth per launch on the four branch shapes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N = 20
def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for dt in ('f32', 'bf16'):
    tdt = torch.bfloat16 if dt == 'bf16' else torch.float32
    for (H, W, C) in ((96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384), (96, 72, 256)):
        P = N * H * W
        x = torch.randn(P, C, device=dev).to(tdt); y = torch.empty_like(x); dy = torch.randn(P, C, device=dev).to(tdt); dx = torch.empty_like(x)
        mean, inv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        ws = torch.empty(L.cdll.fami_bn_workspace(C) // 4, device=dev)
        nb = x.numel() * x.element_size()
        t1 = timeit(lambda: L.call('fami_bn_stats_' + dt, x.data_ptr(), P, C, mean.data_ptr(), inv.data_ptr(), None, None, 0.1, 1e-5, ws.data_ptr(), st))
        t2 = timeit(lambda: L.call('fami_bn_apply_' + dt, x.data_ptr(), mean.data_ptr(), inv.data_ptr(), g.data_ptr(), b.data_ptr(), None, y.data_ptr(), P, C, 1, st))
        t3 = timeit(lambda: L.call('fami_bn_bwd_' + dt, dy.data_ptr(), x.data_ptr(), y.data_ptr(), mean.data_ptr(), inv.data_ptr(), g.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, P, C, 1, 0, 0, 0, ws.data_ptr(), st))
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        slots = torch.zeros(L.cdll.fami_bn_slots_bytes(C) // 4, device=dev)     # never re-zeroed here: timing only
        t4 = timeit(lambda: L.call('fami_bn_train_fwd_' + dt, x.data_ptr(), None, y.data_ptr(), g.data_ptr(), b.data_ptr(), mean.data_ptr(), inv.data_ptr(), rm.data_ptr(), rv.data_ptr(), P, C, 1, 0.1, 1e-5, ws.data_ptr(), st))
        t5 = timeit(lambda: L.call('fami_bn_train_fwd2_' + dt, x.data_ptr(), None, y.data_ptr(), g.data_ptr(), b.data_ptr(), mean.data_ptr(), inv.data_ptr(), rm.data_ptr(), rv.data_ptr(), P, C, 1, 0.1, 1e-5, slots.data_ptr(), st))
        mean.zero_(); inv.fill_(1.0)
        t6 = timeit(lambda: L.call('fami_bn_bwd2_' + dt, dy.data_ptr(), x.data_ptr(), y.data_ptr(), mean.data_ptr(), inv.data_ptr(), g.data_ptr(), b.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, P, C, 1, 0, 0, 0, slots.data_ptr(), st))
        t7 = timeit(lambda: L.call('fami_bn_bwd2_' + dt, dy.data_ptr(), x.data_ptr(), y.data_ptr(), mean.data_ptr(), inv.data_ptr(), g.data_ptr(), b.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, P, C, 2, 0, 0, 0, slots.data_ptr(), st))
        print('   train_fwd 3-launch %5.1f us, 2-launch %5.1f us | bwd 3-launch %5.1f us, 2-launch (mask from y) %5.1f us, (mask from x) %5.1f us' % (t4, t5, t3, t6, t7))
        print('%s %3dx%-3d C=%-3d %6.1f MB | stats %5.1f us (%4.0f GB/s) | apply %5.1f us (%4.0f GB/s, r+w) | bwd %5.1f us (%4.0f GB/s: 2x(dy,x,y) + dx)' %
              (dt, H, W, C, nb / 1e6, t1, nb / t1 / 1e3, t2, 2 * nb / t2 / 1e3, t3, 7 * nb / t3 / 1e3))

"""16x16x4-tile vs 32x32x2-tile f32 implicit GEMM on the HRNet-W48 branch shapes (GPU box).
usage: python tools/bench_c32.py [sweep]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N = 20
shapes = ((96, 72, 48, 3), (48, 36, 96, 3), (24, 18, 192, 3), (12, 9, 384, 3), (96, 72, 64, 1), (96, 72, 256, 1))
sweep = len(sys.argv) > 1


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(n): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for H, W, C, k in shapes:
    w = torch.randn(C, C, k, k, device=dev) * 0.05
    x = torch.randn(N, H, W, C, device=dev); y = torch.empty_like(x)
    pad = k // 2
    fl = 2.0 * N * H * W * C * k * k * C
    for mode, ep in ((0, 'fami_conv2d_fwd_f32'), (1, 'fami_conv2d_dgrad_f32')):
        wp = torch.empty(L.cdll.fami_packed_weight_elems(C, C, k, k, mode), device=dev)
        L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), C, C, k, k, mode, st)
        if mode == 0:
            fn = lambda: L.call(ep, x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, k, k, 1, pad, 1, 0, 0, st)
        else:
            fn = lambda: L.call(ep, x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, k, k, 1, pad, 1, 0, st)
        res = []
        L.call('fami_conv_tune', 16, 0, 0); res.append(('t16', timeit(fn)))
        L.call('fami_conv_tune', 0, 0, 0); res.append(('t32auto', timeit(fn)))
        if sweep:
            for nt in (1, 2, 3):
                if nt > (C + 31) // 32: continue
                for ks in (1, 2, 4):
                    L.call('fami_conv_tune', 32, nt, ks); res.append(('nt%dks%d' % (nt, ks), timeit(fn)))
        L.call('fami_conv_tune', 0, 0, 0)
        print('%dx%d C=%3d k=%d %s: ' % (H, W, C, k, 'fwd' if mode == 0 else 'dgr') +
              '  '.join('%s %.1fus %.0fTF' % (n_, us, fl / us / 1e6) for n_, us in res), flush=True)

"""Does the last partial round of workgroups (the 'tail') dominate the f32 conv kernel?  Times the 48->48 3x3 conv at
pixel counts just below / at / just above a whole number of resident rounds (run by hand on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
C = 48
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
for dt in ('f32', 'bf16'):
    tdt = torch.bfloat16 if dt == 'bf16' else torch.float32
    if dt == 'bf16':
        wp = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, 3, 3, 0), device=dev, dtype=tdt)
        L.call('fami_pack_conv_weight_bf16', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, st)
    else:
        wp = torch.empty(L.cdll.fami_packed_weight_elems(C, C, 3, 3, 0), device=dev)
        L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, st)
    for (N, H, W) in ((15, 96, 72), (16, 64, 128), (17, 96, 72), (18, 96, 72), (19, 96, 72), (20, 96, 72), (32, 64, 64), (32, 64, 128), (33, 64, 128), (40, 96, 72)):
        x = torch.randn(N, H, W, C, device=dev).to(tdt); y = torch.empty_like(x)
        if dt == 'bf16':
            fn = lambda: L.call('fami_conv2d_fwd_bf16', x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, 0, st)
        else:
            fn = lambda: L.call('fami_conv2d_fwd_f32', x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, st)
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): fn()
        e1.record(s); e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        P = N * H * W
        print('%s N=%2d %3dx%-3d P=%7d  wgs(64px)=%6.1f  %7.1f us  %6.1f TF' % (dt, N, H, W, P, P / 64, us, 2.0 * P * C * 9 * C / us / 1e6))

"""Fixed per-launch overhead vs K-loop time of the implicit-GEMM conv: 1x1 and 3x3 at the same pixel count (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N, H, W = 20, 96, 72
for dt in ('f32', 'bf16'):
    tdt = torch.bfloat16 if dt == 'bf16' else torch.float32
    for C, k in ((48, 1), (48, 3), (96, 1), (96, 3), (192, 1), (192, 3)):
        w = torch.randn(C, C, k, k, device=dev) * 0.05
        x = torch.randn(N, H, W, C, device=dev).to(tdt); y = torch.empty_like(x)
        pad = k // 2
        if dt == 'bf16':
            wp = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, k, k, 0), device=dev, dtype=tdt)
            L.call('fami_pack_conv_weight_bf16', w.data_ptr(), wp.data_ptr(), C, C, k, k, 0, st)
            fn = lambda: L.call('fami_conv2d_fwd_bf16', x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, k, k, 1, pad, 1, 0, 0, 0, st)
        else:
            wp = torch.empty(L.cdll.fami_packed_weight_elems(C, C, k, k, 0), device=dev)
            L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), C, C, k, k, 0, st)
            fn = lambda: L.call('fami_conv2d_fwd_f32', x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, k, k, 1, pad, 1, 0, 0, st)
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): fn()
        e1.record(s); e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 2.0 * N * H * W * C * k * k * C
        peak = 2500.0 if dt == 'bf16' else 157.3
        print('%s C=%3d %dx%d  %7.1f us  %6.1f TF  (MFMA-only time %.1f us, in+out %.1f MB -> %.1f us at 5 TB/s)' %
              (dt, C, k, k, us, fl / us / 1e6, fl / peak / 1e6, 2 * x.numel() * x.element_size() / 1e6, 2 * x.numel() * x.element_size() / 5e6))

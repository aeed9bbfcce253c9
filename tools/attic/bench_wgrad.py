"""LDS-staged vs scalar-operand weight-gradient kernels (GPU box).  usage: python tools/bench_wgrad.py [f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N = 20
DT = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
cast = (lambda t: t.bfloat16()) if DT == 'bf16' else (lambda t: t)
for (H, W, C) in ((96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)):
    x = cast(torch.randn(N, H, W, C, device=dev)); dy = cast(torch.randn(N, H, W, C, device=dev))
    dw = torch.empty(C, C, 3, 3, device=dev)
    ref = None
    out = []
    for mode in (0, 1):      # 0 = scalar-operand kernels; 1 = LDS-staged kernel
        L.cdll.fami_conv_tune_wgrad_lds(mode)
        nb = L.cdll.fami_conv2d_wgrad_workspace(N, H, W, C, C, 3, 3, 1, 1, 1)
        ws = torch.empty(nb // 4, device=dev)
        fn = lambda: L.call('fami_conv2d_wgrad_' + DT, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, H, W, C, C, 3, 3, 1, 1, 1, 0, st)
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): fn()
        e1.record(s); e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        if ref is None: ref = dw.clone()
        err = ((dw - ref).abs().max() / ref.abs().max()).item()
        out.append('mode%d %.1fus (err %.1e)' % (mode, us, err))
    L.cdll.fami_conv_tune_wgrad_lds(1)
    print('%3dx%-3d C=%-3d ' % (H, W, C) + '  '.join(out))

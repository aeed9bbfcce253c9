"""Per-tap f32 weight-gradient kernels (GPU box): general (knob 50) against linear-address (51) form, pixel-split sweep.
   usage: python tools/bench_wgrad_ps.py [ps ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N = 20
PS = [int(a) for a in sys.argv[1:]] or [256, 384, 512, 768]
for (H, W, C) in ((96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)):
    x = torch.randn(N, H, W, C, device=dev); dy = torch.randn(N, H, W, C, device=dev)
    dw = torch.empty(C, C, 3, 3, device=dev)
    out = []
    L.cdll.fami_conv_tune_wgrad_lds(0)
    for knob in (50, 53, 51):
        L.cdll.fami_conv_tune_wgrad_lds(knob)
        for ps in PS:
            L.cdll.fami_conv_tune_wgrad_lds(1000 + (ps // 2 if knob == 51 else ps))
            nb = L.cdll.fami_conv2d_wgrad_workspace(N, H, W, C, C, 3, 3, 1, 1, 1)
            ws = torch.empty(nb // 4 + 1024, device=dev)
            fn = lambda: L.call('fami_conv2d_wgrad_f32', x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel() * 4, N, H, W, C, C, 3, 3, 1, 1, 1, 0, st)
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(20): fn()
            e1.record(s); e1.synchronize()
            out.append('%s/ps%d %.1f' % ({50: "gen", 51: "lin16", 52: "lin9", 53: "lin8"}[knob], ps, e0.elapsed_time(e1) / 20 * 1e3))
    L.cdll.fami_conv_tune_wgrad_lds(1000); L.cdll.fami_conv_tune_wgrad_lds(1)    # the LDS-staged f32 kernel wherever eligible
    nb = L.cdll.fami_conv2d_wgrad_workspace(N, H, W, C, C, 3, 3, 1, 1, 1)
    ws = torch.empty(nb // 4 + 1024, device=dev)
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(20): fn()
    e1.record(s); e1.synchronize()
    out.append('lds %.1f' % (e0.elapsed_time(e1) / 20 * 1e3))
    L.cdll.fami_conv_tune_wgrad_lds(-1)
    print('%3dx%-3d C=%-3d ' % (H, W, C) + '  '.join(out), flush=True)

"""f32 wgrad: input-channel tiles per workgroup sweep (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N = 20
for (H, W, C) in ((96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)):
    x = torch.randn(N, H, W, C, device=dev); dy = torch.randn(N, H, W, C, device=dev)
    dw = torch.empty(C, C, 3, 3, device=dev)
    out = []
    for mt in (0, 128, 256, 384, 768):
        L.cdll.fami_conv_tune_wgrad_lds(1000 + mt)
        nb = L.cdll.fami_conv2d_wgrad_workspace(N, H, W, C, C, 3, 3, 1, 1, 1)
        ws = torch.empty(nb // 4, device=dev)
        fn = lambda: L.call('fami_conv2d_wgrad_f32', x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, H, W, C, C, 3, 3, 1, 1, 1, 0, st)
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): fn()
        e1.record(s); e1.synchronize()
        out.append('ps%d %.1fus ws %.0fMB' % (mt, e0.elapsed_time(e1) / 20 * 1e3, nb / 1e6))
    L.cdll.fami_conv_tune_wgrad_lds(1000)
    print('%3dx%-3d C=%-3d ' % (H, W, C) + '  '.join(out))

"""Pipeline-depth x tile sweep of the implicit-GEMM conv on the four dominant shapes (run by hand on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N = 20
for dt in ('f32', 'bf16'):
    tdt = torch.bfloat16 if dt == 'bf16' else torch.float32
    for (H, W, C) in ((96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)):
        w = torch.randn(C, C, 3, 3, device=dev) * 0.05
        x = torch.randn(N, H, W, C, device=dev).to(tdt); y = torch.empty_like(x)
        if dt == 'bf16':
            wp = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, 3, 3, 0), device=dev, dtype=tdt)
            L.call('fami_pack_conv_weight_bf16', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, st)
            fn = lambda: L.call('fami_conv2d_fwd_bf16', x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, 0, st)
        else:
            wp = torch.empty(L.cdll.fami_packed_weight_elems(C, C, 3, 3, 0), device=dev)
            L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, st)
            fn = lambda: L.call('fami_conv2d_fwd_f32', x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, st)
        res = []
        for cfg in ((0, 0, 0), (1, 3, 1), (2, 3, 1), (1, 3, 2), (2, 3, 2), (1, 3, 4), (1, 4, 1), (2, 4, 1), (1, 4, 2), (2, 4, 2), (1, 4, 4), (2, 4, 4)):
            for stg in (2, 3, 4):
                L.cdll.fami_conv_tune(*cfg); L.cdll.fami_conv_tune_stages(stg)
                try:
                    for _ in range(3): fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                    for _ in range(20): fn()
                    e1.record(s); e1.synchronize()
                    us = e0.elapsed_time(e1) / 20 * 1e3
                    res.append((us, cfg, stg))
                except Exception as e:
                    pass
        L.cdll.fami_conv_tune(0, 0, 0); L.cdll.fami_conv_tune_stages(0)
        res.sort()
        fl = 2.0 * N * H * W * C * 9 * C
        print('%s %3dx%-3d C=%-3d ' % (dt, H, W, C) + '  '.join('%s/st%d %.1fus %.0fTF' % (c, g, u, fl / u / 1e6) for u, c, g in res[:6]) +
              '   | default ' + ' '.join('st%d %.1fus' % (g, u) for u, c, g in sorted(res, key=lambda r: r[2]) if c == (0, 0, 0)))

"""Stride-2 input gradient (f32): general tap walk (fami_conv_tune_stages(110)) against parity-class walk (111)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N = 20
for (H, W, Ci, Co) in ((192, 144, 64, 64), (96, 72, 48, 48), (96, 72, 48, 96), (48, 36, 96, 96), (48, 36, 96, 192), (24, 18, 192, 384)):
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
    dy = torch.randn(N, Ho, Wo, Co, device=dev)
    wp = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 1), device=dev)
    L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), Co, Ci, 3, 3, 1, st)
    outs, ts = [], []
    for knob in (110, 111):
        L.cdll.fami_conv_tune_stages(knob)
        dx = torch.full((N, H, W, Ci), 7.0, device=dev)
        fn = lambda: L.call('fami_conv2d_dgrad_f32', dy.data_ptr(), wp.data_ptr(), None, dx.data_ptr(), N, H, W, Ci, Co, 3, 3, 2, 1, 1, 0, st)
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): fn()
        e1.record(s); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3); outs.append(dx)
    L.cdll.fami_conv_tune_stages(111)
    print('%3dx%-3d %3d->%-3d s2 dgrad: general %.1f us  parity %.1f us  max |diff| %.2e (|dx| max %.2f)' %
          (H, W, Ci, Co, ts[0], ts[1], (outs[0] - outs[1]).abs().max().item(), outs[0].abs().max().item()), flush=True)

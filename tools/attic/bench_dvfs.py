"""Is the f32 conv power/clock limited?  Same launch on random vs all-zero operands (zeros toggle no datapath bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N, H, W = 20, 96, 72
for lds in (1, 0):
    L.cdll.fami_conv_tune_lds(lds)
    for dt in ('f32', 'bf16'):
        tdt = torch.bfloat16 if dt == 'bf16' else torch.float32
        for C in (48, 192):
            for fill in ('randn', 'zeros'):
                w = (torch.randn(C, C, 3, 3, device=dev) * 0.05) if fill == 'randn' else torch.zeros(C, C, 3, 3, device=dev)
                x = (torch.randn(N, H, W, C, device=dev) if fill == 'randn' else torch.zeros(N, H, W, C, device=dev)).to(tdt)
                y = torch.empty_like(x)
                if dt == 'bf16':
                    wp = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, 3, 3, 0), device=dev, dtype=tdt)
                    L.call('fami_pack_conv_weight_bf16', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, st)
                    fn = lambda: L.call('fami_conv2d_fwd_bf16', x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, 0, st)
                else:
                    wp = torch.empty(L.cdll.fami_packed_weight_elems(C, C, 3, 3, 0), device=dev)
                    L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, st)
                    fn = lambda: L.call('fami_conv2d_fwd_f32', x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, st)
                for _ in range(5): fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(30): fn()
                e1.record(s); e1.synchronize()
                us = e0.elapsed_time(e1) / 30 * 1e3
                print('lds=%d %s C=%3d %-5s %8.1f us %7.1f TF' % (lds, dt, C, fill, us, 2.0 * N * H * W * C * 9 * C / us / 1e6))

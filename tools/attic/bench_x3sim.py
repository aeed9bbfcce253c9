"""Cost simulation of a split-operand (3 x bf16 planes, 6 products) f32 conv on the LDS-staged kernel (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N = 20
def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(30): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3
for H, W, C in ((96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)):
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    out = []
    for dt in ('f32', 'bf16'):
        tdt = torch.bfloat16 if dt == 'bf16' else torch.float32
        x = torch.randn(N, H, W, C, device=dev).to(tdt); y = torch.empty_like(x)
        if dt == 'bf16':
            wp = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, 3, 3, 0), device=dev, dtype=tdt)
            L.call('fami_pack_conv_weight_bf16', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, st)
            fn = lambda: L.call('fami_conv2d_fwd_bf16', x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, 0, st)
            modes = ((0, 'direct'), (1, 'lds'), (2, 'lds-x3sim'))
        else:
            wp = torch.empty(L.cdll.fami_packed_weight_elems(C, C, 3, 3, 0), device=dev)
            L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, st)
            fn = lambda: L.call('fami_conv2d_fwd_f32', x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, st)
            modes = ((0, 'direct'), (1, 'lds'))
        for m, nm in modes:
            L.cdll.fami_conv_tune_lds(m)
            out.append('%s %s %.1f' % (dt, nm, timeit(fn)))
        L.cdll.fami_conv_tune_lds(-1)
    print('%dx%d C=%d: %s' % (H, W, C, '  '.join(out)), flush=True)

"""Wave-quantisation check: 48->48 3x3 @96x72 forward time against the frame count (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
H, W, C = 96, 72, 48
for dt in ('f32', 'bf16'):
    tdt = torch.bfloat16 if dt == 'bf16' else torch.float32
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    if dt == 'bf16':
        wp = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, 3, 3, 0), device=dev, dtype=tdt)
        L.call('fami_pack_conv_weight_bf16', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, st)
    else:
        wp = torch.empty(L.cdll.fami_packed_weight_elems(C, C, 3, 3, 0), device=dev)
        L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, st)
    out = []
    for N in (8, 12, 16, 17, 18, 19, 20, 21, 22, 24, 28, 32, 36, 38, 40, 44):
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
        out.append('N=%d tiles=%d %.1fus (%.2f us/frame, %.0f TF)' % (N, N * H * W // 16, us, us / N, 2.0 * N * H * W * C * C * 9 / us / 1e6))
    print(dt + '\n  ' + '\n  '.join(out), flush=True)

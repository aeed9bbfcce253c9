"""Micro-benchmark of the conv kernels on the four dominant HRNet-W48 shapes (run by hand on the GPU box)."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib()
dev = torch.device('cuda:0')
s = torch.cuda.current_stream(dev)
N = int(os.environ.get('N', 20))
DT = os.environ.get('DTYPE', 'f32')
BF = DT == 'bf16'
tdt = torch.bfloat16 if BF else torch.float32
SHAPES = [(96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)]

def time_it(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us

def run(mode, H, W, C, cfg):
    x = torch.randn(N, H, W, C, device=dev).to(tdt); y = torch.empty_like(x)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    if BF:
        wp = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, 3, 3, mode), device=dev, dtype=tdt)
        L.call('fami_pack_conv_weight_bf16', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, mode, s.cuda_stream)
    else:
        wp = torch.empty(L.cdll.fami_packed_weight_elems(C, C, 3, 3, mode), device=dev)
        L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, mode, s.cuda_stream)
    L.cdll.fami_conv_tune(*cfg)
    if mode == 0 and BF:
        fn = lambda: L.call('fami_conv2d_fwd_bf16', x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, 0, s.cuda_stream)
    elif BF:
        fn = lambda: L.call('fami_conv2d_dgrad_bf16', x.data_ptr(), wp.data_ptr(), y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, s.cuda_stream)
    elif mode == 0:
        fn = lambda: L.call('fami_conv2d_fwd_f32', x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, s.cuda_stream)
    else:
        fn = lambda: L.call('fami_conv2d_dgrad_f32', x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, s.cuda_stream)
    try:
        us = time_it(fn)
    except Exception as e:
        return None
    finally:
        L.cdll.fami_conv_tune(0, 0, 0)
    return us

flops = lambda H, W, C: 2.0 * N * H * W * C * 9 * C
cfgs = [(0, 0, 0), (4, 3, 1), (2, 3, 1), (1, 3, 1), (2, 6, 1), (1, 6, 1), (2, 3, 2), (1, 3, 2), (2, 6, 2), (1, 6, 2), (1, 3, 4), (2, 3, 4), (1, 6, 4), (2, 6, 4), (2, 4, 1), (4, 4, 1)]
if BF:
    cfgs = [(0, 0, 0), (4, 3, 1), (2, 3, 1), (1, 3, 1), (4, 4, 1), (2, 4, 1), (1, 4, 1), (4, 2, 1), (2, 2, 1), (2, 3, 2), (1, 3, 2), (2, 4, 2), (1, 4, 2), (1, 3, 4), (2, 3, 4), (1, 4, 4), (2, 4, 4)]
for (H, W, C) in SHAPES:
    for mode in (0, 1):
        res = []
        for cfg in cfgs:
            us = run(mode, H, W, C, cfg)
            if us: res.append((cfg, us, flops(H, W, C) / us / 1e6))
        print('%3dx%-3d C=%-3d %s: ' % (H, W, C, 'fwd ' if mode == 0 else 'dgrd') + '  '.join('%s %.0fus %.0fTF' % (c, u, t) for c, u, t in res))
# wgrad
for (H, W, C) in SHAPES:
    x = torch.randn(N, H, W, C, device=dev).to(tdt); dy = torch.randn(N, H, W, C, device=dev).to(tdt)
    dw = torch.empty(C, C, 3, 3, device=dev)
    nb = L.cdll.fami_conv2d_wgrad_workspace(N, H, W, C, C, 3, 3, 1, 1, 1)
    ws = torch.empty(nb // 4, device=dev)
    fn = lambda: L.call('fami_conv2d_wgrad_' + DT, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, H, W, C, C, 3, 3, 1, 1, 1, 0, s.cuda_stream)
    us = time_it(fn)
    print('%3dx%-3d C=%-3d wgrad: %.0fus %.0fTF (ws %.1f MB)' % (H, W, C, us, flops(H, W, C) / us / 1e6, nb / 1e6))

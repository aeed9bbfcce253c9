import os, sys
sys.path.insert(0, '/root/repo')
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
def timeit(fn, reps=40):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
p = lambda t: None if t is None else t.data_ptr()
import ctypes
for (N, H, W, Ci, Co) in [(20, 96, 72, 48, 48), (20, 48, 36, 96, 96)]:
    x = torch.randn(N, H, W, Ci, device=dev).bfloat16(); dy = torch.randn(N, H, W, Co, device=dev).bfloat16()
    dw = torch.empty(Co, Ci, 3, 3, device=dev)
    geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
    nb = L.cdll.fami_conv2d_wgrad_workspace(*geo); ws = torch.empty(nb // 4 + 4, device=dev)
    desc = (ctypes.c_long * 16)()
    full = lambda: L.call('fami_conv2d_wgrad_bf16', p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, st)
    defer = lambda: L.call('fami_conv2d_wgrad_defer_bf16', p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, desc, st)
    out = ['%s kernel+reduce %.1f' % ((N, H, W, Ci, Co), timeit(full))]
    for abl, nm in ((0, 'kernel only'), (1, 'no K loop'), (2, 'no slab store'), (4, 'no restaging'), (3, 'no K, no slab'), (7, 'skeleton')):
        L.cdll.fami_conv_tune_wgrad_lds(22000 + abl)
        out.append('%s %.1f' % (nm, timeit(defer)))
    L.cdll.fami_conv_tune_wgrad_lds(-1)
    print(' | '.join(out), flush=True)

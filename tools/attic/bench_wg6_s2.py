import os, sys
sys.path.insert(0, '/root/repo')
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
p = lambda t: t.data_ptr()
def timeit(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (N, H, W, Ci, Co) in ((20, 96, 72, 48, 96), (20, 96, 72, 48, 48), (20, 48, 36, 96, 192), (20, 48, 36, 96, 96), (20, 24, 18, 192, 384)):
    x = torch.randn(N, H, W, Ci, device=dev).bfloat16(); dy = torch.randn(N, H // 2, W // 2, Co, device=dev).bfloat16()
    dw = torch.empty(Co, Ci, 3, 3, device=dev); geo = (N, H, W, Ci, Co, 3, 3, 2, 1, 1)
    nb = L.cdll.fami_conv2d_wgrad_workspace(*geo); ws = torch.empty(nb // 4 + 4, device=dev)
    run = lambda: L.call('fami_conv2d_wgrad_bf16', p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, st)
    r = []
    for code in (23004, 23005):
        L.cdll.fami_conv_tune_wgrad_lds(-1); L.cdll.fami_conv_tune_wgrad_lds(code); r.append(timeit(run))
    L.cdll.fami_conv_tune_wgrad_lds(-1)
    print('s2 wgrad %dx%d %d->%d: wg16 %.1f us, wg6 %.1f us' % (H, W, Ci, Co, r[0], r[1]), flush=True)

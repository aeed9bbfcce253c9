"""conv_wgrad6_kernel (DMA-staged 48-channel 16-bit weight gradient, conv_wg16.hip) against conv_wgrad16_kernel: result against
fp64 on the same 16-bit operands, time per launch incl. the slab reduce, rows-per-band sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
p = lambda t: None if t is None else t.data_ptr()
def timeit(fn, reps=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for DT in ('bf16', 'f16'):
    tdt = {'bf16': torch.bfloat16, 'f16': torch.float16}[DT]
    for (N, H, W, C) in ((20, 96, 72, 48), (20, 48, 36, 96), (20, 24, 18, 192), (20, 12, 9, 384), (4, 96, 72, 48)):
        torch.manual_seed(2)
        x = torch.randn(N, H, W, C, device=dev).to(tdt); dy = (torch.randn(N, H, W, C, device=dev) * 0.1).to(tdt)
        dw = torch.empty(C, C, 3, 3, device=dev)
        geo = (N, H, W, C, C, 3, 3, 1, 1, 1)
        nb = L.cdll.fami_conv2d_wgrad_workspace(*geo)
        ws = torch.empty(nb // 4 + 4, device=dev)
        run = lambda: L.call('fami_conv2d_wgrad_' + DT, p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, st)
        xd = x.double().permute(0, 3, 1, 2).requires_grad_(False)
        wref = torch.zeros(C, C, 3, 3, device=dev, dtype=torch.double, requires_grad=True)
        F.conv2d(xd, wref, padding=1).backward(dy.double().permute(0, 3, 1, 2))
        ref = wref.grad
        res = []
        for name, code in (('wg16', 23000), ('wg6', 23001)):
            L.cdll.fami_conv_tune_wgrad_lds(-1); L.cdll.fami_conv_tune_wgrad_lds(code)
            dw.zero_(); run(); torch.cuda.synchronize()
            err = ((dw.double() - ref).abs().max() / ref.abs().max()).item()
            res.append('%s err %.1e %.1f us' % (name, err, timeit(run)))
        for tg in (64, 96, 160, 240, 480):
            L.cdll.fami_conv_tune_wgrad_lds(23400 + tg); res.append('tg%d %.1f' % (tg, timeit(run)))
        L.cdll.fami_conv_tune_wgrad_lds(-1)
        print('%s N%d %dx%d C%d | ' % (DT, N, H, W, C) + ' | '.join(res), flush=True)

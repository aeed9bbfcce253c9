"""16-bit weight gradient of the 3x3 stride-1 convolutions (incl. its slab reduce) on the HRNet-W48 branch shapes:
round-1 kernel (conv_wgrad_h_kernel) vs conv_wg16.hip with its tiles-per-run / workgroup-target sweeps.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N = int(os.environ.get('FB_N', 20))
def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
p = lambda t: None if t is None else t.data_ptr()
tdt = torch.bfloat16
SHAPES = [(96, 72, 48, 48, 3, 1, 1, 1), (48, 36, 96, 96, 3, 1, 1, 1), (24, 18, 192, 192, 3, 1, 1, 1), (12, 9, 384, 384, 3, 1, 1, 1),
          (96, 72, 192, 48, 3, 1, 1, 1), (96, 72, 96, 48, 3, 1, 1, 1),
          # round 3, generalised geometry: stride 2 (fuse / transition chains), dilation 3 (DCN predictors, B = 4 frames), 1x1
          (96, 72, 48, 48, 3, 2, 1, 1), (96, 72, 48, 96, 3, 2, 1, 1), (48, 36, 96, 192, 3, 2, 1, 1), (192, 144, 64, 64, 3, 2, 1, 1),
          (96, 72, 48, 216, 3, 1, 3, 3), (96, 72, 48, 108, 3, 1, 3, 3), (96, 72, 64, 256, 1, 1, 0, 1), (96, 72, 256, 64, 1, 1, 0, 1),
          (48, 36, 96, 48, 1, 1, 0, 1), (12, 9, 384, 48, 1, 1, 0, 1)]
for (H, W, Ci, Co, k, stn, pad, dil) in SHAPES:
    Nn = 4 if dil == 3 else N
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // stn + 1, (W + 2 * pad - dil * (k - 1) - 1) // stn + 1
    x = torch.randn(Nn, H, W, Ci, device=dev).to(tdt); dy = torch.randn(Nn, Ho, Wo, Co, device=dev).to(tdt)
    dw = torch.empty(Co, Ci, k, k, device=dev)
    geo = (Nn, H, W, Ci, Co, k, k, stn, pad, dil)
    res = []
    def run():
        nb = L.cdll.fami_conv2d_wgrad_workspace(*geo)
        ws = torch.empty(nb // 4 + 4, device=dev)
        return timeit(lambda: L.call('fami_conv2d_wgrad_bf16', p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, st))
    L.cdll.fami_conv_tune_wgrad_lds(20000); res.append(('r1', run()))
    L.cdll.fami_conv_tune_wgrad_lds(20001)
    for bt in (0, 16, 12, 8):
        L.cdll.fami_conv_tune_wgrad_lds(20100 + bt); res.append(('wg16/bt%d' % bt, run()))
    L.cdll.fami_conv_tune_wgrad_lds(20100)
    for tg in (192, 256, 384, 512):
        L.cdll.fami_conv_tune_wgrad_lds(21000 + tg); res.append(('tg%d' % tg, run()))
    L.cdll.fami_conv_tune_wgrad_lds(-1)
    print('%3dx%-3d %3d->%-3d k%d s%d d%d | ' % (H, W, Ci, Co, k, stn, dil) + ' | '.join('%s %.1f' % r for r in res), flush=True)

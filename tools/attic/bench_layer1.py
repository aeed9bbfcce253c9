"""Stem / layer1 kernels of the step (the serial head and tail: one lane, tensors of up to 70 MB in bf16): time per launch
alone against the HBM floor of the launch (bytes / 5 TB/s) and its MFMA floor.  Forward, input gradient, weight gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
p = lambda t: None if t is None else t.data_ptr()
DT = os.environ.get('DT', 'bf16'); tdt = {'bf16': torch.bfloat16, 'f32': torch.float32}[DT]; sz = 2 if DT == 'bf16' else 4
sfx = '_' + DT
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N = 20
for (name, H, W, Ci, Co, k, stn, pad) in (('stem conv1 3->64 s2', 384, 288, 3, 64, 3, 2, 1), ('stem conv2 64->64 s2', 192, 144, 64, 64, 3, 2, 1),
                                          ('l1 1x1 64->64', 96, 72, 64, 64, 1, 1, 0), ('l1 3x3 64->64', 96, 72, 64, 64, 3, 1, 1),
                                          ('l1 1x1 64->256', 96, 72, 64, 256, 1, 1, 0), ('l1 1x1 256->64', 96, 72, 256, 64, 1, 1, 0),
                                          ('tr 3x3 256->48', 96, 72, 256, 48, 3, 1, 1), ('tr 3x3 256->96 s2', 96, 72, 256, 96, 3, 2, 1)):
    Ho, Wo = (H + 2 * pad - k) // stn + 1, (W + 2 * pad - k) // stn + 1
    Cx = Ci if Ci >= 4 else 4
    x = torch.randn(N, H, W, Ci, device=dev).to(tdt); y = torch.empty(N, Ho, Wo, Co, device=dev, dtype=tdt)
    dy = torch.randn(N, Ho, Wo, Co, device=dev).to(tdt); dx = torch.empty_like(x)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05; dw = torch.empty_like(w)
    geo = (N, H, W, Ci, Co, k, k, stn, pad, 1)
    pe = getattr(L.cdll, 'fami_packed_weight_elems' + ('_bf16' if DT == 'bf16' else ''))
    wp0 = torch.empty(pe(Co, Ci, k, k, 0), device=dev, dtype=tdt); wp1 = torch.empty(pe(Co, Ci, k, k, 1), device=dev, dtype=tdt)
    L.call('fami_pack_conv_weight' + sfx, p(w), p(wp0), Co, Ci, k, k, 0, st); L.call('fami_pack_conv_weight' + sfx, p(w), p(wp1), Co, Ci, k, k, 1, st)
    nb = L.cdll.fami_conv2d_wgrad_workspace(*geo); ws = torch.empty(nb // 4 + 4, device=dev)
    if DT == 'bf16':
        fwd = lambda: L.call('fami_conv2d_fwd_bf16', p(x), p(wp0), None, p(y), *geo, 0, 0, 0, st)
        bwd = lambda: L.call('fami_conv2d_dgrad_bf16', p(dy), p(wp1), p(dx), *geo, 0, st)
    else:
        fwd = lambda: L.call('fami_conv2d_fwd_f32', p(x), p(wp0), None, None, p(y), *geo, 0, 0, st)
        bwd = lambda: L.call('fami_conv2d_dgrad_f32', p(dy), p(wp1), None, p(dx), *geo, 0, st)
    wg = lambda: L.call('fami_conv2d_wgrad' + sfx, p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, st)
    fl = 2.0 * N * Ho * Wo * Ci * Co * k * k
    byts = (x.numel() + y.numel()) * sz
    tf, tb, tw = timeit(fwd), (timeit(bwd) if Ci >= 8 else float('nan')), timeit(wg)
    print('%s %-22s %6.2f GFLOP %6.1f MB | HBM floor %5.1f us, MFMA floor %5.1f us | fwd %6.1f  dgrad %6.1f  wgrad(+reduce) %6.1f us' %
          (DT, name, fl / 1e9, byts / 1e6, byts / 5e6, fl / (2.5e9 if DT == 'bf16' else 4.17e8), tf, tb, tw), flush=True)

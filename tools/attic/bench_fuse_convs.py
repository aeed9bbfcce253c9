"""The fuse-layer / transition convolutions of the f32 step that still run on the exact-f32 MFMA kernels (stride-2 3x3, 1x1): time per
launch alone (forward, input gradient, weight gradient + reduce) against the launch's HBM floor (bytes / 5 TB/s) and its exact-f32
MFMA floor (157.3 TFLOP/s).  DT=f32 | bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
p = lambda t: None if t is None else t.data_ptr()
DT = os.environ.get('DT', 'f32'); tdt = {'bf16': torch.bfloat16, 'f32': torch.float32}[DT]; sz = 2 if DT == 'bf16' else 4
sfx = '_' + DT
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N = 20
SHAPES = (('s2 48->48 @96x72', 96, 72, 48, 48, 3, 2, 1), ('s2 48->96 @96x72', 96, 72, 48, 96, 3, 2, 1), ('s2 48->48 @48x36', 48, 36, 48, 48, 3, 2, 1),
          ('s2 48->192 @48x36', 48, 36, 48, 192, 3, 2, 1), ('s2 96->96 @48x36', 48, 36, 96, 96, 3, 2, 1), ('s2 96->192 @48x36', 48, 36, 96, 192, 3, 2, 1),
          ('s2 48->384 @24x18', 24, 18, 48, 384, 3, 2, 1), ('s2 96->384 @24x18', 24, 18, 96, 384, 3, 2, 1), ('s2 192->384 @24x18', 24, 18, 192, 384, 3, 2, 1),
          ('1x1 96->48 @48x36', 48, 36, 96, 48, 1, 1, 0), ('1x1 192->48 @24x18', 24, 18, 192, 48, 1, 1, 0), ('1x1 384->48 @12x9', 12, 9, 384, 48, 1, 1, 0),
          ('1x1 192->96 @24x18', 24, 18, 192, 96, 1, 1, 0), ('1x1 384->96 @12x9', 12, 9, 384, 96, 1, 1, 0), ('1x1 384->192 @12x9', 12, 9, 384, 192, 1, 1, 0))
for (name, H, W, Ci, Co, k, stn, pad) in SHAPES:
    Ho, Wo = (H + 2 * pad - k) // stn + 1, (W + 2 * pad - k) // stn + 1
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
    tf, tb, tw = timeit(fwd), timeit(bwd), timeit(wg)
    print('%s %-20s %6.2f GFLOP %5.1f MB | HBM floor %5.1f us, f32-MFMA floor %5.1f us | fwd %6.1f  dgrad %6.1f  wgrad(+reduce) %6.1f us' %
          (DT, name, fl / 1e9, byts / 1e6, byts / 5e6, fl / 157.3e6, tf, tb, tw), flush=True)

"""The convolutions of the alignment head's serial chain (Alignment_V15.py:144-158: dilated 3x3 offset / mask predictors over the
48-channel 96x72 map, B = 8 frames), per launch and alone: forward, input gradient, weight gradient (with its slab reduce)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
p = lambda t: None if t is None else t.data_ptr()
def timeit(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
DT = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
tdt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[DT]
for (N, H, W, Ci, Co, pad, dil) in ((8, 96, 72, 48, 216, 3, 3), (8, 96, 72, 48, 108, 3, 3), (8, 96, 72, 48, 48, 1, 1), (8, 96, 72, 96, 48, 1, 1)):
    x = torch.randn(N, H, W, Ci, device=dev).to(tdt); y = torch.empty(N, H, W, Co, device=dev, dtype=tdt)
    dy = torch.randn(N, H, W, Co, device=dev).to(tdt); dx = torch.empty_like(x)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05; dw = torch.empty_like(w)
    sfx = '_' + DT
    elems = L.cdll.fami_packed_weight_elems if DT == 'f32' else L.cdll.fami_packed_weight_elems_bf16
    wp0 = torch.empty(elems(Co, Ci, 3, 3, 0), device=dev, dtype=tdt); wp1 = torch.empty(elems(Co, Ci, 3, 3, 1), device=dev, dtype=tdt)
    L.call('fami_pack_conv_weight' + sfx, p(w), p(wp0), Co, Ci, 3, 3, 0, st); L.call('fami_pack_conv_weight' + sfx, p(w), p(wp1), Co, Ci, 3, 3, 1, st)
    geo = (N, H, W, Ci, Co, 3, 3, 1, pad, dil)
    ws = torch.empty(L.cdll.fami_conv2d_wgrad_workspace(*geo) // 4 + 16, device=dev)
    if DT == 'f32':
        fwd = lambda: L.call('fami_conv2d_fwd_f32', p(x), p(wp0), None, None, p(y), *geo, 0, 0, st)
        bwd = lambda: L.call('fami_conv2d_dgrad_f32', p(dy), p(wp1), None, p(dx), *geo, 0, st)
    else:
        fwd = lambda: L.call('fami_conv2d_fwd' + sfx, p(x), p(wp0), None, p(y), *geo, 0, 0, 0, st)
        bwd = lambda: L.call('fami_conv2d_dgrad' + sfx, p(dy), p(wp1), p(dx), *geo, 0, st)
    wg = lambda: L.call('fami_conv2d_wgrad' + sfx, p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, st)
    gf = 2.0 * N * H * W * Ci * Co * 9 / 1e9
    print('%s %d -> %d dil %d @%dx%d x%d  %.2f GFLOP | fwd %.1f  dgrad %.1f  wgrad(+reduce) %.1f us' % (DT, Ci, Co, dil, H, W, N, gf, timeit(fwd), timeit(bwd), timeit(wg)), flush=True)

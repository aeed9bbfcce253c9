"""f32 3x3 stride-1 convolution (forward and input gradient) of the HRNet-W48 branch shapes: conv_t4.hip's split-product band
kernel against conv_t5.hip's persistent unit-pipelined kernel (rows-per-band and grid sweeps).  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
shapes = [(20, 96, 72, 48, 48), (20, 48, 36, 96, 96), (20, 24, 18, 192, 192), (20, 12, 9, 384, 384), (4, 96, 72, 48, 48),
          (4, 96, 72, 192, 48), (4, 96, 72, 96, 48)]
sweep = os.environ.get('T5_SWEEP', '1') == '1'
for (N, H, W, Ci, Co) in shapes:
    x = torch.randn(N, H, W, Ci, device=dev); y = torch.empty(N, H, W, Co, device=dev)
    dy = torch.randn(N, H, W, Co, device=dev); dx = torch.empty_like(x)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
    geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
    wp0 = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 0), device=dev); wp1 = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 1), device=dev)
    L.call('fami_pack_conv_weight_f32', p(w), p(wp0), Co, Ci, 3, 3, 0, st); L.call('fami_pack_conv_weight_f32', p(w), p(wp1), Co, Ci, 3, 3, 1, st)
    fwd = lambda: L.call('fami_conv2d_fwd_f32', p(x), p(wp0), None, None, p(y), *geo, 0, 0, st)
    bwd = lambda: L.call('fami_conv2d_dgrad_f32', p(dy), p(wp1), None, p(dx), *geo, 0, st)
    res = []
    L.cdll.fami_conv_tune_lds(-1); L.cdll.fami_conv_tune_lds(7000); res.append(('band', timeit(fwd), timeit(bwd)))
    yb = y.clone()
    L.cdll.fami_conv_tune_lds(7001); res.append(('t5', timeit(fwd), timeit(bwd)))
    same = torch.equal(yb, y)
    if sweep:
        for R in (1, 2, 3, 4, 6, 8, 12, 16):
            if R > H or (R * W + 15) // 16 > 18: continue
            L.cdll.fami_conv_tune_lds(7100 + R); res.append(('t5/R%d' % R, timeit(fwd), timeit(bwd)))
        L.cdll.fami_conv_tune_lds(7100)
        for G in (128, 192, 240, 512, 792):
            L.cdll.fami_conv_tune_lds(7500 + G // 8); res.append(('t5/G%d' % G, timeit(fwd), timeit(bwd)))
    L.cdll.fami_conv_tune_lds(-1)
    gf = 2.0 * N * H * W * Ci * Co * 9 / 1e9
    print('f32 N%-2d %3dx%-3d %3d->%-3d %.2f GFLOP bitwise=%s | ' % (N, H, W, Ci, Co, gf, same) + ' | '.join('%s %.1f/%.1f' % r for r in res), flush=True)

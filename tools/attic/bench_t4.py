"""3x3 stride-1 convolution (forward and input gradient) of the four HRNet-W48 branch shapes in bf16: direct kernel, the
round-1 LDS kernel, the register-blocked LDS kernel (conv_t4.hip) with its tiles-per-band sweep.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N = int(os.environ.get('FB_N', 20))
def timeit(fn, reps=40):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
p = lambda t: None if t is None else t.data_ptr()
DT = os.environ.get('DT', 'bf16')
tdt = {'bf16': torch.bfloat16, 'f32': torch.float32}[DT]
shapes = [(96, 72, 48, 48), (48, 36, 96, 96), (24, 18, 192, 192), (12, 9, 384, 384), (96, 72, 192, 48), (96, 72, 96, 48), (96, 72, 256, 48), (96, 72, 64, 64)]
if os.environ.get('T4_SHAPES'):
    shapes = shapes[:int(os.environ['T4_SHAPES'])]
for (H, W, Ci, Co) in shapes:
    x = torch.randn(N, H, W, Ci, device=dev).to(tdt); y = torch.empty(N, H, W, Co, device=dev, dtype=tdt)
    dy = torch.randn(N, H, W, Co, device=dev).to(tdt); dx = torch.empty_like(x)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
    geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
    if DT == 'bf16':
        wp0 = torch.empty(L.cdll.fami_packed_weight_elems_bf16(Co, Ci, 3, 3, 0), device=dev, dtype=tdt); wp1 = torch.empty(L.cdll.fami_packed_weight_elems_bf16(Co, Ci, 3, 3, 1), device=dev, dtype=tdt)
        L.call('fami_pack_conv_weight_bf16', p(w), p(wp0), Co, Ci, 3, 3, 0, st); L.call('fami_pack_conv_weight_bf16', p(w), p(wp1), Co, Ci, 3, 3, 1, st)
        fwd = lambda: L.call('fami_conv2d_fwd_bf16', p(x), p(wp0), None, p(y), *geo, 0, 0, 0, st)
        bwd = lambda: L.call('fami_conv2d_dgrad_bf16', p(dy), p(wp1), p(dx), *geo, 0, st)
    else:
        wp0 = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 0), device=dev); wp1 = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 1), device=dev)
        L.call('fami_pack_conv_weight_f32', p(w), p(wp0), Co, Ci, 3, 3, 0, st); L.call('fami_pack_conv_weight_f32', p(w), p(wp1), Co, Ci, 3, 3, 1, st)
        fwd = lambda: L.call('fami_conv2d_fwd_f32', p(x), p(wp0), None, None, p(y), *geo, 0, 0, st)
        bwd = lambda: L.call('fami_conv2d_dgrad_f32', p(dy), p(wp1), None, p(dx), *geo, 0, st)
    res = []
    L.cdll.fami_conv_tune_lds(0); res.append(('direct', timeit(fwd), timeit(bwd)))
    L.cdll.fami_conv_tune_lds(1); L.cdll.fami_conv_tune_lds(10); res.append(('lds-r1', timeit(fwd), timeit(bwd)))
    L.cdll.fami_conv_tune_lds(11)
    if DT == 'f32':   # exact-f32 MFMA instance of the register-blocked kernel, then the split-product instance (bf16 matrix pipe)
        L.cdll.fami_conv_tune_lds(30); L.cdll.fami_conv_tune_lds(21); L.cdll.fami_conv_tune_lds(112)
        res.append(('t4-exact/bt12', timeit(fwd), timeit(bwd)))
        L.cdll.fami_conv_tune_lds(31)
        L.cdll.fami_conv_tune_lds(52)
        for bt in (0, 12):
            L.cdll.fami_conv_tune_lds(100 + bt); res.append(('s3/bt%d' % bt, timeit(fwd), timeit(bwd)))
        L.cdll.fami_conv_tune_lds(100); L.cdll.fami_conv_tune_lds(61)       # producer / consumer waves, two LDS buffers, bands of <= 8 tiles
        res.append(('s3pc', timeit(fwd), timeit(bwd)))
        L.cdll.fami_conv_tune_lds(60)
        L.cdll.fami_conv_tune_lds(53)       # three pixel tiles per wave: bands of <= 24 tiles
        for bt in (0, 18):
            L.cdll.fami_conv_tune_lds(100 + bt); res.append(('s3m3/bt%d' % bt, timeit(fwd), timeit(bwd)))
        L.cdll.fami_conv_tune_lds(52)
        L.cdll.fami_conv_tune_lds(100); L.cdll.fami_conv_tune_lds(61)       # producer / consumer waves, two LDS buffers, bands of <= 8 tiles
        res.append(('s3pc', timeit(fwd), timeit(bwd)))
        L.cdll.fami_conv_tune_lds(60)
        L.cdll.fami_conv_tune_lds(53)       # three pixel tiles per wave: bands of <= 24 tiles
        for bt in (0, 18):
            L.cdll.fami_conv_tune_lds(100 + bt); res.append(('s3m3/bt%d' % bt, timeit(fwd), timeit(bwd)))
        L.cdll.fami_conv_tune_lds(52)
    else:
        for bt in (0, 16, 12, 8, 6, 4):
            L.cdll.fami_conv_tune_lds(100 + bt); res.append(('t4/bt%d' % bt, timeit(fwd), timeit(bwd)))
    L.cdll.fami_conv_tune_lds(-1)
    gf = 2.0 * N * H * W * Ci * Co * 9 / 1e9
    print(DT + ' %3dx%-3d %3d->%-3d %.2f GFLOP | ' % (H, W, Ci, Co, gf) + ' | '.join('%s %.1f/%.1f' % r for r in res), flush=True)

"""f32 weight gradient of the 3x3 stride-1 convolutions (incl. the slab reduce) on the HRNet-W48 branch shapes: exact-f32
MFMA kernels vs the split-product kernel on the bf16 matrix pipe (conv_wgs3.hip), tiles-per-run / workgroup-target sweeps;
the error of both against fp64 for the shapes small enough.  GPU box."""
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
SHAPES = [(96, 72, 48, 48), (48, 36, 96, 96), (24, 18, 192, 192), (12, 9, 384, 384), (96, 72, 192, 48), (96, 72, 96, 48), (96, 72, 64, 64),
          (96, 72, 256, 48)]
for (H, W, Ci, Co) in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(N, H, W, Ci, device=dev) + 0.3; dy = torch.randn(N, H, W, Co, device=dev) * 1e-3
    dw = torch.empty(Co, Ci, 3, 3, device=dev)
    geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
    res, outs = [], {}
    def run(tag=None):
        nb = L.cdll.fami_conv2d_wgrad_workspace(*geo)
        ws = torch.empty(nb // 4 + 4, device=dev)
        fn = lambda: L.call('fami_conv2d_wgrad_f32', p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, st)
        t = timeit(fn)
        if tag: outs[tag] = dw.clone()
        return t
    L.cdll.fami_conv_tune_wgrad_lds(30000); res.append(('exact', run('exact')))
    L.cdll.fami_conv_tune_wgrad_lds(30001)
    for bt in (0, 9, 8, 6, 4):
        L.cdll.fami_conv_tune_wgrad_lds(30100 + bt); res.append(('s3/bt%d' % bt, run('split' if bt == 0 else None)))
    L.cdll.fami_conv_tune_wgrad_lds(30100)
    for tg in (192, 384, 512):
        L.cdll.fami_conv_tune_wgrad_lds(31000 + tg); res.append(('tg%d' % tg, run()))
    L.cdll.fami_conv_tune_wgrad_lds(-1)
    err = ''
    if Ci * Co <= 96 * 96:       # fp64 reference (autograd of conv2d in double, on the GPU)
        xd = x.double().permute(0, 3, 1, 2); dyd = dy.double().permute(0, 3, 1, 2)
        ref = torch.nn.grad.conv2d_weight(xd, (Co, Ci, 3, 3), dyd, padding=1)
        m = ref.abs().max()
        err = ' | vs fp64: exact %.2e split %.2e' % tuple(((outs[k].double() - ref).abs().max() / m).item() for k in ('exact', 'split'))
    else:
        err = ' | split vs exact %.2e' % ((outs['split'] - outs['exact']).abs().max() / outs['exact'].abs().max()).item()
    print('%3dx%-3d %3d->%-3d | ' % (H, W, Ci, Co) + ' | '.join('%s %.1f' % r for r in res) + err, flush=True)

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
for (N, H, W, Ci, Co) in [(20, 96, 72, 48, 48), (20, 48, 36, 96, 96), (20, 24, 18, 192, 192), (20, 12, 9, 384, 384), (4, 96, 72, 48, 48)]:
    x = torch.randn(N, H, W, Ci, device=dev).bfloat16(); y = torch.empty(N, H, W, Co, device=dev, dtype=torch.bfloat16)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
    geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
    wp0 = torch.empty(L.cdll.fami_packed_weight_elems_bf16(Co, Ci, 3, 3, 0), device=dev, dtype=torch.bfloat16)
    L.call('fami_pack_conv_weight_bf16', p(w), p(wp0), Co, Ci, 3, 3, 0, st)
    fwd = lambda: L.call('fami_conv2d_fwd_bf16', p(x), p(wp0), None, p(y), *geo, 0, 0, 0, st)
    L.cdll.fami_conv_tune_lds(-1); t0 = timeit(fwd); y0 = y.clone()
    L.cdll.fami_conv_tune_lds(7011); L.cdll.fami_conv_tune_lds(7600); L.cdll.fami_conv_tune_lds(7401); t1 = timeit(fwd)
    print((N, H, W, Ci, Co), 'band %.1f us  t5 %.1f us  equal=%s maxdiff=%.3g' % (t0, t1, torch.equal(y0, y), (y0.float() - y.float()).abs().max().item()))
    L.cdll.fami_conv_tune_lds(-1)

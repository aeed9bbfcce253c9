"""igemm_h tile shapes for the wide 1x1 convolutions of stage 1 (bf16, 20 frames @96x72): forward 64 -> 256 and the input gradient 64 -> 256 channels
with accumulate, per forced (MT, NT, KS) and stage count.  fami_conv_tune / fami_conv_tune_stages are process-wide test shims."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
p = lambda t: None if t is None else t.data_ptr()
def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N, H, W = 20, 96, 72
tdt = torch.bfloat16
for (name, Ci, Co) in (('64->256', 64, 256), ('256->64', 256, 64)):
    x = torch.randn(N, H, W, Ci, device=dev).to(tdt); y = torch.empty(N, H, W, Co, device=dev, dtype=tdt)
    dy = torch.randn(N, H, W, Co, device=dev).to(tdt); dx = torch.zeros_like(x)
    w = torch.randn(Co, Ci, 1, 1, device=dev) * 0.05
    geo = (N, H, W, Ci, Co, 1, 1, 1, 0, 1)
    wp0 = torch.empty(L.cdll.fami_packed_weight_elems_bf16(Co, Ci, 1, 1, 0), device=dev, dtype=tdt); wp1 = torch.empty(L.cdll.fami_packed_weight_elems_bf16(Co, Ci, 1, 1, 1), device=dev, dtype=tdt)
    L.call('fami_pack_conv_weight_bf16', p(w), p(wp0), Co, Ci, 1, 1, 0, st); L.call('fami_pack_conv_weight_bf16', p(w), p(wp1), Co, Ci, 1, 1, 1, st)
    fwd = lambda: L.call('fami_conv2d_fwd_bf16', p(x), p(wp0), None, p(y), *geo, 0, 0, 0, st)
    bwd = lambda: L.call('fami_conv2d_dgrad_bf16', p(dy), p(wp1), p(dx), *geo, 0, st)
    bwa = lambda: L.call('fami_conv2d_dgrad_bf16', p(dy), p(wp1), p(dx), *geo, 1, st)
    for stages in (2, 3, 4):
        L.cdll.fami_conv_tune_stages(stages)
        for (mt, nt, ks) in ((0, 0, 0), (4, 4, 1), (2, 4, 1), (1, 4, 1), (4, 2, 1), (2, 2, 1), (4, 3, 1), (2, 3, 1), (4, 1, 1), (2, 4, 2), (1, 4, 2)):
            L.cdll.fami_conv_tune(mt, nt, ks)
            try:
                r = (timeit(fwd), timeit(bwd), timeit(bwa))
            except Exception as e:
                r = None
                lib().cdll.fami_conv_tune(0, 0, 0)
            print('%s stages %d tile (%d,%d,%d): %s' % (name, stages, mt, nt, ks, 'fwd %.1f  dgrad %.1f  dgrad+acc %.1f us' % r if r else 'no instance'), flush=True)
    L.cdll.fami_conv_tune(0, 0, 0); L.cdll.fami_conv_tune_stages(0)

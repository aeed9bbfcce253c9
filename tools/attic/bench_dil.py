"""The DCN predictors' dilated 3x3 convolutions (48 -> 216 / 108, dilation 3, B = 4 frames of 96x72; forward and input gradient):
the implicit-GEMM kernels against the band kernels of conv_t4.hip in their dilated form (fami_conv_tune_lds(40 / 41)).  GPU box."""
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
for DT in ('f32', 'bf16'):
    tdt = {'bf16': torch.bfloat16, 'f32': torch.float32}[DT]
    for (N, H, W, Ci, Co) in [(4, 96, 72, 48, 216), (4, 96, 72, 48, 108), (4, 128, 96, 48, 216)]:
        x = torch.randn(N, H, W, Ci, device=dev).to(tdt); y = torch.empty(N, H, W, Co, device=dev, dtype=tdt)
        dy = torch.randn(N, H, W, Co, device=dev).to(tdt); dx = torch.empty_like(x)
        w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
        bias = torch.randn(Co, device=dev)
        geo = (N, H, W, Ci, Co, 3, 3, 1, 3, 3)
        if DT == 'bf16':
            wp0 = torch.empty(L.cdll.fami_packed_weight_elems_bf16(Co, Ci, 3, 3, 0), device=dev, dtype=tdt); wp1 = torch.empty(L.cdll.fami_packed_weight_elems_bf16(Co, Ci, 3, 3, 1), device=dev, dtype=tdt)
            L.call('fami_pack_conv_weight_bf16', p(w), p(wp0), Co, Ci, 3, 3, 0, st); L.call('fami_pack_conv_weight_bf16', p(w), p(wp1), Co, Ci, 3, 3, 1, st)
            fwd = lambda: L.call('fami_conv2d_fwd_bf16', p(x), p(wp0), p(bias), p(y), *geo, 0, 0, 0, st)
            bwd = lambda: L.call('fami_conv2d_dgrad_bf16', p(dy), p(wp1), p(dx), *geo, 0, st)
        else:
            wp0 = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 0), device=dev); wp1 = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 1), device=dev)
            L.call('fami_pack_conv_weight_f32', p(w), p(wp0), Co, Ci, 3, 3, 0, st); L.call('fami_pack_conv_weight_f32', p(w), p(wp1), Co, Ci, 3, 3, 1, st)
            fwd = lambda: L.call('fami_conv2d_fwd_f32', p(x), p(wp0), p(bias), None, p(y), *geo, 0, 0, st)
            bwd = lambda: L.call('fami_conv2d_dgrad_f32', p(dy), p(wp1), None, p(dx), *geo, 0, st)
        L.cdll.fami_conv_tune_lds(-1); L.cdll.fami_conv_tune_lds(40); t0 = (timeit(fwd), timeit(bwd)); y0, dx0 = y.float().clone(), dx.float().clone()
        L.cdll.fami_conv_tune_lds(41); t1 = (timeit(fwd), timeit(bwd))
        ey = ((y.float() - y0).abs().max() / y0.abs().max()).item(); ex = ((dx.float() - dx0).abs().max() / dx0.abs().max()).item()
        L.cdll.fami_conv_tune_lds(-1)
        print('%s %s igemm fwd %.1f dgrad %.1f us | band fwd %.1f dgrad %.1f us | rel diff y %.2e dx %.2e' % (DT, (N, H, W, Ci, Co), t0[0], t0[1], t1[0], t1[1], ey, ex), flush=True)

"""Micro-benchmark of the DCN / shift kernels (run by hand on the GPU box): HIP-event time and the
algorithmic-bytes bandwidth of SURVEY.md 8d."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib()
dev = torch.device('cuda:0')
s = torch.cuda.current_stream(dev)
B = int(os.environ.get('B', 4))
H, W, C, G = int(os.environ.get('H', 96)), int(os.environ.get('W', 72)), 48, 12


def time_it(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


x = torch.randn(B, H, W, C, device=dev)
off = torch.randn(B, H, W, 18 * G, device=dev) * float(os.environ.get('OFFSTD', 1.0))
msk = torch.randn(B, H, W, 9 * G, device=dev)
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
bias = torch.zeros(C, device=dev)
y = torch.empty(B, H, W, C, device=dev)
wp = torch.empty(L.cdll.fami_dcn_packed_weight_elems(C, C, 3, 3, G), device=dev)
L.call('fami_dcn_pack_weight_bf16', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, G, s.cuda_stream)   # the two fp32 images + the 16-bit image fami_dcn_fwd_bf16 contracts with
P = B * H * W
fwd_bytes = (C + 27 * G + C) * P * 4.0
ys = []
for gather, pf, nm in ((0, 0, 'lds-column'), (1, 0, 'direct 1x7'), (2, 0, 'lds-window'), (0, 0, 'lds-column'), (1, 0, 'direct 1x7'), (2, 0, 'lds-window')):
    L.cdll.fami_dcn_tune(gather); L.cdll.fami_dcn_tune(16 + pf)
    us = time_it(lambda: L.call('fami_dcn_fwd_f32', x.data_ptr(), off.data_ptr(), msk.data_ptr(), wp.data_ptr(), bias.data_ptr(),
                                y.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, s.cuda_stream), reps=50)
    ys.append(y.clone())
    print('dcn fwd %-15s %8.1f us  %7.1f GB/s algorithmic (%.1f MB)' % (nm, us, fwd_bytes / us / 1e3, fwd_bytes / 1e6))
print('max |direct - lds-column| = %.3g, |window - lds-column| = %.3g (|y| max %.3g)' % ((ys[1] - ys[0]).abs().max().item(), (ys[2] - ys[0]).abs().max().item(), ys[0].abs().max().item()))
xb, ob, mb, yb = x.bfloat16(), off.bfloat16(), msk.bfloat16(), y.bfloat16()
for gather, pf, nm in ((0, 0, 'lds-column'), (1, 0, 'direct 1x7'), (2, 0, 'lds-window')):
    L.cdll.fami_dcn_tune(gather); L.cdll.fami_dcn_tune(16 + pf)
    us = time_it(lambda: L.call('fami_dcn_fwd_bf16', xb.data_ptr(), ob.data_ptr(), mb.data_ptr(), wp.data_ptr(), bias.data_ptr(),
                                yb.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, s.cuda_stream), reps=50)
    print('dcn fwd bf16 %-15s %8.1f us  %7.1f GB/s algorithmic' % (nm, us, fwd_bytes / 2 / us / 1e3))
L.cdll.fami_dcn_tune(-1); L.cdll.fami_dcn_tune(16)

dy = torch.randn(B, H, W, C, device=dev)
col = torch.empty(P, max(C * 9, L.cdll.fami_dcn_bwd_col_width(C, C, G, 3, 3, 1, 3, 4, 0)), device=dev)      # (the register-fed kernel pads its column order)
gx = torch.zeros(B, H, W, C, device=dev)
goff = torch.empty_like(off)
gmsk = torch.empty_like(msk)
wpb = torch.empty(L.cdll.fami_dcn_packed_weight_bwd_elems(C, C, 3, 3, G), device=dev)
L.call('fami_dcn_pack_weight_bwd_f32', w.data_ptr(), wpb.data_ptr(), C, C, 3, 3, G, s.cuda_stream)
bwd_bytes = (2 * C + 54 * G + C) * P * 4.0
for sc in (0, 1):
  L.cdll.fami_dcn_tune(512 + sc)
  print('scatter mode %d (0 = f32 compare-and-swap region, 1 = 64-bit fixed-point region)' % sc)
  for name, a_col, a_gx in (('bwd fused full', col, gx), ('bwd fused no-gx', col, None), ('bwd fused no-gx no-col', None, None),
                          ('bwd fused gx no-col', None, gx)):
    us = time_it(lambda: L.call('fami_dcn_bwd_f32', x.data_ptr(), off.data_ptr(), msk.data_ptr(), dy.data_ptr(), wpb.data_ptr(),
                                None if a_col is None else a_col.data_ptr(), None if a_gx is None else a_gx.data_ptr(),
                                goff.data_ptr(), gmsk.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, 0, s.cuda_stream))
    print('%-24s %8.1f us  %7.1f GB/s algorithmic' % (name, us, bwd_bytes / us / 1e3))
  gx.zero_()
  L.call('fami_dcn_bwd_f32', x.data_ptr(), off.data_ptr(), msk.data_ptr(), dy.data_ptr(), wpb.data_ptr(), col.data_ptr(), gx.data_ptr(),
         goff.data_ptr(), gmsk.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, 0, s.cuda_stream)
  ys.append(gx.clone())
print('gx: max |fixed-point - f32 region| = %.3g (|gx| max %.3g)' % ((ys[-1] - ys[-2]).abs().max().item(), ys[-1].abs().max().item()))
L.cdll.fami_dcn_tune(513)

t = torch.randn(B, 2, device=dev)
o = torch.empty_like(x)
us = time_it(lambda: L.call('fami_shift_bilinear_fwd_f32', x.data_ptr(), t.data_ptr(), o.data_ptr(), B, H, W, C, s.cuda_stream))
print('shift fwd          %8.1f us  %7.1f GB/s algorithmic' % (us, 2 * C * P * 4.0 / us / 1e3))

"""Ablation of dcn_fwd_win_kernel (run by hand on the GPU box): fami_dcn_tune(64 + bits) switches phases off
(1 no MFMA, 2 no LDS gather, 4 no offset stream, 8 no window fill); fami_dcn_tune(32 + R) forces the offset reach."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib()
dev = torch.device('cuda:0')
s = torch.cuda.current_stream(dev)
B = int(os.environ.get('B', 4))
H, W, C, G = int(os.environ.get('H', 96)), int(os.environ.get('W', 72)), 48, 12


def time_it(fn, reps=50):
    for _ in range(5): fn()
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
L.call('fami_dcn_pack_weight_f32', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, G, s.cuda_stream)
P = B * H * W
fwd_bytes = (C + 27 * G + C) * P * 4.0
run = lambda: L.call('fami_dcn_fwd_f32', x.data_ptr(), off.data_ptr(), msk.data_ptr(), wp.data_ptr(), bias.data_ptr(),
                     y.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, s.cuda_stream)
L.cdll.fami_dcn_tune(2)
for R, KS in ((4, 1), (4, 2)):
    L.cdll.fami_dcn_tune(32 + R); L.cdll.fami_dcn_tune(256 + KS)
    for abl in (64, 31, 63, 30, 1, 2, 4, 8):
        L.cdll.fami_dcn_tune(64 + abl)
        us = time_it(run)
        print('KS=%d R=%d abl=%2d (%s%s%s%s)  %7.1f us  %7.1f GB/s' % (KS, R, abl, 'M' if not abl & 1 else '-', 'G' if not abl & 2 else '-',
              'O' if not abl & 4 else '-', ('F' if not abl & 8 else '-') + ('W' if not abl & 16 else '-') + ('L' if not abl & 32 else '-'), us, fwd_bytes / us / 1e3))
L.cdll.fami_dcn_tune(64); L.cdll.fami_dcn_tune(32); L.cdll.fami_dcn_tune(-1)

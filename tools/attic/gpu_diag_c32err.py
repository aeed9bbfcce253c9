"""fp32 rounding error of the 16x16x4-tile and 32x32x2-tile 1x1 conv kernels against an fp64 reference (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream(dev).cuda_stream
torch.manual_seed(0)
for Ci, Co in ((64, 256), (256, 64), (256, 256)):
    N, H, W = 2, 32, 24
    x = torch.randn(N, H, W, Ci, device=dev); w = torch.randn(Co, Ci, 1, 1, device=dev) * 0.1
    ref = (x.double().reshape(-1, Ci) @ w.double().reshape(Co, Ci).t())
    cpu = (x.cpu().reshape(-1, Ci) @ w.cpu().reshape(Co, Ci).t()).double().to(dev)
    wp = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 1, 1, 0), device=dev)
    L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), Co, Ci, 1, 1, 0, st)
    out = []
    for mt in (16, 0):
        L.cdll.fami_conv_tune(mt, 0, 0)
        y = torch.empty(N, H, W, Co, device=dev)
        L.call('fami_conv2d_fwd_f32', x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), N, H, W, Ci, Co, 1, 1, 1, 0, 1, 0, 0, st)
        e = (y.double().reshape(-1, Co) - ref).abs()
        out.append('%s: max %.3e rms %.3e' % ('16-tile' if mt else '32-tile', e.max().item(), e.pow(2).mean().sqrt().item()))
    L.cdll.fami_conv_tune(0, 0, 0)
    e = (cpu - ref).abs()
    print('Ci=%d Co=%d  %s | cpu fp32: max %.3e rms %.3e' % (Ci, Co, ' | '.join(out), e.max().item(), e.pow(2).mean().sqrt().item()))

"""GPU box: random-shape comparison of the split-product f32 3x3 convolution kernels (forward, input gradient, weight gradient;
conv_t4.hip S3 / conv_wgs3.hip, incl. the producer / consumer form) against the exact-f32 MFMA kernels through the C ABI."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream(dev).cuda_stream
p = lambda t: None if t is None else t.data_ptr()
random.seed(int(os.environ.get('SEED', 1)))
bad = 0
n = int(os.environ.get('CASES', 150))
for it in range(n):
    N = random.randint(1, 5); H = random.randint(3, 40); W = random.randint(3, 40)
    Ci = random.choice([4, 8, 16, 20, 32, 48, 64, 80, 96, 144]); Co = random.choice([48, 64, 96, 128, 144, 192])
    acc = random.randint(0, 1); use_bias = random.randint(0, 1); pc = random.choice([60, 60, 62]); mt = random.choice([52, 53])
    torch.manual_seed(it)
    x = torch.randn(N, H, W, Ci, device=dev); dy = torch.randn(N, H, W, Co, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.1
    bias = torch.randn(Co, device=dev) if use_bias else None
    geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
    wp0 = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 0), device=dev); wp1 = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 1), device=dev)
    L.call('fami_pack_conv_weight_f32', p(w), p(wp0), Co, Ci, 3, 3, 0, st); L.call('fami_pack_conv_weight_f32', p(w), p(wp1), Co, Ci, 3, 3, 1, st)
    y0 = torch.randn(N, H, W, Co, device=dev); dx0 = torch.randn(N, H, W, Ci, device=dev); dw0 = torch.randn(Co, Ci, 3, 3, device=dev)
    out = {}
    for knob in (30, 31):
        L.cdll.fami_conv_tune_lds(-1); L.cdll.fami_conv_tune_lds(knob); L.cdll.fami_conv_tune_lds(pc); L.cdll.fami_conv_tune_lds(mt)
        L.cdll.fami_conv_tune_wgrad_lds(-1); L.cdll.fami_conv_tune_wgrad_lds(30000 + knob - 30)
        y = y0.clone(); dx = dx0.clone(); dw = dw0.clone()
        L.call('fami_conv2d_fwd_f32', p(x), p(wp0), p(bias), None, p(y), *geo, 0, acc, st)
        L.call('fami_conv2d_dgrad_f32', p(dy), p(wp1), None, p(dx), *geo, acc, st)
        nb = L.cdll.fami_conv2d_wgrad_workspace(*geo)
        ws = torch.empty(nb // 4 + 4, device=dev)
        L.call('fami_conv2d_wgrad_f32', p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, acc, st)
        torch.cuda.synchronize()
        out[knob] = (y, dx, dw)
    errs = [((out[30][k] - out[31][k]).abs().max() / (out[30][k].abs().max() + 1e-20)).item() for k in range(3)]
    if max(errs) > 1e-5 or any(e != e for e in errs):
        bad += 1
        print('MISMATCH', (N, H, W, Ci, Co), 'acc', acc, 'bias', use_bias, 'pc', pc, 'mt', mt, ['%.2e' % e for e in errs], flush=True)
L.cdll.fami_conv_tune_lds(-1); L.cdll.fami_conv_tune_wgrad_lds(-1)
print('%d cases, %d mismatches' % (n, bad))

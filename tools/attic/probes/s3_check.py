"""GPU box: split-product f32 3x3 convolution against the exact-f32 path, shape by shape (forward and input gradient)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream(dev).cuda_stream
p = lambda t: None if t is None else t.data_ptr()
shapes = [(2, 24, 18, 192, 192), (2, 16, 12, 256, 48), (2, 24, 18, 48, 48), (2, 12, 9, 96, 96), (2, 12, 9, 384, 384), (1, 5, 7, 20, 48),
          (2, 33, 21, 64, 64), (20, 96, 72, 48, 48), (2, 48, 36, 96, 96), (1, 24, 18, 16, 48), (1, 24, 18, 32, 48), (1, 24, 18, 48, 48), (1, 6, 5, 16, 48)]
for bt in (0,) if not os.environ.get('S3_MT') else (0, 20):
    for (N, H, W, Ci, Co) in shapes:
        torch.manual_seed(1)
        x = torch.randn(N, H, W, Ci, device=dev); dy = torch.randn(N, H, W, Co, device=dev)
        w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.1
        geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
        wp0 = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 0), device=dev); wp1 = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 1), device=dev)
        L.call('fami_pack_conv_weight_f32', p(w), p(wp0), Co, Ci, 3, 3, 0, st); L.call('fami_pack_conv_weight_f32', p(w), p(wp1), Co, Ci, 3, 3, 1, st)
        out = {}
        for knob in (30, 31):
            L.cdll.fami_conv_tune_lds(-1); L.cdll.fami_conv_tune_lds(knob); L.cdll.fami_conv_tune_lds(100 + bt)
            if os.environ.get('S3_MT'): L.cdll.fami_conv_tune_lds(50 + int(os.environ['S3_MT']))
            if os.environ.get('S3_PC') and knob == 31: L.cdll.fami_conv_tune_lds(61)
            if os.environ.get('S3_MT'): L.cdll.fami_conv_tune_lds(50 + int(os.environ['S3_MT']))
            if os.environ.get('S3_PC') and knob == 31: L.cdll.fami_conv_tune_lds(61)
            y = torch.zeros(N, H, W, Co, device=dev); dx = torch.zeros(N, H, W, Ci, device=dev)
            L.call('fami_conv2d_fwd_f32', p(x), p(wp0), None, None, p(y), *geo, 0, 0, st)
            L.call('fami_conv2d_dgrad_f32', p(dy), p(wp1), None, p(dx), *geo, 0, st)
            torch.cuda.synchronize()
            out[knob] = (y, dx)
        r64 = ''
        if N * H * W * Ci * Co <= 2 * 48 * 36 * 96 * 96 and bt == 0:     # fp64 reference on the host
            ref = torch.nn.functional.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu(), padding=1).permute(0, 2, 3, 1)
            m = ref.abs().max()
            r64 = ' | vs fp64: exact %.2e split %.2e' % tuple(((out[k][0].double().cpu() - ref).abs().max() / m).item() for k in (30, 31))
        ef = ((out[30][0] - out[31][0]).abs().max() / out[30][0].abs().max()).item()
        eb = ((out[30][1] - out[31][1]).abs().max() / out[30][1].abs().max()).item()
        # which pixels / channels are off
        bad = (out[30][0] - out[31][0]).abs() > 1e-3
        where = ''
        if bad.any():
            idx = bad.nonzero()
            where = ' bad n %s y %s x %s co %s' % tuple(sorted(set(idx[:, k].tolist()))[:12] for k in range(4))
        print('bt%d %s fwd %.2e dgrad %.2e%s%s' % (bt, (N, H, W, Ci, Co), ef, eb, r64, where), flush=True)
L.cdll.fami_conv_tune_lds(-1)

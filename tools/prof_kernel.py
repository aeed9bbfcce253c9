"""One kernel, a few launches: the workload rocprofv3 --pmc is pointed at (tools/pmc_run.sh).
    python tools/prof_kernel.py conv_f32|conv_bf16|dgrad_f32|wgrad_f32|wgrad_bf16|dcn_f32|dcn_bf16|dcnbwd_f32 [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib()
dev = torch.device('cuda:0')
s = torch.cuda.current_stream(dev)
what = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, H, W, C = int(os.environ.get('N', 20)), int(os.environ.get('H', 96)), int(os.environ.get('W', 72)), int(os.environ.get('C', 48))
kind, dt = what.rsplit('_', 1)
tdt = torch.bfloat16 if dt == 'bf16' else torch.float32
st = s.cuda_stream
for code in [c for c in os.environ.get('DCN_TUNE', '').split(',') if c]:      # e.g. DCN_TUNE=2 (window kernel), 2,95 (+ ablation bits)
    L.cdll.fami_dcn_tune(int(code))
for code in [c for c in os.environ.get('LDS_TUNE', '').split(',') if c]:      # e.g. LDS_TUNE=112 (12 tiles per band), 30 (no split-product f32)
    L.cdll.fami_conv_tune_lds(int(code))
if os.environ.get('XCD'):
    L.cdll.fami_conv_tune_xcd(int(os.environ['XCD']))
if kind in ('conv', 'dgrad', 'wgrad'):
    x = torch.randn(N, H, W, C, device=dev).to(tdt)
    y = torch.randn(N, H, W, C, device=dev).to(tdt)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    mode = 1 if kind == 'dgrad' else 0
    if dt == 'bf16':
        wp = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, 3, 3, mode), device=dev, dtype=tdt)
        L.call('fami_pack_conv_weight_bf16', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, mode, st)
    else:
        wp = torch.empty(L.cdll.fami_packed_weight_elems(C, C, 3, 3, mode), device=dev)
        L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, mode, st)
    if kind == 'conv' and dt == 'bf16':
        fn = lambda: L.call('fami_conv2d_fwd_bf16', x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, 0, st)
    elif kind == 'conv':
        fn = lambda: L.call('fami_conv2d_fwd_f32', x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, st)
    elif kind == 'dgrad' and dt == 'bf16':
        fn = lambda: L.call('fami_conv2d_dgrad_bf16', x.data_ptr(), wp.data_ptr(), y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, st)
    elif kind == 'dgrad':
        fn = lambda: L.call('fami_conv2d_dgrad_f32', x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, st)
    else:
        dw = torch.empty(C, C, 3, 3, device=dev)
        nb = L.cdll.fami_conv2d_wgrad_workspace(N, H, W, C, C, 3, 3, 1, 1, 1)
        ws = torch.empty(nb // 4, device=dev)
        fn = lambda: L.call('fami_conv2d_wgrad_' + dt, x.data_ptr(), y.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, H, W, C, C, 3, 3, 1, 1, 1, 0, st)
else:
    B, G = int(os.environ.get('B', 4)), 12
    x = torch.randn(B, H, W, C, device=dev).to(tdt)
    off = torch.randn(B, H, W, 18 * G, device=dev).to(tdt)
    msk = torch.randn(B, H, W, 9 * G, device=dev).to(tdt)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    bias = torch.zeros(C, device=dev)
    y = torch.randn(B, H, W, C, device=dev).to(tdt)
    if kind == 'dcn':
        wp = torch.empty(L.cdll.fami_dcn_packed_weight_elems(C, C, 3, 3, G), device=dev)
        L.call('fami_dcn_pack_weight_' + dt, w.data_ptr(), wp.data_ptr(), C, C, 3, 3, G, st)
        fn = lambda: L.call('fami_dcn_fwd_' + dt, x.data_ptr(), off.data_ptr(), msk.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, st)
    else:
        wpb = torch.empty(L.cdll.fami_dcn_packed_weight_bwd_elems(C, C, 3, 3, G), device=dev)
        L.call('fami_dcn_pack_weight_bwd_f32', w.data_ptr(), wpb.data_ptr(), C, C, 3, 3, G, st)
        col = torch.empty(B * H * W, max(C * 9, L.cdll.fami_dcn_bwd_col_width(C, C, G, 3, 3, 1, 3, x.element_size(), 0)), device=dev, dtype=tdt)
        gx = torch.zeros(B, H, W, C, device=dev)
        goff, gmsk = torch.empty_like(off), torch.empty_like(msk)
        fn = lambda: L.call('fami_dcn_bwd_' + dt, x.data_ptr(), off.data_ptr(), msk.data_ptr(), y.data_ptr(), wpb.data_ptr(), col.data_ptr(), gx.data_ptr(), goff.data_ptr(), gmsk.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, 0, st)
for _ in range(reps):
    fn()
torch.cuda.synchronize()

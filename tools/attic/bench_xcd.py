"""Natural vs XCD-contiguous workgroup->tile order of the implicit-GEMM conv kernels (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
N = 20
KNOB = os.environ.get("KNOB", "xcd")   # which fami_conv_tune_<knob>(0|1) to compare
VALS = [int(v) for v in os.environ.get("VALS", "0,1").split(",")]   # the two knob values (e.g. KNOB=stages VALS=100,101)
RESET = int(os.environ.get("RESET", "-1"))
for dt in ('f32', 'bf16'):
    tdt = torch.bfloat16 if dt == 'bf16' else torch.float32
    for H, W, C, k in ((96, 72, 48, 3), (48, 36, 96, 3), (24, 18, 192, 3), (12, 9, 384, 3), (96, 72, 64, 1), (96, 72, 256, 1)):
        w = torch.randn(C, C, k, k, device=dev) * 0.05
        x = torch.randn(N, H, W, C, device=dev).to(tdt); y = torch.empty_like(x)
        out = []
        for mode in (0, 1):
            if dt == 'bf16':
                wp = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, k, k, mode), device=dev, dtype=tdt)
                L.call('fami_pack_conv_weight_bf16', w.data_ptr(), wp.data_ptr(), C, C, k, k, mode, st)
                fn = (lambda: L.call('fami_conv2d_fwd_bf16', x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, k, k, 1, k // 2, 1, 0, 0, 0, st)) if mode == 0 else \
                     (lambda: L.call('fami_conv2d_dgrad_bf16', x.data_ptr(), wp.data_ptr(), y.data_ptr(), N, H, W, C, C, k, k, 1, k // 2, 1, 0, st))
            else:
                wp = torch.empty(L.cdll.fami_packed_weight_elems(C, C, k, k, mode), device=dev)
                L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), C, C, k, k, mode, st)
                fn = (lambda: L.call('fami_conv2d_fwd_f32', x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, k, k, 1, k // 2, 1, 0, 0, st)) if mode == 0 else \
                     (lambda: L.call('fami_conv2d_dgrad_f32', x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, k, k, 1, k // 2, 1, 0, st))
            for xcd in VALS:
                getattr(L.cdll, "fami_conv_tune_" + KNOB)(xcd)
                for _ in range(3): fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(30): fn()
                e1.record(s); e1.synchronize()
                out.append(('%s ' + KNOB + '%d %.1f') % ('fwd' if mode == 0 else 'dgr', xcd, e0.elapsed_time(e1) / 30 * 1e3))
        getattr(L.cdll, "fami_conv_tune_" + KNOB)(RESET)
        print('%s %dx%d C=%d k=%d: %s' % (dt, H, W, C, k, '  '.join(out)), flush=True)

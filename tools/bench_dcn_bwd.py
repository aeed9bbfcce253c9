"""DCN backward kernel, ablations (GPU box): fami_dcn_tune(1024 + bits), bits: 1 = no region flush, 2 = no LDS adds,
4 = constant fixed-point scale (no maxima pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev)
B, H, W, C, G = int(os.environ.get('B', 4)), 96, 72, 48, 12
def time_it(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for code in [c for c in os.environ.get('DCN_TUNE', '').split(',') if c]:      # e.g. DCN_TUNE=4128 (32 KB region budget), applied before the pack
    L.cdll.fami_dcn_tune(int(code))
for dt, tdt in (('f32', torch.float32), ('bf16', torch.bfloat16)):
    x = torch.randn(B, H, W, C, device=dev).to(tdt)
    off = (torch.randn(B, H, W, 18 * G, device=dev) * float(os.environ.get('OFFSTD', 1.0))).to(tdt)
    msk = torch.randn(B, H, W, 9 * G, device=dev).to(tdt)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    dy = torch.randn(B, H, W, C, device=dev).to(tdt)
    col = torch.empty(B * H * W, max(C * 9, L.cdll.fami_dcn_bwd_col_width(C, C, G, 3, 3, 1, 3, x.element_size(), 0)), device=dev).to(tdt)
    gx = torch.zeros(B, H, W, C, device=dev)
    goff = torch.empty_like(off); gmsk = torch.empty_like(msk)
    wpb = torch.empty(L.cdll.fami_dcn_packed_weight_bwd_elems(C, C, 3, 3, G), device=dev)
    L.call('fami_dcn_pack_weight_bwd_f32', w.data_ptr(), wpb.data_ptr(), C, C, 3, 3, G, s.cuda_stream)
    out = []
    for regfed in (1, 0):          # the register-fed kernel (round 5 default) against the general one, same launch
        L.cdll.fami_dcn_tune(2048 + regfed)
        us = time_it(lambda: L.call('fami_dcn_bwd_' + dt, x.data_ptr(), off.data_ptr(), msk.data_ptr(), dy.data_ptr(), wpb.data_ptr(),
                                    col.data_ptr(), gx.data_ptr(), goff.data_ptr(), gmsk.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, 0, s.cuda_stream))
        out.append('%s %.1f' % ('regfed' if regfed else 'general', us))
        if regfed:
            us = time_it(lambda: L.call('fami_dcn_bwd_' + dt, x.data_ptr(), off.data_ptr(), msk.data_ptr(), dy.data_ptr(), wpb.data_ptr(),
                                        col.data_ptr(), None, goff.data_ptr(), gmsk.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, 0, s.cuda_stream))
            out.append('regfed-no-gx %.1f' % us)
            us = time_it(lambda: L.call('fami_dcn_bwd_' + dt, x.data_ptr(), off.data_ptr(), msk.data_ptr(), dy.data_ptr(), wpb.data_ptr(),
                                        None, gx.data_ptr(), goff.data_ptr(), gmsk.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, 0, s.cuda_stream))
            out.append('regfed-no-col %.1f' % us)
    L.cdll.fami_dcn_tune(2049)
    L.cdll.fami_dcn_tune(8192)          # WITHOUT the LDS staging of the offset / mask gradients (default: staged in LDS, written with consecutive lanes on consecutive elements
    us = time_it(lambda: L.call('fami_dcn_bwd_' + dt, x.data_ptr(), off.data_ptr(), msk.data_ptr(), dy.data_ptr(), wpb.data_ptr(),
                                col.data_ptr(), gx.data_ptr(), goff.data_ptr(), gmsk.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, 0, s.cuda_stream))
    out.append('regfed-unstaged %.1f' % us)
    L.cdll.fami_dcn_tune(8193)
    for abl in (0, 1, 2, 3, 4, 7):
        L.cdll.fami_dcn_tune(1024 + abl)
        us = time_it(lambda: L.call('fami_dcn_bwd_' + dt, x.data_ptr(), off.data_ptr(), msk.data_ptr(), dy.data_ptr(), wpb.data_ptr(),
                                    col.data_ptr(), gx.data_ptr(), goff.data_ptr(), gmsk.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, 0, s.cuda_stream))
        out.append('abl%d %.1f' % (abl, us))
    L.cdll.fami_dcn_tune(1024)
    us = time_it(lambda: L.call('fami_dcn_bwd_' + dt, x.data_ptr(), off.data_ptr(), msk.data_ptr(), dy.data_ptr(), wpb.data_ptr(),
                                col.data_ptr(), None, goff.data_ptr(), gmsk.data_ptr(), B, H, W, C, C, G, 3, 3, 1, 3, 3, 0, s.cuda_stream))
    out.append('no-gx %.1f' % us)
    print(dt, ' '.join(out), flush=True)

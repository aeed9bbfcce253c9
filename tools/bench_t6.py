"""conv_t6.hip (weight-resident, DMA-staged 16-bit 3x3 kernel) against conv_t4.hip on the 48-channel shapes: results against
an fp64 torch convolution of the same 16-bit inputs, then time per launch (forward / input gradient), rows-per-band sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
p = lambda t: None if t is None else t.data_ptr()
def timeit(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for DT in ('bf16', 'f16'):
    tdt = {'bf16': torch.bfloat16, 'f16': torch.float16}[DT]
    SH48 = ((20, 96, 72, 48, 48), (20, 48, 36, 96, 96), (20, 24, 18, 192, 192), (20, 12, 9, 384, 384), (4, 96, 72, 48, 48), (20, 64, 64, 48, 96), (2, 48, 36, 96, 48))
    SH64 = ((20, 96, 72, 64, 64), (20, 48, 36, 128, 128), (20, 24, 18, 256, 256), (20, 12, 9, 512, 512), (4, 96, 72, 64, 64))      # T6_SHAPES=w64: HRNet-W64's branches, stage 1's 64 -> 64
    for (N, H, W, Ci, Co) in (SH64 if os.environ.get('T6_SHAPES') == 'w64' else SH48):
        torch.manual_seed(1)
        x = torch.randn(N, H, W, Ci, device=dev).to(tdt); y = torch.empty(N, H, W, Co, device=dev, dtype=tdt)
        dy = torch.randn(N, H, W, Co, device=dev).to(tdt); dx = torch.empty_like(x)
        w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05)
        bias = torch.randn(Co, device=dev) * 0.1
        geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
        wp0 = torch.empty(getattr(L.cdll, 'fami_packed_weight_elems_' + DT)(Co, Ci, 3, 3, 0), device=dev, dtype=tdt)
        wp1 = torch.empty(getattr(L.cdll, 'fami_packed_weight_elems_' + DT)(Co, Ci, 3, 3, 1), device=dev, dtype=tdt)
        L.call('fami_pack_conv_weight_' + DT, p(w), p(wp0), Co, Ci, 3, 3, 0, st); L.call('fami_pack_conv_weight_' + DT, p(w), p(wp1), Co, Ci, 3, 3, 1, st)
        fwd = lambda b=None: L.call('fami_conv2d_fwd_' + DT, p(x), p(wp0), p(b), p(y), *geo, 0, 0, 0, st)
        bwd = lambda acc=0: L.call('fami_conv2d_dgrad_' + DT, p(dy), p(wp1), p(dx), *geo, acc, st)
        wq = w.to(tdt).double()
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), wq, bias.double(), padding=1).permute(0, 2, 3, 1)
        refd = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), wq, padding=1).permute(0, 2, 3, 1)
        out = {}
        for name, knob in (('t4', 8000), ('t6', 8001)):
            L.cdll.fami_conv_tune_lds(-1); L.cdll.fami_conv_tune_lds(knob); L.cdll.fami_conv_tune_lds(knob + 500)
            fwd(bias); bwd()
            torch.cuda.synchronize()
            ef = ((y.double() - ref).abs().max() / ref.abs().max()).item()
            eb = ((dx.double() - refd).abs().max() / refd.abs().max()).item()
            out[name] = (y.clone(), dx.clone())
            print('%s %s N%d %dx%d %d->%d: fwd err %.2e dgrad err %.2e (of max; storage rounding 4e-3 bf16 / 5e-4 f16)' % (DT, name, N, H, W, Ci, Co, ef, eb), flush=True)
        print('   t6 vs t4 max |diff| fwd %.3e dgrad %.3e' % ((out['t4'][0].float() - out['t6'][0].float()).abs().max().item(),
                                                           (out['t4'][1].float() - out['t6'][1].float()).abs().max().item()))
        res = []
        L.cdll.fami_conv_tune_lds(8000); L.cdll.fami_conv_tune_lds(8500); res.append(('t4', timeit(fwd), timeit(bwd)))
        L.cdll.fami_conv_tune_lds(-1); res.append(('t6/t7', timeit(fwd), timeit(bwd)))
        L.cdll.fami_conv_tune_lds(8201); res.append(('t6/MT1', timeit(fwd), timeit(bwd))); L.cdll.fami_conv_tune_lds(8200)
        for rb in (2, 4, 6, 8, 12):
            if rb <= H and H % rb == 0:
                L.cdll.fami_conv_tune_lds((8100 if Ci == 48 else 8600) + rb); res.append(('RB%d' % rb, timeit(fwd), timeit(bwd)))
        if Ci != 48:
            L.cdll.fami_conv_tune_lds(-1)
            for tg in (60, 80, 160, 240):
                L.cdll.fami_conv_tune_lds(8700 + tg); res.append(('tg%d' % tg, timeit(fwd), timeit(bwd)))
        L.cdll.fami_conv_tune_lds(-1)
        gf = 2.0 * N * H * W * Ci * Co * 9 / 1e9
        print('   %.2f GFLOP | ' % gf + ' | '.join('%s %.1f/%.1f' % r for r in res), flush=True)

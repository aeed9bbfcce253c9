"""The BatchNorm backward's statistics pass as the producing input-gradient convolution's epilogue (EpiBN mode 2) against the
two-launch BatchNorm backward behind a plain input gradient, per branch shape: time per (conv, BatchNorm backward) pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
DT = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
tdt = {'bf16': torch.bfloat16, 'f16': torch.float16}[DT]
for (N, H, W, C) in ((24, 96, 72, 48), (24, 48, 36, 96), (24, 24, 18, 192), (24, 12, 9, 384), (24, 96, 72, 64)):
    torch.manual_seed(1)
    P = N * H * W
    z = torch.randn(N, H, W, C, device=dev).to(tdt); dy = torch.randn(N, H, W, C, device=dev).to(tdt)
    dx = torch.empty_like(z); dz = torch.empty_like(z)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    wp1 = torch.empty(getattr(L.cdll, 'fami_packed_weight_elems_' + DT)(C, C, 3, 3, 1), device=dev, dtype=tdt)
    L.call('fami_pack_conv_weight_' + DT, p(w), p(wp1), C, C, 3, 3, 1, st)
    mean = z.float().mean((0, 1, 2)); invstd = (z.float().var((0, 1, 2), unbiased=False) + 1e-5).rsqrt()
    gamma = torch.rand(C, device=dev) + 0.5; beta = torch.randn(C, device=dev) * 0.1
    gg = torch.empty(C, device=dev); gb = torch.empty(C, device=dev)
    nb = L.cdll.fami_bn_slots_bytes(C)
    slots = torch.zeros(nb, device=dev, dtype=torch.uint8)
    geo = (N, H, W, C, C, 3, 3, 1, 1, 1)
    def plain():
        L.call('fami_conv2d_dgrad_' + DT, p(dy), p(wp1), p(dz), *geo, 0, st)
    def bn2():
        slots.zero_()
        L.call('fami_bn_bwd2_' + DT, p(dz), p(z), None, p(mean), p(invstd), p(gamma), p(beta), p(dx), p(gg), p(gb), None, P, C, 2, 0, 0, 0, p(slots), st)
    def fused():
        slots.zero_()
        L.call('fami_conv2d_dgrad_bnstats_' + DT, p(dy), p(wp1), p(dz), *geo, 0, p(z), None, p(mean), p(invstd), p(gamma), p(beta), 2, p(slots), st)
    def apply():
        L.call('fami_bn_bwd_apply_slots_' + DT, p(dz), p(z), p(mean), p(invstd), p(gamma), p(beta), p(dx), p(gg), p(gb), None, P, C, 0, 0, 0, p(slots), st)
    zero = timeit(lambda: slots.zero_())
    tp, tb, tf, ta = timeit(plain), timeit(bn2) - zero, timeit(fused) - zero, timeit(apply)
    print('%s N%d %dx%d C%d: dgrad %.1f + bn_bwd2 %.1f = %.1f us | dgrad+stats %.1f + apply %.1f = %.1f us' % (DT, N, H, W, C, tp, tb, tp + tb, tf, ta, tf + ta), flush=True)

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
BF = torch.bfloat16
for (N, H, W) in ((20, 96, 72), (4, 96, 72)):
    Ci = Co = 48; P = N * H * W
    z = torch.randn(N, H, W, Ci, device=dev).to(BF)
    w2 = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
    wp2 = torch.empty(L.cdll.fami_packed_weight_elems_bf16(Co, Ci, 3, 3, 0), device=dev, dtype=BF)
    L.call('fami_pack_conv_weight_bf16', p(w2), p(wp2), Co, Ci, 3, 3, 0, st)
    gamma, beta = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
    rm, rv = torch.zeros(Ci, device=dev), torch.ones(Ci, device=dev)
    xs = torch.zeros(L.cdll.fami_bn_slots_bytes(Ci) // 8, device=dev, dtype=torch.float64)
    ws = torch.empty(L.cdll.fami_bn_workspace(Ci) // 4 + 16, device=dev)
    a = torch.empty_like(z); y = torch.empty(N, H, W, Co, device=dev, dtype=BF)
    mean, inv = torch.empty(Ci, device=dev), torch.empty(Ci, device=dev)
    ys = torch.zeros(L.cdll.fami_bn_slots_bytes(Co) // 8, device=dev, dtype=torch.float64)
    piv = torch.zeros(Co, device=dev)
    geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
    # fill xs once with plausible stats
    L.call('fami_conv2d_fwd_stats_bf16', p(z), p(wp2), None, p(y), *geo, p(xs), p(piv), st)
    t_apply = timeit(lambda: L.call('fami_bn_apply_slots_bf16', p(z), None, p(a), p(gamma), p(beta), p(mean), p(inv), p(rm), p(rv), P, Ci, 1, 0.1, 1e-5, p(xs), st))
    t_conv = timeit(lambda: L.call('fami_conv2d_fwd_stats_bf16', p(a), p(wp2), None, p(y), *geo, p(ys), p(piv), st))
    t_both = timeit(lambda: (L.call('fami_bn_apply_slots_bf16', p(z), None, p(a), p(gamma), p(beta), p(mean), p(inv), p(rm), p(rv), P, Ci, 1, 0.1, 1e-5, p(xs), st),
                             L.call('fami_conv2d_fwd_stats_bf16', p(a), p(wp2), None, p(y), *geo, p(ys), p(piv), st)))
    t_in = timeit(lambda: L.call('fami_conv2d_fwd_bnin_bf16', p(z), p(wp2), None, p(y), p(a), N, H, W, Ci, Co, p(ys), p(piv), p(xs), P, p(gamma), p(beta), p(mean), p(inv), p(rm), p(rv), 0.1, 1e-5, st))
    print('N=%d: apply %.1f us, conv+stats %.1f us, both back to back %.1f us | BN inside the conv launch %.1f us' % (N, t_apply, t_conv, t_both, t_in))

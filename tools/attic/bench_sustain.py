"""Does a kernel's per-launch time depend on how long the chip has been under load (clock / power management)?  The
dominant f32 convolution, 40 launches against 4000 and 20000 back-to-back launches (HIP events around the whole run)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
p = lambda t: None if t is None else t.data_ptr()
def timeit(fn, reps):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N, H, W, Ci, Co = 20, 96, 72, 48, 48
nbuf = int(os.environ.get('NBUF', 1))
xs = [torch.randn(N, H, W, Ci, device=dev) for _ in range(nbuf)]; ys = [torch.empty(N, H, W, Co, device=dev) for _ in range(nbuf)]
w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
wp0 = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 0), device=dev)
L.call('fami_pack_conv_weight_f32', p(w), p(wp0), Co, Ci, 3, 3, 0, st)
k = [0]
def fwd():
    i = k[0] % nbuf; k[0] += 1
    L.call('fami_conv2d_fwd_f32', p(xs[i]), p(wp0), None, None, p(ys[i]), *geo, 0, 0, st)
for code, name in ((7000, 'band'), (7001, 't5')):
    L.cdll.fami_conv_tune_lds(-1); L.cdll.fami_conv_tune_lds(code)
    for reps in (40, 4000, 20000, 40):
        torch.cuda.synchronize(); time.sleep(0.5 if reps == 40 else 0.0)
        print('%s nbuf=%d reps=%6d  %.1f us/launch' % (name, nbuf, reps, timeit(fwd, reps)), flush=True)

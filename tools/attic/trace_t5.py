"""s_memtime trace of one workgroup of conv_t5.hip (library built with -DFAMI_T5_TRACE): per unit and wave the cycles between
the stamps 0 unit start, 1 past the barrier, 2 weight DMA issued, 3 patch stored / next fetch issued, 4 taps done."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
p = lambda t: None if t is None else t.data_ptr()
N, H, W, Ci, Co = [int(v) for v in os.environ.get('SHAPE', '20,96,72,48,48').split(',')]
x = torch.randn(N, H, W, Ci, device=dev); y = torch.empty(N, H, W, Co, device=dev)
w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
wp0 = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 0), device=dev)
L.call('fami_pack_conv_weight_f32', p(w), p(wp0), Co, Ci, 3, 3, 0, st)
dbg = torch.zeros(8 * 40 * 8, dtype=torch.int64, device=dev)
import ctypes
fn = L.cdll.fami_conv_t5_debug; fn.argtypes = [ctypes.c_void_p]; fn.restype = None
for it in range(3):
    fn(dbg.data_ptr() if it == 2 else None)
    L.call('fami_conv2d_fwd_f32', p(x), p(wp0), None, None, p(y), N, H, W, Ci, Co, 3, 3, 1, 1, 1, 0, 0, st)
torch.cuda.synchronize()
d = dbg.cpu().view(8, 40, 8).numpy()
t0 = d[:, 0, 0].min()
for wv in (0, 2, 5):
    print('wave %d: unit: start(rel) | barrier wait | dma issue | store+fetch | taps' % wv)
    for u in range(0, 20):
        r = d[wv, u]
        if r[4] == 0: break
        print('  u%-2d %8d | %6d | %5d | %6d | %6d   (unit total %d)' % (u, r[0] - t0, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], (d[wv, u + 1, 0] - r[0]) if d[wv, u + 1, 0] else 0))

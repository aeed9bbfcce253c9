"""s_memtime stamps of one workgroup of conv_wg16.hip (library built with -DFAMI_WG16_TRACE).  Stamps: start, after first fetch
issue, after first stash, after barrier; then per run: after next fetch issue, after K-loop half 1, after stash, after K-loop half 2
(before barrier), after barrier; last: after slab store."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream(dev); st = s.cuda_stream
p = lambda t: None if t is None else t.data_ptr()
N, H, W, Ci, Co = 20, 96, 72, 48, 48
x = torch.randn(N, H, W, Ci, device=dev).bfloat16(); dy = torch.randn(N, H, W, Co, device=dev).bfloat16()
dw = torch.empty(Co, Ci, 3, 3, device=dev)
geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
nb = L.cdll.fami_conv2d_wgrad_workspace(*geo); ws = torch.empty(nb // 4 + 4, device=dev)
dbg = torch.zeros(8 * 32, dtype=torch.int64, device=dev)
fn = L.cdll.fami_wgrad16_debug; fn.argtypes = [ctypes.c_void_p]; fn.restype = None
for it in range(3):
    fn(dbg.data_ptr() if it == 2 else None)
    L.call('fami_conv2d_wgrad_bf16', p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, st)
torch.cuda.synchronize()
d = dbg.cpu().view(8, 32).numpy()
for wv in (0, 3, 7):
    r = [int(v) for v in d[wv] if v]
    print('wave %d: ' % wv + ' '.join('%d' % (b - a) for a, b in zip(r[:-1], r[1:])) + '  total %d' % (r[-1] - r[0]))

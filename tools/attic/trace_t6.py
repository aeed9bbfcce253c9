"""s_memtime trace of one conv_t6 workgroup (library built with -DFAMI_T6_TRACE): stamps per wave in cycles from the first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream(dev).cuda_stream
N, H, W, C = 20, 96, 72, 48
x = torch.randn(N, H, W, C, device=dev).bfloat16(); y = torch.empty_like(x)
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
wp = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, 3, 3, 0), device=dev, dtype=torch.bfloat16)
L.call('fami_pack_conv_weight_bf16', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, st)
dbg = torch.zeros(8 * 64, device=dev, dtype=torch.int64)
import ctypes
L.cdll.fami_conv_t6_debug(ctypes.c_void_p(dbg.data_ptr()))
for _ in range(5):
    L.call('fami_conv2d_fwd_bf16', x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, 0, st)
torch.cuda.synchronize()
d = dbg.cpu().view(8, 64)
t0 = int(d[:, 0].min())
names = ['start', 'w-dma', 'x-dma', 'consts'] + sum([['u%d top' % u, 'u%d barrier' % u, 'u%d emitted' % u, 'u%d multiplied' % u] for u in range(4)], [])
for wv in range(8):
    row = [int(v) - t0 for v in d[wv] if int(v) != 0]
    nm = names[:len(row) - 1] + ['last emit']
    print('wave %d: ' % wv + ' '.join('%s=%d' % (n, v) for n, v in zip(nm, row)))

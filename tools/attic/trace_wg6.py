"""s_memtime trace of one conv_wgrad6 workgroup (library built with -DFAMI_WG6_TRACE)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream(dev).cuda_stream
N, H, W, C = 20, 96, 72, 48
x = torch.randn(N, H, W, C, device=dev).bfloat16(); dy = torch.randn(N, H, W, C, device=dev).bfloat16()
dw = torch.empty(C, C, 3, 3, device=dev)
geo = (N, H, W, C, C, 3, 3, 1, 1, 1)
nb = L.cdll.fami_conv2d_wgrad_workspace(*geo); ws = torch.empty(nb // 4 + 4, device=dev)
dbg = torch.zeros(8 * 64, device=dev, dtype=torch.int64)
L.cdll.fami_wgrad6_debug(ctypes.c_void_p(dbg.data_ptr()))
for _ in range(5):
    L.call('fami_conv2d_wgrad_bf16', x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel() * 4, *geo, 0, st)
torch.cuda.synchronize()
d = dbg.cpu().view(8, 64); t0 = int(d[:, 0].min())
for wv in range(8):
    row = [int(v) - t0 for v in d[wv] if int(v) != 0]
    names = ['start', 'dma0'] + sum([['u%d top' % u, 'u%d open' % u, 'u%d dma' % u] for u in range((len(row) - 4) // 3)], []) + ['k done', 'stored']
    print('wave %d: ' % wv + ' '.join('%s=%d' % (n, v) for n, v in zip(names, row)))

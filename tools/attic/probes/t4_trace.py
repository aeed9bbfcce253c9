"""GPU box, FAMI_T4_TRACE build of conv_t4.hip: phase timestamps (s_memtime ticks) of one workgroup of the f32 3x3
convolution CI->48 @96x72 N=20."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream(dev).cuda_stream
p = lambda t: None if t is None else t.data_ptr()
N, H, W, Ci, Co = 20, 96, 72, int(os.environ.get('CI', 48)), 48
x = torch.randn(N, H, W, Ci, device=dev); y = torch.empty(N, H, W, Co, device=dev)
w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.1
wp0 = torch.empty(L.cdll.fami_packed_weight_elems(Co, Ci, 3, 3, 0), device=dev)
L.call('fami_pack_conv_weight_f32', p(w), p(wp0), Co, Ci, 3, 3, 0, st)
for code in [c for c in os.environ.get('LDS_TUNE', '112').split(',') if c]:
    L.cdll.fami_conv_tune_lds(int(code))
dbg = torch.zeros(8 * 32 * 8, dtype=torch.int64, device=dev)
L.cdll.fami_conv_t4_debug.argtypes = [ctypes.c_void_p]
for it in range(3):
    dbg.zero_()
    L.cdll.fami_conv_t4_debug(dbg.data_ptr())
    L.call('fami_conv2d_fwd_f32', p(x), p(wp0), None, None, p(y), N, H, W, Ci, Co, 3, 3, 1, 1, 1, 0, 0, st)
    torch.cuda.synchronize()
L.cdll.fami_conv_t4_debug(None)
d = dbg.cpu().view(8, 32, 8)
t0 = d[:, 0, 0].min().item()
nchunk = (Ci + 15) // 16
for wv in range(8):
    row = []
    for c in range(nchunk):
        s = d[wv, c]
        row.append('c%d@%d: bar %d st %d bar %d taps %d' % (c, s[0] - t0, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3]))
    print('wave', wv, ' | '.join(row))
print('workgroup total (ticks):', (d[:, nchunk - 1, 4].max() - t0).item())

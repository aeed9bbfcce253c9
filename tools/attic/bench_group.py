"""Do four stream lanes pack the chip as well as one fat launch would?  The 48-channel 3x3 convolution @96x72:
(a) 4R launches of N = 20 frames on one stream, (b) R launches on each of four streams (one hipGraph, fork / join),
(c) R launches of N = 80 frames on one stream (the same work as one launch each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fami_pose_amd._lib import lib
L = lib(); dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
R = 16
def make(dt, N, H, W, C):
    tdt = torch.bfloat16 if dt == 'bf16' else torch.float32
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    x = torch.randn(N, H, W, C, device=dev).to(tdt); y = torch.empty_like(x)
    if dt == 'bf16':
        wp = torch.empty(L.cdll.fami_packed_weight_elems_bf16(C, C, 3, 3, 0), device=dev, dtype=tdt)
        L.call('fami_pack_conv_weight_bf16', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, torch.cuda.current_stream().cuda_stream)
        return (x, y, wp), lambda st: L.call('fami_conv2d_fwd_bf16', x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, 0, st)
    wp = torch.empty(L.cdll.fami_packed_weight_elems(C, C, 3, 3, 0), device=dev)
    L.call('fami_pack_conv_weight_f32', w.data_ptr(), wp.data_ptr(), C, C, 3, 3, 0, torch.cuda.current_stream().cuda_stream)
    return (x, y, wp), lambda st: L.call('fami_conv2d_fwd_f32', x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), N, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, st)
def timed(g, reps=20):
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def capture(body):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        body(s, warm=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        body(s, warm=False)
    return g
side = [torch.cuda.Stream(dev) for _ in range(3)]
for dt in ('bf16', 'f32'):
    for shapes in ([(96, 72, 48)] * 4, [(96, 72, 48), (48, 36, 96), (24, 18, 192), (12, 9, 384)]):
        keep = []
        fns = []
        for (H, W, C) in shapes:
            k, f = make(dt, 20, H, W, C); keep.append(k); fns.append(f)
        kb, fbig = make(dt, 80, *shapes[0]); keep.append(kb)
        torch.cuda.synchronize()
        def serial(s, warm):
            for r in range(R):
                for f in fns: f(s.cuda_stream)
        def lanes(s, warm):
            cur = torch.cuda.current_stream()
            for q in side: q.wait_stream(cur)
            for r in range(R):
                for i, f in enumerate(fns):
                    f((cur if i == 0 else side[i - 1]).cuda_stream)
            for q in side: cur.wait_stream(q)
        def big(s, warm):
            for r in range(R): fbig(s.cuda_stream)
        ga, gb, gc = capture(serial), capture(lanes), capture(big)
        a, b, c = timed(ga) / R, timed(gb) / R, timed(gc) / R
        print('%s %s: serial 4 launches %.1f us | 4 lanes %.1f us | one N=80 launch of the first shape %.1f us' % (dt, [s[2] for s in shapes], a, b, c))

"""Un-profiled wall time of the step's phases under hipGraph replay (rocprofv3's kernel trace serialises the stream lanes
and inflates gaps between tiny kernels, so the latency-bound sections cannot be read from it): forward only, forward +
loss + backward (no optimizer), the full step; HRNet alone (forward, forward + backward).  usage: phase_times.py f32|bf16"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fami_pose_amd.engine import Engine, T, _p
from fami_pose_amd.train import Trainer
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dtype = sys.argv[1]
args = types.SimpleNamespace(width=48, img_w=288, img_h=384, sup=4, freeze_backbone=False, dtype=dtype, deterministic=False)
kf, sup, joints, vis = bench.synth_batch(4, 4, 384, 288, 17, dev, 19970808)
model = bench.build(args, dev)
tr = Trainer(model, lr=1e-3, use_mi=True, use_graph=True, targets_from_joints=True)
for _ in range(3): tr.step(kf, sup, joints, vis)

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

def graph_of(fn):
    side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2): fn()
    torch.cuda.current_stream(dev).wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    return g

res = {'full step': timeit(lambda: tr.step(kf, sup, joints, vis))}
adt = model.act_dtype
def fwd_only(record):
    eng = Engine(dev, grad_views=tr.views, record=record, dtype=adt)
    tr.packer.run(eng.stream); eng.prepacked = tr.packer.views
    return eng, model._body(eng, kf, sup)
g1 = graph_of(lambda: fwd_only(False)); res['forward (incl. weight pack)'] = timeit(g1.replay)
def hr(record, back):
    eng = Engine(dev, grad_views=tr.views, record=record, dtype=adt)
    tr.packer.run(eng.stream); eng.prepacked = tr.packer.views
    hm, feats, _ = model.hrnet.run(eng, eng.frames(kf, sup))
    if back:
        feats[0].grad = torch.ones_like(feats[0].data); eng.backward()
g2 = graph_of(lambda: hr(False, False)); res['HRNet forward'] = timeit(g2.replay)
g3 = graph_of(lambda: hr(True, True)); res['HRNet forward + backward'] = timeit(g3.replay)
def fb():
    tr._forward_backward(kf, sup, joints, vis)
g4 = graph_of(fb); res['forward + loss + backward'] = timeit(g4.replay)
for lanes in ('0',):
    os.environ['FAMI_LANES'] = lanes
    g5 = graph_of(fb); res['forward + loss + backward, FAMI_LANES=%s' % lanes] = timeit(g5.replay)
    g6 = graph_of(lambda: hr(True, True)); res['HRNet forward + backward, FAMI_LANES=%s' % lanes] = timeit(g6.replay)
    del os.environ['FAMI_LANES']
for k, v in res.items(): print('%s %-50s %7.2f ms' % (dtype, k, v))

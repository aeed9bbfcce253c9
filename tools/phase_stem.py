"""The serial stem + layer1 stretch of the HRNet backbone (hrnet.py:632-650: conv1, conv2, four Bottlenecks) ALONE: un-profiled wall
time under hipGraph replay (forward, forward + backward), next to the full step -- the stretch runs before the branches fork and
its backward after they have joined, so its time adds to the step one for one.  Under `rocprofv3 --kernel-trace --stats` the same
script gives the per-kernel times of the stretch (a serial chain: no lane contention).
usage: phase_stem.py f32|bf16 [nostep]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fami_pose_amd.engine import Engine
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

adt = model.act_dtype
hr = model.hrnet
def stretch(record, back, upto=4):
    eng = Engine(dev, grad_views=tr.views, record=record, dtype=adt)
    tr.packer.run(eng.stream); eng.prepacked = tr.packer.views
    x = eng.frames(kf, sup)
    eng.wlane_scope = eng.stem_wlane; eng.wlane_pair = True; eng.serial_scope = True
    x = eng.conv_bn(x, hr.conv1, hr.bn1, relu=True)
    x = eng.conv_bn(x, hr.conv2, hr.bn2, relu=True)
    for blk in list(hr.layer1)[:upto]:
        x = blk.run(eng, x)
    eng.wlane_scope = False; eng.wlane_pair = False; eng.serial_scope = False
    if back:
        x.grad = torch.ones_like(x.data); eng.backward()
def pack_only():
    eng = Engine(dev, grad_views=tr.views, record=False, dtype=adt)
    tr.packer.run(eng.stream)
res = {}
if len(sys.argv) < 3:
    res['full step'] = timeit(lambda: tr.step(kf, sup, joints, vis))
res['weight pack alone'] = timeit(graph_of(pack_only).replay)
res['stem + layer1 forward (incl. pack)'] = timeit(graph_of(lambda: stretch(False, False)).replay)
res['stem + layer1 forward + backward (incl. pack)'] = timeit(graph_of(lambda: stretch(True, True)).replay)
res['stem only forward + backward (incl. pack)'] = timeit(graph_of(lambda: stretch(True, True, 0)).replay)
res['stem + first Bottleneck forward + backward (incl. pack)'] = timeit(graph_of(lambda: stretch(True, True, 1)).replay)
for k, v in res.items(): print('%s %-58s %7.3f ms' % (dtype, k, v))
if 'full step' in res:
    s = res['stem + layer1 forward + backward (incl. pack)'] - res['weight pack alone']
    print('%s stem + layer1 stretch = %.3f ms = %.1f %% of the step' % (dtype, s, 100 * s / res['full step']))

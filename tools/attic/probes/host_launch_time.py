"""Host time of one Trainer.step() call in graph mode (the hipGraphLaunch of ~2400 kernel nodes is asynchronous): how far is the step from being
bound by the host's enqueue?  usage: host_launch_time.py <dtype>"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch
import bench
from fami_pose_amd.train import Trainer
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
args = types.SimpleNamespace(width=48, img_w=288, img_h=384, sup=4, freeze_backbone=False, dtype=dtype, deterministic=False)
kf, sup, joints, vis = bench.synth_batch(4, 4, 384, 288, 17, dev, 19970808)
tr = Trainer(bench.build(args, dev), lr=1e-3, use_mi=True, use_graph=True, targets_from_joints=True)
for _ in range(5): tr.step(kf, sup, joints, vis)
torch.cuda.synchronize()
# (a) the device idle when the call is made: pure host time of the call
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.step(kf, sup, joints, vis); ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
# (b) back to back: a call returns when the runtime has accepted the launch (it may wait for queue space)
t0 = time.perf_counter(); tb = []
for _ in range(20):
    t1 = time.perf_counter(); tr.step(kf, sup, joints, vis); tb.append((time.perf_counter() - t1) * 1e3)
host_total = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize(); total = (time.perf_counter() - t0) * 1e3
ts.sort(); 
print('%s: host time of a step() call with the device idle: median %.2f ms (min %.2f, max %.2f); 20 calls back to back: host returned after %.1f ms, device done after %.1f ms (%.2f ms per step); per-call host times %s' %
      (dtype, ts[len(ts) // 2], ts[0], ts[-1], host_total, total, total / 20, ' '.join('%.1f' % t for t in tb)))

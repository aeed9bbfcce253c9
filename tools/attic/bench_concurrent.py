"""How much of the GPU does one training step leave idle?  Two independent Trainers replay their step graphs on two
streams at once; aggregate throughput vs one alone bounds what intra-step (branch-level) concurrency could win."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fami_pose_amd.train import Trainer
dev = torch.device('cuda:0')
import argparse
args = argparse.Namespace(width=48, img_w=288, img_h=384, sup=4, freeze_backbone=False, dtype=os.environ.get('DTYPE', 'f32'))
kf, sup, joints, vis = bench.synth_batch(4, 4, 384, 288, 17, dev, 1)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
trainers = []
for s in streams:
    with torch.cuda.stream(s):
        m = bench.build(args, dev)
        t = Trainer(m, lr=1e-3, use_graph=True, targets_from_joints=True)
        for _ in range(3):
            t.step(kf, sup, joints, vis)
        trainers.append(t)
torch.cuda.synchronize()
def run(active, steps=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        for i in active:
            with torch.cuda.stream(streams[i]):
                trainers[i].step(kf, sup, joints, vis)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
a = run([0]); b = run([0, 1])
print('%s: one trainer %.1f ms/step; two concurrent %.1f ms per pair -> %.2fx aggregate throughput' % (args.dtype, a, b, 2 * a / b))

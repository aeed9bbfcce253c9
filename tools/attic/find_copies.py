"""Where do the device-to-device copies of a training step come from?  Runs two eager steps under torch.profiler with
python stacks and prints the call sites of aten::copy_ / Memcpy events (run by hand on the GPU box)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch
import bench
from fami_pose_amd.train import Trainer
args = argparse.Namespace(batch=4, sup=4, width=48, img_h=384, img_w=288, freeze_backbone=False, dtype=os.environ.get('DTYPE', 'f32'))
dev = torch.device('cuda:0')
tr = Trainer(bench.build(args, dev), lr=1e-3, use_mi=True, use_graph=False, targets_from_joints=True)
kf, sup, joints, vis = bench.synth_batch(4, 4, 384, 288, 17, dev, 1)
for _ in range(2):
    tr.step(kf, sup, joints, vis)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(kf, sup, joints, vis)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::zero_', 'aten::fill_', 'aten::zeros', 'aten::cat', 'aten::add_', 'aten::mul_', 'aten::sum'):
        st = [s for s in (e.stack or []) if 'fami' in s or 'bench' in s][:3]
        cnt[(e.name, ' <- '.join(st))] += 1
for (n, st), c in cnt.most_common(30):
    print('%5d  %-16s %s' % (c, n, st))
names = collections.Counter(e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
for n, c in names.most_common(12):
    print('%6d  %s' % (c, n[:100]))

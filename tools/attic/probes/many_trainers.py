"""Many live Trainers in one process (each captures its own hipGraph): torch hands streams out of a 32-entry round-robin pool, so sooner or later a
capture stream IS one of the engine's cached lane / weight-gradient streams.  Prints the handles per capture; stops at the first failure."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch
import bench
from fami_pose_amd.train import Trainer
from fami_pose_amd.engine import Engine
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
args = types.SimpleNamespace(width=48, img_w=96, img_h=128, sup=2, freeze_backbone=False, dtype=sys.argv[1] if len(sys.argv) > 1 else 'bf16', deterministic=False)
kf, sup, joints, vis = bench.synth_batch(2, 2, 128, 96, 17, dev, 1)
keep = []
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 16):
    tr = Trainer(bench.build(args, dev), lr=1e-3, use_mi=True, use_graph=True, targets_from_joints=True)
    try:
        for _ in range(3): tr.step(kf, sup, joints, vis)
        torch.cuda.synchronize()
    except Exception as e:
        print('trainer %d FAILED: %s' % (i, str(e).split('\n')[0]), flush=True)
        side = [hex(t.cuda_stream) for t in Engine._side_pool.get(dev, [])]
        wg = {k: hex(v.cuda_stream) for k, v in Engine._wgrad_pool.items()}
        print('  side pool', side, '\n  wgrad pool', wg, flush=True)
        break
    side = [hex(t.cuda_stream) for t in Engine._side_pool.get(dev, [])]
    wg = [hex(v.cuda_stream) for v in Engine._wgrad_pool.values()]
    print('trainer %d ok: loss %.4f  side pool %s  wgrad pool %s' % (i, tr.loss_value(), side, wg), flush=True)
    keep.append(tr)

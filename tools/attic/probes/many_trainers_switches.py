"""tools/probes/many_trainers.py with a different host-switch setting per Trainer (the sequence that ended in "capturing stream has unjoined
work" at the tenth Trainer of an ab_env run), small images."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch
import bench
from fami_pose_amd.train import Trainer
from fami_pose_amd.engine import Engine
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
args = types.SimpleNamespace(width=48, img_w=96, img_h=128, sup=2, freeze_backbone=False, dtype='bf16', deterministic=False)
kf, sup, joints, vis = bench.synth_batch(2, 2, 128, 96, 17, dev, 1)
SEQ = [{}, {'FAMI_WGRAD_LANE': '1'}, {'FAMI_STEM_WGRAD_LANES': '2'}, {'FAMI_STEM_WGRAD_LANE': '0'}, {'FAMI_HEAD_WGRAD_LANE': '0'}, {'FAMI_XBN': '1'},
       {'FAMI_PERSIST_LANES': '0'}, {'FAMI_PACK_EARLY': '0'}, {'FAMI_MI_LANES': '0'}, {'FAMI_REGRESSOR_LANES': '0'}, {'FAMI_FUSE_LANES': '0'}, {}, {'FAMI_LANES': '0'}, {}]
keep = []
for i, env in enumerate(SEQ * 2):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        tr = Trainer(bench.build(args, dev), lr=1e-3, use_mi=True, use_graph=True, targets_from_joints=True)
        for _ in range(3): tr.step(kf, sup, joints, vis)
        torch.cuda.synchronize()
        print('trainer %d %s ok: loss %.4f' % (i, env, tr.loss_value()), flush=True)
        keep.append(tr)
    except Exception as e:
        print('trainer %d %s FAILED: %s' % (i, env, str(e).split('\n')[0]), flush=True)
        print('  side pool', [hex(t.cuda_stream) for t in Engine._side_pool.get(dev, [])], ' wgrad pool', {k: hex(v.cuda_stream) for k, v in Engine._wgrad_pool.items()}, flush=True)
        break
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

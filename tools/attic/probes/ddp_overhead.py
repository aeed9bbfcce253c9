"""Per-GPU cost of the data-parallel launch plan itself: the headline workload with a ONE-rank RCCL group (Trainer(force_ddp=True): bucket
graphs, an all-reduce per 32 MB slice that moves nothing, no early Adam, a join per HRNet module) against the plain single-GPU step, graphs
replayed alternately on one box.  usage: ddp_overhead.py <dtype> [rounds]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch
import torch.distributed as dist
import bench
from fami_pose_amd.train import Trainer
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dtype = sys.argv[1]; rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
args = types.SimpleNamespace(width=48, img_w=288, img_h=384, sup=4, freeze_backbone=False, dtype=dtype, deterministic=False)
kf, sup, joints, vis = bench.synth_batch(4, 4, 384, 288, 17, dev, 19970808)
trs = []
for name, force, env in (('single', False, {}), ('ddp_overlap', True, {}), ('ddp_serial', True, {'FAMI_DDP_PLAN': 'serial'})):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    tr = Trainer(bench.build(args, dev), lr=1e-3, use_mi=True, use_graph=True, targets_from_joints=True, force_ddp=force, bucket_mb=32)
    for _ in range(3): tr.step(kf, sup, joints, vis)
    torch.cuda.synchronize()
    trs.append((name, tr))
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
res = {n: [] for n, _ in trs}
for r in range(rounds):
    for name, tr in (trs if r % 2 == 0 else trs[::-1]):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): tr.step(kf, sup, joints, vis)
        torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / 10 * 1e3)
for n, v in res.items():
    v = sorted(v)
    print('%s %-12s median %.3f ms  min %.3f  (%s)' % (dtype, n, v[len(v) // 2], v[0], ' '.join('%.2f' % x for x in v)), flush=True)
dist.destroy_process_group()

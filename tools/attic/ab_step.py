"""Interleaved A/B of whole training steps under two tuning states, in ONE process (graphs are captured once per
state; timing alternates A,B,A,B so clock / thermal drift hits both).  usage: python tools/ab_step.py f32|bf16 A B
   A, B: comma lists of name=int tune calls, e.g.  xcd=1  xcd=3   (fami_conv_tune_<name>); UPPERCASE names are
   environment variables set while that state's Trainer is built and captured (e.g. FAMI_WGRAD_LANE=1)"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fami_pose_amd._lib import lib
from fami_pose_amd.train import Trainer
L = lib().cdll
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dtype = sys.argv[1]
states = [dict((kv.split('=')[0], kv.split('=')[1]) for kv in a.split(',')) for a in sys.argv[2:4]]
args = types.SimpleNamespace(width=48, img_w=288, img_h=384, sup=4, freeze_backbone=False, dtype=dtype, deterministic=False)
kf, sup, joints, vis = bench.synth_batch(4, 4, 384, 288, 17, dev, 19970808)
trainers = []
for st in states:
    for k, v in st.items():
        if k.isupper():
            os.environ[k] = v
        elif k == 'tile':                     # tile=mt:nt:ks -> fami_conv_tune(mt, nt, ks)
            L.fami_conv_tune(*[int(t) for t in v.split(':')])
        elif k == 'bnsmall':
            L.fami_bn_tune_small(int(v))
        else:
            getattr(L, 'fami_conv_tune_' + k.split('#')[0])(int(v))      # 'lds#2=118': a second call of the same knob
    tr = Trainer(bench.build(args, dev), lr=1e-3, use_mi=os.environ.get('AB_NO_MI') != '1', use_graph=True, targets_from_joints=True)
    for _ in range(3):
        tr.step(kf, sup, joints, vis)
    trainers.append(tr)
    for k in st:
        if k.isupper():
            del os.environ[k]
torch.cuda.synchronize()
res = [[], []]
for rnd in range(int(os.environ.get('AB_ROUNDS', '6'))):
    for i, tr in enumerate(trainers):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            tr.step(kf, sup, joints, vis)
        torch.cuda.synchronize(); res[i].append((time.perf_counter() - t0) / 8 * 1e3)
for st, r in zip(states, res):
    print(dtype, st, ' '.join('%.2f' % v for v in r), '| median %.2f ms' % sorted(r)[len(r) // 2])

"""A/B of library knobs on ONE box with less noise than bench.py runs: captures the training step once per setting (hipGraph) and
replays the graphs alternately.  usage: ab_env.py <dtype> <rounds> "<name>:<tune call>,<tune call>" ...   tune call = lds:N | wgrad:N | dcn:N | bn:N | stages:N | env:NAME=VALUE
e.g. ab_env.py bf16 6 "t256:wgrad:21256" "t128:wgrad:21128" """
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fami_pose_amd._lib import lib
from fami_pose_amd.train import Trainer
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dtype, rounds = sys.argv[1], int(sys.argv[2])
L = lib().cdll
# workload: the headline's by default; AB_H / AB_W / AB_SUP / AB_WIDTH / AB_BATCH select another BASELINE config
EH, EW, ES, EC, EB = (int(os.environ.get(k, d)) for k, d in (('AB_H', 384), ('AB_W', 288), ('AB_SUP', 4), ('AB_WIDTH', 48), ('AB_BATCH', 4)))
args = types.SimpleNamespace(width=EC, img_w=EW, img_h=EH, sup=ES, freeze_backbone=False, dtype=dtype, deterministic=False)
kf, sup, joints, vis = bench.synth_batch(EB, ES, EH, EW, 17, dev, 19970808)
trs = []
for spec in sys.argv[3:]:
    name, calls = spec.split(':', 1)
    L.fami_tune_reset()
    envs = []
    for c in calls.split(','):
        if not c: continue
        kind, val = c.split(':')
        if kind == 'env':                      # env:NAME=VALUE (read when the Trainer / Engine of this setting is built)
            k, v = val.split('=')
            envs.append((k, os.environ.get(k)))
            os.environ[k] = v
        else:
            {'lds': L.fami_conv_tune_lds, 'wgrad': L.fami_conv_tune_wgrad_lds, 'dcn': L.fami_dcn_tune, 'bn': L.fami_bn_tune_small, 'stages': L.fami_conv_tune_stages}[kind](int(val))
    model = bench.build(args, dev)
    tr = Trainer(model, lr=1e-3, use_mi=True, use_graph=True, targets_from_joints=True)
    for _ in range(3): tr.step(kf, sup, joints, vis)      # capture happens under this setting
    torch.cuda.synchronize()
    print('%s: statistics passes in convolution epilogues %s' % (name, dict(tr.last_nfused)), flush=True)
    trs.append((name, tr))
    for k, old in envs:
        if old is None: os.environ.pop(k, None)
        else: os.environ[k] = old
L.fami_tune_reset()
res = {n: [] for n, _ in trs}
for r in range(rounds):
    for name, tr in (trs if r % 2 == 0 else trs[::-1]):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): tr.step(kf, sup, joints, vis)
        torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / 10 * 1e3)
for n, v in res.items():
    v = sorted(v)
    print('%s %-10s median %.3f ms  min %.3f  (%s)' % (dtype, n, v[len(v) // 2], v[0], ' '.join('%.2f' % x for x in v)))

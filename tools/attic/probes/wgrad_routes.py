"""GPU box: which weight-gradient geometries of one training step does the pipelined kernel (conv_wg16 / conv_wgs3) NOT take?
usage: python tools/probes/wgrad_routes.py bf16|f32"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, types
import bench
from fami_pose_amd import engine as E
from fami_pose_amd._lib import lib
from fami_pose_amd.train import Trainer
dt = sys.argv[1]
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
L = lib().cdll
seen = collections.Counter()
orig = E.Engine.wgrad
def spy(self, x_data, dy, g, geo, acc, xbn=None):
    N, H, W, Ci, Co, kh, kw, st, pad, dil = geo
    if dt == 'f32':
        took = kh == 3 and kw == 3 and st == 1 and pad == 1 and dil == 1 and getattr(L, '_Z19fami_wgrad_s3_slabsiiiii')(N, H, W, Ci, Co) > 0
    else:
        took = kh == kw and getattr(L, '_Z18fami_wgrad16_slabsiiiiiiiii')(N, H, W, Ci, Co, kh, st, pad, dil) > 0
    seen[(took, geo)] += 1
    return orig(self, x_data, dy, g, geo, acc, xbn)
E.Engine.wgrad = spy
args = types.SimpleNamespace(width=48, img_w=288, img_h=384, sup=4, freeze_backbone=False, dtype=dt, deterministic=False)
kf, sup, joints, vis = bench.synth_batch(4, 4, 384, 288, 17, dev, 1)
tr = Trainer(bench.build(args, dev), lr=1e-3, use_graph=False, targets_from_joints=True)
tr.step(kf, sup, joints, vis)
seen.clear()
tr.step(kf, sup, joints, vis)
torch.cuda.synchronize()
tot = sum(seen.values()); no = [(g, c) for (t, g), c in seen.items() if not t]
print(dt, 'weight gradients per step:', tot, ' not on the pipelined kernel:', sum(c for _, c in no))
for g, c in sorted(no, key=lambda t: -t[1] * t[0][0] * t[0][1] * t[0][2] * t[0][3] * t[0][4] * t[0][5] * t[0][6]):
    N, H, W, Ci, Co, kh, kw, st, pad, dil = g
    print('  x%-3d N=%d %dx%d %d->%d k%d s%d p%d d%d  %.2f GFLOP each' % (c, N, H, W, Ci, Co, kh, st, pad, dil, 2e-9 * N * (H // st) * (W // st) * Ci * Co * kh * kw))

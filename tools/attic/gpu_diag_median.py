"""Distribution of the parameter-gradient error (HIP fp32 vs an fp64 oracle, CPU fp32 beside it) under kernel-selection
switches (diagnostics, run by hand on the GPU box).  usage: S=7 H=128 W=96 python tools/gpu_diag_median.py"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fami_pose_amd as fp
from fami_pose_amd._lib import lib
from fami_pose_amd.loss import JointMSELoss
from oracle import model as om, ops as oops

dev = torch.device('cuda:0')
S, H, W, B = int(os.environ.get('S', 7)), int(os.environ.get('H', 128)), int(os.environ.get('W', 96)), int(os.environ.get('B', 2))
orc = om.realistic_init_(om.AlignmentOracle(om.make_cfg(48), True, S, (H, W)), 3 + S)
gen = torch.Generator().manual_seed(50 + S)
kf, sup = torch.randn(B, 3, H, W, generator=gen), torch.randn(B, 3 * S, H, W, generator=gen)
tgt = torch.rand(B, 17, H // 4, W // 4, generator=gen)
w = (torch.rand(B, 17, 1, generator=gen) < 0.8).float()
f0, k0, mi0 = orc(kf, sup)
oops.total_loss(f0, tgt, w, mi0).backward()
orc64 = copy.deepcopy(orc).double()
orc64.zero_grad()
f64, _, mi64 = orc64(kf.double(), sup.double())
oops.total_loss(f64, tgt.double(), w.double(), mi64).backward()
ref, ref64 = dict(orc.named_parameters()), dict(orc64.named_parameters())
L = lib().cdll


def run(tag):
    model = fp.build_model(fp.default_cfg(48, image_size=(W, H), num_sup=S), 'train')
    model.load_state_dict(orc.state_dict())
    model = model.to(dev)
    f1, k1, mi1 = model(kf.to(dev), sup.to(dev))
    l1 = JointMSELoss()(f1, tgt.to(dev), w.to(dev)) + 0.5 * (-0.1 * mi1[0] + 0.1 * mi1[1] + mi1[2] - mi1[3] + mi1[4] - mi1[5])
    l1.backward()
    mine = dict(model.named_parameters())
    ec, eh, names = [], [], []
    for name, p64 in ref64.items():
        if p64.grad is None:
            continue
        g64 = p64.grad
        s64 = g64.abs().max().item()
        if s64 < 1e-9:
            continue
        ec.append((ref[name].grad.double() - g64).abs().max().item() / s64)
        eh.append((mine[name].grad.cpu().double() - g64).abs().max().item() / s64)
        names.append(name)
    ec, eh = np.array(ec), np.array(eh)
    kinds = {'conv.weight': [i for i, n in enumerate(names) if n.endswith('weight') and ref[n].dim() == 4],
             'bn/bias/other': [i for i, n in enumerate(names) if not (n.endswith('weight') and ref[n].dim() == 4)]}
    print('%-28s median hip %.4f cpu %.4f | p90 hip %.4f cpu %.4f | max hip %.3f cpu %.3f' %
          (tag, np.median(eh), np.median(ec), np.percentile(eh, 90), np.percentile(ec, 90), eh.max(), ec.max()))
    for k, idx in kinds.items():
        print('      %-14s n=%4d median hip %.4f cpu %.4f' % (k, len(idx), np.median(eh[idx]), np.median(ec[idx])), flush=True)


run('defaults')
L.fami_conv_tune_wgrad_lds(0); run('wgrad scalar kernels'); L.fami_conv_tune_wgrad_lds(-1)
L.fami_conv_tune(16, 0, 0); run('16x16 igemm only'); L.fami_conv_tune(0, 0, 0)
L.fami_conv_tune_wgrad_lds(1); run('wgrad lds everywhere'); L.fami_conv_tune_wgrad_lds(-1)
run('defaults again')

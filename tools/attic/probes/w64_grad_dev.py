"""GPU box: deviation of the ill-conditioned W64 gradients (tests/test_model_gpu.py::test_config5_w64_full_size (a)) from the
CPU oracle with the exact-f32 MFMA convolutions and with the split-product ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
import test_model_gpu as tm
from fami_pose_amd._lib import lib
om, fp, oops = tm.om, tm.fp, tm.oops
dev = torch.device('cuda:0')
S, H, W, B = 4, 384, 288, 1
orc = om.realistic_init_(om.AlignmentOracle(om.make_cfg(64), True, S, (H, W), dcn_groups=16), 64)
gen = torch.Generator().manual_seed(164)
kf, sup = torch.randn(B, 3, H, W, generator=gen), torch.randn(B, 3 * S, H, W, generator=gen)
tgt = torch.rand(B, 17, H // 4, W // 4, generator=gen)
w = (torch.rand(B, 17, 1, generator=gen) < 0.8).float()
f0, k0, mi0 = orc(kf, sup)
l0 = oops.total_loss(f0, tgt, w, mi0)
l0.backward()
from fami_pose_amd.loss import JointMSELoss
ref = dict(orc.named_parameters())
for knob in (30, 31):
    lib().cdll.fami_conv_tune_lds(-1); lib().cdll.fami_conv_tune_lds(knob)
    model = fp.build_model(fp.default_cfg(64, image_size=(W, H), num_sup=S), 'train')
    model.load_state_dict(orc.state_dict()); model = model.to(dev)
    f1, k1, mi1 = model(kf.to(dev), sup.to(dev))
    l1 = JointMSELoss()(f1, tgt.to(dev), w.to(dev)) + 0.5 * (-0.1 * mi1[0] + 0.1 * mi1[1] + mi1[2] - mi1[3] + mi1[4] - mi1[5])
    l1.backward()
    mine = dict(model.named_parameters())
    dev_ = []
    for name, p in ref.items():
        if p.grad is None or p.grad.abs().max().item() < 1e-9:
            continue
        a, b = mine[name].grad.double().abs().sum().item(), p.grad.double().abs().sum().item()
        dev_.append((abs(a - b) / b, name))
    dev_.sort(reverse=True)
    print('knob', knob, 'heatmap err %.2e' % (f1.cpu() - f0).abs().max().item(), 'loss', l1.item(), l0.item())
    for d, n in dev_[:8]:
        print('   %.4f %s' % (d, n))

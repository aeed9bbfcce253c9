import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, numpy as np
import fami_pose_amd as fp
from oracle import model as om
from fami_pose_amd.train import Trainer
dev = torch.device('cuda:0')
orc = om.realistic_init_(om.AlignmentOracle(om.make_cfg(48), True, 2, (128, 96)), 5)
m2 = fp.build_model(fp.default_cfg(48, image_size=(96, 128), num_sup=2), 'train'); m2.load_state_dict(orc.state_dict())
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
tr = Trainer(m2.to(dev).set_compute_dtype(mode), lr=1e-3, use_graph=False, targets_from_joints=True)
gen = torch.Generator().manual_seed(9)
kf2, sup2 = torch.randn(2, 3, 128, 96, generator=gen).to(dev), torch.randn(2, 6, 128, 96, generator=gen).to(dev)
joints = (torch.rand(2, 17, 2, generator=gen) * torch.tensor([96.0, 128.0])).to(dev)
vis = (torch.rand(2, 17, generator=gen) < 0.8).float().to(dev)
tr.step(kf2, sup2, joints, vis)
torch.cuda.synchronize()
print('loss', tr.loss_value(), 'parts', tr.loss_parts.tolist())
bad = [n for n, p in tr.model.named_parameters() if id(p) in tr.views and not torch.isfinite(tr.views[id(p)]).all()]
print(len(bad), 'params with non-finite grads:', bad[:30])
badp = [n for n, p in tr.model.named_parameters() if not torch.isfinite(p).all()]
print(len(badp), 'non-finite params', badp[:10])

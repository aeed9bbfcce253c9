"""fp16 mode: head-gradient error against the fp32 oracle by loss scale (GPU box; imports tests helpers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import test_model_gpu as tm
from oracle import ops as oops
from fami_pose_amd.loss import JointMSELoss
dev = torch.device('cuda:0')
S, H, W, B = 4, 384, 288, 2
names = ('agg_final_layer.weight', 'init_feature_agg_block.layers.2.conv2.weight', 'dcn_4.weight', 'dcn_1.weight',
         'dcn_mask_4.conv.weight', 'dcn_offset_4.conv.weight', 'combined_feat_layers.layers.0.conv1.weight')
for mode, scales in (('f16', (64.0, 4096.0, 65536.0)), ('bf16', (1.0,))):
    model, orc = tm._pair(48, S, (H, W), 'train', 31)
    model = model.to(dev).set_compute_dtype(mode)
    gen = torch.Generator().manual_seed(131)
    kf, sup = torch.randn(B, 3, H, W, generator=gen), torch.randn(B, 3 * S, H, W, generator=gen)
    tgt = torch.rand(B, 17, H // 4, W // 4, generator=gen)
    w = (torch.rand(B, 17, 1, generator=gen) < 0.8).float()
    orc.zero_grad()
    fo, ko, mio = orc(kf, sup)
    oops.total_loss(fo, tgt, w, mio).backward()
    ref = dict(orc.named_parameters())
    for ls in scales:
        model.zero_grad()
        f1, k1, mi1 = model(kf.to(dev), sup.to(dev))
        l1 = JointMSELoss()(f1, tgt.to(dev), w.to(dev)) + 0.5 * (-0.1 * mi1[0] + 0.1 * mi1[1] + mi1[2] - mi1[3] + mi1[4] - mi1[5])
        (l1 * ls).backward()
        mine = dict(model.named_parameters())
        print(mode, 'loss scale', ls, ' '.join('%s %.3f' % (n.split('.')[0] + '.' + n.split('.')[-2][:5], ((mine[n].grad.cpu() / ls - ref[n].grad).norm() / ref[n].grad.norm()).item()) for n in names if n in ref and ref[n].grad is not None), flush=True)

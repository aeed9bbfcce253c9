"""Stage-by-stage comparison of the HIP model against the oracle (diagnostics, run by hand on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fami_pose_amd as fp
from fami_pose_amd.engine import Engine
from oracle import model as om

dev = torch.device('cuda:0')
torch.manual_seed(0)
S, H, W, B = int(os.environ.get('S', 2)), int(os.environ.get('H', 256)), int(os.environ.get('W', 192)), int(os.environ.get('B', 2))
gain = float(os.environ.get('GAIN', 0.7))
cfg = fp.default_cfg(48, image_size=(W, H), num_sup=S)
model = fp.build_model(cfg, 'train')
orc = om.AlignmentOracle(om.make_cfg(48), True, S, (H, W))
om.realistic_init_(orc, 1, offset_std=float(os.environ.get("OFFSTD", 1.0)))
model.load_state_dict(orc.state_dict())
model = model.to(dev)
model.set_compute_dtype(os.environ.get('DTYPE', 'f32'))
kf, sup = torch.randn(B, 3, H, W), torch.randn(B, 3 * S, H, W)
f0, k0, mi0, aux0 = orc(kf, sup, return_aux=True)
eng = Engine(dev, record=False, dtype=model.act_dtype)
outs, _ = model._body(eng, kf.to(dev), sup.to(dev))
a = eng.aux
def cmp(name, t, ref):
    x = t.data if hasattr(t, 'data') and not torch.is_tensor(t) else t
    if x.dim() == 4 and x.shape != ref.shape:
        x = x.permute(0, 3, 1, 2)
    d = (x.float().cpu() - ref).abs().max().item()
    print('%-10s max|ref| %.3e  maxabs diff %.3e  rel %.3e' % (name, ref.abs().max().item(), d, d / ref.abs().max().item()))
import torch.nn.functional as Fn
fr = torch.cat([kf] + list(torch.chunk(sup, S, 1)), 0)
with torch.no_grad():
    st1 = Fn.relu(orc.hrnet.bn1(orc.hrnet.conv1(fr)))
cmp('kf_feat', a['kf_feat'], aux0['kf_feat'])
for i, (t, r) in enumerate(zip(a['shifts'], aux0['shifts'])):
    cmp('shift%d' % i, t, r)
    print('   ', t.data.cpu().tolist(), r.tolist())
cmp('agg_sup', a['agg_sup'], aux0['agg_sup'])
cmp('aligned', a['aligned'], aux0['aligned'])
cmp('all_agg', a['all_agg'], aux0['all_agg'])
cmp('final', outs[0], f0)
cmp('kf_hm', outs[1], k0)
for i in range(6):
    print('mi%d %.6e vs %.6e' % (i + 1, outs[2 + i].item(), mi0[i].item()))

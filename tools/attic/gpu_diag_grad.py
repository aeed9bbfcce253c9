"""Stage-by-stage gradient comparison HIP vs oracle (diagnostics, run by hand on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fami_pose_amd as fp
from fami_pose_amd.engine import Engine
from oracle import model as om, ops as oops

dev = torch.device('cuda:0')
S, H, W, B = int(os.environ.get('S', 2)), int(os.environ.get('H', 384)), int(os.environ.get('W', 288)), int(os.environ.get('B', 2))
seed = int(os.environ.get('SEED', 5))
USE_MI = int(os.environ.get('MI', 1))
orc = om.realistic_init_(om.AlignmentOracle(om.make_cfg(48), True, S, (H, W)), seed)
model = fp.build_model(fp.default_cfg(48, image_size=(W, H), num_sup=S), 'train')
model.load_state_dict(orc.state_dict())
model = model.to(dev)
gen = torch.Generator().manual_seed(50 + S)
kf, sup = torch.randn(B, 3, H, W, generator=gen), torch.randn(B, 3 * S, H, W, generator=gen)
tgt = torch.rand(B, 17, H // 4, W // 4, generator=gen)
w = (torch.rand(B, 17, 1, generator=gen) < 0.8).float()
f0, k0, mi0, aux0 = orc(kf, sup, return_aux=True)
for k in ('agg_sup', 'aligned', 'all_agg', 'kf_feat'):
    aux0[k].retain_grad()
for t in aux0['shifts']:
    t.retain_grad()
f0.retain_grad()
l0 = oops.total_loss(f0, tgt, w, mi0 if USE_MI else [])
l0.backward()

tune = os.environ.get("TUNE")
if tune:
    eng_l = __import__("fami_pose_amd._lib", fromlist=["lib"]).lib()
    eng_l.cdll.fami_conv_tune(*[int(v) for v in tune.split(",")])
eng = Engine(dev, record=True)
outs, seeds = model._body(eng, kf.to(dev), sup.to(dev))
from fami_pose_amd.loss import JointMSELoss
outs_t = [o.detach().requires_grad_(True) for o in outs]
l1 = JointMSELoss()(outs_t[0], tgt.to(dev), w.to(dev))
if USE_MI:
    mi1 = outs_t[2:]
    l1 = l1 + 0.5 * (-0.1 * mi1[0] + 0.1 * mi1[1] + mi1[2] - mi1[3] + mi1[4] - mi1[5])
l1.backward()
for fn, o in zip(seeds, outs_t):
    if o.grad is not None:
        fn(o.grad)
eng.backward()
a = eng.aux
print('loss', l0.item(), l1.item())

def cmp(name, g1, g0):
    if g1 is None or g0 is None:
        print('%-46s missing (%s, %s)' % (name, g1 is None, g0 is None)); return
    if g1.dim() == 4 and g1.shape != g0.shape:
        g1 = g1.permute(0, 3, 1, 2)
    g1 = g1.cpu()
    d = (g1 - g0).abs().max().item()
    print('%-46s max|ref| %.3e  maxdiff %.3e  rel %.3e' % (name, g0.abs().max().item(), d, d / (g0.abs().max().item() + 1e-30)))

cmp('d final', a['final'].grad, f0.grad)
cmp('d all_agg', a['all_agg'].grad, aux0['all_agg'].grad)
cmp('d aligned', a['aligned'].grad, aux0['aligned'].grad)
cmp('d agg_sup', a['agg_sup'].grad, aux0['agg_sup'].grad)
cmp('d kf_feat', a['kf_feat'].grad, aux0['kf_feat'].grad)
for i, (t1, t0) in enumerate(zip(a['shifts'], aux0['shifts'])):
    cmp('d shift%d' % i, t1.grad, t0.grad)
ref = dict(orc.named_parameters())
names = [n for n, _ in model.named_parameters() if not n.startswith('hrnet.')]
names += ['hrnet.stage4.2.fuse_layers.0.1.0.weight', 'hrnet.stage4.2.branches.0.3.conv2.weight', 'hrnet.stage4.0.branches.3.0.conv1.weight',
          'hrnet.stage3.0.branches.1.0.conv1.weight', 'hrnet.stage2.0.branches.1.3.bn2.bias', 'hrnet.layer1.0.conv1.weight', 'hrnet.conv1.weight']
mp = dict(model.named_parameters())
for n in names:
    g1 = eng.param_grads.get(id(mp[n]))
    cmp(n, g1, ref[n].grad)

# ---- which side is closer to the truth?  fp64 oracle as the arbiter
import copy
orc64 = copy.deepcopy(orc).double()
orc64.zero_grad()
f64, k64, mi64, aux64 = orc64(kf.double(), sup.double(), return_aux=True)
for k in ('agg_sup', 'aligned', 'all_agg', 'kf_feat'):
    aux64[k].retain_grad()
l64 = oops.total_loss(f64, tgt.double(), w.double(), mi64 if USE_MI else [])
l64.backward()
print('\nfp64 arbiter: |final32 - final64| cpu %.3e hip %.3e' % ((f0.double() - f64).abs().max().item(), (outs[0].cpu().double() - f64).abs().max().item()))
def cmp3(name, g_hip, g_cpu, g64):
    if g_hip.dim() == 4 and g_hip.shape != g64.shape:
        g_hip = g_hip.permute(0, 3, 1, 2)
    s = g64.abs().max().item() + 1e-300
    print('%-46s cpu32-vs-64 %.3e   hip-vs-64 %.3e   hip-vs-cpu32 %.3e' % (
        name, (g_cpu.double() - g64).abs().max().item() / s, (g_hip.cpu().double() - g64).abs().max().item() / s,
        (g_hip.cpu() - g_cpu).abs().max().item() / s))
for k in ('all_agg', 'aligned', 'agg_sup', 'kf_feat'):
    cmp3('d ' + k, a[k].grad, aux0[k].grad, aux64[k].grad)
ref64 = dict(orc64.named_parameters())
for n in ['init_feature_agg_block.layers.2.conv2.weight', 'init_feature_agg_block.layers.0.conv1.weight', 'dcn_4.weight', 'dcn_offset_2.conv.weight',
          'sup_agg_block.layers.0.conv1.weight', 'feat_global_offset_layers.2.conv.weight', 'hrnet.stage3.0.branches.1.0.conv1.weight', 'hrnet.conv1.weight']:
    cmp3(n, eng.param_grads[id(mp[n])], ref[n].grad, ref64[n].grad)

# ---- ranking of parameters by (hip error) / (cpu error), both vs fp64
rows = []
for n, p64 in ref64.items():
    if p64.grad is None or n not in mp or eng.param_grads.get(id(mp[n])) is None:
        continue
    g64 = p64.grad; s64 = g64.abs().max().item()
    if s64 < 1e-9: continue
    e_cpu = (ref[n].grad.double() - g64).abs().max().item() / s64
    e_hip = (eng.param_grads[id(mp[n])].cpu().double() - g64).abs().max().item() / s64
    rows.append((e_hip / (e_cpu + 1e-6), e_hip, e_cpu, n))
rows.sort(reverse=True)
import numpy as np
print('\nmedian e_hip %.3e  median e_cpu %.3e' % (np.median([r[1] for r in rows]), np.median([r[2] for r in rows])))
for r in rows[:25]:
    print('ratio %6.1f  hip %.3e cpu %.3e  %s' % r)

"""The HIP path (through the C ABI) against the REFERENCE-generated block goldens g1 / g2 / g4 / g8
(tests/golden/, written by oracle/gen_golden.py from the reference's own modules): small, well-conditioned
graphs where 1e-4 relative is achievable, so a localised backward bug that the whole-model outlier band
(test_model_gpu.py::test_model_vs_oracle) would absorb fails here.

  g1  BasicBlock / Bottleneck / ChainOfBasicBlocks / conv_bn_relu: y, gx, per-parameter gradient sums in train mode,
      the eval-mode forward on the updated statistics, running statistics afterwards       (posetimation/layers/basic_model.py:25-148, basic_layer.py:13-73)
  g2  HighResolutionModule with 2 / 3 / 4 branches, multi- and single-scale output   (backbones/hrnet.py:17-172)
  g4  HRNetPlus-W48 384x288 heatmaps / features / argmax       (backbones/hrnet.py:521-690)
  g8  the two MI estimators, values and gradients              (zoo/Alignment/Alignment_V15.py:250-277)
Every run uses the library's DEFAULT kernel routes (no tune override).
"""
import os
import warnings

import numpy as np
import pytest
import torch

import fami_pose_amd as fp
from fami_pose_amd import modules as M
from oracle import model as om

pytestmark = pytest.mark.gpu
warnings.filterwarnings('ignore')
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-4


def gold(name):
    return np.load(os.path.join(GOLD, name))


def relerr(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max()) / max(float(np.abs(b).max()), 1e-30)


def nhwc(t, dev):
    return t.permute(0, 2, 3, 1).contiguous().to(dev)


def nchw(t):
    return t.permute(0, 3, 1, 2).float().cpu()


G1 = {
    'basic': (lambda: om.Basic(16, 16), lambda: M.BasicBlock(16, 16)),
    'neck': (lambda: om.Neck(32, 8), lambda: M.Bottleneck(32, 8)),
    'neck_ds': (lambda: om.Neck(16, 8, 1, om._proj(16, 32)), lambda: M.Bottleneck(16, 8, 1, M._shortcut(16, 32))),
    'chain': (lambda: om.BlockChain(24, 16, 2), lambda: M.ChainOfBasicBlocks(24, 16, 2)),
    'cbr': (lambda: om.ConvUnit(16, 16, 3, 2, 1, 1), lambda: M.conv_bn_relu(16, 16, 3, 2, 1, 1)),
    'cbr_dil': (lambda: om.ConvUnit(12, 20, 3, 1, 3, 3, bn=False, relu=False),
                lambda: M.conv_bn_relu(12, 20, 3, 1, 3, 3, has_bn=False, has_relu=False)),
}


@pytest.mark.parametrize('idx,name', list(enumerate(G1)))
def test_g1_blocks_on_the_hip_path(dev, idx, name):
    from fami_pose_amd.engine import Engine, T
    g = gold('g1_blocks.npz')
    orc = om.realistic_init_(G1[name][0](), 100 + idx)        # the generator's seeded init (== the reference module's)
    inner = G1[name][1]()
    inner.load_state_dict(orc.state_dict())
    inner = inner.to(dev)
    x0 = torch.from_numpy(g[name + '.x'])
    names = sorted(k for k, _ in orc.named_parameters())
    has_bn = any(isinstance(m, torch.nn.BatchNorm2d) for m in inner.modules())
    for mode in ('train', 'eval'):                            # same order as the generator: eval sees the updated statistics
        inner.train(mode == 'train')
        # backward through an eval-mode BatchNorm is outside the training hot path (the engine raises): forward only there
        bwd = mode == 'train' or not has_bn
        eng = Engine(dev, record=bwd)
        xt = T(nhwc(x0, dev), bwd)
        y = inner.run(eng, xt)
        assert relerr(nchw(y.data), g['%s.%s.y' % (name, mode)]) < TOL, (name, mode, 'y')
        if not bwd:
            continue
        gy = torch.randn(nchw(y.data).shape, generator=torch.Generator().manual_seed(300 + idx))
        y.grad = nhwc(gy, dev)
        eng.backward()
        torch.cuda.synchronize(dev)
        assert relerr(nchw(xt.grad), g['%s.%s.gx' % (name, mode)]) < TOL, (name, mode, 'gx')
        p = dict(inner.named_parameters())
        gabs = np.array([eng.param_grads[id(p[k])].double().abs().sum().item() for k in names])
        gsum = np.array([eng.param_grads[id(p[k])].double().sum().item() for k in names])
        ref_abs = g['%s.%s.gabs' % (name, mode)]
        assert np.allclose(gabs, ref_abs, rtol=TOL, atol=TOL * np.abs(ref_abs).max()), (name, mode, 'gabs', gabs, ref_abs)
        assert np.allclose(gsum, g['%s.%s.gsum' % (name, mode)], rtol=1e-3, atol=2e-4 * np.abs(ref_abs).max()), \
            (name, mode, 'gsum')
    sd = inner.state_dict()
    for k in sd:
        if 'running' in k:
            assert relerr(sd[k], g['%s.after.%s' % (name, k)]) < 1e-5, (name, k)


@pytest.mark.parametrize('nb', [2, 3, 4])
@pytest.mark.parametrize('mso', [True, False])
def test_g2_hrmodule_on_the_hip_path(dev, nb, mso):
    from fami_pose_amd.engine import Engine, T
    g = gold('g2_hrmodule.npz')
    tag = 'nb%d_%s' % (nb, 'multi' if mso else 'single')
    ch = [8 * 2 ** b for b in range(nb)]
    orc = om.realistic_init_(om.HRModule(ch, [1] * nb, mso), 400 + nb)
    inner = M.HighResolutionModule(ch, [1] * nb, mso)
    inner.load_state_dict(orc.state_dict())
    inner = inner.to(dev).train()
    eng = Engine(dev, record=False)
    ys = inner.run(eng, [T(nhwc(torch.from_numpy(g['%s.x%d' % (tag, b)]), dev), False) for b in range(nb)])
    torch.cuda.synchronize(dev)
    assert len(ys) == (nb if mso else 1)
    for b, y in enumerate(ys):
        assert relerr(nchw(y.data), g['%s.y%d' % (tag, b)]) < TOL, (tag, b)


def test_g4_hrnetplus_w48_on_the_hip_path(dev):
    g = gold('g4_hrnetplus_w48.npz')
    orc = om.realistic_init_(om.HRNetOracle(om.make_cfg(48), plus=True), 48)
    net = fp.HRNetPlus(fp.default_cfg(48, image_size=(288, 384)), True)
    net.load_state_dict(orc.state_dict())
    net = net.to(dev).train()
    x = torch.randn(2, 3, 384, 288, generator=torch.Generator().manual_seed(int(g['x_seed'])))
    with torch.no_grad():
        hm, feats = net(x.to(dev))
    hm = hm.cpu()
    assert len(feats) == 1
    # 293 convolutions + train-mode BatchNorm deep: 1e-3 of the map's maximum (north_star's heatmap tolerance), indices exact
    assert relerr(hm[:, 0], g['hm_j0']) < 1e-3 and relerr(hm[:, 9], g['hm_j9']) < 1e-3
    assert relerr(feats[0][:, 5], g['feat0_c5']) < 1e-3
    flat = hm.reshape(2, hm.shape[1], -1)
    assert np.array_equal(flat.argmax(2).numpy(), g['argmax'])          # bit-exact keypoint indices
    assert relerr(flat.max(2).values, g['maxval']) < 1e-3
    assert relerr(hm.double().abs().sum((2, 3)), g['abssum']) < TOL
    assert abs(feats[0].double().abs().sum().item() - float(g['feat0_abssum'])) <= TOL * float(g['feat0_abssum'])


def test_g8_mi_estimators_on_the_hip_path(dev):
    from fami_pose_amd.engine import Engine, T
    g = gold('g8_mi.npz')
    gen = torch.Generator().manual_seed(int(g['seed']))
    feat = torch.randn(2, 48, 96, 72, generator=gen) * 0.5
    f2 = torch.randn(2, 48, 96, 72, generator=gen) * 0.5
    yy = torch.rand(2, 17, 96, 72, generator=gen) * 0.8
    orc = om.realistic_init_(om.AlignmentOracle(om.make_cfg(48), True, 4, (384, 288)), 15)
    fl = orc.hrnet.final_layer
    eng = Engine(dev, record=True)
    # feat_label_mi_estimation: A = final_layer(feat).detach(), Bt = the heatmap ; feat_feat: A = F1.detach(), Bt = F2
    a = eng.conv(T(nhwc(feat, dev), False), fl.weight.detach().to(dev), fl.bias.detach().to(dev), 1, fl.padding[0], 1,
                 out_f32=True)
    yt, f2t = T(nhwc(yy, dev), True), T(nhwc(f2, dev), True)
    m1, seed1 = eng.softmax_kl(eng.to_nchw(a), yt, 0.05)
    m2, seed2 = eng.softmax_kl(feat.to(dev), f2t, 0.05)
    assert m1.item() == pytest.approx(float(g['feat_label']), rel=TOL, abs=1e-10)
    assert m2.item() == pytest.approx(float(g['feat_feat']), rel=TOL, abs=1e-10)
    seed1(3.0)
    seed2(-2.0)
    gy, gf2 = nchw(yt.grad), nchw(f2t.grad)
    assert torch.isfinite(gy).all() and torch.isfinite(gf2).all()
    if bool(g['gy_finite']):
        assert relerr(gy[:, 3], g['gy_c3']) < 2 * TOL
        assert gy.double().abs().sum().item() == pytest.approx(float(g['gy_abssum']), rel=2 * TOL)
    if bool(g['gf2_finite']):
        assert relerr(gf2[:, 7], g['gf2_c7']) < 2 * TOL
        assert gf2.double().abs().sum().item() == pytest.approx(float(g['gf2_abssum']), rel=2 * TOL)

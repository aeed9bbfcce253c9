"""The oracle (CPU restatement) against the golden vectors generated from the REFERENCE's own modules
(oracle/gen_golden.py, run in the build container; fixtures under tests/golden/).  CPU only.

Every fixture stores inputs (or their seed), the seed of realistic_init_ and the reference's outputs; the
generator asserted that the seeded init gives identical state_dicts on the reference and oracle modules.
The restatement is the same sequence of torch fp32 ops, so most comparisons are exact; the tolerance 2e-6
(relative to the tensor's max) only absorbs thread-count dependent summation order inside MKL-DNN.
"""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import model as om, ops as oops

warnings.filterwarnings('ignore')
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 2e-6


def gold(name):
    return np.load(os.path.join(GOLD, name))


def close(a, b, tol=TOL):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(float(np.abs(b).max()), 1e-30)
    err = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) / scale
    assert err <= tol, 'relative error %.3e > %.1e' % (err, tol)


def _fwd_bwd(mod, x, gy_seed, train):
    mod.train(train)
    x = x.clone().requires_grad_(True)
    y = mod(x)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(gy_seed))
    mod.zero_grad()
    y.backward(gy)
    names = sorted(k for k, _ in mod.named_parameters())
    p = dict(mod.named_parameters())
    return y, x.grad, np.array([p[k].grad.double().sum().item() for k in names]), \
        np.array([p[k].grad.double().abs().sum().item() for k in names])


# ------------------------------------------------------------------ G1
G1 = {
    'basic': lambda: om.Basic(16, 16),
    'neck': lambda: om.Neck(32, 8),
    'neck_ds': lambda: om.Neck(16, 8, 1, om._proj(16, 32)),
    'chain': lambda: om.BlockChain(24, 16, 2),
    'cbr': lambda: om.ConvUnit(16, 16, 3, 2, 1, 1),
    'cbr_dil': lambda: om.ConvUnit(12, 20, 3, 1, 3, 3, bn=False, relu=False),
}


@pytest.mark.parametrize('idx,name', list(enumerate(G1)))
def test_g1_blocks(idx, name):
    g = gold('g1_blocks.npz')
    mod = om.realistic_init_(G1[name](), 100 + idx)
    x = torch.from_numpy(g[name + '.x'])
    for mode in ('train', 'eval'):
        y, gx, gsum, gabs = _fwd_bwd(mod, x, 300 + idx, mode == 'train')
        close(y, g['%s.%s.y' % (name, mode)])
        close(gx, g['%s.%s.gx' % (name, mode)], 1e-5)
        close(gabs, g['%s.%s.gabs' % (name, mode)], 1e-5)
        assert np.allclose(gsum, g['%s.%s.gsum' % (name, mode)], rtol=1e-4, atol=1e-4 * np.abs(gabs).max())
    sd = mod.state_dict()
    for k in sd:
        if 'running' in k:
            close(sd[k], g['%s.after.%s' % (name, k)])


# ------------------------------------------------------------------ G2
@pytest.mark.parametrize('nb', [2, 3, 4])
@pytest.mark.parametrize('mso', [True, False])
def test_g2_hrmodule(nb, mso):
    g = gold('g2_hrmodule.npz')
    tag = 'nb%d_%s' % (nb, 'multi' if mso else 'single')
    ch = [8 * 2 ** b for b in range(nb)]
    mod = om.realistic_init_(om.HRModule(ch, [1] * nb, mso), 400 + nb).train()
    ys = mod([torch.from_numpy(g['%s.x%d' % (tag, b)]) for b in range(nb)])
    assert len(ys) == (nb if mso else 1)
    for b, y in enumerate(ys):
        close(y, g['%s.y%d' % (tag, b)])


# ------------------------------------------------------------------ G3 / G4
def _check_hm(hm, g, tol=TOL):
    B, J = hm.shape[:2]
    flat = hm.reshape(B, J, -1)
    assert np.array_equal(flat.argmax(2).numpy(), g['argmax'])          # bit-exact indices
    close(flat.max(2).values, g['maxval'], tol)
    close(hm.double().sum((2, 3)), g['sum'], 1e-5)
    close(hm.double().abs().sum((2, 3)), g['abssum'], 1e-5)


def test_g3_hrnet_w32_config1():
    """BASELINE.json configs[0]: HRNet-W32 256x192 single-frame heatmap forward + MSE on CPU."""
    g = gold('g3_hrnet_w32.npz')
    net = om.realistic_init_(om.HRNetOracle(om.make_cfg(32), plus=False), 32).eval()
    x = torch.randn(1, 3, 256, 192, generator=torch.Generator().manual_seed(int(g['x_seed'])))
    with torch.no_grad():
        hm, feats = net(x)
    close(hm, g['hm'])
    _check_hm(hm, g)
    gen = torch.Generator().manual_seed(int(g['tw_seed']))
    tgt = torch.rand(1, 17, 64, 48, generator=gen)
    w = (torch.rand(1, 17, 1, generator=gen) < 0.8).float()
    assert abs(oops.joint_mse(hm, tgt, w).item() - float(g['mse'])) <= 1e-6 * abs(float(g['mse']))
    for i, f in enumerate(feats):
        assert abs(f.double().abs().sum().item() - float(g['feat%d_abssum' % i])) <= 1e-5 * float(g['feat%d_abssum' % i])


def test_g4_hrnetplus_w48():
    g = gold('g4_hrnetplus_w48.npz')
    net = om.realistic_init_(om.HRNetOracle(om.make_cfg(48), plus=True), 48).train()
    x = torch.randn(2, 3, 384, 288, generator=torch.Generator().manual_seed(int(g['x_seed'])))
    with torch.no_grad():
        hm, feats = net(x)
    close(hm[:, 0], g['hm_j0'])
    close(hm[:, 9], g['hm_j9'])
    close(feats[0][:, 5], g['feat0_c5'])
    _check_hm(hm, g)
    assert len(feats) == 1          # last stage-4 module fuses to branch 0 only
    assert abs(feats[0].double().abs().sum().item() - float(g['feat0_abssum'])) <= 1e-5 * float(g['feat0_abssum'])


# ------------------------------------------------------------------ G5 / G6 / G7
@pytest.mark.parametrize('name', ['b4', 'b1'])
def test_g5_mse(name):
    g = gold('g5_mse.npz')
    p, t, w = (torch.from_numpy(g[name + k]) for k in ('.pred', '.gt', '.w'))
    assert abs(oops.joint_mse(p, t, w).item() - float(g[name + '.loss'])) < 1e-6
    assert abs(oops.joint_mse(p, t, w, True, False).item() - float(g[name + '.loss_nodiv'])) < 2e-5
    assert abs(oops.joint_mse(p, t, w, False, True).item() - float(g[name + '.loss_nowt'])) < 1e-6
    pp = p.clone().requires_grad_(True)
    oops.joint_mse(pp, t, w).backward()
    close(pp.grad, g[name + '.gpred'], 1e-5)


@pytest.mark.parametrize('sigma', [3, 2])
def test_g6_targets(sigma):
    g = gold('g6_targets.npz')
    t, w = oops.generate_heatmaps(g['joints'], g['vis'], sigma, np.array([288, 384]), np.array([72, 96]), 17)
    assert np.array_equal(t, g['s%d.target' % sigma])          # exact
    assert np.array_equal(w, g['s%d.weight' % sigma])
    assert w[4, 0] == 0 and w[5, 0] == 0 and w[11, 0] == 0.5 and t[11].max() == 0 and t[10].max() == 0


def test_g7_decode():
    g = gold('g7_decode.npz')
    preds, maxvals = oops.get_max_preds(g['out'])
    assert np.array_equal(preds, g['preds']) and np.array_equal(maxvals, g['maxvals'])
    assert tuple(preds[0, 0]) == (0, 0) and tuple(preds[0, 1]) == (0, 0) and tuple(preds[0, 2]) == (7, 5)
    acc, avg, cnt, pred = oops.accuracy(g['out'], g['tgt'])
    assert np.array_equal(acc, g['acc']) and avg == float(g['avg']) and cnt == int(g['cnt'])
    assert np.array_equal(pred, g['pred'])
    assert np.array_equal(oops.argmax_indices(g['out']), g['out'].reshape(3, 17, -1).argmax(2))


# ------------------------------------------------------------------ G8 / G9 / G10 / G11
@pytest.fixture(scope='module')
def oracle_v15():
    return om.realistic_init_(om.AlignmentOracle(om.make_cfg(48), True, 4, (384, 288)), 15)


def test_g8_mi(oracle_v15):
    g = gold('g8_mi.npz')
    gen = torch.Generator().manual_seed(int(g['seed']))
    feat = torch.randn(2, 48, 96, 72, generator=gen) * 0.5
    f2 = (torch.randn(2, 48, 96, 72, generator=gen) * 0.5).requires_grad_(True)
    yy = (torch.rand(2, 17, 96, 72, generator=gen) * 0.8).requires_grad_(True)
    fl = oracle_v15.hrnet.final_layer
    m1 = oops.feat_label_mi(feat, yy, fl.weight, fl.bias)
    m2 = oops.feat_feat_mi(feat, f2)
    assert m1.item() == pytest.approx(float(g['feat_label']), rel=1e-6, abs=1e-12)
    assert m2.item() == pytest.approx(float(g['feat_feat']), rel=1e-6, abs=1e-12)
    (3.0 * m1 - 2.0 * m2).backward()
    if bool(g['gy_finite']):
        close(yy.grad[:, 3], g['gy_c3'], 1e-4)
        assert yy.grad.double().abs().sum().item() == pytest.approx(float(g['gy_abssum']), rel=1e-4)
    if bool(g['gf2_finite']):
        close(f2.grad[:, 7], g['gf2_c7'], 1e-4)
        assert f2.grad.double().abs().sum().item() == pytest.approx(float(g['gf2_abssum']), rel=1e-4)
    assert torch.isfinite(yy.grad).all() and torch.isfinite(f2.grad).all()


def test_g9_alignment_v15_train_and_eval(oracle_v15):
    g = gold('g9_alignment_v15.npz')
    gen = torch.Generator().manual_seed(int(g['seed']))
    kf = torch.randn(1, 3, 384, 288, generator=gen)
    sup = torch.randn(1, 12, 384, 288, generator=gen)
    tgt = torch.rand(1, 17, 96, 72, generator=gen)
    w = (torch.rand(1, 17, 1, generator=gen) < 0.8).float()
    m = oracle_v15.train()
    final, kf_hm, mi = m(kf, sup)
    close(final, g['final'], 1e-5)
    close(kf_hm, g['kf_hm'], 1e-5)
    assert np.array_equal(final.reshape(1, 17, -1).argmax(2).numpy(), g['final_argmax'])
    assert np.array_equal(kf_hm.reshape(1, 17, -1).argmax(2).numpy(), g['kf_argmax'])
    assert np.allclose([x.item() for x in mi], g['mi'], rtol=1e-5, atol=1e-10)
    loss = oops.total_loss(final, tgt, w, mi)
    assert loss.item() == pytest.approx(float(g['loss']), rel=1e-5)
    m.zero_grad()
    loss.backward()
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert len(grads) == int(g['n_params_with_grad'])
    close(grads['agg_final_layer.weight'], g['grad.agg_final_layer.weight'], 1e-4)
    close(grads['dcn_1.bias'], g['grad.dcn_1.bias'], 1e-4)
    for key in g.files:
        if key.startswith('grad.') and key.endswith('.abssum'):
            name = key[5:-7]
            assert grads[name].double().abs().sum().item() == pytest.approx(float(g[key]), rel=2e-4), name
    sd = m.state_dict()
    close(sd['hrnet.bn1.running_mean'], g['after.hrnet.bn1.running_mean'], 1e-5)
    close(sd['sup_agg_block.layers.0.bn1.running_var'], g['after.sup_agg_block.layers.0.bn1.running_var'], 1e-5)

    val = om.realistic_init_(om.AlignmentOracle(om.make_cfg(48), False, 4, (384, 288)), 15).eval()
    with torch.no_grad():
        res = val(kf, sup)
    assert len(res) == 2
    close(res[0], g['eval.final'], 1e-5)
    close(res[1], g['eval.kf_hm'], 1e-5)
    assert np.array_equal(res[0].reshape(1, 17, -1).argmax(2).numpy(), g['eval.final_argmax'])


def test_g10_state_dict_keys(oracle_v15):
    want = [ln.split() for ln in open(os.path.join(GOLD, 'g10_state_dict_keys.txt')) if ln.strip()]
    sd = oracle_v15.state_dict()
    assert [k for k, _, _ in want] == list(sd.keys())
    for k, shp, dt in want:
        assert ('x'.join(map(str, sd[k].shape)) or 'scalar') == shp and str(sd[k].dtype) == 'torch.' + dt, k
    assert len(want) == 1929


# ------------------------------------------------------------------ un-pinned third-party ops: two independent formulations
def test_dcn_gather_vs_gridsample():
    """torchvision deform_conv2d semantics: the gather restatement against an independent F.grid_sample one."""
    gen = torch.Generator().manual_seed(11)
    B, C, H, W, G = 2, 24, 13, 10, 6
    x = torch.randn(B, C, H, W, generator=gen)
    off = torch.randn(B, 18 * G, H, W, generator=gen) * 2.5      # reaches well outside the map
    msk = torch.randn(B, 9 * G, H, W, generator=gen)
    wt = torch.randn(20, C, 3, 3, generator=gen) * 0.1
    bias = torch.randn(20, generator=gen)
    a = oops.deform_conv2d(x, off, msk, wt, bias, 1, 3, 3)
    b = oops.deform_conv2d_gridsample(x, off, msk, wt, bias, 1, 3, 3)
    close(a, b, 1e-5)
    # zero offsets + unit mask == ordinary dilated convolution
    c = oops.deform_conv2d(x, torch.zeros_like(off), torch.ones_like(msk), wt, bias, 1, 3, 3)
    close(c, torch.nn.functional.conv2d(x, wt, bias, 1, 3, 3), 1e-5)


def test_warp_translate_vs_gridsample_and_integer_shift():
    gen = torch.Generator().manual_seed(12)
    src = torch.randn(3, 8, 12, 9, generator=gen)
    t = torch.tensor([[0.3, -1.7], [4.25, 2.5], [-20.0, 0.0]])
    close(oops.warp_translate(src, t), oops.warp_translate_gridsample(src, t), 1e-5)
    ti = torch.tensor([[2.0, 1.0]] * 3)       # integer translation: out[y,x] = src[y-1, x-2], zero filled
    out = oops.warp_translate(src, ti)
    assert torch.equal(out[:, :, 1:, 2:], src[:, :, :-1, :-2]) and out[:, :, 0].abs().max() == 0
    M = torch.tensor([[[1.0, 0.0, 2.0], [0.0, 1.0, 1.0]]]).repeat(3, 1, 1)
    assert torch.equal(oops.warp_affine_like(src, M, (12, 9)), out)


def test_warp_translate_legacy_align_corners_false():
    """MODEL.WARP_ALIGN_CORNERS False = kornia <= 0.4's default at Alignment_V15.py:135: the restatement (translation
    scaled by W/(W-1), H/(H-1)) against that release's own pipeline spelled with torch (normalize_homography for
    [0, W-1], inverse, affine_grid + grid_sample with align_corners=False), values and both gradients."""
    gen = torch.Generator().manual_seed(13)
    src = torch.randn(3, 8, 12, 9, generator=gen, dtype=torch.float64, requires_grad=True)
    t = torch.tensor([[0.3, -1.7], [4.25, 2.5], [-3.6, 0.4]], dtype=torch.float64, requires_grad=True)
    a = oops.warp_translate(src, t, align_corners=False)
    b = oops.warp_translate_legacy_gridsample(src, t)
    close(a.detach(), b.detach(), 1e-9)
    g = torch.randn(a.shape, generator=gen, dtype=torch.float64)
    ga = torch.autograd.grad((a * g).sum(), [src, t])
    gb = torch.autograd.grad((b * g).sum(), [src, t])
    close(ga[0], gb[0], 1e-9)
    close(ga[1], gb[1], 1e-8)
    # and it differs from the pixel-exact form by exactly the scaled shift
    c = oops.warp_translate(src, t * torch.tensor([9 / 8, 12 / 11], dtype=torch.float64))
    close(a.detach(), c.detach(), 1e-12)


def test_g12_final_preds():
    """get_final_preds: argmax + quarter-pixel shift + inverse affine (heatmaps_process.py:47-73)."""
    g = gold('g12_final_preds.npz')
    preds, maxvals = oops.get_final_preds(g['hm'].copy(), g['center'], g['scale'])
    assert np.array_equal(maxvals, g['maxvals'])
    assert np.abs(preds - g['preds']).max() < 1e-9
    # closed form of the rot-0 inverse affine: uniform scale 200*scale[0]/W about the heatmap centre
    c, s = g['center'][0], g['scale'][0]
    want = c + (np.array([5.0, 0.0]) - np.array([36.0, 48.0])) * (200.0 * s[0] / 72)
    assert np.abs(preds[0, 1] - want).max() < 1e-4
    assert np.abs(preds[0, 0] - (c - np.array([36.0, 48.0]) * (200.0 * s[0] / 72))).max() < 1e-4


# ------------------------------------------------------------------ input pipeline
def test_input_pipeline_transforms_match_reference_golden():
    """dark_get_affine_transform / exec_affine_transform / fliplr_joints against vectors produced by the reference's own
    datasets/process modules (oracle/gen_golden.py::g13_input_pipeline)."""
    from oracle import ops as O
    g = np.load(os.path.join(GOLD, 'g13_input_pipeline.npz'))
    for b in range(4):
        t = O.dark_get_affine_transform(g['centers'][b], g['scales'][b], float(g['rots'][b]), g['image_size'])
        ti = O.dark_get_affine_transform(g['centers'][b], g['scales'][b], float(g['rots'][b]), g['image_size'], inv=1)
        assert np.array_equal(t, g['trans'][b]) and np.array_equal(ti, g['trans_inv'][b])
        for j in range(17):
            assert np.array_equal(O.exec_affine_transform(g['joints'][b, j, 0:2], t), g['pts'][b, j])
        fj, fv = O.fliplr_joints(g['joints'][b], g['vis'][b], 1280, [[3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]])
        assert np.array_equal(fj, g['flip_joints'][b]) and np.array_equal(fv, g['flip_vis'][b])
        # the inverse transform really inverts (cv2_invert_affine is what warpAffine applies)
        mi = O.cv2_invert_affine(t)
        p = np.array([100.0, 50.0, 1.0])
        assert np.allclose(mi @ np.append(t @ p, 1.0), p[:2], atol=1e-8)


def test_warp_affine_restatement_vs_float_bilinear():
    """PARITY UNPINNED for cv2.warpAffine (third-party, absent): the fixed-point restatement is held against an
    independent float64 bilinear resampling of the same inverse map -- sub-pixel positions are quantised to 1/32 pixel
    and the result is rounded to 8 bits, so the two agree within a few grey levels everywhere (max |d| bounded by the
    local gradient x 1/32 px + 0.5) and exactly for an integer translation."""
    from oracle import ops as O
    rng = np.random.RandomState(3)
    src = (rng.rand(60, 80, 3) * 255).astype(np.uint8)
    src = np.round(np.stack([np.convolve(np.convolve(src[..., c].astype(np.float64).ravel(), np.ones(5) / 5, 'same')
                                         .reshape(60, 80).T.ravel(), np.ones(5) / 5, 'same').reshape(80, 60).T for c in range(3)], -1)).astype(np.uint8)
    M = O.dark_get_affine_transform([40.0, 30.0], [0.3, 0.4], 20.0, [48, 64])
    out = O.cv2_warp_affine_u8(src, M, (48, 64))
    Mi = O.cv2_invert_affine(M)
    ys, xs = np.mgrid[0:64, 0:48].astype(np.float64)
    sx = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]
    sy = Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = sx - x0, sy - y0

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < 60) & (xx >= 0) & (xx < 80)
        return src[np.clip(yy, 0, 59), np.clip(xx, 0, 79)].astype(np.float64) * ok[..., None]
    ref = (tap(y0, x0) * ((1 - fy) * (1 - fx))[..., None] + tap(y0, x0 + 1) * ((1 - fy) * fx)[..., None] +
           tap(y0 + 1, x0) * (fy * (1 - fx))[..., None] + tap(y0 + 1, x0 + 1) * (fy * fx)[..., None])
    d = np.abs(out.astype(np.float64) - ref)
    assert d.max() < 6.0 and d.mean() < 0.6, (d.max(), d.mean())
    # integer translation: pure copy with zero border
    T = np.array([[1.0, 0.0, 3.0], [0.0, 1.0, -2.0]])
    o2 = O.cv2_warp_affine_u8(src, T, (80, 60))
    assert np.array_equal(o2[0:58, 3:80], src[2:60, 0:77]) and o2[58:].max() == 0 and o2[:, :3].max() == 0
    # flip = warp of the mirrored image
    assert np.array_equal(O.cv2_warp_affine_u8(src, M, (48, 64), flip=True), O.cv2_warp_affine_u8(src[:, ::-1].copy(), M, (48, 64)))
    t = O.to_tensor_normalize(out, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    assert t.shape == (3, 64, 48) and t.dtype == torch.float32
    assert float(t[0, 5, 7]) == pytest.approx((out[5, 7, 0] / 255.0 - 0.485) / 0.229, rel=1e-6)

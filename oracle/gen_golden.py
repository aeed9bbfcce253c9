#!/usr/bin/env python
"""Oracle (test infrastructure, BUILD CONTAINER ONLY): generate the golden vectors of
tests/golden/ by running the REFERENCE's own modules (imported from /root/reference through
oracle/ref_harness.py) on seeded inputs.

    python -m oracle.gen_golden            # writes tests/golden/*.npz|*.txt

What a fixture holds is data only: inputs (or the seed they are drawn from), the seed of the
weight initialisation (oracle.model.realistic_init_, applied to the reference module) and the
reference's outputs.  tests/test_oracle.py rebuilds the same weights on the oracle's restatement
and must reproduce the outputs; that pins the restatement to the reference.
For every fixture this script first asserts that the seeded init yields identical state_dicts on
the reference module and on the oracle module (same keys, same values), so "seed" is a faithful
stand-in for "weights".

The two third-party ops (torchvision DeformConv2d, kornia warp_affine) are not importable here:
the reference class runs with oracle/ops.py injected for them -- parity there is UNPINNED
(oracle/__init__.py).
"""
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import model as om, ops as oops, ref_harness as rh  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
warnings.filterwarnings('ignore')


def _np(t):
    return t.detach().cpu().numpy()


def _same_state(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys()), 'state_dict keys differ'
    for k in sa:
        assert sa[k].shape == sb[k].shape and torch.equal(sa[k], sb[k]), k


def _save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024))


def _fwd_bwd(mod, x, gy_seed, train):
    mod.train(train)
    x = x.clone().requires_grad_(True)
    y = mod(x)
    g = torch.Generator().manual_seed(gy_seed)
    gy = torch.randn(y.shape, generator=g)
    mod.zero_grad()
    y.backward(gy)
    gsum = {k: p.grad.double().sum().item() for k, p in mod.named_parameters()}
    gabs = {k: p.grad.double().abs().sum().item() for k, p in mod.named_parameters()}
    return y, x.grad, gsum, gabs


# ------------------------------------------------------------------ G1 blocks
def g1_blocks(ns):
    cases = {
        'basic': (lambda: ns.BasicBlock(16, 16), lambda: om.Basic(16, 16), (2, 16, 12, 9)),
        'neck': (lambda: ns.Bottleneck(32, 8), lambda: om.Neck(32, 8), (2, 32, 12, 9)),
        'neck_ds': (lambda: ns.Bottleneck(16, 8, 1, nn.Sequential(nn.Conv2d(16, 32, 1, bias=False),
                                                                 nn.BatchNorm2d(32, momentum=0.1))),
                    lambda: om.Neck(16, 8, 1, om._proj(16, 32)), (2, 16, 8, 8)),
        'chain': (lambda: ns.ChainOfBasicBlocks(24, 16, num_blocks=2), lambda: om.BlockChain(24, 16, 2),
                  (2, 24, 24, 18)),
        'cbr': (lambda: ns.conv_bn_relu(16, 16, 3, 2, 1, 1), lambda: om.ConvUnit(16, 16, 3, 2, 1, 1), (3, 16, 13, 9)),
        'cbr_dil': (lambda: ns.conv_bn_relu(12, 20, 3, 1, 3, 3, has_bn=False, has_relu=False),
                    lambda: om.ConvUnit(12, 20, 3, 1, 3, 3, bn=False, relu=False), (2, 12, 16, 12)),
    }
    out = {}
    for i, (name, (mk_ref, mk_orc, shp)) in enumerate(cases.items()):
        ref, orc = mk_ref(), mk_orc()
        om.realistic_init_(ref, 100 + i)
        om.realistic_init_(orc, 100 + i)
        _same_state(ref, orc)
        x = torch.randn(shp, generator=torch.Generator().manual_seed(200 + i))
        out[name + '.x'] = _np(x)
        for mode in ('train', 'eval'):
            y, gx, gsum, gabs = _fwd_bwd(ref, x, 300 + i, mode == 'train')
            out['%s.%s.y' % (name, mode)] = _np(y)
            out['%s.%s.gx' % (name, mode)] = _np(gx)
            out['%s.%s.gsum' % (name, mode)] = np.array([gsum[k] for k in sorted(gsum)])
            out['%s.%s.gabs' % (name, mode)] = np.array([gabs[k] for k in sorted(gabs)])
        sd = ref.state_dict()      # running stats after ONE train-mode forward (momentum 0.1)
        for k in sd:
            if 'running' in k:
                out['%s.after.%s' % (name, k)] = _np(sd[k])
    _save('g1_blocks.npz', **out)


# ------------------------------------------------------------------ G2 HighResolutionModule
def g2_hrmodule(ns):
    out = {}
    for nb in (2, 3, 4):
        ch = [8 * 2 ** b for b in range(nb)]
        for mso in (True, False):
            tag = 'nb%d_%s' % (nb, 'multi' if mso else 'single')
            ref = ns.HighResolutionModule(nb, ns.BasicBlock, [1] * nb, list(ch), list(ch), 'SUM', mso)
            orc = om.HRModule(ch, [1] * nb, mso)
            om.realistic_init_(ref, 400 + nb)
            om.realistic_init_(orc, 400 + nb)
            _same_state(ref, orc)
            g = torch.Generator().manual_seed(500 + nb)
            xs = [torch.randn(2, ch[b], 16 >> b, 16 >> b, generator=g) for b in range(nb)]
            ref.train()
            ys = ref([x.clone() for x in xs])
            for b, x in enumerate(xs):
                out['%s.x%d' % (tag, b)] = _np(x)
            for b, y in enumerate(ys):
                out['%s.y%d' % (tag, b)] = _np(y)
    _save('g2_hrmodule.npz', **out)


# ------------------------------------------------------------------ G3 HRNet-W32 (BASELINE config 1) / G4 HRNetPlus-W48
def _hm_summary(hm):
    B, J = hm.shape[:2]
    flat = hm.reshape(B, J, -1)
    return dict(argmax=flat.argmax(2).numpy().astype(np.int64), maxval=_np(flat.max(2).values),
                sum=_np(hm.double().sum((2, 3))), abssum=_np(hm.double().abs().sum((2, 3))))


def g3_hrnet_w32(ns):
    cfg = rh.ref_cfg(32)
    ref = ns.HRNet(cfg, True)
    orc = om.HRNetOracle(om.make_cfg(32), plus=False)
    om.realistic_init_(ref, 32)
    om.realistic_init_(orc, 32)
    _same_state(ref, orc)
    x = torch.randn(1, 3, 256, 192, generator=torch.Generator().manual_seed(3200))
    ref.eval()
    with torch.no_grad():
        hm, feats = ref(x)
    g = torch.Generator().manual_seed(3201)
    tgt = torch.rand(1, 17, 64, 48, generator=g)
    w = (torch.rand(1, 17, 1, generator=g) < 0.8).float()
    mse = ns.JointMSELoss()(hm, tgt, w)
    out = dict(hm=_np(hm), mse=np.array(mse.item()), x_seed=np.array(3200), tw_seed=np.array(3201))
    for i, f in enumerate(feats):
        out['feat%d_sum' % i] = np.array(f.double().sum().item())
        out['feat%d_abssum' % i] = np.array(f.double().abs().sum().item())
    out.update(_hm_summary(hm))
    _save('g3_hrnet_w32.npz', **out)


def g4_hrnetplus_w48(ns):
    cfg = rh.ref_cfg(48)
    ref = ns.HRNetPlus(cfg, True)
    orc = om.HRNetOracle(om.make_cfg(48), plus=True)
    om.realistic_init_(ref, 48)
    om.realistic_init_(orc, 48)
    _same_state(ref, orc)
    x = torch.randn(2, 3, 384, 288, generator=torch.Generator().manual_seed(4800))
    ref.train()
    with torch.no_grad():
        hm, feats = ref(x)
    out = dict(x_seed=np.array(4800), hm_j0=_np(hm[:, 0]), hm_j9=_np(hm[:, 9]), feat0_c5=_np(feats[0][:, 5]))
    for i, f in enumerate(feats):
        out['feat%d_sum' % i] = np.array(f.double().sum().item())
        out['feat%d_abssum' % i] = np.array(f.double().abs().sum().item())
    out.update(_hm_summary(hm))
    _save('g4_hrnetplus_w48.npz', **out)


# ------------------------------------------------------------------ G5 MSE / G6 targets / G7 decode
def g5_mse(ns):
    out = {}
    g = torch.Generator().manual_seed(5)
    for name, B in (('b4', 4), ('b1', 1)):          # b1 exercises the squeeze() path of mse_loss.py:29-33
        p, t = torch.randn(B, 17, 12, 9, generator=g), torch.rand(B, 17, 12, 9, generator=g)
        w = (torch.rand(B, 17, 1, generator=g) < 0.7).float()
        out[name + '.pred'], out[name + '.gt'], out[name + '.w'] = _np(p), _np(t), _np(w)
        out[name + '.loss'] = np.array(ns.JointMSELoss()(p, t, w).item())
        out[name + '.loss_nodiv'] = np.array(ns.JointMSELoss(True, False)(p, t, w).item())
        out[name + '.loss_nowt'] = np.array(ns.JointMSELoss(False, True)(p, t, w).item())
        pp = p.clone().requires_grad_(True)
        ns.JointMSELoss()(pp, t, w).backward()
        out[name + '.gpred'] = _np(pp.grad)
    _save('g5_mse.npz', **out)


def g6_targets(ns):
    img, hm = np.array([288, 384]), np.array([72, 96])
    rng = np.random.RandomState(6)
    J = 17
    joints = np.zeros((J, 3), np.float32)
    joints[:, 0] = rng.uniform(0, 288, J)
    joints[:, 1] = rng.uniform(0, 384, J)
    # hand-placed edge cases: centre, corners, just outside, far outside, half-integer rounding, negative
    joints[0, :2] = (144, 192)
    joints[1, :2] = (0, 0)
    joints[2, :2] = (287.9, 383.9)
    joints[3, :2] = (-30, 100)        # patch partially inside on the left?  mu_x = int(-7.5+0.5) = -7 -> ul=-16, br=3
    joints[4, :2] = (-60, -60)        # fully outside -> weight 0
    joints[5, :2] = (330, 100)        # right outside: mu_x = 83 -> ul = 74 >= 72 -> weight 0
    joints[6, :2] = (6.0, 10.0)       # x/4 + .5 = 2.0 exactly
    joints[7, :2] = (5.99, 9.99)
    joints[8, :2] = (-1.0, -3.0)      # int() truncation toward zero on negatives: int(-0.25+0.5)=0, int(-0.75+0.5)=0
    joints[9, :2] = (-3.0, -5.0)      # int(-0.75+.5)=0 ; int(-1.25+.5) = 0 (trunc toward zero, not floor)
    vis = np.ones((J, 3), np.float32)
    vis[10] = 0
    vis[11] = 0.5                     # weight 0.5 is NOT > 0.5: no patch drawn but weight stays 0.5
    out = {'joints': joints, 'vis': vis}
    for sigma in (3, 2):
        t, w = ns.generate_heatmaps(joints, vis, sigma, img, hm, J)
        out['s%d.target' % sigma], out['s%d.weight' % sigma] = t, w
    _save('g6_targets.npz', **out)


def g7_decode(ns):
    rng = np.random.RandomState(7)
    B, J, H, W = 3, 17, 24, 18
    out_hm = rng.randn(B, J, H, W).astype(np.float32)
    tgt = rng.rand(B, J, H, W).astype(np.float32)
    out_hm[0, 0] = -np.abs(out_hm[0, 0])             # all-negative map -> coords zeroed
    out_hm[0, 1] = 0.0                               # all ties at 0 -> index 0, max 0 -> zeroed
    out_hm[0, 2] = 0.0
    out_hm[0, 2, 5, 7] = out_hm[0, 2, 9, 3] = 2.5    # tie: first (row-major) wins
    tgt[1, 3] = 0.0
    tgt[1, 3, 0, 5] = 1.0                            # target at y<=1 -> ignored by accuracy
    tgt[1, 4] = 0.0
    tgt[1, 4, 10, 1] = 1.0                           # target at x<=1 -> ignored
    for j in range(5, 12):                           # plant near-hits so accuracy is not trivially 0
        for b in range(B):
            iy, ix = np.unravel_index(tgt[b, j].argmax(), (H, W))
            out_hm[b, j, min(iy + (j % 2), H - 1), ix] = 9.0
    preds, maxvals = ns.get_max_preds(out_hm)
    acc, avg, cnt, pred = ns.accuracy(out_hm, tgt)
    _save('g7_decode.npz', out=out_hm, tgt=tgt, preds=preds, maxvals=maxvals, acc=acc, avg=np.array(avg),
          cnt=np.array(cnt), pred=pred)


def g12_final_preds(ns):
    """get_final_preds.  cv2 is absent: the reference runs with oracle.ops.cv2_get_affine_transform
    injected as cv2.getAffineTransform (ref_harness), so this pins everything but that one third-party call."""
    rng = np.random.RandomState(12)
    B, J, H, W = 3, 17, 96, 72
    hm = rng.randn(B, J, H, W).astype(np.float32) * 0.1
    for b in range(B):
        for j in range(J):
            y, x = rng.randint(0, H), rng.randint(0, W)
            hm[b, j, y, x] += 1.0 + rng.rand()
            if 0 < x < W - 1:
                hm[b, j, y, x + 1] += 0.5 * rng.rand()
    hm[0, 0] = -np.abs(hm[0, 0])                    # no positive maximum -> coordinates zeroed before the transform
    hm[0, 1] = 0.0
    hm[0, 1, 0, 5] = 1.0                            # border maxima: no quarter-pixel shift
    hm[0, 2] = 0.0
    hm[0, 2, 40, 71] = 1.0
    hm[0, 3] = 0.0
    hm[0, 3, 50, 30] = 1.0                          # symmetric neighbours: sign(0) = 0
    center = np.array([[150.5, 200.25], [512.0, 300.0], [80.0, 1000.5]], np.float32)
    scale = np.array([[1.2, 1.6], [2.5, 3.3333], [0.45, 0.6]], np.float32)
    preds, maxvals = ns.get_final_preds(hm.copy(), center, scale)
    _save('g12_final_preds.npz', hm=hm, center=center, scale=scale, preds=preds, maxvals=maxvals)


def g13_input_pipeline(ns):
    """Crop transform, joint transform and flip bookkeeping of the training pipeline, from the
    reference's own datasets/process/affine_transform.py and pose_process.py (cv2.getAffineTransform injected as for
    g12).  The image resampling itself (cv2.warpAffine) is third-party and stays unpinned."""
    import importlib
    at = importlib.import_module('datasets.process.affine_transform')
    pp = importlib.import_module('datasets.process.pose_process')
    rng = np.random.RandomState(13)
    image_size = np.array([288, 384])
    centers = np.array([[640.5, 360.25], [100.0, 80.0], [1200.75, 700.0], [333.0, 512.5]], np.float64)
    scales = np.array([[1.8, 2.4], [0.6, 0.8], [2.7, 3.6], [1.2375, 1.65]], np.float64)
    rots = np.array([0.0, 27.5, -44.0, 90.0])
    trans = np.stack([at.dark_get_affine_transform(c.copy(), s.copy(), r, image_size) for c, s, r in zip(centers, scales, rots)])
    trans_inv = np.stack([at.dark_get_affine_transform(c.copy(), s.copy(), r, image_size, inv=1)
                          for c, s, r in zip(centers, scales, rots)])
    joints = np.zeros((4, 17, 3), np.float32)
    joints[:, :, 0] = rng.uniform(0, 1280, (4, 17))
    joints[:, :, 1] = rng.uniform(0, 720, (4, 17))
    vis = np.zeros((4, 17, 3), np.float32)
    vis[:, :, :2] = (rng.rand(4, 17, 1) < 0.8).astype(np.float32)
    pts = np.stack([np.stack([at.exec_affine_transform(joints[b, j, 0:2], trans[b]) for j in range(17)]) for b in range(4)])
    flip_pairs = [[3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
    fj, fv = [], []
    for b in range(4):
        a, v = pp.fliplr_joints(joints[b].copy(), vis[b].copy(), 1280, flip_pairs)
        fj.append(a)
        fv.append(v)
    _save('g13_input_pipeline.npz', image_size=image_size, centers=centers, scales=scales, rots=rots, trans=trans,
          trans_inv=trans_inv, joints=joints, vis=vis, pts=pts, flip_joints=np.stack(fj), flip_vis=np.stack(fv))


# ------------------------------------------------------------------ G8 MI / G9 whole model / G10 keys / G11 init stats
def g9_alignment(ns):
    cfg = rh.ref_cfg(48)
    torch.manual_seed(9)
    ref = ns.Alignment_V15(cfg, 'train')

    # G11: reference init statistics (Alignment_V15.py:185-214) before re-initialisation
    stats = {}
    sd = ref.state_dict()
    for key in ('hrnet.conv1.weight', 'hrnet.stage3.2.branches.1.0.conv1.weight', 'dcn_offset_1.conv.weight',
                'sup_agg_block.layers.0.conv1.weight', 'agg_final_layer.weight', 'dcn_1.weight',
                'feat_global_offset_layers.7.weight'):
        stats[key + '.std'] = np.array(sd[key].double().std().item())
        stats[key + '.mean'] = np.array(sd[key].double().mean().item())
    for key in ('dcn_offset_1.conv.bias', 'agg_final_layer.bias', 'dcn_1.bias', 'feat_global_offset_layers.7.bias',
                'hrnet.bn1.bias'):
        stats[key + '.absmax'] = np.array(sd[key].abs().max().item())
    stats['hrnet.bn1.weight.min'] = np.array(sd['hrnet.bn1.weight'].min().item())
    stats['hrnet.bn1.weight.max'] = np.array(sd['hrnet.bn1.weight'].max().item())
    _save('g11_init_stats.npz', **stats)

    # G10: state_dict keys + shapes
    with open(os.path.join(OUT, 'g10_state_dict_keys.txt'), 'w') as f:
        for k, v in sd.items():
            f.write('%s %s %s\n' % (k, 'x'.join(map(str, v.shape)) or 'scalar', str(v.dtype).replace('torch.', '')))
    f32 = [v for v in sd.values() if v.dtype == torch.float32]
    print('g10_state_dict_keys.txt      %d keys, %d params+buffers' % (len(sd), sum(v.numel() for v in f32)))

    orc = om.AlignmentOracle(om.make_cfg(48), True, 4, (384, 288))
    om.realistic_init_(ref, 15)
    om.realistic_init_(orc, 15)
    _same_state(ref, orc)

    # G8: the two MI estimators, extracted from the reference class, incl. gradient wrt the target side
    g = torch.Generator().manual_seed(8)
    Bm = 2
    feat = torch.randn(Bm, 48, 96, 72, generator=g) * 0.5
    f2 = (torch.randn(Bm, 48, 96, 72, generator=g) * 0.5).requires_grad_(True)
    yy = (torch.rand(Bm, 17, 96, 72, generator=g) * 0.8).requires_grad_(True)
    m1 = ref.feat_label_mi_estimation(feat, yy)
    m2 = ref.feat_feat_mi_estimation(feat, f2)
    (3.0 * m1 - 2.0 * m2).backward()
    _save('g8_mi.npz', seed=np.array(8), feat_label=np.array(m1.item()), feat_feat=np.array(m2.item()),
          gy_c3=_np(yy.grad[:, 3]), gf2_c7=_np(f2.grad[:, 7]),
          gy_abssum=np.array(yy.grad.double().abs().sum().item()),
          gf2_abssum=np.array(f2.grad.double().abs().sum().item()),
          gy_finite=np.array(bool(torch.isfinite(yy.grad).all())),
          gf2_finite=np.array(bool(torch.isfinite(f2.grad).all())))

    # G9: whole model, train-mode 3-tuple (+ loss and gradients) and the 2-tuple of a val-phase instance
    g = torch.Generator().manual_seed(90)
    B = 1
    kf = torch.randn(B, 3, 384, 288, generator=g)
    sup = torch.randn(B, 12, 384, 288, generator=g)
    tgt = torch.rand(B, 17, 96, 72, generator=g)
    w = (torch.rand(B, 17, 1, generator=g) < 0.8).float()
    ref.train()
    final, kf_hm, mi = ref(kf, sup)
    loss = oops.total_loss(final, tgt, w, mi)   # loss assembly is the build's restatement (core fn not importable)
    ref.zero_grad()
    loss.backward()
    grads = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
    out = dict(seed=np.array(90), init_seed=np.array(15), final=_np(final), kf_hm=_np(kf_hm),
               mi=np.array([m.item() for m in mi]), loss=np.array(loss.item()),
               final_argmax=_hm_summary(final.detach())['argmax'], kf_argmax=_hm_summary(kf_hm.detach())['argmax'])
    for k in ('agg_final_layer.weight', 'dcn_1.weight', 'dcn_offset_3.conv.weight', 'dcn_mask_4.conv.bias',
              'feat_global_offset_layers.9.weight', 'sup_agg_block.layers.0.conv1.weight', 'hrnet.conv1.weight',
              'hrnet.stage4.2.fuse_layers.0.3.0.weight', 'hrnet.layer1.0.bn1.weight'):
        out['grad.' + k + '.abssum'] = np.array(grads[k].double().abs().sum().item())
        out['grad.' + k + '.sum'] = np.array(grads[k].double().sum().item())
    out['grad.agg_final_layer.weight'] = _np(grads['agg_final_layer.weight'])
    out['grad.dcn_1.bias'] = _np(grads['dcn_1.bias'])
    out['n_params_with_grad'] = np.array(len(grads))
    # running stats moved by the train-mode forward
    out['after.hrnet.bn1.running_mean'] = _np(ref.state_dict()['hrnet.bn1.running_mean'])
    out['after.sup_agg_block.layers.0.bn1.running_var'] = _np(ref.state_dict()['sup_agg_block.layers.0.bn1.running_var'])

    val = ns.Alignment_V15(cfg, 'validate')
    om.realistic_init_(val, 15)
    val.eval()
    with torch.no_grad():
        res = val(kf, sup)
    assert len(res) == 2
    out['eval.final'], out['eval.kf_hm'] = _np(res[0]), _np(res[1])
    out['eval.final_argmax'] = _hm_summary(res[0])['argmax']
    _save('g9_alignment_v15.npz', **out)


def g14_posetrack_json(ns_unused):
    """The reference's PoseTrack JSON writer (PoseTrack_Alignment.evaluate, :883-1017) run on synthetic predictions:
    inputs and the written files are stored as one JSON fixture (data only)."""
    import json
    import tempfile
    import types
    ns = rh.load_posetrack_evaluate()
    rng = np.random.RandomState(14)
    out = {'cases': []}
    for is18 in (False, True):
        with tempfile.TemporaryDirectory() as tmp:
            annot = os.path.join(tmp, 'annot')
            os.makedirs(annot)
            zf = 6 if is18 else 8
            vids = [('images/bonn/000001_bonn', 5, 'images'), ('images/mpii_5sec/024159_mpii', 4, 'annolist')]
            annots = {}
            for vid, nfr, style in vids:
                fname = vid.split('/')[-1] + '.json'
                first = vid + '/' + str(0 if is18 else 1).zfill(zf) + '.jpg'
                if style == 'images':
                    data = {'images': [{'file_name': first, 'nframes': nfr}]}
                else:
                    data = {'annolist': [{'image': [{'name': first}]}] + [{'image': [{'name': 'x'}]} for _ in range(nfr - 1)]}
                annots[fname] = data
                with open(os.path.join(annot, fname), 'w') as f:
                    json.dump(data, f)
            # detections: video 1 frames {first+1: 2 people, first+3: 1 person}; video 2 frame {first: 1 person}
            f0 = 0 if is18 else 1
            names = ['/data/posetrack/' + vids[0][0] + '/' + str(f0 + 1).zfill(zf) + '.jpg',
                     '/data/posetrack/' + vids[0][0] + '/' + str(f0 + 3).zfill(zf) + '.jpg',
                     '/data/posetrack/' + vids[1][0] + '/' + str(f0).zfill(zf) + '.jpg']
            fmap = {names[0]: [0, 2], names[1]: [1], names[2]: [3]}
            preds = np.concatenate([rng.uniform(0, 500, (4, 17, 2)), rng.uniform(0, 1, (4, 17, 1))], 2)
            boxes = np.concatenate([rng.uniform(50, 400, (4, 2)), rng.uniform(0.5, 2, (4, 2)),
                                    rng.uniform(1e3, 1e5, (4, 1)), rng.uniform(0.3, 1, (4, 1))], 1)
            fake = types.SimpleNamespace(phase='validate', annotation_dir=annot, is_posetrack18=is18)
            outdir = os.path.join(tmp, 'out')
            ns.PoseTrack_Alignment.evaluate(fake, None, preds, outdir, boxes, fmap)
            files = {}
            resdir = os.path.join(outdir, 'val_set_json_results')
            for fn in sorted(os.listdir(resdir)):
                with open(os.path.join(resdir, fn)) as f:
                    files[fn] = json.load(f)
            out['cases'].append({'is_posetrack18': is18, 'annotations': annots, 'filenames_map': fmap,
                                 'preds': preds.tolist(), 'boxes': boxes.tolist(), 'written': files})
    with open(os.path.join(OUT, 'g14_posetrack_json.json'), 'w') as f:
        json.dump(out, f)
    print('g14_posetrack_json.json', os.path.getsize(os.path.join(OUT, 'g14_posetrack_json.json')), 'bytes')


def main():
    assert rh.available(), 'needs /root/reference (build container only)'
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ns = rh.load()
    if len(sys.argv) > 1:                 # regenerate selected fixtures only: python -m oracle.gen_golden g12_final_preds
        for name in sys.argv[1:]:
            globals()[name](ns)
        return
    g12_final_preds(ns)
    g13_input_pipeline(ns)
    g1_blocks(ns)
    g2_hrmodule(ns)
    g3_hrnet_w32(ns)
    g4_hrnetplus_w48(ns)
    g5_mse(ns)
    g6_targets(ns)
    g7_decode(ns)
    g9_alignment(ns)
    g14_posetrack_json(ns)


if __name__ == '__main__':
    main()

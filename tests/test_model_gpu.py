"""Whole-model parity of the HIP path (through the C ABI) on the MI355X:
  * against the committed golden vectors generated from the REFERENCE class (tests/golden/g3, g9),
  * against the CPU oracle on the same seeded inputs (BASELINE configs 2-like 3-frame case, generalised heads),
  * at BASELINE's full size through size-independent properties (eval-mode batch independence, run-to-run
    determinism, hipGraph replay == eager launch sequence, a Trainer step == oracle + torch.optim.Adam).
Tolerances (fp32): heatmaps <= 1e-3 absolute on O(1) heatmaps (north_star), argmax indices bit-exact.
"""
import os
import warnings

import numpy as np
import pytest
import torch

import fami_pose_amd as fp
from oracle import model as om, ops as oops

pytestmark = pytest.mark.gpu
warnings.filterwarnings('ignore')
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
HM_TOL = 1e-3


def _argmax(hm):
    return hm.reshape(hm.shape[0], hm.shape[1], -1).argmax(2).cpu().numpy()


def _pair(width, S, hw, phase, seed, freeze=False):
    """HIP model + oracle carrying the same realistic-scale weights."""
    H, W = hw
    orc = om.realistic_init_(om.AlignmentOracle(om.make_cfg(width), phase == 'train', S, (H, W)), seed)
    model = fp.build_model(fp.default_cfg(width, image_size=(W, H), num_sup=S, freeze_backbone=freeze), phase)
    model.load_state_dict(orc.state_dict())
    return model, orc


def test_g9_alignment_v15_golden(dev):
    """Reference class outputs (train 3-tuple, loss, gradients; val-phase 2-tuple) on the 5-frame 384x288 W48 model."""
    g = np.load(os.path.join(GOLD, 'g9_alignment_v15.npz'))
    gen = torch.Generator().manual_seed(int(g['seed']))
    kf = torch.randn(1, 3, 384, 288, generator=gen)
    sup = torch.randn(1, 12, 384, 288, generator=gen)
    tgt = torch.rand(1, 17, 96, 72, generator=gen)
    w = (torch.rand(1, 17, 1, generator=gen) < 0.8).float()
    model, _ = _pair(48, 4, (384, 288), 'train', int(g['init_seed']))
    model = model.to(dev)
    final, kf_hm, mi = model(kf.to(dev), sup.to(dev))
    assert (final.cpu() - torch.from_numpy(g['final'])).abs().max().item() < HM_TOL
    assert (kf_hm.cpu() - torch.from_numpy(g['kf_hm'])).abs().max().item() < HM_TOL
    assert np.array_equal(_argmax(final), g['final_argmax'])          # bit-exact keypoint indices
    assert np.array_equal(_argmax(kf_hm), g['kf_argmax'])
    assert np.allclose([m.item() for m in mi], g['mi'], rtol=2e-3, atol=1e-9)
    from fami_pose_amd.loss import JointMSELoss
    loss = JointMSELoss()(final, tgt.to(dev), w.to(dev)) + \
        0.5 * (-0.1 * mi[0] + 0.1 * mi[1] + mi[2] - mi[3] + mi[4] - mi[5])
    assert loss.item() == pytest.approx(float(g['loss']), rel=1e-4)
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    ref = torch.from_numpy(g['grad.agg_final_layer.weight'])
    assert ((grads['agg_final_layer.weight'].cpu() - ref).abs().max() / ref.abs().max()).item() < 1e-3
    # a bias gradient is sum_pixels dy: O(1e-5) left after cancellation of O(1e-3) terms -> looser relative bound
    ref = torch.from_numpy(g['grad.dcn_1.bias'])
    assert ((grads['dcn_1.bias'].cpu() - ref).abs().max() / ref.abs().max()).item() < 5e-2
    # The golden gradients are the reference's own fp32-CPU values.  Backward through the train-mode BatchNorm
    # chains is ill-conditioned in fp32 (the fp32 CPU path itself is several 1e-3 away from an fp64 evaluation,
    # see test_model_vs_oracle, which arbitrates with fp64), so norms are held to 1e-2 here.
    bad = []
    for key in g.files:
        if key.startswith('grad.') and key.endswith('.abssum'):
            name = key[5:-7]
            got, want = grads[name].double().abs().sum().item(), float(g[key])
            if abs(got - want) > 1e-2 * want:
                bad.append((name, got, want))
    assert not bad, bad
    sd = model.state_dict()
    assert (sd['hrnet.bn1.running_mean'].cpu() - torch.from_numpy(g['after.hrnet.bn1.running_mean'])).abs().max() < 1e-5
    assert int(sd['hrnet.bn1.num_batches_tracked']) == 1

    val, _ = _pair(48, 4, (384, 288), fp.VAL_PHASE, int(g['init_seed']))
    val = val.to(dev)
    with torch.no_grad():
        res = val(kf.to(dev), sup.to(dev))
    assert len(res) == 2
    assert (res[0].cpu() - torch.from_numpy(g['eval.final'])).abs().max().item() < HM_TOL
    assert (res[1].cpu() - torch.from_numpy(g['eval.kf_hm'])).abs().max().item() < HM_TOL
    assert np.array_equal(_argmax(res[0]), g['eval.final_argmax'])


def test_g3_hrnet_w32_config1_golden(dev):
    """BASELINE configs[0] (HRNet-W32 256x192 single-frame heatmaps + MSE) on the HIP path vs the reference."""
    g = np.load(os.path.join(GOLD, 'g3_hrnet_w32.npz'))
    orc = om.realistic_init_(om.HRNetOracle(om.make_cfg(32), plus=False), 32)
    cfg = fp.default_cfg(32, name='HRNet', image_size=(192, 256))
    net = fp.build_model(cfg, fp.VAL_PHASE)
    net.load_state_dict(orc.state_dict())
    net = net.to(dev)
    x = torch.randn(1, 3, 256, 192, generator=torch.Generator().manual_seed(int(g['x_seed'])))
    with torch.no_grad():
        hm, feats = net(x.to(dev))
    assert (hm.cpu() - torch.from_numpy(g['hm'])).abs().max().item() < HM_TOL
    assert np.array_equal(_argmax(hm), g['argmax'])
    gen = torch.Generator().manual_seed(int(g['tw_seed']))
    tgt = torch.rand(1, 17, 64, 48, generator=gen)
    w = (torch.rand(1, 17, 1, generator=gen) < 0.8).float()
    from fami_pose_amd.loss import JointMSELoss
    assert JointMSELoss()(hm, tgt.to(dev), w.to(dev)).item() == pytest.approx(float(g['mse']), rel=1e-4)
    assert len(feats) == 4
    for i, f in enumerate(feats):
        assert f.double().abs().sum().item() == pytest.approx(float(g['feat%d_abssum' % i]), rel=1e-4)


@pytest.mark.parametrize('S,hw,B', [(2, (192, 128), 2), (7, (128, 96), 2), (1, (256, 192), 1)])
def test_model_vs_oracle(dev, S, hw, B):
    """BASELINE configs[1]'s graph (3-frame W48) and generalised heads (7 / 1 supporting frames, other input sizes)
    against the CPU oracle: forward, loss, and gradients of head, DCN, translation regressor and backbone -- every
    parameter, arbitrated by an fp64 evaluation.  (Sizes keep the fp64 CPU pass short: the full 384x288 resolution is
    held to the oracle by test_train_gpu.py::test_bench_workload_train_step_matches_the_oracle at the bench's own batch
    and by the reference-generated golden g9; the GPU suite has to fit the driver's time limit.)"""
    H, W = hw
    model, orc = _pair(48, S, hw, 'train', 3 + S)
    model = model.to(dev)
    gen = torch.Generator().manual_seed(50 + S)
    kf, sup = torch.randn(B, 3, H, W, generator=gen), torch.randn(B, 3 * S, H, W, generator=gen)
    tgt = torch.rand(B, 17, H // 4, W // 4, generator=gen)
    w = (torch.rand(B, 17, 1, generator=gen) < 0.8).float()
    f0, k0, mi0 = orc(kf, sup)
    l0 = oops.total_loss(f0, tgt, w, mi0)
    l0.backward()
    f1, k1, mi1 = model(kf.to(dev), sup.to(dev))
    from fami_pose_amd.loss import JointMSELoss
    l1 = JointMSELoss()(f1, tgt.to(dev), w.to(dev)) + 0.5 * (-0.1 * mi1[0] + 0.1 * mi1[1] + mi1[2] - mi1[3] + mi1[4] - mi1[5])
    l1.backward()
    assert (f1.cpu() - f0).abs().max().item() < HM_TOL and (k1.cpu() - k0).abs().max().item() < HM_TOL
    assert np.array_equal(_argmax(f1), _argmax(f0.detach())) and np.array_equal(_argmax(k1), _argmax(k0.detach()))
    assert l1.item() == pytest.approx(l0.item(), rel=1e-4)
    # Gradients: fp32 backward through ~100 train-mode BatchNorms is ill-conditioned (each BN backward subtracts the
    # common mode of the incoming gradient), so two correct fp32 implementations disagree at the 1e-2 level deep in
    # the net (the low-resolution 384-channel branch, K = 3456 per output, worst).  An fp64 evaluation of the oracle
    # arbitrates: against it the HIP path must be as accurate as the reference's fp32 CPU path over the ~1900
    # parameters -- median and 90th-percentile error within 3x of the CPU path's, and no single parameter off by more
    # than max(25 % of its gradient's max magnitude, 2x the CPU path's own error).  The amplification is chaotic: two
    # kernel selections with IDENTICAL rounding error per conv (tools/gpu_diag_c32err.py: same rms against fp64)
    # land at 1.4x and 2.0x the CPU median (tools/gpu_diag_median.py), so the factor is a noise band, not a precision
    # claim; a wrong kernel gives O(1) errors (100x the median) everywhere.  Measured worst HIP outlier 0.13-0.28 where
    # the CPU path is at 0.05-0.28.
    import copy
    orc64 = copy.deepcopy(orc).double()
    orc64.zero_grad()
    f64, _, mi64 = orc64(kf.double(), sup.double())
    oops.total_loss(f64, tgt.double(), w.double(), mi64).backward()
    assert (f1.cpu().double() - f64).abs().max().item() < 2 * (f0.double() - f64).abs().max().item() + 1e-4
    ref, ref64, mine = dict(orc.named_parameters()), dict(orc64.named_parameters()), dict(model.named_parameters())
    bad, e_cpus, e_hips = [], [], []
    for name, p64 in ref64.items():
        if p64.grad is None:
            assert mine[name].grad is None or float(mine[name].grad.abs().max()) == 0.0, name
            continue
        g64 = p64.grad
        s64 = g64.abs().max().item()
        if s64 < 1e-9:          # conv biases in front of a train-mode BN: exactly-zero gradient up to rounding
            continue
        e_cpu = (ref[name].grad.double() - g64).abs().max().item() / s64
        e_hip = (mine[name].grad.cpu().double() - g64).abs().max().item() / s64
        e_cpus.append(e_cpu)
        e_hips.append(e_hip)
        if e_hip > max(0.25, 2 * e_cpu):
            bad.append((name, e_hip, e_cpu))
    assert len(e_hips) > 900 and not bad, bad[:20]
    assert np.median(e_hips) <= 3 * np.median(e_cpus) + 1e-4, (np.median(e_hips), np.median(e_cpus))
    assert np.percentile(e_hips, 90) <= 3 * np.percentile(e_cpus, 90) + 1e-4, (np.percentile(e_hips, 90), np.percentile(e_cpus, 90))


def test_full_size_gradients_elementwise_vs_fp64(dev):
    """Per-ELEMENT gradient parity at the headline resolution (5-frame 384x288 W48, one clip) against an fp64 evaluation
    of the oracle, for a fixed subset of parameters: the head's output layer, the four DCN weights, and one convolution per
    HRNet stage (layer1, stages 2-4, on branches 0-3).  The norm bands elsewhere in this file catch a wrong kernel; this
    one bounds the error of every element: the HIP gradient's distance from fp64 -- relative RMS over the tensor and the
    largest element error relative to the tensor's largest element -- must stay within 2x the reference arithmetic's own
    distance (the fp32 CPU oracle against the same fp64 values) plus a floor of 2e-5 / 1e-4.  Backward through ~100
    train-mode BatchNorms amplifies fp32 rounding (test_model_vs_oracle), so the bound is relative to what fp32 itself
    achieves on each tensor, not absolute."""
    import copy
    S, H, W, B = 4, 384, 288, 1
    model, orc = _pair(48, S, (H, W), 'train', 17)
    model = model.to(dev)
    gen = torch.Generator().manual_seed(417)
    kf, sup = torch.randn(B, 3, H, W, generator=gen), torch.randn(B, 3 * S, H, W, generator=gen)
    tgt = torch.rand(B, 17, H // 4, W // 4, generator=gen)
    w = (torch.rand(B, 17, 1, generator=gen) < 0.8).float()
    f0, _, mi0 = orc(kf, sup)
    oops.total_loss(f0, tgt, w, mi0).backward()
    orc64 = copy.deepcopy(orc).double()
    orc64.zero_grad()
    f64, _, mi64 = orc64(kf.double(), sup.double())
    oops.total_loss(f64, tgt.double(), w.double(), mi64).backward()
    f1, _, mi1 = model(kf.to(dev), sup.to(dev))
    from fami_pose_amd.loss import JointMSELoss
    l1 = JointMSELoss()(f1, tgt.to(dev), w.to(dev)) + 0.5 * (-0.1 * mi1[0] + 0.1 * mi1[1] + mi1[2] - mi1[3] + mi1[4] - mi1[5])
    l1.backward()
    names = ['agg_final_layer.weight', 'dcn_1.weight', 'dcn_2.weight', 'dcn_3.weight', 'dcn_4.weight',
             'hrnet.layer1.0.conv1.weight', 'hrnet.stage2.0.branches.0.0.conv1.weight',
             'hrnet.stage3.0.branches.1.0.conv1.weight', 'hrnet.stage4.0.branches.2.0.conv1.weight',
             'hrnet.stage4.2.branches.3.3.conv2.weight']
    ref, ref64, mine = dict(orc.named_parameters()), dict(orc64.named_parameters()), dict(model.named_parameters())
    bad = []
    for n in names:
        g64 = ref64[n].grad
        gc_, gh = ref[n].grad.double(), mine[n].grad.cpu().double()
        assert gh.shape == g64.shape
        rms_c, rms_h = ((gc_ - g64).norm() / g64.norm()).item(), ((gh - g64).norm() / g64.norm()).item()
        mx = g64.abs().max().item()
        max_c, max_h = (gc_ - g64).abs().max().item() / mx, (gh - g64).abs().max().item() / mx
        print('elementwise grad %-45s rms: HIP %.3e CPU-fp32 %.3e | max: HIP %.3e CPU-fp32 %.3e' % (n, rms_h, rms_c, max_h, max_c))
        if rms_h > 2 * rms_c + 2e-5 or max_h > 2 * max_c + 1e-4:
            bad.append((n, rms_h, rms_c, max_h, max_c))
    assert not bad, bad


@pytest.mark.parametrize('width,S,hw,B', [(48, 7, (512, 384), 1), (64, 4, (128, 96), 2), (32, 2, (128, 96), 2)])
def test_baseline_configs_4_5_forward(dev, width, S, hw, B):
    """BASELINE configs[3] (W48, 512x384, 7 supporting frames: 128x96 feature maps, 336-channel sup_agg input) and the
    widths of configs[0]/[4] (W32: 8 DCN offset groups, W64: 16) -- forward + loss parity against the oracle with the
    generalised head."""
    H, W = hw
    G = 12 if width % 48 == 0 else width // 4
    orc = om.realistic_init_(om.AlignmentOracle(om.make_cfg(width), True, S, (H, W), dcn_groups=G), 40 + width)
    model = fp.build_model(fp.default_cfg(width, image_size=(W, H), num_sup=S), 'train')
    assert model.G == G
    model.load_state_dict(orc.state_dict())
    model = model.to(dev)
    gen = torch.Generator().manual_seed(60 + width)
    kf, sup = torch.randn(B, 3, H, W, generator=gen), torch.randn(B, 3 * S, H, W, generator=gen)
    tgt = torch.rand(B, 17, H // 4, W // 4, generator=gen)
    w = (torch.rand(B, 17, 1, generator=gen) < 0.8).float()
    f0, k0, mi0 = orc(kf, sup)
    l0 = oops.total_loss(f0, tgt, w, mi0)
    l0.backward()
    f0, k0 = f0.detach(), k0.detach()
    f1, k1, mi1 = model(kf.to(dev), sup.to(dev))
    assert (f1.cpu() - f0).abs().max().item() < HM_TOL and (k1.cpu() - k0).abs().max().item() < HM_TOL
    assert np.array_equal(_argmax(f1), _argmax(f0)) and np.array_equal(_argmax(k1), _argmax(k0))
    from fami_pose_amd.loss import JointMSELoss
    l1 = JointMSELoss()(f1, tgt.to(dev), w.to(dev)) + 0.5 * (-0.1 * mi1[0] + 0.1 * mi1[1] + mi1[2] - mi1[3] + mi1[4] - mi1[5])
    assert l1.item() == pytest.approx(l0.item(), rel=1e-4)
    l1.backward()
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in model.parameters())
    # backward parity: the head's output layer at 1e-3, every parameter's gradient norm within 2e-2 of the CPU fp32
    # path's (fp32 backward through the train-mode BatchNorm chains is ill-conditioned -- see test_model_vs_oracle, which
    # arbitrates the W48 cases with fp64; the norms catch a wrong kernel or a wrong generalised-head shape)
    ref, mine = dict(orc.named_parameters()), dict(model.named_parameters())
    g0, g1 = ref['agg_final_layer.weight'].grad, mine['agg_final_layer.weight'].grad.cpu()
    assert ((g1 - g0).abs().max() / g0.abs().max()).item() < 1e-3
    bad = []
    for name, p in ref.items():
        if p.grad is None or p.grad.abs().max().item() < 1e-9:
            continue
        a, b = mine[name].grad.double().abs().sum().item(), p.grad.double().abs().sum().item()
        # the translation regressor's gradient enters through d(shift)/d(tx, ty): a whole map of signed products summed
        # into 2 numbers per frame (cancellation), so the fp32 CPU path itself is only good to a few percent there
        if abs(a - b) > (1e-1 if name.startswith('feat_global_offset_layers') else 2e-2) * b:
            bad.append((name, a, b))
    if bad:
        # a norm outside the 2 % band around the CPU fp32 path is arbitrated by an fp64 evaluation of the oracle: the HIP
        # norm must be within 3x the CPU path's own distance from the truth, or within 5 % of the truth.  Measured on
        # W32 / S=2 with the second accumulator set of the linear-address implicit GEMM (another summation order inside
        # every 3x3 convolution): stage3.3.branches.0.2.bn1.weight fp64 0.013714, CPU 0.013619 (-0.7 %), HIP 0.013343
        # (-2.7 %); stage4.0.fuse_layers.0.2.1.weight 0.012050 / 0.012008 / 0.011751 -- with the single accumulator set the
        # same two norms sit just inside the 2 % band.  test_model_vs_oracle tolerates element errors of 25 % of the
        # gradient's magnitude on the same chaotic amplification (see its comment); a wrong kernel is off by O(1).
        import copy
        orc64 = copy.deepcopy(orc).double()
        orc64.zero_grad()
        f64, _, mi64 = orc64(kf.double(), sup.double())
        oops.total_loss(f64, tgt.double(), w.double(), mi64).backward()
        ref64 = dict(orc64.named_parameters())
        worse = []
        for name, a, b in bad:
            c = ref64[name].grad.abs().sum().item()
            if abs(a - c) > max(3 * abs(b - c), 5e-2 * c):
                worse.append((name, a, b, c))
        assert not worse, worse[:10]


@pytest.mark.parametrize('S,B', [(4, 4), (2, 8)], ids=['config3_5frame_b4', 'config2_3frame_b8'])
def test_full_size_properties(dev, S, B):
    """BASELINE's full per-GPU batches -- 4 five-frame clips (the bench workload) and config 2 as stated: 8 three-frame
    clips, fp32 -- at 384x288: properties that need no CPU reference."""
    model, _ = _pair(48, S, (384, 288), fp.VAL_PHASE, 21)
    model = model.to(dev)
    gen = torch.Generator().manual_seed(77)
    kf, sup = torch.randn(B, 3, 384, 288, generator=gen).to(dev), torch.randn(B, 3 * S, 384, 288, generator=gen).to(dev)
    with torch.no_grad():
        a = model(kf, sup)
        b = model(kf, sup)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])              # deterministic (no atomics in forward)
        one = model(kf[2:3], sup[2:3])                                          # eval BN: clips are independent
    assert (one[0] - a[0][2:3]).abs().max().item() < 1e-4
    assert (one[1] - a[1][2:3]).abs().max().item() < 1e-4
    assert np.array_equal(_argmax(one[0]), _argmax(a[0][2:3]))
    assert torch.isfinite(a[0]).all() and a[0].shape == (B, 17, 96, 72)


def test_trainer_step_matches_oracle_adam(dev):
    """Trainer.step (fwd, on-device targets, MSE + MI, bwd, Adam in flat arenas) vs oracle + torch.optim.Adam, and
    the hipGraph replay vs the eager launch sequence."""
    from fami_pose_amd.train import Trainer
    S, H, W, B = 2, 128, 96, 2
    gen = torch.Generator().manual_seed(9)
    kf, sup = torch.randn(B, 3, H, W, generator=gen), torch.randn(B, 3 * S, H, W, generator=gen)
    joints = torch.rand(B, 17, 2, generator=gen) * torch.tensor([W, H], dtype=torch.float32)
    vis = (torch.rand(B, 17, generator=gen) < 0.8).float()

    model, orc = _pair(48, S, (H, W), 'train', 5)
    opt = torch.optim.Adam(orc.parameters(), lr=1e-3)
    tg = np.zeros((B, 17, H // 4, W // 4), np.float32)
    tw = np.zeros((B, 17, 1), np.float32)
    for b in range(B):
        j3 = np.concatenate([joints[b].numpy(), np.zeros((17, 1), np.float32)], 1)
        v3 = np.repeat(vis[b].numpy()[:, None], 3, 1)
        tg[b], tw[b] = oops.generate_heatmaps(j3, v3, 3, np.array([W, H]), np.array([W // 4, H // 4]), 17)
    f0, k0, mi0 = orc(kf, sup)
    l0 = oops.total_loss(f0, torch.from_numpy(tg), torch.from_numpy(tw), mi0)
    opt.zero_grad()
    l0.backward()
    g_ref = orc.agg_final_layer.weight.grad.clone()
    opt.step()

    tr = Trainer(model.to(dev), lr=1e-3, use_graph=False, targets_from_joints=True)
    tr.step(kf.to(dev), sup.to(dev), joints.to(dev), vis.to(dev))
    assert tr.loss_value() == pytest.approx(l0.item(), rel=1e-4)
    g1 = tr.views[id(model.agg_final_layer.weight)].cpu()
    assert ((g1 - g_ref).abs().max() / g_ref.abs().max()).item() < 1e-3
    # Adam's first step is lr*g/(|g|+eps): compare where the gradient is not at the noise floor
    p0, p1 = orc.agg_final_layer.weight.data, model.agg_final_layer.weight.data.cpu()
    big = g_ref.abs() > 1e-3 * g_ref.abs().max()
    assert (p1 - p0)[big].abs().max().item() < 1e-5
    rm0, rm1 = orc.hrnet.bn1.running_mean, model.hrnet.bn1.running_mean.cpu()
    assert (rm0 - rm1).abs().max().item() < 1e-5

    # graph replay == eager, three steps from identical starts
    losses = []
    for use_graph in (False, True):
        m2, _ = _pair(48, S, (H, W), 'train', 5)
        t2 = Trainer(m2.to(dev), lr=1e-3, use_graph=use_graph, targets_from_joints=True)
        ls = []
        for _ in range(3):                  # the capture warm-up is rolled back: step k means the same in both modes
            t2.step(kf.to(dev), sup.to(dev), joints.to(dev), vis.to(dev))
            ls.append(t2.loss_value())
        losses.append(ls)
    assert all(np.isfinite(losses[0])) and all(np.isfinite(losses[1]))
    assert losses[0][0] == pytest.approx(l0.item(), rel=1e-4)
    assert losses[0][2] < losses[0][0]                                           # the step optimises
    assert losses[1] == pytest.approx(losses[0], rel=1e-2)                       # graph replay == eager, step by step


def _half_emulation(orc, dtype):
    """The oracle with every conv operand / conv output / BN output rounded to the 16-bit storage type (fp32
    accumulation, fp32 heatmap heads): a plain restatement of '16-bit storage + 16-bit MFMA' for the reference graph."""
    import copy
    emu = copy.deepcopy(orc)
    rb = lambda t: t.to(dtype).float()
    with torch.no_grad():
        for m in emu.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.copy_(rb(m.weight))
    for n, m in emu.named_modules():
        if isinstance(m, torch.nn.Conv2d):
            m.register_forward_pre_hook(lambda mod, inp: tuple(rb(i) for i in inp))
            if not n.endswith('final_layer'):
                m.register_forward_hook(lambda mod, inp, out: rb(out))
        elif isinstance(m, torch.nn.BatchNorm2d):
            m.register_forward_hook(lambda mod, inp, out: rb(out))
    return emu


def _keypoint_agreement(hm, ref):
    """-> (share of joints whose flat argmax index equals the reference's, PCK@0.5 of hm against ref as the target)."""
    a, b = _argmax(hm), _argmax(ref)
    _, pck, _, _ = oops.accuracy(hm.numpy(), ref.numpy())
    return float((a == b).mean()), float(pck)


@pytest.mark.parametrize('mode', ['bf16', 'f16'])
def test_half_mode_vs_oracle(dev, mode):
    """BASELINE config 3's arithmetic (bf16) and config 5's (fp16): 16-bit activations + 16-bit MFMA convolutions; fp32
    accumulation, master weights, BN statistics, heatmaps, losses.  bf16 keeps 8 significand bits and this randomly
    initialised 300-layer net amplifies rounding noise: a plain bf16 emulation of the REFERENCE graph on CPU is itself
    ~0.2-0.35 (relative RMS) away from its fp32 evaluation; fp16 (11 bits) is ~8x closer.  The criterion is therefore
    relative to that emulation: the HIP path must be no further from the fp32 oracle than 1.5x the emulation's
    distance (per output, relative RMS) plus a floor of 1e-2 (bf16) / 2e-3 (fp16), and its loss within 10 % / 2 %.
    The 1e-3 / bit-exact-argmax contract belongs to the fp32 mode, tested above; the 16-bit kernels are held
    individually to 1e-2 / 1.5e-3 in tests/test_kernels_half_gpu.py."""
    tdt, floor, ltol = (torch.bfloat16, 1e-2, 0.1) if mode == 'bf16' else (torch.float16, 2e-3, 0.02)
    S, H, W, B = (4, 384, 288, 2) if mode == 'bf16' else (4, 384, 288, 1)   # (the fp16 CPU emulation is 2x slower per pixel: one clip)
    model, orc = _pair(48, S, (H, W), 'train', 31)
    model = model.to(dev).set_compute_dtype(mode)
    gen = torch.Generator().manual_seed(131)
    kf, sup = torch.randn(B, 3, H, W, generator=gen), torch.randn(B, 3 * S, H, W, generator=gen)
    tgt = torch.rand(B, 17, H // 4, W // 4, generator=gen)
    w = (torch.rand(B, 17, 1, generator=gen) < 0.8).float()
    emu = _half_emulation(orc, tdt)
    with torch.no_grad():
        f0, k0, mi0 = orc(kf, sup)
        fe, ke, _ = emu(kf, sup)
        l0 = oops.total_loss(f0, tgt, w, mi0)
    f1, k1, mi1 = model(kf.to(dev), sup.to(dev))
    assert f1.dtype == torch.float32 and k1.dtype == torch.float32
    rms = lambda a, b: ((a - b).norm() / b.norm()).item()
    for hip, emul, ref in ((f1, fe, f0), (k1, ke, k0)):
        e_hip, e_emu = rms(hip.detach().cpu(), ref), rms(emul, ref)
        assert e_hip <= 1.5 * e_emu + floor, (e_hip, e_emu)
    # keypoint level, reported here and ASSERTED in test_half_mode_keypoints_on_a_fitted_model: a randomly initialised net
    # emits noise heatmaps whose argmax flips under any rounding (measured: bf16 0.32 agreement for the HIP path AND for the
    # CPU emulation of the reference graph in bf16; fp16 0.82 / 0.88 on 17 joints), so on THIS model the only meaningful
    # statement is "no worse than a plain 16-bit port of the reference": within 0.15 (2-3 joints) of the emulation
    for name, hip, emul, ref in (('final', f1, fe, f0), ('kf', k1, ke, k0)):
        a_hip, p_hip = _keypoint_agreement(hip.detach().cpu(), ref)
        a_emu, p_emu = _keypoint_agreement(emul, ref)
        print('half-mode keypoints (random init) %s %s: HIP agreement %.4f PCK %.4f | CPU emulation agreement %.4f PCK %.4f'
              % (mode, name, a_hip, p_hip, a_emu, p_emu))
        assert a_hip >= a_emu - 0.15 and p_hip >= p_emu - 0.15, (name, a_hip, a_emu, p_hip, p_emu)
    from fami_pose_amd.loss import JointMSELoss
    l1 = JointMSELoss()(f1, tgt.to(dev), w.to(dev)) + 0.5 * (-0.1 * mi1[0] + 0.1 * mi1[1] + mi1[2] - mi1[3] + mi1[4] - mi1[5])
    assert l1.item() == pytest.approx(l0.item(), rel=ltol)
    # fp16 activation gradients need the caller's loss scale, as with any fp16 autocast training loop
    ls = 1.0 if mode == 'bf16' else 4096.0
    (l1 * ls).backward()
    g = model.agg_final_layer.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.dtype == torch.float32
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in model.parameters())
    if mode == 'f16':
        # head gradients against the fp32 oracle's: 11-bit storage noise, not underflow, must be what separates them
        orc.zero_grad()
        fo, ko, mio = orc(kf, sup)
        oops.total_loss(fo, tgt, w, mio).backward()
        # 5e-2 at the output layer.  Deeper weights see the 11-bit noise of the whole 16-bit FORWARD (the feature maps
        # entering the head differ by a few percent) amplified by the train-mode BatchNorm backward (the effect
        # test_model_vs_oracle arbitrates with fp64 in the fp32 mode).  tools/diag_f16_grad.py: 0.078 / 0.308 for the two
        # weights below at loss scales 64, 4096 and 65536 alike (so it is not underflow), 0.23 / 0.84 in bf16.
        for name, tol in (('agg_final_layer.weight', 5e-2), ('init_feature_agg_block.layers.2.conv2.weight', 1.2e-1),
                          ('dcn_4.weight', 4.5e-1)):
            g0 = dict(orc.named_parameters())[name].grad
            g1 = dict(model.named_parameters())[name].grad.cpu() / ls
            assert ((g1 - g0).norm() / g0.norm()).item() < tol, name
    # training in the 16-bit mode optimises (the Trainer applies its own static loss scale in fp16)
    from fami_pose_amd.train import Trainer
    m2, _ = _pair(48, 2, (128, 96), 'train', 5)
    tr = Trainer(m2.to(dev).set_compute_dtype(mode), lr=1e-3, use_graph=False, targets_from_joints=True)
    assert tr.loss_scale == (8192.0 if mode == 'f16' else 1.0)
    gen = torch.Generator().manual_seed(9)
    kf2, sup2 = torch.randn(2, 3, 128, 96, generator=gen).to(dev), torch.randn(2, 6, 128, 96, generator=gen).to(dev)
    joints = (torch.rand(2, 17, 2, generator=gen) * torch.tensor([96.0, 128.0])).to(dev)
    vis = (torch.rand(2, 17, generator=gen) < 0.8).float().to(dev)
    ls = []
    for _ in range(4):
        tr.step(kf2, sup2, joints, vis)
        ls.append(tr.loss_value())
    assert all(np.isfinite(ls)) and ls[-1] < ls[0]


@pytest.mark.parametrize('mode', ['bf16', 'f16'])
def test_half_mode_keypoints_on_a_fitted_model(dev, mode):
    """Keypoint-level criterion for BASELINE config 3's arithmetic (bf16) and config 5's (fp16) at 384x288: what a pose
    estimator is judged by is where its heatmaps peak, and that only means something on a model whose heatmaps HAVE peaks.
    So the fp32 HIP model is first fitted to one batch of two 5-frame clips (Trainer, Adam, MSE + MI, on-device Gaussian
    targets) until its heatmaps peak at the joints; then the SAME weights run in fp32 on the CPU oracle (the reference
    arithmetic), in fp32 on the HIP path and in the 16-bit mode on the HIP path, train-mode BatchNorm on the same batch:
      * the 16-bit heatmaps peak within two heatmap pixels of the fp32 oracle's peak for >= 95 % of the visible joints, never
        further than three (exact-index and one-pixel agreement are printed);
      * |PCK@0.5(16-bit) - PCK@0.5(fp32 oracle)| <= 0.02 against the ground-truth targets (`accuracy`, evaluate.py:39-75);
      * the fp32 HIP path keeps the bit-exact index contract on the fitted weights too."""
    from fami_pose_amd.train import Trainer
    S, H, W, B = 4, 384, 288, 2
    model, orc = _pair(48, S, (H, W), 'train', 23)
    model = model.to(dev).set_deterministic(True)      # the FIT is reproducible run to run (fixed-point DCN input gradient): so are the rates below
    gen = torch.Generator().manual_seed(523)
    kf, sup = torch.randn(B, 3, H, W, generator=gen), torch.randn(B, 3 * S, H, W, generator=gen)
    joints = torch.rand(B, 17, 2, generator=gen) * torch.tensor([W - 32.0, H - 32.0]) + 16.0      # peaks away from the border
    vis = (torch.rand(B, 17, generator=gen) < 0.8).float()
    tr = Trainer(model, lr=1e-3, use_graph=True, targets_from_joints=True)
    args = (kf.to(dev), sup.to(dev), joints.to(dev), vis.to(dev))
    first = None
    for it in range(FIT_STEPS):
        tr.step(*args)
        if it == 0:
            first = tr.loss_value()
    last = tr.loss_value()
    pck_fit = tr.accuracy()[0][1]
    print('fitted model: loss %.5f -> %.5f after %d steps, training PCK %.3f' % (first, last, FIT_STEPS, pck_fit))
    assert last < 0.5 * first and pck_fit >= 0.9, (first, last, pck_fit)        # the heatmaps peak at the joints now
    del tr
    # ground-truth heatmaps (the oracle's generate_heatmaps, pinned by golden g6) for the PCK
    tg = np.zeros((B, 17, H // 4, W // 4), np.float32)
    for b in range(B):
        j3 = np.concatenate([joints[b].numpy(), np.zeros((17, 1), np.float32)], 1)
        v3 = np.repeat(vis[b].numpy()[:, None], 3, 1)
        tg[b], _ = oops.generate_heatmaps(j3, v3, 3, np.array([W, H]), np.array([W // 4, H // 4]), 17)
    seen = vis.numpy() > 0.5
    orc.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    orc.train()
    with torch.no_grad():
        f0, k0, _ = orc(kf, sup)
        sd = {k: v.clone() for k, v in model.state_dict().items()}         # (train-mode forwards advance the running statistics)
        f32hm, _, _ = model(kf.to(dev), sup.to(dev))
        model.load_state_dict(sd)
        model.set_compute_dtype(mode)
        f16hm, _, _ = model(kf.to(dev), sup.to(dev))
    i0, i32, i16 = _argmax(f0), _argmax(f32hm), _argmax(f16hm)
    assert np.array_equal(i32[seen], i0[seen])                               # fp32 HIP path: bit-exact indices on fitted weights
    Wh = W // 4
    d = np.maximum(np.abs(i16 // Wh - i0 // Wh), np.abs(i16 % Wh - i0 % Wh))[seen]      # Chebyshev distance of the peaks, pixels
    exact, near, near2 = float((d == 0).mean()), float((d <= 1).mean()), float((d <= 2).mean())
    _, pck0, _, _ = oops.accuracy(f0.numpy(), tg)
    _, pck16, _, _ = oops.accuracy(f16hm.cpu().numpy(), tg)
    print('fitted-model keypoints %s: %d visible joints, same argmax index as the fp32 oracle %.4f, within one / two heatmap pixels '
          '%.4f / %.4f, largest distance %d px; PCK %.4f vs fp32 %.4f' % (mode, int(seen.sum()), exact, near, near2, int(d.max()), pck16, pck0))
    # A fitted sigma = 3 peak is nearly flat at its top (the neighbour of the maximum is within 5 % of it), and 16-bit storage
    # perturbs every activation of a 300-layer net: the EXACT index is not stable under bf16 (measured 0.25 - 0.4 exact
    # agreement, 0.86 - 1.0 within one pixel over four runs of the same tree -- the fit itself is not run-to-run reproducible:
    # float atomics in the DCN input gradient -- with PCK 1.0 on both sides every time; fp16 0.89 exact, 0.96 - 1.0 within one).
    # What is asserted is what the reference scores a pose estimator by: the PCK against the ground truth must not move, and no
    # peak may wander further than the decode's own refinement plus a pixel (the PCK radius here is 4.8 x 3.6 heatmap pixels).
    assert abs(pck16 - pck0) <= 0.02, (pck16, pck0)
    assert near2 >= 0.95 and d.max() <= 3, (exact, near, near2, int(d.max()))


FIT_STEPS = 150


def test_config5_w64_full_size(dev):
    """BASELINE configs[4]: HRNet-W64 (64/128/256/512 channels, 16 DCN offset groups), 5-frame 384x288.
    (a) fp32 mode against the CPU oracle at the north_star contract: heatmaps <= 1e-3, argmax bit-exact, loss 1e-4,
        head / DCN / backbone gradient norms within 2e-2 and the head's last layer within 1e-3.  The norm band is wide
        because the global-offset regressor's BatchNorm backward is ill-conditioned at this size: it turns 1e-6-level
        rounding differences of the convolutions into 1e-2-level changes of a few gradient tensors, run to run
        (atomic order) as well as path to path -- tools/probes/w64_grad_dev.py on MI355X: largest deviation from the
        oracle 0.7 % with the exact-f32 MFMA convolutions, 0.7-1.0 % (weights) with the split-product ones, whose
        rounding error is 2-2.5x larger (test_split_product_f32_conv_is_as_accurate_as_the_f32_mfma);
    (b) fp16 mode (the config's arithmetic) against the fp16 emulation of the reference graph, as in
        test_half_mode_vs_oracle."""
    S, H, W, B = 4, 384, 288, 1
    orc = om.realistic_init_(om.AlignmentOracle(om.make_cfg(64), True, S, (H, W), dcn_groups=16), 64)
    model = fp.build_model(fp.default_cfg(64, image_size=(W, H), num_sup=S), 'train')
    assert model.G == 16 and model.C == 64
    model.load_state_dict(orc.state_dict())
    model = model.to(dev)
    gen = torch.Generator().manual_seed(164)
    kf, sup = torch.randn(B, 3, H, W, generator=gen), torch.randn(B, 3 * S, H, W, generator=gen)
    tgt = torch.rand(B, 17, H // 4, W // 4, generator=gen)
    w = (torch.rand(B, 17, 1, generator=gen) < 0.8).float()
    f0, k0, mi0 = orc(kf, sup)
    l0 = oops.total_loss(f0, tgt, w, mi0)
    l0.backward()
    from fami_pose_amd.loss import JointMSELoss

    def hip_loss(f, mi):
        return JointMSELoss()(f, tgt.to(dev), w.to(dev)) + 0.5 * (-0.1 * mi[0] + 0.1 * mi[1] + mi[2] - mi[3] + mi[4] - mi[5])
    f1, k1, mi1 = model(kf.to(dev), sup.to(dev))
    assert (f1.cpu() - f0).abs().max().item() < HM_TOL and (k1.cpu() - k0).abs().max().item() < HM_TOL
    assert np.array_equal(_argmax(f1), _argmax(f0.detach())) and np.array_equal(_argmax(k1), _argmax(k0.detach()))
    l1 = hip_loss(f1, mi1)
    assert l1.item() == pytest.approx(l0.item(), rel=1e-4)
    l1.backward()
    ref, mine = dict(orc.named_parameters()), dict(model.named_parameters())
    g0, g1 = ref['agg_final_layer.weight'].grad, mine['agg_final_layer.weight'].grad.cpu()
    assert ((g1 - g0).abs().max() / g0.abs().max()).item() < 1e-3
    bad = []
    for name, p in ref.items():
        if p.grad is None or p.grad.abs().max().item() < 1e-9:
            continue
        a, b = mine[name].grad.double().abs().sum().item(), p.grad.double().abs().sum().item()
        if abs(a - b) > (5e-2 if p.numel() <= 64 else 2e-2) * b:
            bad.append((name, a, b))
    assert not bad, bad[:10]
    # (b) the fp16 arithmetic of the config
    model.zero_grad()
    model.set_compute_dtype('f16')
    emu = _half_emulation(orc, torch.float16)
    with torch.no_grad():
        fe, ke, _ = emu(kf, sup)
    f2, k2, mi2 = model(kf.to(dev), sup.to(dev))
    rms = lambda a, b: ((a - b).norm() / b.norm()).item()
    for hip, emul, refo in ((f2, fe, f0.detach()), (k2, ke, k0.detach())):
        e_hip, e_emu = rms(hip.detach().cpu(), refo), rms(emul, refo)
        assert e_hip <= 1.5 * e_emu + 2e-3, (e_hip, e_emu)
    l2 = hip_loss(f2, mi2)
    assert l2.item() == pytest.approx(l0.item(), rel=0.02)
    (l2 * 4096.0).backward()
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in model.parameters())
    g2 = mine['agg_final_layer.weight'].grad.cpu() / 4096.0
    assert ((g2 - g0).norm() / g0.norm()).item() < 5e-2


def test_ddp_path_single_rank_rccl(dev):
    """The N>1 code path on real hardware with ONE rank: RCCL process group, bucketed async all-reduce fired from the
    backward hooks, scale, Adam -- must reproduce the plain single-GPU step bit for bit (an all-reduce over one rank is
    the identity).  Runs in a CHILD process (tests/_ddp_single_rank.py): creating and destroying an RCCL process group
    inside the long-lived pytest process left the HIP runtime in a state where a hipGraph instantiated by a LATER test
    crashed in hipGraphLaunch (hip::Graph::UpdateStreams; reproducible only with this test earlier in the same process,
    gone without it) -- a training process keeps its one process group for life, so the child mirrors real use."""
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29517')
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ddp_single_rank.py')],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'DDP_SINGLE_RANK_OK' in r.stdout


def test_predict_end_to_end(dev):
    """Inference step (val-phase forward + on-device get_final_preds) against oracle forward + oracle decode."""
    from fami_pose_amd.evaluate import predict
    S, H, W, B = 4, 384, 288, 2
    model, orc = _pair(48, S, (H, W), fp.VAL_PHASE, 77)
    model = model.to(dev)
    orc.eval()
    gen = torch.Generator().manual_seed(78)
    kf, sup = torch.randn(B, 3, H, W, generator=gen), torch.randn(B, 3 * S, H, W, generator=gen)
    center = np.array([[320.0, 240.5], [100.25, 400.0]], np.float32)
    scale = np.array([[1.5, 2.0], [0.9, 1.2]], np.float32)
    with torch.no_grad():
        f0, _ = orc(kf, sup)
    p0, m0 = oops.get_final_preds(f0.numpy().copy(), center, scale)
    preds, maxvals, hm = predict(model, kf.to(dev), sup.to(dev), center, scale)
    assert (hm.cpu() - f0).abs().max().item() < HM_TOL
    assert np.abs(maxvals.cpu().numpy() - m0).max() < HM_TOL
    # same argmax and same quarter-pixel decision on every joint -> image coordinates agree to fp32 rounding
    assert np.abs(preds.cpu().numpy().astype(np.float64) - p0).max() < 1e-2

"""The train step around the model on the MI355X (engine/core/functions/alignment_mi_function_term6_1.py:104-174,
engine/defaults/trainer.py:54-89, engine/defaults/checkpoints.py:45-107, posetimation/optimizer/scheduler.py:14-35):
per-step PCK on the device, checkpoint save / resume into a fresh trainer, the epoch's last (smaller) batch under
graph replay, the LR schedule under graph replay, frozen -> trained -> frozen weight images.

Bitwise comparisons run with `set_deterministic(True)`: the DCN input-gradient scatter then adds 64-bit fixed-point
integers instead of floats, and every other kernel of the step already has a fixed summation order."""
import os

import numpy as np
import pytest
import torch

import fami_pose_amd as fp
from oracle import model as om, ops as oops

pytestmark = pytest.mark.gpu

S, H, W, B = 2, 128, 96, 2


def _model(seed=5, freeze=False, phase='train'):
    orc = om.realistic_init_(om.AlignmentOracle(om.make_cfg(48), phase == 'train', S, (H, W)), seed)
    model = fp.build_model(fp.default_cfg(48, image_size=(W, H), num_sup=S, freeze_backbone=freeze), phase)
    model.load_state_dict(orc.state_dict())
    return model, orc


def _batch(dev, seed=9, b=B):
    gen = torch.Generator().manual_seed(seed)
    kf, sup = torch.randn(b, 3, H, W, generator=gen), torch.randn(b, 3 * S, H, W, generator=gen)
    joints = torch.rand(b, 17, 2, generator=gen) * torch.tensor([W, H], dtype=torch.float32)
    vis = (torch.rand(b, 17, generator=gen) < 0.8).float()
    return tuple(t.to(dev) for t in (kf, sup, joints, vis))


def test_pck_inside_the_step_matches_reference_accuracy(dev):
    """core fn :159-163 calls accuracy(pred, target) and accuracy(kf_bb, target) every iteration on host copies of four
    heatmap stacks; Trainer.step produces the same numbers with three launches each and no copy."""
    from fami_pose_amd.train import Trainer
    model, _ = _model()
    tr = Trainer(model.to(dev), use_graph=False, targets_from_joints=True)
    kf, sup, joints, vis = _batch(dev)
    final, kf_hm = tr.step(kf, sup, joints, vis)[:2]
    tg = np.zeros((B, 17, H // 4, W // 4), np.float32)
    for b in range(B):
        j3 = np.concatenate([joints[b].cpu().numpy(), np.zeros((17, 1), np.float32)], 1)
        v3 = np.repeat(vis[b].cpu().numpy()[:, None], 3, 1)
        tg[b], _ = oops.generate_heatmaps(j3, v3, 3, np.array([W, H]), np.array([W // 4, H // 4]), 17)
    got = tr.accuracy()
    for (acc, avg, cnt), hm in zip(got, (final, kf_hm)):
        a0, avg0, cnt0, _ = oops.accuracy(hm.cpu().numpy(), tg)
        assert cnt == cnt0 and avg == pytest.approx(avg0, abs=1e-6)
        assert np.allclose(acc, a0, atol=1e-6)
    # a crafted case with known answers: perfect prediction, a miss, an invisible joint, an ignored target at x <= 1
    from fami_pose_amd._lib import lib
    hm_t = torch.zeros(1, 4, 24, 18)
    hm_p = torch.zeros(1, 4, 24, 18)
    hm_t[0, 0, 10, 9] = 1.0; hm_p[0, 0, 10, 9] = 0.7           # hit
    hm_t[0, 1, 10, 9] = 1.0; hm_p[0, 1, 20, 2] = 0.9           # miss
    hm_t[0, 2, 5, 1] = 1.0; hm_p[0, 2, 5, 1] = 1.0             # target at x = 1: ignored
    hm_p[0, 3] = -1.0                                          # no target at all (all-zero map -> (0, 0)): ignored
    out = torch.zeros(7, device=dev)
    iws = torch.empty(8, dtype=torch.int64, device=dev)
    mws = torch.empty(8, device=dev)
    hm_pd, hm_td = hm_p.to(dev), hm_t.to(dev)                  # named: a temporary would be freed (and its block reused) before the launch
    lib().call('fami_pck_accuracy_f32', hm_pd.data_ptr(), hm_td.data_ptr(), out.data_ptr(), iws.data_ptr(),
               mws.data_ptr(), 1, 4, 24, 18, 0.5, torch.cuda.current_stream(dev).cuda_stream)
    a0, avg0, cnt0, _ = oops.accuracy(hm_p.numpy(), hm_t.numpy())
    assert out.cpu().tolist() == pytest.approx(list(a0) + [avg0, cnt0])
    assert out.cpu().tolist() == pytest.approx([0.5, 1.0, 0.0, -1.0, -1.0, 0.5, 2.0])


def _run(dev, steps, use_graph, resume_from=None, save_at=None, tmp=None, batches=None):
    from fami_pose_amd import checkpoint as ck
    from fami_pose_amd.train import Trainer
    model, _ = _model()
    model = model.to(dev).set_deterministic(True)
    tr = Trainer(model, lr=1e-3, use_graph=use_graph, targets_from_joints=True)
    start = 0
    if resume_from is not None:
        _, _, start = ck.resume(model, tr, resume_from)
    losses = []
    for k in range(start, steps):
        tr.step(*batches[k])
        losses.append(tr.loss_parts.clone())
        if save_at is not None and k + 1 == save_at:
            ck.save_checkpoint(k, tmp, model, tr)       # begin_epoch = k -> resume continues at k + 1
    return tr, losses


def test_deterministic_steps_and_checkpoint_resume(dev, tmp_path):
    """(a) two identical runs are bitwise identical (deterministic DCN backward); (b) hipGraph replay is bitwise the
    eager launch sequence; (c) train 2 steps, save, resume into a FRESH model + Trainer, take step 3: parameters, Adam
    moments, BatchNorm buffers and loss equal the uninterrupted 3-step run bit for bit (checkpoints.py:45-107)."""
    from fami_pose_amd import checkpoint as ck
    batches = [_batch(dev, 9 + k) for k in range(3)]
    tr_a, la = _run(dev, 3, False, batches=batches)
    tr_b, lb = _run(dev, 3, False, batches=batches)
    assert torch.equal(tr_a.flat, tr_b.flat) and all(torch.equal(x, y) for x, y in zip(la, lb))
    tr_g, lg = _run(dev, 3, True, batches=batches)
    assert torch.equal(tr_a.flat, tr_g.flat) and all(torch.equal(x, y) for x, y in zip(la, lg))
    # the capture warm-up is rolled back completely: three steps -> counters at 3 (one BN call per step for the stem)
    assert int(tr_g.model.hrnet.bn1.num_batches_tracked) == 3 == int(tr_a.model.hrnet.bn1.num_batches_tracked)
    assert int(tr_g.model.feat_global_offset_layers[1].bn.num_batches_tracked) == 3 * S

    folder = str(tmp_path / 'ckpt')
    _run(dev, 2, False, save_at=2, tmp=folder, batches=batches)
    path = ck.get_latest_checkpoint(folder)
    assert os.path.basename(path) == 'epoch_1_state.pth'
    for use_graph in (False, True):
        tr_r, lr_ = _run(dev, 3, use_graph, resume_from=path, batches=batches)
        assert len(lr_) == 1 and torch.equal(lr_[0], la[2])
        assert torch.equal(tr_r.flat, tr_a.flat)
        assert torch.equal(tr_r.opt.m, tr_a.opt.m) and torch.equal(tr_r.opt.v, tr_a.opt.v)
        assert torch.equal(tr_r.opt.state, tr_a.opt.state)
        for (n1, b1), (n2, b2) in zip(tr_r.model.named_buffers(), tr_a.model.named_buffers()):
            assert n1 == n2 and torch.equal(b1, b2), n1


def test_smaller_last_batch_under_graph_replay(dev):
    """The reference DataLoader has no drop_last: the last batch of an epoch is smaller.  A graph-mode Trainer keeps
    one captured plan per batch shape; the sequence B=2, B=1, B=2 must equal the eager sequence bit for bit."""
    from fami_pose_amd.train import Trainer
    seq = [_batch(dev, 20, 2), _batch(dev, 21, 1), _batch(dev, 22, 2)]
    res = []
    for use_graph in (False, True):
        model, _ = _model()
        tr = Trainer(model.to(dev).set_deterministic(True), use_graph=use_graph, targets_from_joints=True)
        ls = []
        for bt in seq:
            out = tr.step(*bt)
            assert out[0].shape[0] == bt[0].shape[0]
            ls.append(tr.loss_parts.clone())
        res.append((tr.flat.clone(), ls))
        if use_graph:
            assert len(tr._cache) == 2
    assert torch.equal(res[0][0], res[1][0]) and all(torch.equal(x, y) for x, y in zip(res[0][1], res[1][1]))
    with pytest.raises(ValueError):
        tr.step(seq[0][0], seq[1][1], seq[0][2], seq[0][3])


def test_no_garbage_collection_inside_a_capture(dev):
    """A Trainer that went out of scope is a reference cycle (its Engine's callbacks point back at it), so its hipGraph is destroyed
    by the cyclic collector -- and a graph destroyed while a stream captures aborts the process (the default bench run: f32 Trainer,
    then the bf16 Trainer's capture).  Trainer._capture collects first and keeps the collector off until the capture has ended."""
    import gc
    from fami_pose_amd.train import Trainer
    bt = _batch(dev, 30, 1)
    model, _ = _model()
    tr = Trainer(model.to(dev), use_graph=True, targets_from_joints=True)
    tr.step(*bt)
    del tr, model                                  # (left to the collector)
    model, _ = _model()
    tr = Trainer(model.to(dev), use_graph=True, targets_from_joints=True)
    seen = []
    inner = tr._forward_backward

    def spy(*a, **k):
        seen.append((torch.cuda.is_current_stream_capturing(), gc.isenabled()))
        return inner(*a, **k)
    tr._forward_backward = spy
    assert gc.isenabled()
    tr.step(*bt)
    assert gc.isenabled()
    assert (True, False) in seen and (True, True) not in seen


def test_lr_schedule_drives_captured_adam(dev):
    """MultiStepLR (scheduler.py:14-35; TRAIN.LR_STEP / LR_FACTOR) writes the device-resident learning rate: the captured
    hipGraph keeps replaying and the very next update is gamma times smaller.  Changing betas re-captures."""
    from fami_pose_amd.train import MultiStepLR, Trainer
    model, _ = _model()
    tr = Trainer(model.to(dev).set_deterministic(True), lr=1e-3, use_graph=True, targets_from_joints=True)
    sched = MultiStepLR(tr, [1, 2], 0.1)
    bt = _batch(dev)
    w = model.agg_final_layer.weight
    deltas = []
    for epoch in range(3):
        before = w.detach().clone()
        tr.step(*bt)
        deltas.append((w.detach() - before).abs().max().item())
        sched.step()
    # Adam's update is ~lr in magnitude for its first steps
    assert deltas[0] == pytest.approx(1e-3, rel=0.05)
    assert deltas[1] == pytest.approx(1e-4, rel=0.35) and deltas[2] == pytest.approx(1e-5, rel=0.5)
    assert tr.opt.state[1].item() == pytest.approx(1e-5, rel=1e-5) and len(tr._cache) == 1
    plan0 = list(tr._cache.values())[0][0]
    tr.opt.set_hyper(betas=(0.8, 0.99))
    tr.step(*bt)
    assert list(tr._cache.values())[0][0] is not plan0          # re-captured with the new kernel arguments


def test_frozen_weight_image_is_not_reused_after_training(dev):
    """freeze -> forward (packed images cached on the frozen parameters) -> unfreeze -> train -> freeze -> forward must
    use the trained weights (the cache key did not see in-place arena updates)."""
    from fami_pose_amd.train import Trainer
    model, _ = _model()
    model = model.to(dev).set_deterministic(True)
    kf, sup, joints, vis = _batch(dev)
    tr = Trainer(model, lr=1e-2, use_graph=False, targets_from_joints=True)     # parameters move into the arena
    model.hrnet.freeze_weight()
    with torch.no_grad():
        y0 = model(kf, sup)[0].clone()
    for p in model.hrnet.parameters():
        p.requires_grad = True
    for _ in range(2):
        tr.step(kf, sup, joints, vis)
    model.hrnet.freeze_weight()
    with torch.no_grad():
        y1 = model(kf, sup)[0].clone()
    fresh, _ = _model()
    fresh.load_state_dict(model.state_dict())
    fresh = fresh.to(dev).set_deterministic(True)         # same launch plan as `model` (bitwise comparison below) ...
    Trainer(fresh, use_graph=False, targets_from_joints=True)   # ... incl. the arena layout: adjacent predictor weights run as one convolution
    fresh.hrnet.freeze_weight()
    with torch.no_grad():
        y2 = fresh(kf, sup)[0]
    assert (y1 - y0).abs().max().item() > 1e-4           # training moved the output
    assert torch.equal(y1, y2)                           # and the re-frozen model runs on the trained weights


def test_evaluate_loop_accumulates_like_the_reference(dev, tmp_path):
    """The validation loop (core fn :222-328) on the HIP path: decode + PCK on device, accumulation arrays as the
    reference fills them, then the PoseTrack JSON files -- against oracle forward + oracle decode + oracle accuracy."""
    import json
    from fami_pose_amd import evaluate as ev
    model, orc = _model(seed=7, phase=fp.VAL_PHASE)
    model = model.to(dev)
    orc.eval()
    gen = torch.Generator().manual_seed(3)
    batches, want_preds, want_acc = [], [], [0.0, 0]
    names_all = ['/d/images/bonn/000001_bonn/0000000%d.jpg' % i for i in (1, 2, 4)]
    k = 0
    for b in (2, 1):
        kf, sup = torch.randn(b, 3, H, W, generator=gen), torch.randn(b, 3 * S, H, W, generator=gen)
        tgt = torch.rand(b, 17, H // 4, W // 4, generator=gen)
        meta = {'image': names_all[k:k + b], 'center': np.random.RandomState(k).uniform(100, 300, (b, 2)).astype(np.float32),
                'scale': np.random.RandomState(k + 9).uniform(0.8, 2.0, (b, 2)).astype(np.float32),
                'score': np.random.RandomState(k + 5).uniform(0.3, 1.0, b)}
        k += b
        batches.append((kf, sup, tgt, meta))
        with torch.no_grad():
            f0, _ = orc(kf, sup)
        p0, m0 = oops.get_final_preds(f0.numpy().copy(), meta['center'], meta['scale'])
        want_preds.append(np.concatenate([p0, m0], 2))
        _, avg, cnt, _ = oops.accuracy(f0.numpy(), tgt.numpy())
        want_acc[0] += avg * cnt
        want_acc[1] += cnt
    acc = ev.evaluate_loop(model, batches, 3)
    want = np.concatenate(want_preds)
    assert np.abs(acc.all_preds[:, :, :2] - want[:, :, :2]).max() < 1e-2 and np.abs(acc.all_preds[:, :, 2] - want[:, :, 2]).max() < 1e-3
    assert acc.acc_cnt[0] == want_acc[1] and acc.accuracy(0) == pytest.approx(want_acc[0] / want_acc[1], abs=1e-6)
    assert list(acc.filenames_map) == names_all and acc.idx == 3
    annot = tmp_path / 'annot'
    annot.mkdir()
    for nm in ('000001_bonn.json', 'other.json'):
        (annot / nm).write_text(json.dumps({'images': [{'file_name': 'images/bonn/%s/00000001.jpg' % nm[:-5], 'nframes': 4}]}))
    written = ev.write_posetrack_results(acc.all_preds, acc.all_boxes, acc.filenames_map, str(annot), str(tmp_path / 'res'))
    (path,) = written
    data = json.load(open(path))['annolist']
    assert [el['imgnum'][0] for el in data] == [1, 2, 3, 4] and data[2]['annorect'][0]['score'] == [0]      # frame 3: dummy


def test_bench_workload_train_step_matches_the_oracle(dev):
    """The headline workload itself (bench.py: HRNet-W48, 4 supporting frames, 384x288, batch 4 per GPU, train phase with
    MI, on-device Gaussian targets, fp32): one Trainer.step against the CPU oracle's forward on the same inputs --
    heatmaps <= 1e-3 absolute (north_star), argmax keypoint indices bit-exact, loss to 1e-4 relative -- and, with the
    deterministic DCN backward, two hipGraph-replayed steps bitwise equal to two eager steps (parameters and both Adam
    moments)."""
    from fami_pose_amd.train import Trainer
    Sx, Hx, Wx, Bx = 4, 384, 288, 4
    orc = om.realistic_init_(om.AlignmentOracle(om.make_cfg(48), True, Sx, (Hx, Wx)), 31)
    gen = torch.Generator().manual_seed(19970808)
    kf, sup = torch.randn(Bx, 3, Hx, Wx, generator=gen), torch.randn(Bx, 3 * Sx, Hx, Wx, generator=gen)
    joints = torch.rand(Bx, 17, 2, generator=gen) * torch.tensor([Wx, Hx], dtype=torch.float32)
    vis = (torch.rand(Bx, 17, generator=gen) < 0.8).float()
    tg = np.zeros((Bx, 17, Hx // 4, Wx // 4), np.float32)
    tw = np.zeros((Bx, 17, 1), np.float32)
    for b in range(Bx):
        j3 = np.concatenate([joints[b].numpy(), np.zeros((17, 1), np.float32)], 1)
        v3 = np.repeat(vis[b].numpy()[:, None], 3, 1)
        tg[b], tw[b] = oops.generate_heatmaps(j3, v3, 3, np.array([Wx, Hx]), np.array([Wx // 4, Hx // 4]), 17)
    orc.zero_grad()
    f0, k0, mi0 = orc(kf, sup)
    l0t = oops.total_loss(f0, torch.from_numpy(tg), torch.from_numpy(tw), mi0)
    l0t.backward()                      # the reference path's fp32 gradients at the full size (keep one)
    l0 = l0t.item()
    f0, k0 = f0.detach(), k0.detach()
    args = tuple(t.to(dev) for t in (kf, sup, joints, vis))

    def trainer(use_graph):
        m = fp.build_model(fp.default_cfg(48, image_size=(Wx, Hx), num_sup=Sx), 'train')
        m.load_state_dict(orc.state_dict())
        m.set_deterministic(True)
        return Trainer(m.to(dev), lr=1e-3, use_graph=use_graph, targets_from_joints=True)

    te = trainer(False)
    outs = te.step(*args)
    final, kf_hm = outs[0].cpu(), outs[1].cpu()
    assert (final - f0).abs().max().item() < 1e-3 and (kf_hm - k0).abs().max().item() < 1e-3
    am = lambda t: t.reshape(Bx, 17, -1).argmax(2).numpy()
    assert np.array_equal(am(final), am(f0)) and np.array_equal(am(kf_hm), am(k0))
    assert te.loss_value() == pytest.approx(l0, rel=1e-4)
    # Full-resolution gradients of this very step against the oracle's fp32 backward: every parameter's gradient norm to
    # 2e-2 (fp32 backward through ~100 train-mode BatchNorms is ill-conditioned: the CPU path itself sits 1e-2 from an
    # fp64 evaluation deep in the net, test_model_vs_oracle arbitrates that at a size fp64 can afford), and, element by
    # element, the head's output-side parameters to 5e-3 of the gradient's maximum (the DCN input-gradient scatter adds in
    # run-dependent order; measured 1e-3 .. 3e-3).
    mine = {n: te.views[id(p)] for n, p in te.model.named_parameters() if id(p) in te.views}
    ref = dict(orc.named_parameters())
    bad = []
    for n, gv in mine.items():
        g0 = ref[n].grad
        if g0 is None or g0.abs().max().item() < 1e-9:
            continue
        want, got = g0.double().abs().sum().item(), gv.double().abs().sum().item()
        if abs(got - want) > 2e-2 * want:
            bad.append((n, got, want))
    assert len(mine) > 900 and not bad, bad[:20]
    # agg_final_layer has no BatchNorm behind it (5e-3 of the gradient's maximum); the DCN layers and their predictors sit in
    # front of three train-mode BasicBlocks and take the scatter's run-dependent summation order: 2e-2 (measured 1e-3 .. 7e-3)
    for n, tol in (('agg_final_layer.weight', 5e-3), ('agg_final_layer.bias', 5e-3), ('dcn_4.weight', 2e-2), ('dcn_3.weight', 2e-2),
                   ('dcn_offset_4.conv.weight', 2e-2), ('dcn_mask_4.conv.weight', 2e-2)):
        g0 = ref[n].grad
        assert ((mine[n].cpu() - g0).abs().max() / g0.abs().max()).item() < tol, n
    te.step(*args)
    tg_ = trainer(True)
    for _ in range(2):
        tg_.step(*args)
    torch.cuda.synchronize(dev)
    assert torch.equal(te.flat, tg_.flat) and torch.equal(te.opt.m, tg_.opt.m) and torch.equal(te.opt.v, tg_.opt.v)


def test_config2_train_step_as_stated_matches_the_oracle(dev):
    """BASELINE config 2 AS STATED -- HRNet-W48, 384x288, 3-frame clips (S = 2, the generalised head), batch 8, fp32, train mode
    (batch statistics over 24 frames): one Trainer.step against the CPU oracle + torch.optim.Adam on the same inputs -- heatmaps
    <= 1e-3 absolute, argmax keypoint indices bit-exact, loss to 1e-4 relative, the output layer's gradient to 1e-3 of its
    maximum and its Adam update to 1e-5, the stem BatchNorm's running mean to 1e-5 (round-5 review: the 141 clips/s record of
    this config was guarded by a finite-loss check alone)."""
    from fami_pose_amd.train import Trainer
    Sx, Hx, Wx, Bx = 2, 384, 288, 8
    orc = om.realistic_init_(om.AlignmentOracle(om.make_cfg(48), True, Sx, (Hx, Wx)), 37)
    gen = torch.Generator().manual_seed(19970808 + 2)
    kf, sup = torch.randn(Bx, 3, Hx, Wx, generator=gen), torch.randn(Bx, 3 * Sx, Hx, Wx, generator=gen)
    joints = torch.rand(Bx, 17, 2, generator=gen) * torch.tensor([Wx, Hx], dtype=torch.float32)
    vis = (torch.rand(Bx, 17, generator=gen) < 0.8).float()
    tg = np.zeros((Bx, 17, Hx // 4, Wx // 4), np.float32)
    tw = np.zeros((Bx, 17, 1), np.float32)
    for b in range(Bx):
        j3 = np.concatenate([joints[b].numpy(), np.zeros((17, 1), np.float32)], 1)
        v3 = np.repeat(vis[b].numpy()[:, None], 3, 1)
        tg[b], tw[b] = oops.generate_heatmaps(j3, v3, 3, np.array([Wx, Hx]), np.array([Wx // 4, Hx // 4]), 17)
    m = fp.build_model(fp.default_cfg(48, image_size=(Wx, Hx), num_sup=Sx), 'train')
    m.load_state_dict(orc.state_dict())
    opt = torch.optim.Adam(orc.parameters(), lr=1e-3)
    opt.zero_grad()
    f0, k0, mi0 = orc(kf, sup)
    l0t = oops.total_loss(f0, torch.from_numpy(tg), torch.from_numpy(tw), mi0)
    l0t.backward()
    g_ref = orc.agg_final_layer.weight.grad.clone()
    p_before = orc.agg_final_layer.weight.data.clone()
    opt.step()
    f0, k0 = f0.detach(), k0.detach()

    tr = Trainer(m.to(dev), lr=1e-3, use_graph=False, targets_from_joints=True)
    outs = tr.step(*(t.to(dev) for t in (kf, sup, joints, vis)))
    final, kf_hm = outs[0].cpu(), outs[1].cpu()
    assert (final - f0).abs().max().item() < 1e-3 and (kf_hm - k0).abs().max().item() < 1e-3
    am = lambda t: t.reshape(Bx, 17, -1).argmax(2).numpy()
    assert np.array_equal(am(final), am(f0)) and np.array_equal(am(kf_hm), am(k0))
    assert tr.loss_value() == pytest.approx(l0t.item(), rel=1e-4)
    g1 = tr.views[id(m.agg_final_layer.weight)].cpu()
    assert ((g1 - g_ref).abs().max() / g_ref.abs().max()).item() < 1e-3
    big = g_ref.abs() > 1e-3 * g_ref.abs().max()      # Adam's first step is lr * g / (|g| + eps): compare above the noise floor
    upd0 = orc.agg_final_layer.weight.data - p_before
    upd1 = m.agg_final_layer.weight.data.cpu() - p_before
    assert (upd1 - upd0)[big].abs().max().item() < 1e-5
    assert (orc.hrnet.bn1.running_mean - m.hrnet.bn1.running_mean.cpu()).abs().max().item() < 1e-5


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_merged_predictors_match_the_two_convolution_path(dev, mode, monkeypatch):
    """The offset and the mask predictor of every DCN layer (Alignment_V15.py:79-100, both applied to the same tensor at :144-158)
    run as ONE 48 -> 324 dilated convolution inside Trainer.step (engine.CatParam over weights the flat arena keeps adjacent; the
    DCN kernels read offsets and masks from one tensor).  Against the two-convolution path (FAMI_MERGE_PREDICTORS=0): the same
    loss and the same gradients on every predictor / DCN / upstream parameter up to the summation order of other kernel
    routes, twelve convolution-family launches and four bias sums fewer, identical state_dict layout."""
    from fami_pose_amd.train import Trainer
    from fami_pose_amd._lib import lib
    bt = _batch(dev, 41)
    grads, losses, calls = [], [], []
    for merge in ('1', '0'):
        monkeypatch.setenv('FAMI_MERGE_PREDICTORS', merge)
        model, _ = _model(7)
        model = model.to(dev).set_deterministic(True).set_compute_dtype(mode)
        tr = Trainer(model, use_graph=False, targets_from_joints=True)
        assert len(tr.cats) == 8                                           # (built either way: the arena layout does not depend on the switch)
        n0 = lib().ncalls
        tr.step(*bt)
        calls.append(lib().ncalls - n0)
        losses.append(tr.loss_value())
        grads.append({n: tr.views[id(p)].clone() for n, p in model.named_parameters() if id(p) in tr.views})
    assert calls[1] - calls[0] >= 12, calls
    tol = 2e-5 if mode == 'f32' else 3e-2
    assert losses[0] == pytest.approx(losses[1], rel=1e-6 if mode == 'f32' else 1e-3)
    for name in ('dcn_offset_1.conv.weight', 'dcn_mask_1.conv.weight', 'dcn_offset_3.conv.bias', 'dcn_mask_4.conv.bias',
                 'dcn_mask_4.conv.weight', 'dcn_2.weight', 'combined_feat_layers.layers.0.conv1.weight',
                 'sup_agg_block.layers.0.conv1.weight'):
        a, b = grads[0][name], grads[1][name]
        assert ((a - b).norm() / b.norm()).item() < tol, name

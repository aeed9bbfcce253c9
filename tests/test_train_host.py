"""Host logic of the train step that needs no GPU: the LR schedule (posetimation/optimizer/scheduler.py:14-35 builds a
torch MultiStepLR from TRAIN.LR_STEP / TRAIN.LR_FACTOR; engine/defaults/trainer.py steps it once per epoch) and the
flat Adam's hyper-parameter versioning that makes a graph-mode Trainer re-capture."""
import pytest
import torch

from fami_pose_amd.train import FlatAdam, MultiStepLR


@pytest.mark.parametrize('milestones,gamma,last_epoch', [([8, 12, 16], 0.1, -1), ([2, 3], 0.5, -1), ([1], 0.1, -1)])
def test_multistep_lr_follows_torch(milestones, gamma, last_epoch):
    p = torch.nn.Parameter(torch.zeros(4))
    ref_opt = torch.optim.Adam([p], lr=1e-3)
    ref = torch.optim.lr_scheduler.MultiStepLR(ref_opt, milestones, gamma, last_epoch=last_epoch)
    opt = FlatAdam(torch.zeros(4), lr=1e-3)
    mine = MultiStepLR(opt, milestones, gamma, last_epoch=last_epoch)
    for epoch in range(21):                      # TRAIN.END_EPOCH = 21 in configs/Alignment/Base_PoseTrack17.yaml
        assert mine.get_last_lr()[0] == pytest.approx(ref.get_last_lr()[0], rel=1e-9), epoch
        assert opt.state[1].item() == pytest.approx(ref.get_last_lr()[0], rel=1e-6)      # the device-resident copy (fp32)
        assert opt.lr == mine.get_last_lr()[0]
        ref_opt.step()
        ref.step()
        mine.step()
    assert mine.last_epoch == ref.last_epoch


def test_multistep_lr_resume_and_state_dict():
    opt = FlatAdam(torch.zeros(4), lr=1e-3)
    a = MultiStepLR(opt, [8, 12, 16], 0.1)
    for _ in range(13):
        a.step()
    sd = a.state_dict()
    opt2 = FlatAdam(torch.zeros(4), lr=1e-3)
    b = MultiStepLR(opt2, [1], 0.5)
    b.load_state_dict(sd)
    assert b.get_last_lr() == a.get_last_lr() and opt2.lr == pytest.approx(1e-5)
    # resuming at an epoch (trainer.py passes last_epoch = begin_epoch - 1 on resume): same closed form
    opt3 = FlatAdam(torch.zeros(4), lr=1e-3)
    c = MultiStepLR(opt3, [8, 12, 16], 0.1, last_epoch=12)
    assert c.last_epoch == 13 and c.get_last_lr()[0] == pytest.approx(1e-5)


def test_flat_adam_hyper_version():
    opt = FlatAdam(torch.zeros(4), lr=1e-3)
    v = opt.hyper_version
    opt.set_hyper(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)       # unchanged: captured graphs stay valid
    assert opt.hyper_version == v
    opt.set_lr(5e-4)                                                    # lr lives in device memory: no re-capture
    assert opt.hyper_version == v and opt.state[1].item() == pytest.approx(5e-4)
    opt.set_hyper(eps=1e-6)
    assert opt.hyper_version == v + 1 and opt.eps == 1e-6


def test_weight_packer_orders_forward_images_first():
    """WeightPacker splits its batch launch by orientation (forward images packed before the forward pass, input-gradient
    images on a side lane beside it): the descriptor array must hold every mode-0 record before the first mode-1 record,
    the two halves must cover every conv weight once each, and the views must tile the arena without overlap."""
    import numpy as np
    from fami_pose_amd.train import WeightPacker
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 16, 3), torch.nn.BatchNorm2d(16), torch.nn.Conv2d(16, 48, 3, 2, 1),
                                torch.nn.Conv2d(48, 17, 1), torch.nn.Linear(4, 2))
    table, off = [], 0
    for p in model.parameters():
        table.append((p, off, p.numel()))
        off += p.numel()
    flat = torch.zeros(off)
    pk = WeightPacker(model, flat, table, torch.float32)
    desc = np.frombuffer(pk.desc.numpy().tobytes(), dtype=[('src', '<i8'), ('dst', '<i8'), ('Co', '<i4'), ('Ci', '<i4'),
                                                             ('taps', '<i4'), ('mode', '<i4')])
    assert pk.n == 6 and pk.n_fwd == 3
    assert list(desc['mode']) == [0, 0, 0, 1, 1, 1]
    convs = [m for m in model if isinstance(m, torch.nn.Conv2d)]
    for half in (desc[:3], desc[3:]):
        assert sorted(zip(half['Co'], half['Ci'], half['taps'])) == sorted((c.out_channels, c.in_channels, c.kernel_size[0] ** 2) for c in convs)
    spans = sorted((v.storage_offset(), v.numel()) for v in pk.views.values())
    assert spans[0][0] == 0 and all(a + n == b for (a, n), (b, _) in zip(spans, spans[1:]))
    assert spans[-1][0] + spans[-1][1] == pk.arena.numel() and len(pk.views) == 6


@pytest.mark.parametrize('N,Ho,Wo,k,pad,dil,MT', [(2, 12, 9, 3, 1, 1, 1), (1, 7, 5, 3, 1, 1, 2), (3, 6, 4, 3, 1, 1, 1), (2, 9, 8, 5, 2, 1, 1),
                                                  (1, 1, 6, 3, 1, 1, 1), (2, 10, 7, 3, 3, 3, 1)])
def test_parity_class_tiling_of_the_stride2_input_gradient(N, Ho, Wo, k, pad, dil, MT):
    """Index arithmetic of ConvArgs.par (conv.hip, stride-2 dgrad by parity class), restated in numpy: the class-major
    tile order must cover every output pixel exactly once, and the taps a class walks must be exactly the taps that can
    reach a pixel of that class ((y + pad - ky*dil) and (x + pad - kx*dil) even) -- so skipping the others drops nothing."""
    import numpy as np
    seen = np.zeros((N, Ho, Wo), dtype=int)
    H0, H1, W0, W1 = (Ho + 1) // 2, Ho // 2, (Wo + 1) // 2, Wo // 2
    tiles = []
    for c in range(4):
        Ha, Wb = (H1 if c >> 1 else H0), (W1 if c & 1 else W0)
        tiles.append(-(-N * Ha * Wb // (MT * 16)))
    for tix in range(sum(tiles)):
        rem, c = tix, 0
        while rem >= tiles[c]:
            rem -= tiles[c]
            c += 1
        ca, cb = c >> 1, c & 1
        Ha, Wb = (H1 if ca else H0), (W1 if cb else W0)
        mine = [t for t in range(k * k) if ((ca + pad - (t // k) * dil) | (cb + pad - (t % k) * dil)) & 1 == 0]
        for ml in range(rem * MT * 16, (rem + 1) * MT * 16):
            if ml >= N * Ha * Wb:
                continue
            n, r = divmod(ml, Ha * Wb)
            yy, xx = divmod(r, Wb)
            y, x = 2 * yy + ca, 2 * xx + cb
            seen[n, y, x] += 1
            reach = [t for t in range(k * k) if (y + pad - (t // k) * dil) % 2 == 0 and (x + pad - (t % k) * dil) % 2 == 0]
            assert reach == mine, (y, x, reach, mine)
    assert (seen == 1).all()


def test_resumed_schedule_does_not_decay_twice(tmp_path):
    """a checkpoint written after the first milestone carries the DECAYED lr; the resumed MultiStepLR must scale
    the stored `initial_lr`, as torch's does (KeyError there when the key is missing), not the decayed one."""
    from types import SimpleNamespace
    from fami_pose_amd.checkpoint import adam_state_dict, load_adam_state_dict
    p = torch.nn.Parameter(torch.zeros(4))
    ref_opt = torch.optim.Adam([p], lr=1e-3)
    ref = torch.optim.lr_scheduler.MultiStepLR(ref_opt, [8, 12, 16], 0.1)
    opt = FlatAdam(torch.zeros(4), lr=1e-3)
    sched = MultiStepLR(opt, [8, 12, 16], 0.1)
    for _ in range(10):
        ref_opt.step()
        ref.step()
        sched.step()
    tr = SimpleNamespace(opt=opt, table=[(p, 0, 4)])
    sd = adam_state_dict(tr)
    assert sd['param_groups'][0]['lr'] == pytest.approx(1e-4) and sd['param_groups'][0]['initial_lr'] == pytest.approx(1e-3)
    # the reference resumes with torch's scheduler on OUR checkpoint: it needs `initial_lr`
    p2 = torch.nn.Parameter(torch.zeros(4))
    ref_opt2 = torch.optim.Adam([p2], lr=1e-3)
    ref_sd = ref_opt2.state_dict()
    ref_sd['param_groups'][0].update(lr=sd['param_groups'][0]['lr'], initial_lr=sd['param_groups'][0]['initial_lr'])
    ref_opt2.load_state_dict(ref_sd)
    ref2 = torch.optim.lr_scheduler.MultiStepLR(ref_opt2, [8, 12, 16], 0.1, last_epoch=10)
    # our resume
    opt2 = FlatAdam(torch.zeros(4), lr=1e-3)
    load_adam_state_dict(SimpleNamespace(opt=opt2, table=[(p, 0, 4)]), sd)
    assert opt2.lr == pytest.approx(1e-4) and opt2.initial_lr == pytest.approx(1e-3)
    mine = MultiStepLR(opt2, [8, 12, 16], 0.1, last_epoch=10)
    for epoch in range(11, 21):
        # rel 1e-6: the checkpoint's lr went through the optimizer's device-resident fp32 state
        assert mine.get_last_lr()[0] == pytest.approx(ref2.get_last_lr()[0], rel=1e-6), epoch
        ref_opt2.step()
        ref2.step()
        mine.step()
    # a torch-written checkpoint that has an initial_lr (the reference attaches its scheduler before saving)
    osd = ref_opt.state_dict()
    assert 'initial_lr' in osd['param_groups'][0]
    opt3 = FlatAdam(torch.zeros(4), lr=1e-3)
    load_adam_state_dict(SimpleNamespace(opt=opt3, table=[(p, 0, 4)]), osd)
    assert MultiStepLR(opt3, [8, 12, 16], 0.1, last_epoch=10).get_last_lr()[0] == pytest.approx(1e-4)


def test_flat_arena_keeps_merged_predictor_weights_adjacent():
    """The offset and the mask predictor of a DCN layer (Alignment_V15.py:79-100) run as ONE convolution on the HIP path
    (engine.CatParam): flatten_parameters must lay the arena out so that the two weights (and the two biases) are adjacent, the
    merged view must equal torch.cat of the parts, and the parameter TABLE must stay in model.parameters() order --
    torch.optim.Adam numbers its state that way, so reference checkpoints keep loading (checkpoint.py)."""
    import fami_pose_amd as fp
    from fami_pose_amd.train import flatten_parameters
    model = fp.build_model(fp.default_cfg(48, image_size=(96, 128), num_sup=2), 'train')
    torch.manual_seed(3)
    for p in model.parameters():
        p.data.normal_()
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    cats = model.merged_predictors()
    assert not any(c.adjacent() for pair in cats.values() for c in pair)      # parameters allocated one by one: two-conv path
    flat, table = flatten_parameters(model)
    assert [id(p) for p, _, _ in table] == [id(p) for p in model.parameters() if p.requires_grad]
    assert sorted(o for _, o, _ in table) != [o for _, o, _ in table]         # (the layout did move something)
    spans = sorted((o, o + n) for _, o, n in table)
    assert spans[0][0] == 0 and spans[-1][1] == flat.numel() and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for n, p in model.named_parameters():
        assert torch.equal(p.detach(), before[n]), n                          # values survive the move
    G = model.G
    for k, (wc, bc) in cats.items():
        assert wc.adjacent() and bc.adjacent()
        assert wc.shape == (27 * G, 48, 3, 3) and bc.shape == (27 * G,)
        off, msk = getattr(model, 'dcn_offset_%d' % k).conv, getattr(model, 'dcn_mask_%d' % k).conv
        assert torch.equal(wc.data, torch.cat([off.weight.data, msk.weight.data], 0))
        assert torch.equal(bc.data, torch.cat([off.bias.data, msk.bias.data], 0))
        assert wc.data.data_ptr() == off.weight.data_ptr() and wc.requires_grad
    # state_dict is untouched by all of this
    assert set(model.state_dict().keys()) >= {'dcn_offset_1.conv.weight', 'dcn_mask_1.conv.weight', 'dcn_offset_4.conv.bias'}
    # a frozen part: no merged view
    model.dcn_mask_2.conv.weight.requires_grad_(False)
    assert not cats[2][0].adjacent()

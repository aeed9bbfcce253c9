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

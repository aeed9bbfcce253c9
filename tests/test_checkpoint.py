"""Checkpoint format compatibility with the reference (engine/defaults/checkpoints.py:17-107): a reference-format
file written from the oracle + torch.optim.Adam loads into the product model + flat Adam and round-trips.  CPU only
(host logic; FlatAdam's arenas are plain tensors, no kernel is launched)."""
import os

import torch

import fami_pose_amd as fp
from fami_pose_amd import checkpoint as ck
from fami_pose_amd.train import FlatAdam, flatten_parameters
from oracle import model as om


class _HostTrainer:
    """The two attributes checkpoint.py needs from train.Trainer, without a GPU."""

    def __init__(self, model):
        self.flat, self.table = flatten_parameters(model)
        self.opt = FlatAdam(self.flat, lr=1e-3)


def test_reference_format_round_trip(tmp_path):
    orc = om.realistic_init_(om.AlignmentOracle(om.make_cfg(48), True, 4, (384, 288)), 3)
    opt = torch.optim.Adam([p for p in orc.parameters() if p.requires_grad], lr=1e-3)
    g = torch.Generator().manual_seed(0)
    for p in orc.parameters():
        p.grad = torch.randn(p.shape, generator=g) * 1e-2
    opt.step()
    opt.step()
    # what the reference's save_checkpoint writes
    path = str(tmp_path / 'epoch_7_state.pth')
    torch.save({'begin_epoch': 7, 'state_dict': {'module.' + k: v for k, v in orc.state_dict().items()},
                'optimizer': [opt.state_dict()]}, path)
    torch.save({}, str(tmp_path / 'epoch_3_state.pth'))
    assert ck.get_latest_checkpoint(str(tmp_path)) == path

    model = fp.build_model(fp.default_cfg(48), fp.TRAIN_PHASE)
    tr = _HostTrainer(model)
    _, _, begin = ck.resume(model, tr, path)
    assert begin == 8
    sd = model.state_dict()
    for k, v in orc.state_dict().items():
        assert torch.equal(sd[k], v), k
    assert model.hrnet.conv1.weight.data_ptr() == tr.flat.data_ptr()      # parameters still live in the arena
    osd = opt.state_dict()
    for i, (p, off, n) in enumerate(tr.table):
        assert torch.equal(tr.opt.m[off:off + n].view(p.shape), osd['state'][i]['exp_avg'])
        assert torch.equal(tr.opt.v[off:off + n].view(p.shape), osd['state'][i]['exp_avg_sq'])
    assert tr.opt.state[0].item() == 2.0 and abs(tr.opt.state[2].item() - (1 - 0.9 ** 2)) < 1e-6

    # write with this repo, read back with torch.optim.Adam (what the reference's resume does)
    out = ck.save_checkpoint(9, str(tmp_path / 'out'), model, tr)
    assert os.path.basename(out) == 'epoch_9_state.pth'
    back = torch.load(out, map_location='cpu')
    assert back['begin_epoch'] == 9 and list(back['state_dict'].keys()) == list(orc.state_dict().keys())
    opt2 = torch.optim.Adam([p for p in orc.parameters() if p.requires_grad], lr=5e-4)
    opt2.load_state_dict(back['optimizer'][0])
    assert abs(opt2.param_groups[0]['lr'] - 1e-3) < 1e-9          # lr lives in an fp32 device scalar
    for a, b in zip(opt.state.values(), opt2.state.values()):
        assert torch.equal(a['exp_avg'], b['exp_avg']) and torch.equal(a['exp_avg_sq'], b['exp_avg_sq'])
        assert float(a['step']) == float(b['step'])

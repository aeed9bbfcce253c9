"""Checkpoint / pretrained-weight compatibility with the reference.

File format of engine/defaults/checkpoints.py:45-67: ``epoch_{n}_state.pth`` holding
``{'begin_epoch': n, 'state_dict': model.state_dict() (module.-prefix stripped), 'optimizer': [optim.state_dict()]}``;
`resume` (:70-107) loads it back and returns ``begin_epoch + 1``; `get_latest_checkpoint` (:17-31) picks the highest
``epoch_N``.  The module tree of this repo has the reference's state_dict keys, so the model part is a plain
``state_dict``; the optimizer part converts between the flat-arena Adam of train.py and a ``torch.optim.Adam``
state_dict (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``, parameters numbered in ``model.parameters()`` order
restricted to ``requires_grad``, as posetimation/optimizer/optimizer.py:18-22,66-68 builds it).

Limitation: the flat Adam is ONE parameter group.  The reference's optional second group (TRAIN.LR_SECOND_GROUP:
a list of two Adams, optimizer.py:24-64) is not representable; `load_adam_state_dict` raises on a parameter-count
mismatch instead of loading a partial state.
"""
import os
import os.path as osp

import torch


def get_latest_checkpoint(checkpoint_save_folder):
    if not osp.isdir(checkpoint_save_folder):
        return None
    best, best_idx = None, None
    for name in sorted(os.listdir(checkpoint_save_folder)):
        if not name.endswith('.pth'):
            continue
        try:
            idx = int(name.split('_')[1])          # "epoch_{n}_state.pth"
        except (IndexError, ValueError):
            continue
        if best_idx is None or idx > best_idx:
            best, best_idx = osp.join(checkpoint_save_folder, name), idx
    return best


def _strip_module(sd):
    return {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}


def adam_state_dict(trainer):
    """Flat-arena Adam (train.FlatAdam) -> torch.optim.Adam.state_dict() layout."""
    opt = trainer.opt
    step = float(opt.state[0].item())
    state = {}
    for i, (p, off, n) in enumerate(trainer.table):
        state[i] = {'step': torch.tensor(step), 'exp_avg': opt.m[off:off + n].view(p.shape).detach().cpu().clone(),
                    'exp_avg_sq': opt.v[off:off + n].view(p.shape).detach().cpu().clone()}
    group = {'lr': float(opt.state[1].item()), 'initial_lr': float(getattr(opt, 'initial_lr', opt.lr)), 'betas': tuple(opt.betas), 'eps': opt.eps, 'weight_decay': opt.wd,
             'amsgrad': False, 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False,
             'fused': None, 'params': list(range(len(trainer.table)))}
    return {'state': state, 'param_groups': [group]}


def load_adam_state_dict(trainer, sd):
    """torch.optim.Adam.state_dict() (e.g. from a reference checkpoint) -> flat-arena Adam."""
    opt = trainer.opt
    if len(sd['param_groups'][0]['params']) != len(trainer.table):
        raise ValueError('optimizer state has %d parameters, the model has %d trainable ones' %
                         (len(sd['param_groups'][0]['params']), len(trainer.table)))
    steps = set()
    with torch.no_grad():
        opt.m.zero_()
        opt.v.zero_()
        for i, (p, off, n) in enumerate(trainer.table):
            st = sd['state'].get(i)
            if st is None:
                continue
            opt.m[off:off + n].copy_(st['exp_avg'].reshape(-1).to(opt.m.device))
            opt.v[off:off + n].copy_(st['exp_avg_sq'].reshape(-1).to(opt.v.device))
            steps.add(float(st['step']))
        if len(steps) > 1:
            raise ValueError('per-parameter step counts differ; the flat Adam keeps one')
        g = sd['param_groups'][0]
        # betas / eps / weight decay are kernel arguments baked into a captured hipGraph: set_hyper bumps the
        # optimizer's hyper_version and a graph-mode Trainer re-captures on its next step()
        opt.set_hyper(betas=g['betas'], eps=g['eps'], weight_decay=g['weight_decay'])
        t = steps.pop() if steps else 0.0
        opt.lr = float(g['lr'])
        # torch writes `initial_lr` once a scheduler has been attached; a checkpoint without it was never scheduled
        opt.initial_lr = float(g.get('initial_lr', g['lr']))
        opt.state.copy_(torch.tensor([t, g['lr'], 1.0 - opt.betas[0] ** t, 1.0 - opt.betas[1] ** t]))


def save_checkpoint(epoch, save_folder, model, optimizer, **kwargs):
    """`optimizer`: a train.Trainer (flat Adam) or any torch optimizer / list of them (reference signature)."""
    os.makedirs(save_folder, exist_ok=True)
    path = osp.join(save_folder, 'epoch_{}_state.pth'.format(epoch))
    sd = {k: v.detach().cpu() for k, v in _strip_module(model.state_dict()).items()}
    opts = optimizer if isinstance(optimizer, list) else [optimizer]
    osd = [adam_state_dict(o) if hasattr(o, 'table') else o.state_dict() for o in opts]
    torch.save({'begin_epoch': epoch, 'state_dict': sd, 'optimizer': osd}, path)
    return path


def resume(model, optimizer, checkpoint_file, **kwargs):
    ckpt = torch.load(checkpoint_file, map_location='cpu')
    begin_epoch = ckpt['begin_epoch'] + 1
    sd = _strip_module(ckpt['state_dict'])
    sd = {(k[7:] if k.startswith('preact.') else k): v for k, v in sd.items()}
    with torch.no_grad():
        own = model.state_dict()
        missing = [k for k in own if k not in sd]
        extra = [k for k in sd if k not in own]
        if missing or extra:
            raise RuntimeError('state_dict mismatch: missing %s unexpected %s' % (missing[:5], extra[:5]))
        for k, v in sd.items():          # in place: parameters may be views into the trainer's flat arena
            own[k].copy_(v.to(own[k].device))
    opts = optimizer if isinstance(optimizer, list) else [optimizer]
    for o, osd in zip(opts, ckpt['optimizer']):
        if hasattr(o, 'table'):
            load_adam_state_dict(o, osd)
        else:
            o.load_state_dict(osd)
    return model, optimizer, begin_epoch

"""On-device heatmap loss, Gaussian targets and decode (NCHW boundary tensors).

Mirrors posetimation/loss/mse_loss.py:13-40 (JointMSELoss),
datasets/process/heatmaps_process.py:146-203 (generate_heatmaps) and :16-44
(get_max_preds), engine/core/utils/evaluate.py:13-75 (accuracy).  The reductions
run in HIP; only the final [B,J]-sized PCK bookkeeping (a few dozen scalars)
is host arithmetic, as in the reference.
"""
import numpy as np
import torch
import torch.nn as nn

from ._lib import lib


def _p(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("fami_pose_amd losses run on the HIP path only (tensor is on %s)" % t.device)


class _WMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, w, scale):
        R, L = pred.shape[0] * pred.shape[1], pred[0, 0].numel()
        out = torch.empty(1, device=pred.device)
        ws = torch.empty(R, device=pred.device)
        lib().call('fami_wmse_fwd_f32', _p(pred), _p(gt), _p(w), _p(out), R, L, float(scale), _p(ws), _stream(pred))
        ctx.save_for_backward(pred, gt, w)
        ctx.scale = scale
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, gt, w = ctx.saved_tensors
        R, L = pred.shape[0] * pred.shape[1], pred[0, 0].numel()
        d = torch.empty_like(pred)
        lib().call('fami_wmse_bwd_f32', _p(pred), _p(gt), _p(w), _p(d), R, L, float(ctx.scale),
                   _p(g.reshape(1).contiguous()), 0, _stream(pred))
        return d, None, None, None


class JointMSELoss(nn.Module):
    """(1/J) sum_j mean_{b,p} (w_bj (pred - gt))^2, one fused reduction instead of the reference's J-step loop."""

    def __init__(self, use_target_weight=True, divided_num_joints=True):
        super().__init__()
        self.use_target_weight = use_target_weight
        self.divided_num_joints = divided_num_joints

    def forward(self, output, target, target_weight):
        _need_cuda(output)
        B, J = output.shape[:2]
        L = output[0, 0].numel()
        scale = 1.0 / (B * L) / (J if self.divided_num_joints else 1)
        w = target_weight.reshape(B * J).float().contiguous() if self.use_target_weight else None
        return _WMSE.apply(output.float().contiguous(), target.float().contiguous(), w, scale)


def generate_heatmaps(joints, joints_vis, sigma, image_size, heatmap_size):
    """joints [B,J,2|3] (pixels, device), joints_vis [B,J] or [B,J,3] -> target [B,J,Hh,Wh], weight [B,J,1].
    image_size / heatmap_size are (w, h) as in the reference."""
    _need_cuda(joints)
    B, J = joints.shape[:2]
    xy = joints[..., :2].float().contiguous()
    vis = (joints_vis[..., 0] if joints_vis.dim() == 3 else joints_vis).float().contiguous()
    Wh, Hh = int(heatmap_size[0]), int(heatmap_size[1])
    target = torch.empty(B, J, Hh, Wh, device=joints.device)
    weight = torch.empty(B, J, device=joints.device)
    lib().call('fami_gauss_target_f32', _p(xy), _p(vis), _p(target), _p(weight), B, J, Hh, Wh, int(image_size[1]),
               int(image_size[0]), int(sigma), _stream(joints))
    return target, weight.reshape(B, J, 1)


def argmax_indices(hm, with_max=False):
    """flat row-major argmax per (b, j), first max on ties -> int64 [B,J]."""
    _need_cuda(hm)
    B, J = hm.shape[:2]
    L = hm[0, 0].numel()
    idx = torch.empty(B, J, dtype=torch.int64, device=hm.device)
    mx = torch.empty(B, J, device=hm.device)
    lib().call('fami_argmax2d_f32', _p(hm.float().contiguous()), _p(idx), _p(mx), B * J, L, _stream(hm))
    return (idx, mx) if with_max else idx


def get_max_preds(hm):
    """-> preds [B,J,2] float32 (x, y; zeroed where max <= 0), maxvals [B,J,1]."""
    idx, mx = argmax_indices(hm, with_max=True)
    W = hm.shape[3]
    preds = torch.stack([(idx % W).float(), torch.div(idx, W, rounding_mode='floor').float()], -1)
    preds = preds * (mx > 0).float().unsqueeze(-1)
    return preds, mx.unsqueeze(-1)


def get_final_preds(batch_heatmaps, center, scale):
    """On-device get_final_preds (datasets/process/heatmaps_process.py:47-73): heatmaps [B,J,H,W] (device),
    center / scale [B,2] -> (preds [B,J,2] image coordinates, maxvals [B,J,1]) as device tensors."""
    _need_cuda(batch_heatmaps)
    B, J, H, W = batch_heatmaps.shape
    dev = batch_heatmaps.device
    c = torch.as_tensor(center, dtype=torch.float32, device=dev).reshape(B, 2).contiguous()
    sc = torch.as_tensor(scale, dtype=torch.float32, device=dev).reshape(B, 2).contiguous()
    preds = torch.empty(B, J, 2, device=dev)
    mx = torch.empty(B, J, device=dev)
    ws = torch.empty(B * J, dtype=torch.int64, device=dev)
    lib().call('fami_final_preds_f32', _p(batch_heatmaps.float().contiguous()), _p(c), _p(sc), _p(preds), _p(mx), _p(ws),
               B, J, H, W, _stream(batch_heatmaps))
    return preds, mx.unsqueeze(-1)


def accuracy(output, target, thr=0.5):
    """PCK on heatmap argmax (evaluate.py:39-75).  -> (acc[J+1], avg_acc, cnt, pred[B,J,2] ndarray)."""
    pred, _ = get_max_preds(output)
    tgt, _ = get_max_preds(target)
    pred, tgt = pred.cpu().numpy(), tgt.cpu().numpy()          # 2*B*J*2 floats cross the bus, not 2 heatmap stacks
    B, J = pred.shape[:2]
    h, w = output.shape[2], output.shape[3]
    norm = np.ones((B, 2)) * np.array([h, w]) / 10
    valid = (tgt[:, :, 0] > 1) & (tgt[:, :, 1] > 1)
    d = np.linalg.norm(pred.astype(np.float32) / norm[:, None, :] - tgt.astype(np.float32) / norm[:, None, :], axis=2)
    dists = np.where(valid, d, -1.0).T
    acc = np.zeros(J + 1)
    avg, cnt = 0.0, 0
    for c in range(J):
        use = dists[c] != -1
        acc[c + 1] = (dists[c][use] < thr).sum() * 1.0 / use.sum() if use.sum() > 0 else -1
        if acc[c + 1] >= 0:
            avg += acc[c + 1]
            cnt += 1
    avg = avg / cnt if cnt != 0 else 0
    if cnt != 0:
        acc[0] = avg
    return acc, avg, cnt, pred

"""Independent anchors for the two ops whose arithmetic lives in third-party packages that are absent from the
reference tree (torchvision.ops.DeformConv2d at Alignment_V15.py:146-158, kornia.geometry.warp_affine at :133-135).
Everywhere else these HIP kernels are checked against this repo's own restatement (oracle/ops.py); the cases below
tie them to something that restatement has no part in:

  * zero offsets + unit mask: a deformable conv IS the plain dilated conv -> ATen's F.conv2d on the CPU (forward and
    the gradients wrt input / weight / bias);
  * integer offsets (a different one per offset group and tap) + arbitrary per-(group, tap) mask: every sample falls on
    a pixel centre -> a sum of masked, integer-shifted, zero-filled copies of the input built with slicing only;
  * warp by an integer translation == shifted copy with zero fill; by a fractional translation of a linear ramp ==
    the ramp evaluated at the translated coordinates (bilinear interpolation reproduces affine functions exactly).
None of these imports `oracle`."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.float().permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _shifted(x, dy, dx):
    """y[..., i, j] = x[..., i + dy, j + dx], zero outside the map (slicing only)."""
    H, W = x.shape[-2:]
    out = torch.zeros_like(x)
    i0, i1 = max(0, -dy), min(H, H - dy)
    j0, j1 = max(0, -dx), min(W, W - dx)
    if i0 < i1 and j0 < j1:
        out[..., i0:i1, j0:j1] = x[..., i0 + dy:i1 + dy, j0 + dx:j1 + dx]
    return out


@pytest.mark.parametrize('direct', [0, 1, 2], ids=['lds-columns', 'register-fed', 'lds-window'])
@pytest.mark.parametrize('C,G,H,W', [(48, 12, 24, 18), (64, 16, 13, 11)])
def test_dcn_zero_offsets_is_dilated_conv(dev, direct, C, G, H, W):
    from fami_pose_amd._lib import lib
    from fami_pose_amd.engine import Engine, T
    torch.manual_seed(C + H)
    B = 2
    conv = nn.Conv2d(C, C, 3, 1, 3, 3)
    x = torch.randn(B, C, H, W, requires_grad=True)
    y = conv(x)                                   # ATen, CPU
    gy = torch.randn_like(y)
    y.backward(gy)
    lib().cdll.fami_dcn_tune(direct)
    try:
        eng = Engine(dev)
        wd, bd = nn.Parameter(conv.weight.detach().to(dev)), nn.Parameter(conv.bias.detach().to(dev))
        xt = T(nhwc(x.detach()).to(dev), True)
        ot = T(torch.zeros(B, H, W, 18 * G, device=dev), True)
        mt = T(torch.ones(B, H, W, 9 * G, device=dev), True)
        yt = eng.dcn(xt, ot, mt, wd, bd, G, 3, 3)
        assert relerr(nchw(yt.data), y) < 2e-5
        yt.grad = nhwc(gy).to(dev)
        eng.backward()
        assert relerr(nchw(xt.grad), x.grad) < 5e-5
        assert relerr(eng.param_grads[id(wd)], conv.weight.grad) < 5e-5
        assert relerr(eng.param_grads[id(bd)], conv.bias.grad) < 5e-5
    finally:
        lib().cdll.fami_dcn_tune(-1)


@pytest.mark.parametrize('direct', [0, 1, 2], ids=['lds-columns', 'register-fed', 'lds-window'])
def test_dcn_integer_offsets_are_shifted_taps(dev, direct):
    from fami_pose_amd._lib import lib
    from fami_pose_amd.engine import Engine, T
    torch.manual_seed(5)
    B, C, G, H, W = 2, 48, 12, 20, 15
    cg = C // G
    x = torch.randn(B, C, H, W)
    w = torch.randn(C, C, 3, 3) * 0.1
    b = torch.randn(C)
    oi = torch.randint(-4, 5, (G, 9, 2))                 # (dy, dx) per (group, tap), integers incl. beyond the halo
    mk = torch.randn(G, 9)
    # offset layout of torchvision.ops.deform_conv2d: channel 2*(g*9 + tap) = dy, +1 = dx; mask channel g*9 + tap
    off = oi.reshape(1, G * 18, 1, 1).float().expand(B, G * 18, H, W).contiguous()
    msk = mk.reshape(1, G * 9, 1, 1).expand(B, G * 9, H, W).contiguous()
    ref = b.reshape(1, C, 1, 1).expand(B, C, H, W).clone()
    for tap in range(9):
        i, j = tap // 3, tap % 3
        col = torch.zeros_like(x)
        for g in range(G):
            dy, dx = int(oi[g, tap, 0]), int(oi[g, tap, 1])
            col[:, g * cg:(g + 1) * cg] = mk[g, tap] * _shifted(x[:, g * cg:(g + 1) * cg], -3 + 3 * i + dy, -3 + 3 * j + dx)
        ref += torch.einsum('oc,bchw->bohw', w[:, :, i, j], col)
    lib().cdll.fami_dcn_tune(direct)
    try:
        eng = Engine(dev, record=False)
        yt = eng.dcn(T(nhwc(x).to(dev)), T(nhwc(off).to(dev)), T(nhwc(msk).to(dev)), nn.Parameter(w.to(dev)),
                     nn.Parameter(b.to(dev)), G, 3, 3)
        assert relerr(nchw(yt.data), ref) < 2e-5
    finally:
        lib().cdll.fami_dcn_tune(-1)


def test_shift_integer_translation_is_zero_filled_copy(dev):
    from fami_pose_amd.engine import Engine, T
    torch.manual_seed(1)
    B, C, H, W = 3, 48, 24, 18
    x = torch.randn(B, C, H, W)
    t = torch.tensor([[2.0, -3.0], [-5.0, 0.0], [0.0, 30.0]])      # (tx, ty); the last one leaves the map entirely
    # warp_affine with M = [[1,0,tx],[0,1,ty]]: out[y, x] = src[y - ty, x - tx]
    ref = torch.stack([_shifted(x[b], -int(t[b, 1]), -int(t[b, 0])) for b in range(B)])
    eng = Engine(dev, record=False)
    yt = eng.shift(T(nhwc(x).to(dev)), T(t.to(dev)))
    assert torch.equal(nchw(yt.data).cpu(), ref)                   # exact: weights are 1 and 0


def test_shift_fractional_translation_of_a_ramp(dev):
    from fami_pose_amd.engine import Engine, T
    B, C, H, W = 2, 48, 24, 18
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    a = torch.linspace(-1, 1, C).reshape(C, 1, 1)
    bb = torch.linspace(0.5, -0.25, C).reshape(C, 1, 1)
    ramp = (a * xs + bb * ys + 0.3).unsqueeze(0).expand(B, C, H, W).contiguous()
    t = torch.tensor([[0.3, -1.7], [4.25, 2.5]])
    eng = Engine(dev, record=False)
    out = nchw(eng.shift(T(nhwc(ramp).to(dev)), T(t.to(dev))).data).cpu()
    for b in range(B):
        tx, ty = float(t[b, 0]), float(t[b, 1])
        sx, sy = xs - tx, ys - ty                                  # source coordinates of every output pixel
        inside = (sx >= 0) & (sx <= W - 1) & (sy >= 0) & (sy <= H - 1)
        want = a * sx + bb * sy + 0.3
        err = ((out[b] - want).abs() * inside).max().item()
        assert err < 1e-4, (b, err)
        assert inside.sum() > 0.5 * H * W
        # fully outside the source map (all four corners): zero padding
        far = (sx <= -1) | (sx >= W) | (sy <= -1) | (sy >= H)
        assert (out[b].abs() * far).max().item() == 0.0

"""BatchNorm statistics folded into the convolution epilogues (conv.hip EpiBN; Engine.conv_bn, the fused branch of
Engine.conv's backward) against (a) the same graph with the stand-alone statistics passes and (b) torch fp32 autograd
of the reference's blocks (posetimation/layers/basic_model.py:25-63 BasicBlock, hrnet.py:724-762 stride-2 transition):
outputs, running statistics, input gradient, every weight / BatchNorm-parameter gradient.  All three storage types,
direct and LDS-staged convolution routes.  Tolerances: f32 vs torch 2e-5 (relative to the tensor's max), fused vs
unfused 5e-6 (same arithmetic, different summation order of the statistics).  16-bit: two valid evaluation orders of a
chain of five BatchNorms differ by many storage ulps after the backward pass (one flipped rounding early on is amplified
by every normalisation behind it), so the fused run is held to the f32 engine run instead: its error may be at most
1.5x the unfused 16-bit run's own error plus a floor of one storage ulp of the tensor's maximum."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

DT = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}


@pytest.fixture(params=[0, 1], ids=['direct', 'lds'])
def lds_mode(request):
    from fami_pose_amd._lib import lib
    lib().cdll.fami_conv_tune_lds(request.param)
    if request.param:
        lib().cdll.fami_conv_tune_lds(21)                   # ... including the (opt-in) f32 instance of the register-blocked kernel
        lib().cdll.fami_conv_tune_lds(112)                  # (explicit tiles per band: the f32 instance otherwise only takes chip-filling launches)
    yield request.param
    lib().cdll.fami_conv_tune_lds(-1)


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


class RefBlock(nn.Module):           # basic_model.py:25-63
    def __init__(self, c):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(c, c, 3, 1, 1, bias=False), nn.BatchNorm2d(c)
        self.conv2, self.bn2 = nn.Conv2d(c, c, 3, 1, 1, bias=False), nn.BatchNorm2d(c)

    def forward(self, x):
        y = torch.relu(self.bn1(self.conv1(x)))
        return torch.relu(self.bn2(self.conv2(y)) + x)


class RefNet(nn.Module):
    def __init__(self, c, c2):
        super().__init__()
        self.a, self.b = RefBlock(c), RefBlock(c)
        self.down = nn.Sequential(nn.Conv2d(c, c2, 3, 2, 1, bias=False), nn.BatchNorm2d(c2), nn.ReLU())
        self.last = nn.Conv2d(c2, 8, 1, bias=True)

    def forward(self, x):
        return self.last(self.down(self.b(self.a(x))))


def _run(dev, ref, x, gy, dtype, fuse, xbn=False, fuse_bwd=None, auto=False, defaults=False):
    """the same graph on the HIP engine -> dict of results (fp32, on the host)"""
    from fami_pose_amd.engine import Engine, T
    from fami_pose_amd.modules import BasicBlock, _cbr, run_cbr
    net = RefNet(ref.a.conv1.in_channels, ref.down[0].out_channels)
    net.load_state_dict(ref.state_dict())
    net = net.to(dev).train()
    blocks = []
    for rb in (net.a, net.b):
        b = BasicBlock(rb.conv1.in_channels, rb.conv1.in_channels)
        b.conv1, b.bn1, b.conv2, b.bn2 = rb.conv1, rb.bn1, rb.conv2, rb.bn2
        blocks.append(b)
    eng = Engine(dev, dtype=dtype)
    if not eng.bn2:
        pytest.skip('two-launch BatchNorm disabled')
    if not defaults:
        eng.fuse_bn_fwd = eng.fuse_bn_bwd = bool(fuse)      # switch the fusion per engine (FAMI_FUSE_BN)
        eng.fuse_bn_fwd3 = False                            # (every convolution class, not only the DMA-staged kernels the 16-bit default fuses into)
        eng.fuse_bn_c64 = 3                                 # (... including the 32-channel-phase kernel's epilogue, which the default leaves out)
        eng.fuse_bn_bwd_auto = bool(auto)                   # (the default's per-kernel choice for the backward statistics)
        if fuse_bwd is not None:
            eng.fuse_bn_bwd = bool(fuse_bwd)
        eng.use_xbn = bool(xbn)
    xt = T(x.permute(0, 2, 3, 1).contiguous().to(dev).to(dtype), True)
    h = xt
    for b in blocks:
        h = b.run(eng, h)
    h = run_cbr(eng, net.down, h)
    y = eng.conv(h, net.last.weight, net.last.bias, 1, 0, 1)
    y.grad = gy.permute(0, 2, 3, 1).contiguous().to(dev).to(dtype)
    eng.backward()
    torch.cuda.synchronize(dev)
    out = {'y': y.data.float().permute(0, 3, 1, 2), 'dx': xt.grad.float().permute(0, 3, 1, 2)}
    for n, p in net.named_parameters():
        out['g.' + n] = eng.param_grads[id(p)].float()
    for n, b in net.named_buffers():
        if b.dtype.is_floating_point:
            out['b.' + n] = b.float()
    return {k: v.detach().cpu().clone() for k, v in out.items()}, eng.nfused


@pytest.mark.parametrize('shape', [(2, 48, 24, 18, 96), (3, 96, 13, 11, 192), (2, 192, 12, 10, 384), (2, 64, 23, 20, 64)],
                         ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f16'])
def test_fused_bn_statistics(dev, shape, dt, lds_mode):
    N, C, H, W, C2 = shape
    torch.manual_seed(sum(shape))
    ref = RefNet(C, C2).train()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
                m.running_mean.normal_(0, 0.2)           # the epilogue's pivot: neither zero nor the batch mean
    x = torch.randn(N, C, H, W) + 0.5
    gy = torch.randn(N, 8, (H + 1) // 2, (W + 1) // 2)
    dtype = DT[dt]
    fused, nf = _run(dev, ref, x, gy, dtype, True)
    plain, npl = _run(dev, ref, x, gy, dtype, False)
    # 5 BatchNorms forward; backward: a.bn1, a.bn2 (residual: last contribution = b.conv1's input gradient), b.bn1,
    # b.bn2 (-> down conv, stride 2: parity-class input gradient), down.bn (-> 1x1 conv) -- the last one only when its
    # quarter-size tensor is above the one-launch small-tensor kernel's limit (P * C > 32768)
    n_exp = 5 if N * ((H + 1) // 2) * ((W + 1) // 2) * C2 > 32768 else 4
    assert nf == {'fwd': n_exp, 'bwd': n_exp, 'xbn': 0} and npl == {'fwd': 0, 'bwd': 0, 'xbn': 0}
    if dt == 'f32':
        for k in fused:
            assert relerr(fused[k], plain[k]) < 5e-6, (k, relerr(fused[k], plain[k]))
    else:
        ref32, _ = _run(dev, ref, x, gy, torch.float32, False)
        ulp = {'bf16': 2.0 ** -8, 'f16': 2.0 ** -11}[dt]
        for k in fused:
            ef, ep = relerr(fused[k], ref32[k]), relerr(plain[k], ref32[k])
            assert ef < 1.5 * ep + ulp, (k, ef, ep)
    if dt == 'f32':
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)
        yr.backward(gy)
        assert relerr(fused['y'], yr) < 2e-5 and relerr(fused['dx'], xr.grad) < 2e-5
        for n, p in ref.named_parameters():
            assert relerr(fused['g.' + n], p.grad) < 5e-5, n
        for n, b in ref.named_buffers():
            if b.dtype.is_floating_point:
                assert relerr(fused['b.' + n], b) < 1e-5, n


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
def test_default_backward_fusion_follows_the_input_gradient_kernel(dev, dt):
    """FAMI_FUSE_BN=auto (the 16-bit default): the backward statistics ride in the input gradient's epilogue exactly where the
    DMA-staged kernels (conv_t6.hip) take it -- the 48-channel kernel at every such site (a.bn1: mask recomputed from the BN
    input; a.bn2: residual, mask from the BN output, accumulating launch; b.bn1), the phased kernel not at all here (no
    96-channel 3x3 in the net) -- and nowhere once those kernels are switched off.  Results against the unfused graph as in
    test_fused_bn_statistics."""
    from fami_pose_amd._lib import lib
    N, C, H, W, C2 = 8, 48, 96, 72, 96
    torch.manual_seed(77)
    ref = RefNet(C, C2).train()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
                m.running_mean.normal_(0, 0.2)
    x = torch.randn(N, C, H, W) + 0.5
    gy = torch.randn(N, 8, (H + 1) // 2, (W + 1) // 2)
    assert lib().cdll.fami_conv_t6_eligible(N, H, W, C, C) == 1
    auto, na = _run(dev, ref, x, gy, DT[dt], True, fuse_bwd=False, auto=True)
    plain, npl = _run(dev, ref, x, gy, DT[dt], True, fuse_bwd=False)
    assert na['bwd'] == 3 and npl['bwd'] == 0 and na['fwd'] == npl['fwd']
    ref32, _ = _run(dev, ref, x, gy, torch.float32, False)
    ulp = {'bf16': 2.0 ** -8, 'f16': 2.0 ** -11}[dt]
    for k in auto:
        ea, ep = relerr(auto[k], ref32[k]), relerr(plain[k], ref32[k])
        assert ea < 1.5 * ep + ulp, (k, ea, ep)
    lib().cdll.fami_conv_tune_lds(8000)               # (the autouse fixture restores the knobs)
    off, no = _run(dev, ref, x, gy, DT[dt], True, fuse_bwd=False, auto=True)
    assert no['bwd'] == 0


@pytest.mark.parametrize('dt', ['bf16', 'f32'])
def test_default_forward_fusion_rule(dev, dt):
    """FAMI_FUSE_BN=auto as an Engine comes up: in 16-bit storage the forward statistics ride only in the epilogues of the DMA-staged 3x3
    kernels (the four 48 -> 48 convolutions of the two blocks; not the stride-2 convolution behind them), the backward statistics in
    the three input gradients those kernels take; in f32 storage every convolution's epilogue carries its forward statistics and the two
    conv1 -> bn1 -> ReLU edges are not materialised (XBN).  Results against the f32 unfused graph."""
    N, C, H, W, C2 = 8, 48, 96, 72, 96
    torch.manual_seed(78)
    ref = RefNet(C, C2).train()
    x = torch.randn(N, C, H, W) + 0.5
    gy = torch.randn(N, 8, (H + 1) // 2, (W + 1) // 2)
    out, nf = _run(dev, ref, x, gy, DT[dt], None, defaults=True)
    if dt == 'bf16':
        assert nf == {'fwd': 4, 'bwd': 3, 'xbn': 0}, nf
    else:
        assert nf['fwd'] == 5 and nf['bwd'] == 0 and nf['xbn'] == 2, nf
    ref32, _ = _run(dev, ref, x, gy, torch.float32, False)
    plain, _ = _run(dev, ref, x, gy, DT[dt], False)
    ulp = {'bf16': 2.0 ** -8, 'f32': 5e-6}[dt]
    for k in out:            # (as in test_fused_bn_statistics: no further from the f32 unfused graph than the unfused graph of the same storage type)
        ea, ep = relerr(out[k], ref32[k]), relerr(plain[k], ref32[k])
        assert ea < 1.5 * ep + ulp, (k, ea, ep)


@pytest.mark.parametrize('shape', [(2, 48, 24, 18, 96), (3, 96, 13, 11, 192), (2, 192, 12, 10, 384), (2, 64, 23, 20, 64),
                                   (4, 48, 96, 72, 96)], ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f16'])
def test_bn_relu_applied_while_the_next_conv_stages_its_input(dev, shape, dt):
    """Engine.conv_bn_relu_into (conv_epi.h XBN): conv1 -> bn1 -> relu -> conv2 of a BasicBlock with the normalised tensor
    never written.  The consumer kernels round the staged values to the storage type exactly as the stand-alone apply pass
    stores them, so the graph computes what the materialising graph computes; the statistics' fp64 atomics are the one
    order-dependent sum (a last-bit difference of mean / invstd can flip a storage rounding), hence a 4-ulp bound on
    every result plus "at least half of the results are bitwise equal" rather than equality everywhere.  f32 storage (the
    split-product kernels): nothing rounds such a difference away, so every result is held to 4e-6 of its maximum."""
    from fami_pose_amd._lib import lib
    N, C, H, W, C2 = shape
    torch.manual_seed(sum(shape) + 1)
    ref = RefNet(C, C2).train()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
                m.running_mean.normal_(0, 0.2)
    x = torch.randn(N, C, H, W) + 0.5
    gy = torch.randn(N, 8, (H + 1) // 2, (W + 1) // 2)
    if dt != 'f32':
        # late round 4: the DMA-staged kernels (conv_t6.hip) take a materialised input, and where they are eligible the engine
        # does not ask for the consumer-side transform; this test holds the transform on the band kernels (conv_t4.hip, conv_wg16.hip),
        # which stay the route of every shape the DMA kernels do not take.  (The autouse fixture restores the knobs.)
        lib().cdll.fami_conv_tune_lds(8000)
        lib().cdll.fami_conv_tune_lds(8500)
        lib().cdll.fami_conv_tune_wgrad_lds(23000)
    assert (lib().cdll.fami_conv2d_xbn_ok_f32 if dt == 'f32' else lib().cdll.fami_conv2d_xbn_ok)(N, H, W, C, C) == 1
    lazy, nl = _run(dev, ref, x, gy, DT[dt], True, xbn=True, fuse_bwd=False)
    mat, nm = _run(dev, ref, x, gy, DT[dt], True, xbn=False, fuse_bwd=False)
    assert nl['xbn'] == 2 and nm['xbn'] == 0 and nl['fwd'] == nm['fwd']
    ulp = {'f32': 1e-6, 'bf16': 2.0 ** -8, 'f16': 2.0 ** -11}[dt]
    for k in lazy:
        assert relerr(lazy[k], mat[k]) < 4 * ulp, (k, relerr(lazy[k], mat[k]))
    same = sum(torch.equal(lazy[k], mat[k]) for k in lazy)
    assert dt == 'f32' or same >= len(lazy) // 2, (same, len(lazy))

"""16-bit activation modes of the HIP kernels -- bf16 (BASELINE config 3) and fp16 (BASELINE config 5) -- through the
C ABI, against torch fp32 evaluated on the SAME 16-bit-rounded operands.  What separates the two is then only (a) the
rounding of each kernel's output to the storage type (relative 2^-9 = 2e-3 per element for bf16, 2^-12 = 2.4e-4 for
fp16) and (b) fp32 summation order.  Tolerances (of the tensor's max magnitude): activations / activation gradients
1e-2 (bf16) and 1.5e-3 (fp16); fp32 results (weight / BN-parameter gradients, statistics, heatmap outputs) 2e-3 / 1e-3."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 1], ids=['direct', 'lds'])
def lds_mode(request):
    """Run the convolution cases on the direct kernels and with the LDS-staged 3x3 kernel enabled."""
    from fami_pose_amd._lib import lib
    lib().cdll.fami_conv_tune_lds(request.param)
    yield request.param
    lib().cdll.fami_conv_tune_lds(-1)
BF = torch.bfloat16          # module globals re-pointed by the `half` fixture below (bf16 | fp16)
ACT_TOL, F32_TOL = 1e-2, 2e-3


@pytest.fixture(params=['bf16', 'f16'], autouse=True)
def half(request):
    """Every test of this module runs once per 16-bit storage type."""
    global BF, ACT_TOL, F32_TOL
    if request.param == 'bf16':
        BF, ACT_TOL, F32_TOL = torch.bfloat16, 1e-2, 2e-3
    else:
        BF, ACT_TOL, F32_TOL = torch.float16, 1.5e-3, 1e-3
    yield request.param
    BF, ACT_TOL, F32_TOL = torch.bfloat16, 1e-2, 2e-3


def _eng(dev):
    from fami_pose_amd.engine import Engine
    return Engine(dev, dtype=BF)


def rb(x):
    """round to the 16-bit storage type, keep fp32 storage (the reference operand)"""
    return x.to(BF).float()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.float().permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


CASES = [
    # N, H, W, Ci, Co, k, stride, pad, dil, bias
    (2, 24, 18, 48, 48, 3, 1, 1, 1, False),
    (2, 12, 9, 96, 96, 3, 1, 1, 1, False),
    (1, 12, 9, 192, 192, 3, 1, 1, 1, False),
    (2, 6, 5, 384, 384, 3, 1, 1, 1, False),      # split-K path
    (2, 32, 24, 3, 64, 3, 2, 1, 1, False),       # stem (Ci = 3: scalar operand fetch)
    (2, 16, 12, 64, 64, 3, 2, 1, 1, False),
    (2, 16, 12, 64, 256, 1, 1, 0, 1, False),
    (2, 16, 12, 256, 48, 3, 1, 1, 1, False),
    (2, 13, 9, 96, 192, 3, 2, 1, 1, False),      # odd sizes, stride 2
    (2, 24, 18, 48, 216, 3, 1, 3, 3, True),      # DCN offset predictor (dilation 3)
    (2, 24, 18, 48, 108, 3, 1, 3, 3, True),      # DCN mask predictor: input gradient over 108 channels (8-byte operand fetch)
    (2, 13, 9, 20, 108, 3, 2, 1, 1, True),       # ... the same fetch, stride 2 (parity-class input gradient), both directions
    (2, 24, 18, 48, 17, 1, 1, 0, 1, True),       # heatmap head (Co = 17, fp32 output)
    (3, 7, 5, 16, 16, 3, 2, 1, 1, True),
    (1, 5, 7, 20, 36, 3, 1, 1, 1, True),         # channel tails
    (20, 96, 72, 48, 48, 3, 1, 1, 1, False),     # the dominant shape at full size (MT = 4 tiles)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_half(dev, case, lds_mode):
    N, H, W, Ci, Co, k, s, p, d, has_bias = case
    torch.manual_seed(hash(case) % 1000)
    conv = nn.Conv2d(Ci, Co, k, s, p, d, bias=has_bias)
    with torch.no_grad():
        conv.weight.copy_(rb(conv.weight))
    x = rb(torch.randn(N, Ci, H, W)).requires_grad_(True)
    y = conv(x)
    gy = rb(torch.randn_like(y))
    y.backward(gy)

    from fami_pose_amd.engine import T
    eng = _eng(dev)
    cd = nn.Conv2d(Ci, Co, k, s, p, d, bias=has_bias).to(dev)
    cd.load_state_dict(conv.state_dict())
    out_f32 = Co == 17
    xt = T(nhwc(x.detach()).to(dev).to(BF), True)
    yt = eng.conv(xt, cd.weight, cd.bias, s, p, d, out_f32=out_f32)
    assert yt.data.dtype == (torch.float32 if out_f32 else BF)
    assert relerr(nchw(yt.data), y) < (F32_TOL if out_f32 else ACT_TOL)
    yt.grad = nhwc(gy).to(dev).to(BF)
    eng.backward()
    assert xt.grad.dtype == BF and relerr(nchw(xt.grad), x.grad) < ACT_TOL
    assert relerr(eng.param_grads[id(cd.weight)], conv.weight.grad) < F32_TOL
    if has_bias:
        assert relerr(eng.param_grads[id(cd.bias)], conv.bias.grad) < F32_TOL


@pytest.mark.parametrize("C,relu,res", [(48, True, True), (96, False, False), (384, True, True)])
def test_bn_half(dev, C, relu, res):
    torch.manual_seed(C)
    N, H, W = 3, 10, 7
    bn = nn.BatchNorm2d(C, momentum=0.1)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.2)
    bd = nn.BatchNorm2d(C, momentum=0.1).to(dev)
    bd.load_state_dict(bn.state_dict())
    x = rb(torch.randn(N, C, H, W) * 2 + 0.7).requires_grad_(True)
    r = rb(torch.randn(N, C, H, W)).requires_grad_(True) if res else None
    y = bn(x)
    if res:
        y = y + r
    if relu:
        y = F.relu(y)
    gy = rb(torch.randn_like(y))
    y.backward(gy)
    from fami_pose_amd.engine import T
    eng = _eng(dev)
    xt = T(nhwc(x.detach()).to(dev).to(BF), True)
    rt = T(nhwc(r.detach()).to(dev).to(BF), True) if res else None
    yt = eng.bn(xt, bd, relu=relu, residual=rt)
    assert yt.data.dtype == BF and relerr(nchw(yt.data), y) < ACT_TOL
    assert relerr(bd.running_mean, bn.running_mean) < 1e-5 and relerr(bd.running_var, bn.running_var) < 1e-5
    yt.grad = nhwc(gy).to(dev).to(BF)
    eng.backward()
    # the HIP backward sees the bf16-rounded y for the ReLU mask; elements with |y| below rounding may flip
    assert relerr(nchw(xt.grad), x.grad) < 2 * ACT_TOL
    assert relerr(eng.param_grads[id(bd.weight)], bn.weight.grad) < 5 * F32_TOL
    assert relerr(eng.param_grads[id(bd.bias)], bn.bias.grad) < 5 * F32_TOL


def test_shift_dcn_half(dev):
    from oracle import ops as O
    from fami_pose_amd.engine import T
    torch.manual_seed(3)
    eng = _eng(dev)
    # shift
    x = rb(torch.randn(2, 48, 24, 18)).requires_grad_(True)
    t = torch.tensor([[0.3, -1.7], [4.25, 2.5]], requires_grad=True)
    y = O.warp_translate(x, t)
    g = rb(torch.randn_like(y))
    y.backward(g)
    xt = T(nhwc(x.detach()).to(dev).to(BF), True)
    tt = T(t.detach().to(dev), True, f32grad=True)
    yt = eng.shift(xt, tt)
    assert relerr(nchw(yt.data), y) < ACT_TOL
    yt.grad = nhwc(g).to(dev).to(BF)
    eng.backward()
    assert relerr(nchw(xt.grad), x.grad) < ACT_TOL and relerr(tt.grad, t.grad) < F32_TOL
    # dcn
    B, C, G, H, W = 2, 48, 12, 12, 9
    x = rb(torch.randn(B, C, H, W)).requires_grad_(True)
    off = rb(torch.randn(B, 18 * G, H, W) * 2.0).requires_grad_(True)
    msk = rb(torch.randn(B, 9 * G, H, W)).requires_grad_(True)
    w = (torch.randn(C, C, 3, 3) * 0.1).requires_grad_(True)
    b = torch.randn(C, requires_grad=True)
    y = O.deform_conv2d(x, off, msk, w, b, 1, 3, 3)
    g = rb(torch.randn_like(y))
    y.backward(g)
    from fami_pose_amd._lib import lib
    gxs = []
    for knob in (513, 514):            # input-gradient scatter: 32-bit fixed-point LDS region (16-bit default), 64-bit region
        lib().cdll.fami_dcn_tune(knob)
        try:
            eng = _eng(dev)
            wd, bd = nn.Parameter(w.detach().to(dev)), nn.Parameter(b.detach().to(dev))
            xt, ot, mt = (T(nhwc(v.detach()).to(dev).to(BF), True) for v in (x, off, msk))
            yt = eng.dcn(xt, ot, mt, wd, bd, G, 3, 3)
            assert yt.data.dtype == BF and relerr(nchw(yt.data), y) < ACT_TOL
            yt.grad = nhwc(g).to(dev).to(BF)
            eng.backward()
            assert relerr(nchw(xt.grad), x.grad) < ACT_TOL
            assert relerr(nchw(ot.grad), off.grad) < ACT_TOL and relerr(nchw(mt.grad), msk.grad) < ACT_TOL
            # col (the weight-gradient operand) is stored in bf16: one extra rounding of the modulated samples
            assert relerr(eng.param_grads[id(wd)], w.grad) < 3 * F32_TOL
            assert relerr(eng.param_grads[id(bd)], b.grad) < F32_TOL
            gxs.append(xt.grad.float().clone())
        finally:
            lib().cdll.fami_dcn_tune(513)
    # 20 bits per contribution at the bound: the two regions differ far below the storage type's resolution, i.e. the
    # stored 16-bit gradients differ by roundings that fall the other way (at most an ulp of the largest value or two)
    assert relerr(gxs[0], gxs[1]) < 8e-3


def test_weight_resident_dma_conv(dev, half):
    """conv_t6.hip (round 4): the 48-channel 3x3 stride-1 convolution with the whole weight image LDS-resident, patch and
    weights copied by LDS DMA (buffer loads: borders / out-of-image rows are out-of-range offsets = zeros), a dense K order
    over (tap, 8-channel granule).  Forward (+ bias), input gradient (plain and accumulating) and the forward BatchNorm
    statistics epilogue against fp64 on the same 16-bit operands, and against the band kernel (conv_t4.hip) it replaces:
    the bench shape, the head's 4-frame shape, 64-pixel rows (no ninth tile), two output-channel blocks, bands of 2 / 4 / 6 / 8
    / 12 rows, units of two and four rows; the phased kernel with 48-channel phases (W48) and with 32-channel phases (W64)."""
    from fami_pose_amd._lib import lib
    L = lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    sfx = '_' + half
    # (N, H, W, Ci, Co, rows per band, rows per unit of the 48-channel kernel | workgroup target of the phase kernel)
    cases = [(20, 96, 72, 48, 48, 0, 0), (4, 96, 72, 48, 48, 0, 0), (3, 12, 72, 48, 48, 6, 0), (2, 16, 64, 48, 96, 0, 0),
             (5, 24, 72, 48, 96, 12, 0), (2, 8, 72, 48, 48, 2, 0), (2, 8, 72, 48, 48, 4, 1), (3, 16, 64, 48, 48, 8, 2), (1, 10, 72, 48, 48, 0, 0),
             # conv3x3_t7_kernel: input channels in phases of 48, several bands per workgroup
             (20, 48, 36, 96, 96, 0, 0), (20, 24, 18, 192, 192, 0, 0), (20, 12, 9, 384, 384, 0, 0), (3, 48, 36, 96, 48, 0, 7),
             (2, 24, 18, 144, 96, 4, 1), (3, 12, 9, 96, 144, 6, 500), (2, 16, 36, 192, 48, 8, 3), (1, 10, 72, 96, 48, 2, 0),
             # ... in phases of 32 (round 5: layers of 64-multiple channels -- HRNet-W64's branches, stage 1's 64 -> 64): swizzled 64-byte positions
             (20, 96, 72, 64, 64, 0, 0), (20, 48, 36, 128, 128, 0, 0), (20, 24, 18, 256, 256, 0, 0), (20, 12, 9, 512, 512, 0, 0), (3, 48, 36, 128, 64, 0, 7),
             (2, 24, 18, 64, 128, 4, 1), (3, 12, 9, 128, 192, 6, 500), (2, 16, 36, 256, 64, 8, 3), (1, 10, 72, 64, 64, 2, 0)]
    try:
        for it, (N, H, W, Ci, Co, rows, mt) in enumerate(cases):
            torch.manual_seed(it)
            x, dy = torch.randn(N, H, W, Ci, device=dev).to(BF), torch.randn(N, H, W, Co, device=dev).to(BF)
            w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.1
            bias = torch.randn(Co, device=dev) if it % 2 else None
            pivot = torch.randn(Co, device=dev) * 0.1
            geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
            wp = [torch.empty(getattr(L.cdll, 'fami_packed_weight_elems' + sfx)(Co, Ci, 3, 3, m), device=dev, dtype=BF) for m in (0, 1)]
            for m in (0, 1):
                L.call('fami_pack_conv_weight' + sfx, p(w), p(wp[m]), Co, Ci, 3, 3, m, st)
            wq = w.to(BF).double()
            ref = F.conv2d(x.double().permute(0, 3, 1, 2), wq, None if bias is None else bias.double(), padding=1).permute(0, 2, 3, 1)
            refd = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), wq, padding=1).permute(0, 2, 3, 1)
            dx0 = torch.randn(N, H, W, Ci, device=dev).to(BF)
            out = {}
            for code in (8000, 8001):
                L.cdll.fami_conv_tune_lds(-1)
                L.cdll.fami_conv_tune_lds(code)
                L.cdll.fami_conv_tune_lds(code + 500)          # (8500 / 8501: the phase kernel)
                if code == 8001:
                    L.cdll.fami_conv_tune_lds(8400)          # no minimum job count
                    if rows:
                        L.cdll.fami_conv_tune_lds((8100 if Ci == 48 else 8600) + rows)
                    if mt:
                        L.cdll.fami_conv_tune_lds(8200 + mt if Ci == 48 else 8700 + mt)
                    assert L.cdll.fami_conv_t6_eligible(N, H, W, Ci, Co) == (1 if Ci == 48 else (2 if Ci % 48 == 0 else 3)), (N, H, W, Ci, Co, rows)
                y, ys, dx, dxa = (torch.empty(N, H, W, Co, device=dev, dtype=BF), torch.empty(N, H, W, Co, device=dev, dtype=BF),
                                  torch.empty(N, H, W, Ci, device=dev, dtype=BF), dx0.clone())
                slots = torch.zeros(L.cdll.fami_bn_slots_bytes(Co) // 8, device=dev, dtype=torch.float64)
                L.call('fami_conv2d_fwd' + sfx, p(x), p(wp[0]), p(bias), p(y), *geo, 0, 0, 0, st)
                L.call('fami_conv2d_fwd_stats' + sfx, p(x), p(wp[0]), p(bias), p(ys), *geo, p(slots), p(pivot), st)
                L.call('fami_conv2d_dgrad' + sfx, p(dy), p(wp[1]), p(dx), *geo, 0, st)
                L.call('fami_conv2d_dgrad' + sfx, p(dy), p(wp[1]), p(dxa), *geo, 1, st)
                torch.cuda.synchronize(dev)
                assert torch.equal(y, ys)
                assert relerr(y, ref) < ACT_TOL and relerr(dx, refd) < ACT_TOL and relerr(dxa, refd + dx0.double()) < ACT_TOL, (it, code)
                # statistics of the values AS STORED, shifted by the pivot: the slot rows hold [ns][2][Co] sums
                ns = 8 if Co <= 96 else 4
                rows_ = slots[:8 * 2 * Co].view(8, 2, Co).sum(0).cpu()      # (unused slot rows stay zero)
                d = ys.double().reshape(-1, Co) - pivot.double()
                assert relerr(rows_[0], d.sum(0)) < 1e-5 and relerr(rows_[1], (d * d).sum(0)) < 1e-5, (it, code)
                piv = slots[8 * 2 * Co:].view(torch.float32)[:Co]
                assert torch.equal(piv.cpu(), pivot.cpu())
                # EpiBN mode 2 (the input gradient also takes the ReLU mask and the two sums of the BatchNorm that produced the
                # convolution's input): mask from the BN output / recomputed from its input, plain and accumulating
                z = torch.randn(N, H, W, Ci, device=dev).to(BF)
                mean, invstd = torch.randn(Ci, device=dev) * 0.2, torch.rand(Ci, device=dev) + 0.5
                gamma, beta = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
                sc = invstd * gamma
                ybn = torch.relu(torch.addcmul(torch.addcmul(beta, -mean, sc), z.float(), sc)).to(BF)     # fma(z, sc, fma(-mean, sc, beta))
                for rmode in (1, 2):
                    for acc in (0, 1):
                        dxm = dx0.clone() if acc else torch.empty(N, H, W, Ci, device=dev, dtype=BF)
                        slots2 = torch.zeros(L.cdll.fami_bn_slots_bytes(Ci) // 8, device=dev, dtype=torch.float64)
                        L.call('fami_conv2d_dgrad_bnstats' + sfx, p(dy), p(wp[1]), p(dxm), *geo, acc, p(z), p(ybn if rmode == 1 else None),
                               p(mean), p(invstd), p(gamma), p(beta), rmode, p(slots2), st)
                        torch.cuda.synchronize(dev)
                        full = (refd + dx0.double()) if acc else refd
                        keep = (ybn.float() > 0) if rmode == 1 else (torch.addcmul(torch.addcmul(beta, -mean, sc), z.float(), sc) > 0)
                        want = torch.where(keep, full, torch.zeros_like(full))
                        assert relerr(dxm, want) < ACT_TOL, (it, code, rmode, acc, relerr(dxm, want))
                        g = dxm.double().reshape(-1, Ci)
                        xh = (z.double().reshape(-1, Ci) - mean.double()) * invstd.double()
                        r2 = slots2[:8 * 2 * Ci].view(8, 2, Ci).sum(0).cpu()
                        assert relerr(r2[0], g.sum(0)) < 1e-5 and relerr(r2[1], (g * xh).sum(0)) < 1e-5, (it, code, rmode, acc)
                out[code] = (y, dx, dxa)
            L.cdll.fami_conv_tune_lds(-1)
            # two kernels, two summation orders: they agree to the storage type's rounding of a few elements
            for k in range(3):
                assert relerr(out[8001][k], out[8000][k].double()) < ACT_TOL, (it, k)
    finally:
        L.cdll.fami_conv_tune_lds(-1)


def test_dma_staged_weight_gradient(dev, half):
    """conv_wgrad6_kernel (conv_wg16.hip, round 4): the 3x3 stride-1 weight gradient in 48 x 48 channel blocks with the X patch and
    the dY rows of a unit (whole rows, about 288 pixels) copied by LDS DMA (out-of-range buffer offsets = the zero border and the
    zero dY rows that pad the last K step), two buffers, one barrier per unit, consecutive units per workgroup (across frames).
    Against fp64 on the same 16-bit operands and against conv_wgrad16_kernel: the four branch shapes of the bench workload, the
    head's 4-frame shape, 64-pixel rows, Ci != Co, forced units per workgroup, accumulate."""
    from fami_pose_amd._lib import lib
    L = lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    sfx = '_' + half
    try:
        for it, (N, H, W, Ci, Co, nu) in enumerate([(20, 96, 72, 48, 48, 0), (4, 96, 72, 48, 48, 0), (20, 48, 36, 96, 96, 0), (20, 24, 18, 192, 192, 0),
                                                    (20, 12, 9, 384, 384, 0), (3, 8, 64, 48, 48, 0), (2, 12, 72, 48, 96, 1), (2, 24, 72, 96, 48, 2),
                                                    (3, 24, 18, 48, 144, 5), (2, 96, 72, 48, 48, 48), (5, 16, 72, 48, 48, 3), (3, 12, 9, 96, 48, 2),
                                                    # 64-channel blocks (stage 1, the 256 -> 48 transition): units of two rows, five K steps
                                                    (20, 96, 72, 64, 64, 0), (4, 96, 72, 256, 48, 0), (2, 24, 72, 128, 64, 3),
                                                    # HRNet-W64's other branches: 64 x 32 blocks with nine / five / four K steps per unit
                                                    (20, 48, 36, 128, 128, 0), (6, 24, 18, 256, 256, 0), (4, 12, 9, 512, 512, 0), (3, 12, 9, 128, 64, 2)]):
            torch.manual_seed(it)
            x = torch.randn(N, H, W, Ci, device=dev).to(BF)
            dy = (torch.randn(N, H, W, Co, device=dev) * 0.1).to(BF)
            geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
            nb = L.cdll.fami_conv2d_wgrad_workspace(*geo)
            ws = torch.empty(nb // 4 + 4, device=dev)
            wref = torch.zeros(Co, Ci, 3, 3, device=dev, dtype=torch.double, requires_grad=True)
            F.conv2d(x.double().permute(0, 3, 1, 2), wref, padding=1).backward(dy.double().permute(0, 3, 1, 2))
            ref = wref.grad
            dw0 = torch.randn(Co, Ci, 3, 3, device=dev)
            out = {}
            for code in (23000, 23001):
                L.cdll.fami_conv_tune_wgrad_lds(-1)
                L.cdll.fami_conv_tune_wgrad_lds(code)
                if code == 23001 and nu:
                    L.cdll.fami_conv_tune_wgrad_lds(23100 + nu)
                dw, dwa = torch.empty(Co, Ci, 3, 3, device=dev), dw0.clone()
                L.call('fami_conv2d_wgrad' + sfx, p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, st)
                L.call('fami_conv2d_wgrad' + sfx, p(x), p(dy), p(dwa), p(ws), ws.numel() * 4, *geo, 1, st)
                torch.cuda.synchronize(dev)
                assert relerr(dw, ref) < 2e-6 and relerr(dwa - dw0, ref) < 2e-5, (it, code, relerr(dw, ref))
                out[code] = dw
            assert relerr(out[23001], out[23000].double()) < 2e-6
    finally:
        L.cdll.fami_conv_tune_wgrad_lds(-1)


def test_dma_staged_wide_1x1_weight_gradient(dev, half):
    """conv_wgrad1_kernel (conv_wg16.hip, round 4): the 1x1 weight gradients of stage 1 (64 <-> 256 channels on the full-resolution
    map) in 64 x 64 channel blocks over the flat pixel axis, units of 288 pixels copied by LDS DMA (the last unit padded with
    out-of-range = zero granules).  Against fp64 on the same 16-bit operands and against the kernel it replaces."""
    from fami_pose_amd._lib import lib
    L = lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    sfx = '_' + half
    try:
        for it, (N, H, W, Ci, Co, tg) in enumerate([(20, 96, 72, 64, 256, 0), (20, 96, 72, 256, 64, 0), (20, 96, 72, 64, 64, 0), (7, 101, 97, 128, 64, 7),
                                                    (9, 96, 80, 64, 128, 500)]):
            torch.manual_seed(it)
            x = torch.randn(N, H, W, Ci, device=dev).to(BF)
            dy = (torch.randn(N, H, W, Co, device=dev) * 0.1).to(BF)
            geo = (N, H, W, Ci, Co, 1, 1, 1, 0, 1)
            nb = L.cdll.fami_conv2d_wgrad_workspace(*geo)
            ws = torch.empty(nb // 4 + 4, device=dev)
            ref = (dy.double().reshape(-1, Co).t() @ x.double().reshape(-1, Ci)).reshape(Co, Ci, 1, 1)
            dw0 = torch.randn(Co, Ci, 1, 1, device=dev)
            out = {}
            for code in (24000, 24001):
                L.cdll.fami_conv_tune_wgrad_lds(-1)
                L.cdll.fami_conv_tune_wgrad_lds(code)
                if code == 24001 and tg:
                    L.cdll.fami_conv_tune_wgrad_lds(24100 + tg)
                dw, dwa = torch.empty(Co, Ci, 1, 1, device=dev), dw0.clone()
                L.call('fami_conv2d_wgrad' + sfx, p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, st)
                L.call('fami_conv2d_wgrad' + sfx, p(x), p(dy), p(dwa), p(ws), ws.numel() * 4, *geo, 1, st)
                torch.cuda.synchronize(dev)
                assert relerr(dw, ref) < 3e-6 and relerr(dwa - dw0, ref) < 3e-5, (it, code, relerr(dw, ref))
                out[code] = dw
            assert relerr(out[24001], out[24000].double()) < 3e-6
    finally:
        L.cdll.fami_conv_tune_wgrad_lds(-1)


def test_dma_staged_weight_gradient_dilated(dev, half):
    """conv_wgrad6_kernel on the dilated predictors of the alignment head (Alignment_V15.py:83-101: 48 -> 216 offsets, 48 -> 108 masks,
    dilation = padding = 3): units of two output rows over 2 + 2 dil patch rows, tap offsets scaled by the dilation, 48-channel
    output blocks with a tail (216 = 4.5 blocks; 108 is not even a whole 8-channel granule: the straddling granule's extra channels
    only reach columns that are not stored).  Against fp64 on the same operands and against conv_wgrad16_kernel."""
    from fami_pose_amd._lib import lib
    L = lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    sfx = '_' + half
    try:
        for it, (N, H, W, Ci, Co, dil) in enumerate([(8, 96, 72, 48, 216, 3), (8, 96, 72, 48, 108, 3), (2, 24, 18, 48, 216, 3), (3, 24, 18, 48, 108, 3),
                                                     (2, 16, 36, 96, 52, 2), (1, 96, 72, 48, 48, 3)]):
            torch.manual_seed(it)
            x = torch.randn(N, H, W, Ci, device=dev).to(BF)
            dy = (torch.randn(N, H, W, Co, device=dev) * 0.1).to(BF)
            geo = (N, H, W, Ci, Co, 3, 3, 1, dil, dil)
            nb = L.cdll.fami_conv2d_wgrad_workspace(*geo)
            ws = torch.empty(nb // 4 + 4, device=dev)
            wref = torch.zeros(Co, Ci, 3, 3, device=dev, dtype=torch.double, requires_grad=True)
            F.conv2d(x.double().permute(0, 3, 1, 2), wref, padding=dil, dilation=dil).backward(dy.double().permute(0, 3, 1, 2))
            ref = wref.grad
            dw0 = torch.randn(Co, Ci, 3, 3, device=dev)
            out = {}
            for code in (23008, 23009):
                L.cdll.fami_conv_tune_wgrad_lds(-1)
                L.cdll.fami_conv_tune_wgrad_lds(code)
                dw, dwa = torch.empty(Co, Ci, 3, 3, device=dev), dw0.clone()
                L.call('fami_conv2d_wgrad' + sfx, p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, st)
                L.call('fami_conv2d_wgrad' + sfx, p(x), p(dy), p(dwa), p(ws), ws.numel() * 4, *geo, 1, st)
                torch.cuda.synchronize(dev)
                assert relerr(dw, ref) < 2e-6 and relerr(dwa - dw0, ref) < 2e-5, (it, code, relerr(dw, ref))
                out[code] = dw
            assert relerr(out[23009], out[23008].double()) < 2e-6
    finally:
        L.cdll.fami_conv_tune_wgrad_lds(-1)


def test_dma_staged_weight_gradient_stride2(dev, half):
    """conv_wgrad6_kernel on the stride-2 3x3 convolutions of the fuse / transition chains (hrnet.py:117-150): the unit's patch is
    st (UR - 1) + 3 input rows, a pixel's tap sits at twice its output coordinates.  Against fp64 and conv_wgrad16_kernel."""
    from fami_pose_amd._lib import lib
    L = lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    sfx = '_' + half
    try:
        for it, (N, H, W, Ci, Co) in enumerate([(20, 96, 72, 48, 96), (20, 48, 36, 96, 192), (20, 24, 18, 192, 384), (3, 16, 24, 48, 48),
                                                (2, 96, 72, 48, 48), (4, 192, 144, 64, 64)]):      # (the stem's second convolution: units of one output row)
            torch.manual_seed(it)
            Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            x = torch.randn(N, H, W, Ci, device=dev).to(BF)
            dy = (torch.randn(N, Ho, Wo, Co, device=dev) * 0.1).to(BF)
            geo = (N, H, W, Ci, Co, 3, 3, 2, 1, 1)
            nb = L.cdll.fami_conv2d_wgrad_workspace(*geo)
            ws = torch.empty(nb // 4 + 4, device=dev)
            wref = torch.zeros(Co, Ci, 3, 3, device=dev, dtype=torch.double, requires_grad=True)
            F.conv2d(x.double().permute(0, 3, 1, 2), wref, padding=1, stride=2).backward(dy.double().permute(0, 3, 1, 2))
            ref = wref.grad
            out = {}
            for code in (23004, 23005):
                L.cdll.fami_conv_tune_wgrad_lds(-1)
                L.cdll.fami_conv_tune_wgrad_lds(code)
                dw = torch.empty(Co, Ci, 3, 3, device=dev)
                L.call('fami_conv2d_wgrad' + sfx, p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, st)
                torch.cuda.synchronize(dev)
                assert relerr(dw, ref) < 3e-6, (it, code, relerr(dw, ref))
                out[code] = dw
            assert relerr(out[23005], out[23004].double()) < 3e-6
    finally:
        L.cdll.fami_conv_tune_wgrad_lds(-1)


def test_stem_conv1_forward(dev, half):
    """conv_stem1_fwd_kernel (conv_stem.hip): the stem's 3 -> 64 stride-2 convolution with K dense over (tap, channel) -- against fp64
    on the same 16-bit operands and against the implicit-GEMM kernel it replaces (fami_conv_tune_lds(9000)); with the BatchNorm
    statistics epilogue the slot rows must hold the shifted sums of the values AS STORED (odd sizes: border taps on every side,
    a ragged last pixel tile)."""
    from fami_pose_amd._lib import lib
    L = lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    sfx = '_' + half
    try:
        for it, (N, H, W, has_bias) in enumerate([(4, 384, 288, False), (3, 38, 26, True), (1, 17, 23, False)]):
            torch.manual_seed(it)
            Ci, Co = 3, 64
            Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            x = torch.randn(N, H, W, Ci, device=dev).to(BF)
            w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.2
            bias = torch.randn(Co, device=dev) * 0.1 if has_bias else None
            pivot = torch.randn(Co, device=dev) * 0.05
            wp = torch.empty(L.cdll.fami_packed_weight_elems_bf16(Co, Ci, 3, 3, 0), device=dev, dtype=BF)
            L.call('fami_pack_conv_weight' + sfx, p(w), p(wp), Co, Ci, 3, 3, 0, st)
            ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.to(BF).double(), None if bias is None else bias.double(), stride=2,
                           padding=1).permute(0, 2, 3, 1)
            geo = (N, H, W, Ci, Co, 3, 3, 2, 1, 1)
            nb = L.cdll.fami_bn_slots_bytes(Co)
            out = {}
            for code in (9000, 9001):
                L.cdll.fami_conv_tune_lds(code)
                y = torch.empty(N, Ho, Wo, Co, device=dev, dtype=BF)
                ys = torch.empty_like(y)
                slots = torch.zeros(nb, device=dev, dtype=torch.uint8)
                L.call('fami_conv2d_fwd' + sfx, p(x), p(wp), p(bias), p(y), *geo, 0, 0, 0, st)
                L.call('fami_conv2d_fwd_stats' + sfx, p(x), p(wp), p(bias), p(ys), *geo, p(slots), p(pivot), st)
                torch.cuda.synchronize(dev)
                assert relerr(y, ref) < ACT_TOL, (it, code, relerr(y, ref))
                assert torch.equal(y, ys), (it, code)
                rows = slots[:8 * 2 * Co * 8].view(torch.float64).view(8, 2, Co).sum(0)
                piv = slots[8 * 2 * Co * 8:8 * 2 * Co * 8 + Co * 4].view(torch.float32)
                assert torch.equal(piv, pivot)
                d = ys.double().reshape(-1, Co) - pivot.double()
                assert relerr(rows[0], d.sum(0)) < 1e-5 and relerr(rows[1], (d * d).sum(0)) < 1e-5, (it, code)
                out[code] = y
            assert relerr(out[9001], out[9000].double()) < ACT_TOL
    finally:
        L.cdll.fami_conv_tune_lds(-1)


def test_stem_conv1_weight_gradient(dev, half):
    """conv_wgrad_stem_kernel (conv_wg16.hip, round 4): the weight gradient of the stem's 3 -> 64 stride-2 convolution (hrnet.py:573-578)
    as a GEMM over the flat output-pixel axis with a gathered im2col tile (K = 27) and DMA-staged dY rows.  Against fp64 and the
    scalar-operand kernel it replaces; the second shape has image rows the kernel must zero-pad on every side."""
    from fami_pose_amd._lib import lib
    L = lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    sfx = '_' + half
    try:
        for it, (N, H, W) in enumerate([(20, 384, 288), (3, 16, 288), (2, 8, 144)]):
            torch.manual_seed(it)
            Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            x = torch.randn(N, H, W, 3, device=dev).to(BF)
            dy = (torch.randn(N, Ho, Wo, 64, device=dev) * 0.1).to(BF)
            geo = (N, H, W, 3, 64, 3, 3, 2, 1, 1)
            nb = L.cdll.fami_conv2d_wgrad_workspace(*geo)
            ws = torch.empty(nb // 4 + 4, device=dev)
            wref = torch.zeros(64, 3, 3, 3, device=dev, dtype=torch.double, requires_grad=True)
            F.conv2d(x.double().permute(0, 3, 1, 2), wref, padding=1, stride=2).backward(dy.double().permute(0, 3, 1, 2))
            ref = wref.grad
            out = {}
            for code in (25000, 25001):
                L.cdll.fami_conv_tune_wgrad_lds(-1)
                L.cdll.fami_conv_tune_wgrad_lds(code)
                dw = torch.empty(64, 3, 3, 3, device=dev)
                L.call('fami_conv2d_wgrad' + sfx, p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, 0, st)
                torch.cuda.synchronize(dev)
                assert relerr(dw, ref) < 3e-6, (it, code, relerr(dw, ref))
                out[code] = dw
            assert relerr(out[25001], out[25000].double()) < 3e-6
    finally:
        L.cdll.fami_conv_tune_wgrad_lds(-1)


def test_backward_pair_is_bitwise_the_two_launch_form(dev, half):
    """fami_conv2d_bwd_pair_* (csrc/conv_pair.h, round 6): the input gradient (plain / with the backward-statistics epilogue, fresh /
    accumulating) and the deferred weight gradient of a 3x3 stride-1 convolution as ONE launch.  The combined kernel runs the two
    single kernels' bodies on disjoint workgroups, so every output -- dx, the fp64 statistics rows up to atomic order, dW after the
    deferred reduce -- must equal the two-call form's; checked on every layer shape of PAIR_SHAPES, with the combined launch
    switched off (the recorded halves replayed as single launches), with a different weight-gradient workgroup target for the
    combined launch, and on a shape no combined instance takes."""
    import ctypes
    from fami_pose_amd._lib import lib
    L = lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    sfx = '_' + half
    nlong = L.cdll.fami_wgrad_reduce_desc_longs()
    shapes = [(20, 96, 72, 48, 48, 1), (4, 96, 72, 48, 48, 1), (4, 96, 72, 96, 48, 1), (20, 48, 36, 96, 96, 1), (20, 24, 18, 192, 192, 1), (24, 24, 18, 192, 192, 1),
              (20, 12, 9, 384, 384, 1), (20, 96, 72, 64, 64, 1), (20, 48, 36, 128, 128, 1), (20, 24, 18, 256, 256, 1), (20, 12, 9, 512, 512, 1),
              (2, 32, 24, 48, 48, 0), (8, 128, 96, 48, 48, 0)]
    try:
        for it, (N, H, W, Ci, Co, want_pair) in enumerate(shapes):
            torch.manual_seed(100 + it)
            geo = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
            L.cdll.fami_tune_reset()
            assert L.cdll.fami_conv2d_bwd_pair_ok(*geo) == want_pair, geo
            x = torch.randn(N, H, W, Ci, device=dev).to(BF)
            dy = (torch.randn(N, H, W, Co, device=dev) * 0.1).to(BF)
            w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
            wpd = torch.empty(getattr(L.cdll, 'fami_packed_weight_elems' + sfx)(Co, Ci, 3, 3, 1), device=dev, dtype=BF)
            L.call('fami_pack_conv_weight' + sfx, p(w), p(wpd), Co, Ci, 3, 3, 1, st)
            z = torch.randn(N, H, W, Ci, device=dev).to(BF)
            mean, invstd = torch.randn(Ci, device=dev) * 0.2, torch.rand(Ci, device=dev) + 0.5
            gamma, beta = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
            sc = invstd * gamma
            ybn = torch.relu(torch.addcmul(torch.addcmul(beta, -mean, sc), z.float(), sc)).to(BF)
            dx0 = torch.randn(N, H, W, Ci, device=dev).to(BF)
            dw0 = torch.randn(Co, Ci, 3, 3, device=dev)
            variants = [(0, 0, 0), (1, 1, 0), (0, 0, 2), (1, 1, 1), (0, 1, 2)] if want_pair else [(0, 0, 0), (1, 1, 2)]
            for (accx, accw, rmode) in variants:          # rmode 0: no statistics epilogue
                res = {}
                for form in ('two', 'pair', 'pair_off', 'pair_tgt'):
                    L.cdll.fami_tune_reset()
                    # (27000: the weight-gradient half keeps the single launch's workgroup target -> the same pixel split, the same sums)
                    L.cdll.fami_conv_tune_wgrad_lds(27000 if form != 'pair_tgt' else 27000 + 136)
                    if form == 'pair_off':
                        L.cdll.fami_conv_tune_lds(8998)
                    ws = torch.empty(L.cdll.fami_conv2d_wgrad_workspace(*geo) // 4 + 4, device=dev)
                    dx = dx0.clone() if accx else torch.empty(N, H, W, Ci, device=dev, dtype=BF)
                    dw = dw0.clone() if accw else torch.empty(Co, Ci, 3, 3, device=dev)
                    slots = torch.zeros(L.cdll.fami_bn_slots_bytes(Ci) // 8, device=dev, dtype=torch.float64) if rmode else None
                    bn = (p(z), p(ybn if rmode == 1 else None), p(mean), p(invstd), p(gamma), p(beta), rmode, p(slots))
                    desc = (ctypes.c_long * nlong)()
                    if form == 'two':
                        if rmode:
                            L.call('fami_conv2d_dgrad_bnstats' + sfx, p(dy), p(wpd), p(dx), *geo, accx, *bn, st)
                        else:
                            L.call('fami_conv2d_dgrad' + sfx, p(dy), p(wpd), p(dx), *geo, accx, st)
                        L.call('fami_conv2d_wgrad_defer' + sfx, p(x), p(dy), p(dw), p(ws), ws.numel() * 4, *geo, accw, desc, st)
                    else:
                        L.call('fami_conv2d_bwd_pair' + sfx, p(x), p(dy), p(wpd), p(dx), p(dw), p(ws), ws.numel() * 4, *geo, accx, accw, desc,
                               *(bn if rmode else (None, None, None, None, None, None, 0, None)), st)
                    L.call('fami_wgrad_reduce_batch', desc, 1, st)
                    torch.cuda.synchronize(dev)
                    res[form] = (dx, dw, slots)
                for form in ('pair', 'pair_off', 'pair_tgt'):
                    assert torch.equal(res[form][0], res['two'][0]), (geo, accx, accw, rmode, form)
                    if form == 'pair_tgt' and want_pair:      # another split of the pixels over workgroups: fp32 summation order differs
                        assert relerr(res[form][1], res['two'][1]) < 1e-5, (geo, form)
                    else:
                        assert torch.equal(res[form][1], res['two'][1]), (geo, accx, accw, rmode, form)
                    if rmode:
                        ns_rows = res[form][2][:8 * 2 * Ci].view(8, 2, Ci).sum(0)
                        assert relerr(ns_rows, res['two'][2][:8 * 2 * Ci].view(8, 2, Ci).sum(0)) < 1e-5, (geo, form)      # (the phased kernel adds its sums with LDS float atomics: wave order)
            # the library's default target for the combined launch (another pixel split: fp32 summation order differs)
            L.cdll.fami_tune_reset()
            ws = torch.empty(L.cdll.fami_conv2d_wgrad_workspace(*geo) // 4 + 4, device=dev)
            dxd, dwd = torch.empty(N, H, W, Ci, device=dev, dtype=BF), torch.empty(Co, Ci, 3, 3, device=dev)
            desc = (ctypes.c_long * nlong)()
            L.call('fami_conv2d_bwd_pair' + sfx, p(x), p(dy), p(wpd), p(dxd), p(dwd), p(ws), ws.numel() * 4, *geo, 0, 0, desc,
                   None, None, None, None, None, None, 0, None, st)
            L.call('fami_wgrad_reduce_batch', desc, 1, st)
            torch.cuda.synchronize(dev)
            # ... and against fp64 once per shape (the two-launch form has its own tests; this guards the test itself)
            wref = torch.zeros(Co, Ci, 3, 3, device=dev, dtype=torch.double, requires_grad=True)
            F.conv2d(x.double().permute(0, 3, 1, 2), wref, padding=1).backward(dy.double().permute(0, 3, 1, 2))
            assert relerr(res['pair'][1] - (dw0 if variants[-1][1] else 0), wref.grad) < F32_TOL, geo
            assert relerr(dwd, wref.grad) < F32_TOL, geo
    finally:
        L.cdll.fami_tune_reset()


def test_batchnorm_inside_the_consumer_convolution_launch(dev, half):
    """fami_conv2d_fwd_bnin_* (round 6): conv1 -> bn1 -> ReLU -> conv2 of a BasicBlock (basic_model.py:34-63) with bn1's apply pass
    inside conv2's launch -- the 48-channel kernel folds the statistics rows conv1's epilogue filled, transforms its patch in LDS and
    stores the normalised rows it owns.  Against fami_bn_apply_slots_* + fami_conv2d_fwd(_stats)_*: the normalised tensor, the
    convolution output, mean / invstd and the running statistics bit for bit, the output statistics rows up to atomic order; on the
    bench shapes (20 and 4 frames), 64-pixel rows, bands at the image border, with and without a bias / an output-statistics epilogue."""
    from fami_pose_amd._lib import lib
    L = lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    sfx = '_' + half
    try:
        for it, (N, H, W, Co, ostats, with_bias) in enumerate([(20, 96, 72, 48, 1, 0), (4, 96, 72, 48, 0, 1), (3, 16, 64, 96, 1, 1), (2, 8, 72, 48, 0, 0),
                                                               (5, 24, 72, 144, 1, 0)]):
            Ci = 48
            torch.manual_seed(700 + it)
            L.cdll.fami_tune_reset()
            L.cdll.fami_conv_tune_lds(8400)          # no minimum job count: the small shapes take the kernel too
            assert L.cdll.fami_conv2d_fwd_bnin_ok(N, H, W, Ci, Co) == 1, (N, H, W, Co)
            P = N * H * W
            x0 = torch.randn(N, H, W, Ci, device=dev).to(BF)
            w1, w2 = torch.randn(Ci, Ci, 3, 3, device=dev) * 0.05, torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
            bias = torch.randn(Co, device=dev) * 0.1 if with_bias else None
            gamma, beta = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
            wp1 = torch.empty(getattr(L.cdll, 'fami_packed_weight_elems' + sfx)(Ci, Ci, 3, 3, 0), device=dev, dtype=BF)
            wp2 = torch.empty(getattr(L.cdll, 'fami_packed_weight_elems' + sfx)(Co, Ci, 3, 3, 0), device=dev, dtype=BF)
            L.call('fami_pack_conv_weight' + sfx, p(w1), p(wp1), Ci, Ci, 3, 3, 0, st)
            L.call('fami_pack_conv_weight' + sfx, p(w2), p(wp2), Co, Ci, 3, 3, 0, st)
            geo1 = (N, H, W, Ci, Ci, 3, 3, 1, 1, 1)
            geo2 = (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
            res = {}
            rm_init, rv_init = torch.randn(Ci, device=dev) * 0.2, torch.rand(Ci, device=dev) + 0.5
            z = torch.empty(N, H, W, Ci, device=dev, dtype=BF)
            xs = torch.zeros(L.cdll.fami_bn_slots_bytes(Ci) // 8, device=dev, dtype=torch.float64)
            L.call('fami_conv2d_fwd_stats' + sfx, p(x0), p(wp1), None, p(z), *geo1, p(xs), p(rm_init), st)       # conv1 + bn1's statistics (ONE run: the rows arrive in atomic order)
            piv = torch.randn(Co, device=dev) * 0.1 if ostats else None
            for form in ('two', 'in'):
                rm, rv = rm_init.clone(), rv_init.clone()
                rm0 = rm.clone()
                a = torch.empty(N, H, W, Ci, device=dev, dtype=BF)
                y = torch.empty(N, H, W, Co, device=dev, dtype=BF)
                mean, invstd = torch.empty(Ci, device=dev), torch.empty(Ci, device=dev)
                ys = torch.zeros(L.cdll.fami_bn_slots_bytes(Co) // 8, device=dev, dtype=torch.float64) if ostats else None
                if form == 'two':
                    L.call('fami_bn_apply_slots' + sfx, p(z), None, p(a), p(gamma), p(beta), p(mean), p(invstd), p(rm), p(rv), P, Ci, 1, 0.1, 1e-5, p(xs), st)
                    if ostats:
                        L.call('fami_conv2d_fwd_stats' + sfx, p(a), p(wp2), p(bias), p(y), *geo2, p(ys), p(piv), st)
                    else:
                        L.call('fami_conv2d_fwd' + sfx, p(a), p(wp2), p(bias), p(y), *geo2, 0, 0, 0, st)
                else:
                    L.call('fami_conv2d_fwd_bnin' + sfx, p(z), p(wp2), p(bias), p(y), p(a), N, H, W, Ci, Co, p(ys), p(piv), p(xs), P,
                           p(gamma), p(beta), p(mean), p(invstd), p(rm), p(rv), 0.1, 1e-5, st)
                torch.cuda.synchronize(dev)
                assert not torch.equal(rm, rm0)
                res[form] = (a, y, mean, invstd, rm, rv, ys)
            for k in range(6):
                assert torch.equal(res['in'][k], res['two'][k]), (N, H, W, Co, k)
            if ostats:
                ns = 8 if Co <= 96 else 4
                rows = lambda t: t[:ns * 2 * Co].view(ns, 2, Co).sum(0)
                assert relerr(rows(res['in'][6]), rows(res['two'][6])) < 1e-12
            # the normalised tensor against torch once (guards the test itself)
            sc = res['two'][3] * gamma
            want = torch.relu(torch.addcmul(torch.addcmul(beta, -res['two'][2], sc), z.float(), sc))
            assert relerr(res['in'][0], want) < ACT_TOL
    finally:
        L.cdll.fami_tune_reset()

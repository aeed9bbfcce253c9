"""Oracle (test infrastructure): pure-PyTorch CPU restatement of the FAMI-Pose
model graph -- HRNet / HRNetPlus backbone and the Alignment_V15 head -- with
state_dict keys equal to the reference module tree.

Reference (relative to /root/reference):
  posetimation/backbones/hrnet.py:17-172 (HighResolutionModule), :186-332 (HRNet),
  :521-690 (HRNetPlus); posetimation/layers/basic_model.py:25-63 (BasicBlock),
  :66-113 (Bottleneck), :128-148 (ChainOfBasicBlocks);
  posetimation/layers/basic_layer.py:13-73 (conv_bn_relu);
  posetimation/zoo/Alignment/Alignment_V15.py:47-183.

The only generalisations over the reference class are the ones SURVEY.md 8a
lists (head width C = STAGE2.NUM_CHANNELS[0], S supporting frames, Linear
in-features from the input size, DCN offset-group count); for W48 / S=4 /
384x288 the graph is the reference's.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

_MOM = 0.1


def _bn(c):
    return nn.BatchNorm2d(c, momentum=_MOM)


class ConvUnit(nn.Module):
    """conv_bn_relu (basic_layer.py:13-73): children named conv / bn."""

    def __init__(self, cin, cout, k, stride, padding, dilation, bias=True, bn=True, relu=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, padding, dilation, bias=bias)
        self.bn = _bn(cout) if bn else None
        self.act = relu

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return F.relu(x) if self.act else x


class Basic(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = _bn(planes)
        # NB the reference passes `stride` to BOTH convs (basic_model.py:41)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = _bn(planes)
        self.downsample = downsample

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + r)


class Neck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = _bn(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = _bn(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + r)


def _proj(cin, cout, stride=1):
    return nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), _bn(cout))


class BlockChain(nn.Module):
    """ChainOfBasicBlocks (basic_model.py:128-148): child named layers."""

    def __init__(self, cin, cout, num_blocks=1):
        super().__init__()
        seq = [Basic(cin, cout, 1, _proj(cin, cout))]
        seq += [Basic(cout, cout) for _ in range(1, num_blocks)]
        self.layers = nn.Sequential(*seq)

    def forward(self, x):
        return self.layers(x)


class Up(nn.Module):
    def __init__(self, f):
        super().__init__()
        self.f = f

    def forward(self, x):
        return F.interpolate(x, scale_factor=self.f, mode='nearest')


class HRModule(nn.Module):
    """HighResolutionModule (hrnet.py:17-172), BASIC blocks, SUM fuse."""

    def __init__(self, chans, num_blocks, multi_scale_output=True):
        super().__init__()
        nb = len(chans)
        self.nb = nb
        self.branches = nn.ModuleList(
            [nn.Sequential(*[Basic(chans[i], chans[i]) for _ in range(num_blocks[i])]) for i in range(nb)])
        fuse = []
        for i in range(nb if multi_scale_output else 1):
            row = []
            for j in range(nb):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(chans[j], chans[i], 1, 1, 0, bias=False),
                                             nn.BatchNorm2d(chans[i]), Up(2 ** (j - i))))
                elif j == i:
                    row.append(None)
                else:
                    steps = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        co = chans[i] if last else chans[j]
                        mods = [nn.Conv2d(chans[j], co, 3, 2, 1, bias=False), nn.BatchNorm2d(co)]
                        if not last:
                            mods.append(nn.ReLU(True))
                        steps.append(nn.Sequential(*mods))
                    row.append(nn.Sequential(*steps))
            fuse.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(fuse) if nb > 1 else None

    def forward(self, xs):
        # the reference overwrites the caller's list in place (hrnet.py:156-157);
        # HRNet.forward's `feature = x3_list` (hrnet.py:323) therefore returns the
        # stage4.0 *branch outputs*, a quirk kept here.
        for i in range(self.nb):
            xs[i] = self.branches[i](xs[i])
        if self.nb == 1:
            return [xs[0]]
        outs = []
        for i in range(len(self.fuse_layers)):
            y = xs[0] if i == 0 else self.fuse_layers[i][0](xs[0])
            for j in range(1, self.nb):
                y = y + (xs[j] if i == j else self.fuse_layers[i][j](xs[j]))
            outs.append(F.relu(y))
        return outs


class HRNetOracle(nn.Module):
    """HRNet (hrnet.py:186-332) when plus=False: forward -> (heatmaps, pre-stage4
    branch list); HRNetPlus (hrnet.py:521-690) when plus=True: forward ->
    (heatmaps, stage4 outputs)."""

    def __init__(self, cfg, plus=True):
        super().__init__()
        ex = cfg['MODEL']['EXTRA']
        self.plus = plus
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = _bn(64)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = _bn(64)
        self.layer1 = nn.Sequential(Neck(64, 64, 1, _proj(64, 256)), *[Neck(256, 64) for _ in range(3)])
        pre = [256]
        for s in (2, 3, 4):
            sc = ex['STAGE%d' % s]
            assert sc['BLOCK'] == 'BASIC' and sc['FUSE_METHOD'] == 'SUM'
            ch = list(sc['NUM_CHANNELS'])
            setattr(self, 'transition%d' % (s - 1), self._transition(pre, ch))
            mods = []
            for m in range(sc['NUM_MODULES']):
                mso = not (s == 4 and m == sc['NUM_MODULES'] - 1)
                mods.append(HRModule(ch, sc['NUM_BLOCKS'], mso))
            setattr(self, 'stage%d' % s, nn.Sequential(*mods))
            setattr(self, 'nb%d' % s, sc['NUM_BRANCHES'])
            pre = ch
        k = ex['FINAL_CONV_KERNEL']
        self.final_layer = nn.Conv2d(pre[0], cfg['MODEL']['NUM_JOINTS'], k, 1, 1 if k == 3 else 0)

    @staticmethod
    def _transition(pre, cur):
        out = []
        for i, c in enumerate(cur):
            if i < len(pre):
                out.append(None if c == pre[i] else
                           nn.Sequential(nn.Conv2d(pre[i], c, 3, 1, 1, bias=False), nn.BatchNorm2d(c), nn.ReLU(True)))
            else:
                steps = []
                for j in range(i + 1 - len(pre)):
                    co = c if j == i - len(pre) else pre[-1]
                    steps.append(nn.Sequential(nn.Conv2d(pre[-1], co, 3, 2, 1, bias=False), nn.BatchNorm2d(co), nn.ReLU(True)))
                out.append(nn.Sequential(*steps))
        return nn.ModuleList(out)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = self.layer1(x)
        ys = [x]
        x3 = None
        for s in (2, 3, 4):
            tr = getattr(self, 'transition%d' % (s - 1))
            xs = []
            for i in range(getattr(self, 'nb%d' % s)):
                if tr[i] is not None:
                    xs.append(tr[i](ys[-1]))
                else:
                    xs.append(ys[i])
            if s == 4:
                x3 = xs
            ys = xs
            for m in getattr(self, 'stage%d' % s):
                ys = m(ys)
        hm = self.final_layer(ys[0])
        return hm, (ys if self.plus else x3)


class DCN(nn.Module):
    """torchvision.ops.DeformConv2d parameter layout (weight [Co,Ci,3,3], bias)."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c, c, 3, 3))
        self.bias = nn.Parameter(torch.zeros(c))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)

    def forward(self, x, off, mask):
        return ops.deform_conv2d(x, off, mask, self.weight, self.bias, 1, 3, 3)


def ceil_half(v, n=5):
    for _ in range(n):
        v = (v + 1) // 2
    return v


class AlignmentOracle(nn.Module):
    """Alignment_V15 (Alignment_V15.py:24-183)."""

    def __init__(self, cfg, train=True, num_sup=4, image_hw=(384, 288), dcn_groups=12, warp_align_corners=True):
        super().__init__()
        self.is_train = train
        self.warp_align_corners = warp_align_corners      # kornia.warp_affine's align_corners (ops.warp_translate)
        self.J = cfg['MODEL']['NUM_JOINTS']
        C = cfg['MODEL']['EXTRA']['STAGE2']['NUM_CHANNELS'][0]
        self.C, self.S = C, num_sup
        self.hrnet = HRNetOracle(cfg, plus=True)
        h5, w5 = ceil_half(image_hw[0] // 4), ceil_half(image_hw[1] // 4)
        self.feat_global_offset_layers = nn.Sequential(
            BlockChain(C, 16, 1),
            *[ConvUnit(16, 16, 3, 2, 1, 1) for _ in range(5)],
            nn.Flatten(), nn.Linear(16 * h5 * w5, 64), nn.Linear(64, 64), nn.Linear(64, 2))
        self.combined_feat_layers = BlockChain(2 * C, C, 1)
        for k in range(1, 5):
            setattr(self, 'dcn_offset_%d' % k, ConvUnit(C, 18 * dcn_groups, 3, 1, 3, 3, bn=False, relu=False))
            setattr(self, 'dcn_mask_%d' % k, ConvUnit(C, 9 * dcn_groups, 3, 1, 3, 3, bn=False, relu=False))
            setattr(self, 'dcn_%d' % k, DCN(C))
        self.sup_agg_block = BlockChain(C * num_sup, C, 2)
        self.init_feature_agg_block = BlockChain(2 * C, C, 3)
        self.agg_final_layer = nn.Conv2d(C, self.J, 3, 1, 1)

    def _dcn(self, k, src, x):
        off = getattr(self, 'dcn_offset_%d' % k)(src)
        msk = getattr(self, 'dcn_mask_%d' % k)(src)
        return getattr(self, 'dcn_%d' % k)(x, off, msk)

    def forward(self, kf_x, sup_x, return_aux=False):
        B, S = kf_x.shape[0], sup_x.shape[1] // 3
        frames = torch.cat([kf_x] + list(torch.chunk(sup_x, S, dim=1)), 0)
        hm, feats = self.hrnet(frames)
        hms = torch.chunk(hm, S + 1, 0)
        fs = torch.chunk(feats[0], S + 1, 0)
        kf_hm, kf = hms[0], fs[0]
        aligned = []
        shifts = []
        for i in range(S):
            t = self.feat_global_offset_layers(fs[1 + i] - kf)
            shifts.append(t)
            aligned.append(ops.warp_translate(fs[1 + i], t, self.warp_align_corners))
        agg_sup = self.sup_agg_block(torch.cat(aligned, 1))
        comb = self.combined_feat_layers(torch.cat([agg_sup, kf], 1))
        comb = self._dcn(1, comb, comb)
        comb = self._dcn(2, comb, comb)
        al = self._dcn(3, comb, agg_sup)
        al = self._dcn(4, al, al)
        allf = self.init_feature_agg_block(torch.cat([kf, al], 1))
        final = self.agg_final_layer(allf)
        aux = dict(shifts=shifts, agg_sup=agg_sup, aligned=al, all_agg=allf, kf_feat=kf)
        if not self.is_train:
            return (final, kf_hm, aux) if return_aux else (final, kf_hm)
        fw, fb = self.hrnet.final_layer.weight, self.hrnet.final_layer.bias
        mi = [ops.feat_label_mi(allf, final, fw, fb), ops.feat_feat_mi(kf, allf),
              ops.feat_label_mi(agg_sup, final, fw, fb), ops.feat_feat_mi(agg_sup, allf),
              ops.feat_label_mi(kf, final, fw, fb), ops.feat_feat_mi(kf, allf)]
        return (final, kf_hm, mi, aux) if return_aux else (final, kf_hm, mi)


def make_cfg(width=48, num_joints=17, final_kernel=1, freeze=False):
    """Minimal attr+item config carrying exactly the keys the model reads
    (values of configs/Alignment/Base_PoseTrack17.yaml:45-87)."""
    w = width
    return {'MODEL': {'NUM_JOINTS': num_joints, 'FREEZE_HRNET_WEIGHTS': freeze, 'PRETRAINED': '',
                      'BACKBONE_PRETRAINED': '', 'INIT_WEIGHTS': True, 'NAME': 'Alignment_V15',
                      'EXTRA': {'FINAL_CONV_KERNEL': final_kernel,
                                'STAGE2': dict(NUM_MODULES=1, NUM_BRANCHES=2, BLOCK='BASIC', NUM_BLOCKS=[4, 4],
                                               NUM_CHANNELS=[w, 2 * w], FUSE_METHOD='SUM'),
                                'STAGE3': dict(NUM_MODULES=4, NUM_BRANCHES=3, BLOCK='BASIC', NUM_BLOCKS=[4, 4, 4],
                                               NUM_CHANNELS=[w, 2 * w, 4 * w], FUSE_METHOD='SUM'),
                                'STAGE4': dict(NUM_MODULES=3, NUM_BRANCHES=4, BLOCK='BASIC', NUM_BLOCKS=[4, 4, 4, 4],
                                               NUM_CHANNELS=[w, 2 * w, 4 * w, 8 * w], FUSE_METHOD='SUM')}}}


def realistic_init_(model, seed=0, conv_gain=1.0, gamma=(0.15, 0.45), offset_std=0.5):
    """Re-initialise at realistic scale (the reference's
    std=0.001 init makes outputs ~1e-10 and any tolerance vacuous).  The scales
    are those of a trained network: O(1) features (small residual-branch BN
    gains), sub-pixel-to-pixel DCN offsets, masks around 1, O(1) heatmaps -- a
    random network with O(10) offsets on O(30) features is chaotic (1e-5 feature
    noise becomes 1e-2 after four deformable layers) and measures conditioning,
    not kernels."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, m in model.named_modules():
            leaf = name.split('.')[-1]
            if isinstance(m, nn.Conv2d):
                fan = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                std = conv_gain * (2.0 / fan) ** 0.5
                if 'dcn_offset' in name:
                    std = offset_std / (fan ** 0.5)
                elif 'dcn_mask' in name:
                    std = 0.2 / (fan ** 0.5)
                elif leaf in ('final_layer', 'agg_final_layer'):
                    std = 0.5 / (fan ** 0.5)
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
                    if 'dcn_mask' in name:
                        m.bias.add_(1.0)
            elif isinstance(m, nn.BatchNorm2d):
                lo, hi = gamma
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * (hi - lo) + lo)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            elif isinstance(m, nn.Linear):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (1.0 / m.in_features) ** 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
            elif hasattr(m, 'weight') and isinstance(getattr(m, 'weight'), nn.Parameter) and m.weight.dim() == 4:
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (1.0 / (m.weight.shape[1] * 9)) ** 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
    return model

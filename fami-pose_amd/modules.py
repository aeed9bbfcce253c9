"""Module tree of the HRNet backbone and its blocks, MI355X-native.

The nn.Module objects here are *parameter containers* with the reference's
attribute names, so `state_dict()` keys equal the reference tree (reference
checkpoints load unchanged: SURVEY.md 8b).  They never run torch compute: each
block's `run(eng, x)` emits HIP kernel launches through the engine tape.

Reference: posetimation/layers/basic_model.py:25-63 (BasicBlock), :66-113
(Bottleneck), :128-148 (ChainOfBasicBlocks); posetimation/layers/basic_layer.py:13-73
(conv_bn_relu); posetimation/backbones/hrnet.py:17-172 (HighResolutionModule),
:186-332 (HRNet), :521-690 (HRNetPlus).
"""
import torch.nn as nn

from . import options

BN_MOMENTUM = 0.1


def _conv(cin, cout, k, stride=1, pad=0, dil=1, bias=False):
    return nn.Conv2d(cin, cout, k, stride, pad, dil, bias=bias)


def run_conv(eng, conv, x, out_f32=False):
    return eng.conv(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], conv.dilation[0], out_f32=out_f32)


class conv_bn_relu(nn.Module):
    def __init__(self, in_planes, out_planes, kernel_size, stride, padding, dilation, has_bias=True, has_bn=True,
                 has_relu=True):
        super().__init__()
        self.conv = _conv(in_planes, out_planes, kernel_size, stride, padding, dilation, has_bias)
        self.bn = nn.BatchNorm2d(out_planes, momentum=BN_MOMENTUM) if has_bn else None
        self.has_relu = has_relu

    def run(self, eng, x):
        if self.bn is not None:
            return eng.conv_bn(x, self.conv, self.bn, relu=self.has_relu)
        assert not self.has_relu
        return run_conv(eng, self.conv, x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv2 = _conv(planes, planes, 3, stride, 1)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample

    def run(self, eng, x):
        res = x
        if self.downsample is not None:
            res = eng.conv_bn(x, self.downsample[0], self.downsample[1])
        y = eng.conv_bn_relu_into(x, self.conv1, self.bn1, self.conv2)
        return eng.conv_bn(y, self.conv2, self.bn2, relu=True, residual=res)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv2 = _conv(planes, planes, 3, stride, 1)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4, momentum=BN_MOMENTUM)
        self.downsample = downsample

    def run(self, eng, x):
        res = x
        if self.downsample is not None:
            res = eng.conv_bn(x, self.downsample[0], self.downsample[1])
        y = eng.conv_bn(x, self.conv1, self.bn1, relu=True)
        y = eng.conv_bn(y, self.conv2, self.bn2, relu=True)
        return eng.conv_bn(y, self.conv3, self.bn3, relu=True, residual=res)


def _shortcut(cin, cout):
    return nn.Sequential(_conv(cin, cout, 1), nn.BatchNorm2d(cout, momentum=BN_MOMENTUM))


class ChainOfBasicBlocks(nn.Module):
    def __init__(self, input_channel, ouput_channel, num_blocks=1):
        super().__init__()
        blocks = [BasicBlock(input_channel, ouput_channel, 1, _shortcut(input_channel, ouput_channel))]
        blocks += [BasicBlock(ouput_channel, ouput_channel) for _ in range(num_blocks - 1)]
        self.layers = nn.Sequential(*blocks)

    def run(self, eng, x):
        for blk in self.layers:
            x = blk.run(eng, x)
        return x


def _cbr(cin, cout, k, stride, relu=True):
    """Sequential(conv, bn[, relu]) as the transitions / fuse paths name them (indices 0, 1)."""
    mods = [_conv(cin, cout, k, stride, 1 if k == 3 else 0), nn.BatchNorm2d(cout)]
    if relu:
        mods.append(nn.ReLU(True))
    return nn.Sequential(*mods)


def run_cbr(eng, seq, x):
    return eng.conv_bn(x, seq[0], seq[1], relu=len(seq) > 2)


class HighResolutionModule(nn.Module):
    def __init__(self, channels, num_blocks, multi_scale_output=True):
        super().__init__()
        nb = len(channels)
        self.num_branches = nb
        self.branches = nn.ModuleList(
            nn.Sequential(*[BasicBlock(channels[b], channels[b]) for _ in range(num_blocks[b])]) for b in range(nb))
        self.fuse_layers = None
        if nb > 1:
            rows = []
            for i in range(nb if multi_scale_output else 1):
                row = []
                for j in range(nb):
                    if j == i:
                        row.append(None)
                    elif j > i:   # 1x1 conv + BN, upsampled 2^(j-i) by the fuse kernel (Interpolate has no params)
                        row.append(nn.Sequential(_conv(channels[j], channels[i], 1), nn.BatchNorm2d(channels[i]),
                                                 nn.Identity()))
                    else:         # chain of stride-2 3x3 conv + BN (+ReLU except last)
                        n = i - j
                        row.append(nn.Sequential(*[
                            _cbr(channels[j], channels[i] if k == n - 1 else channels[j], 3, 2, relu=k < n - 1)
                            for k in range(n)]))
                rows.append(nn.ModuleList(row))
            self.fuse_layers = nn.ModuleList(rows)

    def run_branches(self, eng, xs):
        """The parallel branches are independent until the fuse: each runs on its own stream lane."""
        ys = []
        forked = eng.fork(self.num_branches)
        for b in range(self.num_branches):
            if forked:
                eng.set_lane(b)
            y = xs[b]
            for blk in self.branches[b]:
                y = blk.run(eng, y)
            ys.append(y)
        if forked:
            eng.join(self.num_branches)
        return ys

    def run_fuse(self, eng, ys):
        """out_i = relu(sum_j f_ij(x_j)).  Every f_ij -- conv chain, BatchNorm statistics, and in backward the gradient
        into x_j -- depends on branch j only, so the terms run on stream lane j; the sums follow the join."""
        nb = self.num_branches
        if nb == 1:
            return ys
        handles = [[None] * nb for _ in self.fuse_layers]
        forked = eng.fork(nb) if eng.fuse_lanes else False
        for j in range(nb):
            if forked:
                eng.set_lane(j)
            for i, row in enumerate(self.fuse_layers):
                if j == i:
                    handles[i][j] = eng.fuse_term(ys[j], None, 0)
                elif j > i:
                    handles[i][j] = eng.conv_fuse_term(ys[j], row[j][0], row[j][1], j - i)
                else:
                    z = ys[j]
                    chain = row[j]
                    for k in range(len(chain) - 1):
                        z = run_cbr(eng, chain[k], z)
                    handles[i][j] = eng.conv_fuse_term(z, chain[-1][0], chain[-1][1], 0)
        if forked:
            eng.join(nb)
        return [eng.fuse(hs) for hs in handles]

    def run_both(self, eng, xs):
        """run_branches + run_fuse inside ONE forked region: lane j runs branch j and then the fuse terms that read it -- no
        join / fork pair between the two halves (a lane join costs the chip an idle gap on every lane but the slowest).
        -> (branch outputs, fused outputs)"""
        nb = self.num_branches
        if nb == 1 or not eng.fuse_lanes or not options.flag('FAMI_MERGE_FORK'):
            ys = self.run_branches(eng, xs)
            return ys, self.run_fuse(eng, ys)
        forked = eng.fork(nb)
        if not forked:
            ys = self.run_branches(eng, xs)
            return ys, self.run_fuse(eng, ys)
        ys = []
        handles = [[None] * nb for _ in self.fuse_layers]
        for j in range(nb):
            eng.set_lane(j)
            y = xs[j]
            for blk in self.branches[j]:
                y = blk.run(eng, y)
            ys.append(y)
            for i, row in enumerate(self.fuse_layers):
                if j == i:
                    handles[i][j] = eng.fuse_term(y, None, 0)
                elif j > i:
                    handles[i][j] = eng.conv_fuse_term(y, row[j][0], row[j][1], j - i)
                else:
                    z = y
                    chain = row[j]
                    for k in range(len(chain) - 1):
                        z = run_cbr(eng, chain[k], z)
                    handles[i][j] = eng.conv_fuse_term(z, chain[-1][0], chain[-1][1], 0)
        eng.join(nb)
        return ys, [eng.fuse(hs) for hs in handles]

    def run_lanes(self, eng, xs, last=False):
        """One module INSIDE a region the caller has forked (HRNetBody.run keeps the lanes of a stage forked across its modules):
        lane j runs branch j and the fuse terms that read it; fused output i is summed on lane i behind an all-to-all of lane
        events (Engine.lanes_sync) -- no join, no sums on lane 0.  -> (branch outputs, fused outputs, joined)"""
        nb = self.num_branches
        ys = []
        handles = [[None] * nb for _ in self.fuse_layers]
        for j in range(nb):
            eng.set_lane(j)
            y = xs[j]
            for blk in self.branches[j]:
                y = blk.run(eng, y)
            ys.append(y)
            for i, row in enumerate(self.fuse_layers):
                if j == i:
                    handles[i][j] = eng.fuse_term(y, None, 0)
                elif j > i:
                    handles[i][j] = eng.conv_fuse_term(y, row[j][0], row[j][1], j - i)
                else:
                    z = y
                    chain = row[j]
                    for k in range(len(chain) - 1):
                        z = run_cbr(eng, chain[k], z)
                    handles[i][j] = eng.conv_fuse_term(z, chain[-1][0], chain[-1][1], 0)
        if last or len(handles) < nb:
            # the stage's last module: its sums follow the stage's join on lane 0.  (With the all-to-all here the backward pass would
            # open the stage with a fork and record the all-to-all's events right behind the fork's waits, no kernel in between --
            # a pattern hipStreamEndCapture does not survive; the same goes for lanes without an output in front of the join.)
            eng.join(nb)
            return ys, [eng.fuse(hs) for hs in handles], True
        eng.lanes_sync(nb)
        outs = []
        for i, hs in enumerate(handles):
            eng.set_lane(i)
            outs.append(eng.fuse(hs))
        return ys, outs, False

    def run(self, eng, xs):
        return self.run_both(eng, xs)[1]


class HRNetBody(nn.Module):
    """Shared constructor / forward of HRNet and HRNetPlus."""

    def __init__(self, cfg, is_train=True, **kwargs):
        super().__init__()
        extra = cfg['MODEL']['EXTRA']
        self.is_train = is_train
        self.conv1 = _conv(3, 64, 3, 2, 1)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = _conv(64, 64, 3, 2, 1)
        self.bn2 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.layer1 = nn.Sequential(Bottleneck(64, 64, 1, _shortcut(64, 256)),
                                    *[Bottleneck(256, 64) for _ in range(3)])
        prev = [256]
        self.stage_branches = {}
        for s in (2, 3, 4):
            sc = extra['STAGE%d' % s]
            if sc['BLOCK'] != 'BASIC' or sc['FUSE_METHOD'] != 'SUM':
                raise ValueError('only BASIC blocks with SUM fusion are on the hot path')
            ch = [int(c) for c in sc['NUM_CHANNELS']]
            if sc['NUM_BRANCHES'] != len(sc['NUM_BLOCKS']):
                raise ValueError('NUM_BRANCHES({}) <> NUM_BLOCKS({})'.format(sc['NUM_BRANCHES'], len(sc['NUM_BLOCKS'])))
            setattr(self, 'transition%d' % (s - 1), self._make_transition(prev, ch))
            n_mod = sc['NUM_MODULES']
            setattr(self, 'stage%d' % s, nn.Sequential(*[
                HighResolutionModule(ch, sc['NUM_BLOCKS'], multi_scale_output=not (s == 4 and m == n_mod - 1))
                for m in range(n_mod)]))
            self.stage_branches[s] = sc['NUM_BRANCHES']
            prev = ch
        k = extra['FINAL_CONV_KERNEL']
        self.final_layer = nn.Conv2d(prev[0], cfg['MODEL']['NUM_JOINTS'], k, 1, 1 if k == 3 else 0)

    @staticmethod
    def _make_transition(prev, cur):
        mods = []
        for i, c in enumerate(cur):
            if i < len(prev):
                mods.append(None if prev[i] == c else _cbr(prev[i], c, 3, 1))
            else:
                n = i + 1 - len(prev)
                mods.append(nn.Sequential(*[_cbr(prev[-1], c if k == n - 1 else prev[-1], 3, 2) for k in range(n)]))
        return nn.ModuleList(mods)

    def freeze_weight(self):
        for p in self.parameters():
            p.requires_grad = False

    def run(self, eng, x):
        """x: engine tensor [N,H,W,3] -> (heatmap T [N,H/4,W/4,J], stage-4 outputs, pre-stage-4 inputs)."""
        outer = eng.wlane_scope
        eng.wlane_scope = outer or eng.stem_wlane     # serial chain: weight gradients on their own lane (Engine.__init__)
        eng.wlane_pair = True                         # ... on two of them, alternately (Engine._enter_wlane)
        eng.serial_scope = True                       # ... and BatchNorm backward statistics in the dgrad epilogues
        x = eng.conv_bn(x, self.conv1, self.bn1, relu=True)
        x = eng.conv_bn(x, self.conv2, self.bn2, relu=True)
        for blk in self.layer1:
            x = blk.run(eng, x)
        eng.wlane_scope = outer
        eng.wlane_pair = False
        eng.serial_scope = False
        ys = [x]
        stage4_in = None
        for s in (2, 3, 4):
            if s == 3:
                eng.join_late_weights()      # the Trainer packs the weight images of stage 3 .. the head beside the stem
            tr = getattr(self, 'transition%d' % (s - 1))
            xs = []
            eng.wlane_scope = outer or eng.stem_wlane
            eng.wlane_pair = True
            for i in range(self.stage_branches[s]):
                if tr[i] is None:
                    xs.append(ys[i])
                else:
                    z = ys[-1]
                    if isinstance(tr[i][0], nn.Sequential):
                        for step in tr[i]:
                            z = run_cbr(eng, step, z)
                    else:
                        z = run_cbr(eng, tr[i], z)
                    xs.append(z)
            eng.wlane_scope = outer
            eng.wlane_pair = False
            ys = xs
            if s == 2:
                eng.mark('stage2')      # backward: every gradient but the stem stretch's is final here (train.Trainer: early Adam)
            nb = self.stage_branches[s]
            persist = eng.persist_lanes and eng.fuse_lanes and nb > 1 and eng.fork(nb)      # the stage's lanes stay forked across its modules
            joined = False
            for mi, mod in enumerate(getattr(self, 'stage%d' % s)):
                if persist and not joined:
                    yb, ys, joined = mod.run_lanes(eng, ys, last=(mi == len(getattr(self, 'stage%d' % s)) - 1))
                else:
                    yb, ys = mod.run_both(eng, ys)
                if s == 4 and mi == 0:
                    # the reference's `feature = x3_list` is overwritten in place by stage4[0]'s
                    # branches (hrnet.py:156-157,323): HRNet.forward returns these branch outputs
                    stage4_in = list(yb)
            if persist and not joined:
                eng.join(nb)
        hm = run_conv(eng, self.final_layer, ys[0], out_f32=True)     # heatmaps are fp32 in every mode
        return hm, ys, stage4_in

"""Alignment_V15 (FAMI-Pose) on the MI355X HIP engine.

Drop-in for posetimation/zoo/Alignment/Alignment_V15.py:24-300: same registry
name, constructor `(cfg, phase, **kwargs)`, `forward(kf_x, sup_x)` ->
`(final_hm, kf_bb_hm, [mi_1..mi_6])` when constructed for the train phase else
`(final_hm, kf_bb_hm)` (:181-183), same state_dict keys, same init_weights
statistics (:185-214) and pretrained remap (:216-240).

Generalisations the reference class lacks: head width
C = STAGE2.NUM_CHANNELS[0], S = MODEL.NUM_SUPPORT_FRAMES supporting frames,
Linear in-features from MODEL.IMAGE_SIZE, DCN offset groups 12 when 12 | C else
C/4 (MODEL.DCN_OFFSET_GROUPS overrides).  For W48 / S=4 / 384x288 the graph is
exactly the reference's.
"""
import logging
import os.path as osp

import torch
import torch.nn as nn

from .. import options
from ..modules import ChainOfBasicBlocks, conv_bn_relu, run_conv
from ..runtime import EngineModule
from .hrnet import HRNetPlus, _hyper_parameters
from .registry import MODEL_REGISTRY, TRAIN_PHASE

MI_TEMPERATURE = 0.05


class DeformConv2d(nn.Module):
    """Parameter container with torchvision.ops.DeformConv2d's layout and default init."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, dilation=1):
        super().__init__()
        self.padding, self.dilation = padding, dilation
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        bound = 1.0 / (in_channels * kernel_size * kernel_size) ** 0.5
        nn.init.uniform_(self.bias, -bound, bound)


def _ceil_half(v, times=5):
    for _ in range(times):
        v = (v + 1) // 2
    return v


@MODEL_REGISTRY.register()
class Alignment_V15(EngineModule):

    @classmethod
    def get_model_hyper_parameters(cls, cfg):
        s = _hyper_parameters(cfg)
        if cfg.LOSS.HEATMAP_MSE.USE:
            s += "_MseLoss_{}".format(cfg.LOSS.HEATMAP_MSE.WEIGHT)
        return s

    def __init__(self, cfg, is_train, **kwargs):
        super().__init__()
        m = cfg['MODEL']
        self.num_joints = m['NUM_JOINTS']
        self.pretrained = m.get('PRETRAINED', '')
        self.is_train = (is_train == TRAIN_PHASE) or (is_train is True)
        self.pretrained_layers = ['*']
        self.hrnet = HRNetPlus(cfg, self.is_train)
        self.freeze_hrnet_weight = m['FREEZE_HRNET_WEIGHTS']
        C = int(m['EXTRA']['STAGE2']['NUM_CHANNELS'][0])
        S = int(m.get('NUM_SUPPORT_FRAMES', 4) or 4)
        img_w, img_h = m.get('IMAGE_SIZE', [288, 384])
        G = m.get('DCN_OFFSET_GROUPS', None) or (12 if C % 48 == 0 else C // 4)
        self.C, self.S, self.G = C, S, G
        # kornia.warp_affine's align_corners at Alignment_V15.py:135 (the call passes none): True = pixel-exact translation
        # (kornia >= 0.5 behaviour, this build's default), False = the kornia <= 0.4 default (translation scaled by W/(W-1))
        self.warp_align_corners = bool(m.get('WARP_ALIGN_CORNERS', True))
        h5, w5 = _ceil_half(img_h // 4), _ceil_half(img_w // 4)

        self.feat_global_offset_layers = nn.Sequential(
            ChainOfBasicBlocks(C, 16, num_blocks=1),
            *[conv_bn_relu(16, 16, 3, 2, 1, 1) for _ in range(5)],
            nn.Flatten(),
            nn.Linear(16 * h5 * w5, 64), nn.Linear(64, 64), nn.Linear(64, 2))
        self.combined_feat_layers = ChainOfBasicBlocks(2 * C, C, num_blocks=1)
        for k in (1, 2, 3, 4):
            setattr(self, 'dcn_offset_%d' % k, conv_bn_relu(C, 2 * 9 * G, 3, 1, 3, 3, has_bn=False, has_relu=False))
            setattr(self, 'dcn_mask_%d' % k, conv_bn_relu(C, 9 * G, 3, 1, 3, 3, has_bn=False, has_relu=False))
            setattr(self, 'dcn_%d' % k, DeformConv2d(C, C, 3, padding=3, dilation=3))
        self.sup_agg_block = ChainOfBasicBlocks(C * S, C, num_blocks=2)
        self.init_feature_agg_block = ChainOfBasicBlocks(2 * C, C, num_blocks=3)
        self.agg_final_layer = nn.Conv2d(C, self.num_joints, 3, 1, 1)
        self.init_weights()
        if self.freeze_hrnet_weight:
            self.hrnet.freeze_weight()

    # ------------------------------------------------------------------ init (Alignment_V15.py:185-248)
    def init_weights(self, *args, **kwargs):
        hrnet_names = set()
        for name, mod in self.named_modules():
            if name.split('.')[0] == 'hrnet':
                hrnet_names.add(name)
            if isinstance(mod, nn.Conv2d):
                nn.init.normal_(mod.weight, std=0.001)
                if mod.bias is not None:
                    nn.init.constant_(mod.bias, 0)
            elif isinstance(mod, nn.BatchNorm2d):
                nn.init.constant_(mod.weight, 1)
                nn.init.constant_(mod.bias, 0)
            elif 'bias' in dict(mod.named_parameters(recurse=False)):
                nn.init.constant_(mod.bias, 0)      # Linear / DeformConv2d keep their default weight init
        if self.pretrained and osp.isfile(self.pretrained):
            sd = torch.load(self.pretrained, map_location='cpu')
            sd = sd.get('state_dict', sd)
            if list(sd.keys())[0].startswith('module.'):
                sd = {k[7:]: v for k, v in sd.items()}
            remapped = {}
            for k, v in sd.items():
                top = k.split('.')[0]
                if top in hrnet_names:
                    remapped[k] = v
                elif 'hrnet.' + top in hrnet_names:     # plain HRNet checkpoint -> hrnet.* keys
                    remapped['hrnet.' + k] = v
            self.load_state_dict(remapped, strict=False)
        elif self.pretrained:
            logging.getLogger(__name__).error('=> please download pre-trained models first!')

    # ------------------------------------------------------------------ forward (Alignment_V15.py:113-183)
    def merged_predictors(self):
        """{k: (CatParam(weights), CatParam(biases))}: the offset and the mask predictor of DCN layer k (Alignment_V15.py:79-100;
        both read the same tensor at :144-158) as ONE convolution with 2GK + GK output channels.  Views, not copies: usable only
        while the parts are adjacent in memory (train.flatten_parameters lays the arena out that way, `adjacent_parameters`)."""
        if getattr(self, '_cat', None) is None:
            from ..engine import CatParam
            self._cat = {}
            for k in (1, 2, 3, 4):
                oc, mc = getattr(self, 'dcn_offset_%d' % k).conv, getattr(self, 'dcn_mask_%d' % k).conv
                self._cat[k] = (CatParam([oc.weight, mc.weight]), CatParam([oc.bias, mc.bias]))
        return self._cat

    def adjacent_parameters(self):
        """parameter groups a flat arena must keep adjacent, in this order (train.flatten_parameters)"""
        return [c.parts for pair in self.merged_predictors().values() for c in pair]

    def _dcn(self, eng, k, src, x):
        d = getattr(self, 'dcn_%d' % k)
        wcat, bcat = self.merged_predictors()[k]
        if eng.cat_usable(wcat, bcat):
            oc = getattr(self, 'dcn_offset_%d' % k).conv
            om = eng.conv(src, wcat, bcat, oc.stride[0], oc.padding[0], oc.dilation[0])
            return eng.dcn(x, om, None, d.weight, d.bias, self.G, d.padding, d.dilation)
        off = getattr(self, 'dcn_offset_%d' % k).run(eng, src)
        msk = getattr(self, 'dcn_mask_%d' % k).run(eng, src)
        return eng.dcn(x, off, msk, d.weight, d.bias, self.G, d.padding, d.dilation)

    def _translation(self, eng, diff):
        seq = self.feat_global_offset_layers
        z = seq[0].run(eng, diff)
        if options.number('FAMI_ABL_REGTAIL'):
            # upper-bound experiment (WRONG results, refused without FAMI_ALLOW_WRONG=1): the regressor ends after its first
            # n - 1 stride-2 stages; what a fused kernel for the rest of the chain could save at most
            from ..engine import T
            for i in range(1, options.number('FAMI_ABL_REGTAIL')):
                z = seq[i].run(eng, z)
            return T(torch.zeros(diff.shape[0], 2, device=eng.dev))
        for i in range(1, 6):
            z = seq[i].run(eng, z)
        z = eng.flatten_chw(z)
        for i in (7, 8, 9):
            z = eng.linear(z, seq[i])
        return z

    def _body(self, eng, kf_x, sup_x):
        B = kf_x.shape[0]
        S = sup_x.shape[1] // 3
        hm, feats, _ = self.hrnet.run(eng, eng.frames(kf_x, sup_x))
        feat = feats[0]
        kf_hm = eng.batch_slice(hm, 0, B, terminal=True)
        kf = eng.batch_slice(feat, 0, B)
        # The S translation regressors SHARE their weights and BatchNorm modules (Alignment_V15.py:125-135 applies the same
        # feat_global_offset_layers to every supporting frame) and are chains of ~40 tiny kernels each: they run on
        # separate stream lanes.  What they share is kept race-free by the engine: weight gradients accumulate in
        # lane-private buffers folded at the join (Engine.pgrad), and the BatchNorm running statistics are not touched
        # inside the lanes but updated afterwards frame by frame, the order the reference updates them in
        # (Engine.defer_bn / apply_deferred_bn).  `sup - kf` stays on lane 0: all S differences send gradient into the
        # same key-frame slice.
        sups = [eng.batch_slice(feat, (1 + i) * B, (2 + i) * B) for i in range(S)]
        diffs = [eng.sub(sup, kf) for sup in sups]
        aligned, shifts = [], []
        forked = eng.fork(min(S, 4)) if eng.regressor_lanes else False
        if forked:
            eng.defer_bn = []
        for i in range(S):
            if forked:
                eng.set_lane(i % 4)
            t = self._translation(eng, diffs[i])
            shifts.append(t)
            aligned.append(eng.shift(sups[i], t, self.warp_align_corners))
        if forked:
            eng.join(min(S, 4))
            eng.apply_deferred_bn()
        eng.wlane_scope = True      # the aggregation / DCN stack is one serial chain: its weight gradients go to their own lane
        agg_sup = self.sup_agg_block.run(eng, eng.concat(aligned))
        comb = self.combined_feat_layers.run(eng, eng.concat([agg_sup, kf]))
        comb = self._dcn(eng, 1, comb, comb)
        comb = self._dcn(eng, 2, comb, comb)
        al = self._dcn(eng, 3, comb, agg_sup)
        al = self._dcn(eng, 4, al, al)
        all_agg = self.init_feature_agg_block.run(eng, eng.concat([kf, al]))
        final = run_conv(eng, self.agg_final_layer, all_agg, out_f32=True)
        eng.wlane_scope = False

        eng.join_side()             # the deferred BatchNorm running-statistics update ran beside the aggregation / DCN stack
        outs = [eng.to_nchw(final), eng.to_nchw(kf_hm)]
        seeds = [lambda g: eng.seed_nchw(final, g), lambda g: eng.seed_nchw(kf_hm, g)]
        eng.aux = {'final': final, 'kf_hm': kf_hm, 'mis': [], 'shifts': shifts, 'agg_sup': agg_sup, 'aligned': al,
                   'all_agg': all_agg, 'kf_feat': kf}
        if self.is_train:
            fl = self.hrnet.final_layer

            def label_mi(f):     # feat_label_mi_estimation: A = hrnet.final_layer(Feat).detach(), Bt = final_hm
                a = eng.conv(_detached(f), fl.weight.detach(), None if fl.bias is None else fl.bias.detach(),
                             fl.stride[0], fl.padding[0], fl.dilation[0], out_f32=True)
                return eng.softmax_kl(eng.to_nchw(a), final, MI_TEMPERATURE)

            def feat_mi(f1, f2):  # feat_feat_mi_estimation: A = F1.detach(), Bt = F2
                return eng.softmax_kl(eng.to_nchw(f1), f2, MI_TEMPERATURE)

            # the six MI terms only READ the head's tensors: three stream lanes (each ~50-90 us of small dependent kernels
            # -- transposes, a 17-channel conv, the row softmax / KL pass; they were ~0.35 ms back to back on the head's
            # serial chain)
            # mi_6 = feat_feat_mi(kf, all_agg) is the SAME term as mi_2 (Alignment_V15.py:171,179 call it twice on the same
            # tensors): computed once; the output list still carries six scalars and each keeps its own gradient seed
            terms = [lambda: label_mi(all_agg), lambda: feat_mi(kf, all_agg), lambda: label_mi(agg_sup),
                     lambda: feat_mi(agg_sup, all_agg), lambda: label_mi(kf)]
            forked = eng.fork(3) if eng.mi_lanes else False
            mis = []
            for i, fn in enumerate(terms):
                if forked:
                    eng.set_lane(i % 3)
                mis.append(fn())
            if forked:
                eng.join(3)
            v6 = eng.empty(1)
            eng.call('fami_axpby_f32', mis[1][0].data_ptr(), None, v6.data_ptr(), 1, 1.0, 0.0)
            mis.append((v6, mis[1][1]))
            eng.aux['mis'] = mis
            eng.aux['mi_same'] = {5: 1}       # term index -> the earlier term it repeats (Trainer: one seed with the summed coefficient)
            for val, seed in mis:
                outs.append(val.reshape(()))
                seeds.append(lambda g, seed=seed: seed(1.0, g.reshape(1).contiguous()))
        return outs, seeds

    def forward(self, kf_x, sup_x, **kwargs):
        res = self._launch(self._body, kf_x, sup_x)
        if self.is_train:
            return res[0], res[1], list(res[2:])
        return res[0], res[1]


def _detached(t):
    from ..engine import T
    return T(t.data, False)

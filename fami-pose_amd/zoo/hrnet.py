"""HRNet / HRNetPlus front ends (posetimation/backbones/hrnet.py:186-332, :521-690)
on the HIP engine.  forward(x[N,3,H,W]) -> (heatmaps [N,J,H/4,W/4], feature list),
NCHW at the boundary like the reference."""
import logging
import os.path as osp

import torch

from ..modules import HRNetBody
from ..runtime import EngineModule
from .registry import MODEL_REGISTRY


def _hyper_parameters(cfg):
    sf = cfg.TRAIN.SCALE_FACTOR
    if not isinstance(sf, list):
        sf = [sf, sf]
    return "bbox_{}_rot_{}_scale_{}-{}".format(cfg.DATASET.BBOX_ENLARGE_FACTOR, cfg.TRAIN.ROT_FACTOR, 1 - sf[0],
                                               1 + sf[1])


class _HRNetFront(HRNetBody, EngineModule):
    plus = True

    def __init__(self, cfg, is_train=True, **kwargs):
        HRNetBody.__init__(self, cfg, is_train, **kwargs)
        self.backbone_pretrained = cfg['MODEL'].get('BACKBONE_PRETRAINED', '')

    @classmethod
    def get_model_hyper_parameters(cls, cfg):
        return _hyper_parameters(cfg)

    def init_weights(self, *args, **kwargs):
        path = self.backbone_pretrained
        if path and osp.isfile(path):
            sd = torch.load(path, map_location='cpu')
            sd = sd.get('state_dict', sd)
            if list(sd.keys())[0].startswith('module.'):
                sd = {k[7:]: v for k, v in sd.items()}
            self.load_state_dict(sd)
        elif path:
            logging.getLogger(__name__).error('=> please download pre-trained models first!')

    def forward(self, x, **kwargs):
        def body(eng, x):
            hm, ys, pre = self.run(eng, eng.from_nchw(x))
            feats = ys if self.plus else pre
            outs = [eng.to_nchw(hm)] + [eng.to_nchw(f) for f in feats]
            tens = [hm] + list(feats)
            seeds = [(lambda g, t=t: eng.seed_nchw(t, g)) for t in tens]
            return outs, seeds
        res = self._launch(body, x)
        return res[0], list(res[1:])


@MODEL_REGISTRY.register()
class HRNet(_HRNetFront):
    plus = False


@MODEL_REGISTRY.register()
class HRNetPlus(_HRNetFront):
    plus = True

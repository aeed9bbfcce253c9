"""Oracle (test infrastructure, BUILD CONTAINER ONLY): import the reference's
own Python modules from /root/reference so the restatement in oracle/ can be
validated against them and golden vectors generated (oracle/gen_golden.py).

Nothing here runs on the GPU box (/root/reference does not exist there) and no
reference source enters the repo: this file only prepares `sys.modules` so the
reference files import in this image, where yacs / torchvision / kornia / cv2
are absent and several reference packages are broken as shipped (SURVEY.md
2.3, 8c, Appendix C).  The two third-party ops the reference calls
(torchvision DeformConv2d, kornia warp_affine) are provided by oracle/ops.py --
that boundary is "parity unpinned" (see oracle/__init__.py).
"""
import importlib
import importlib.util
import os
import sys
import types

REF = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REF, 'posetimation'))


class AttrDict(dict):
    """cfg stand-in: attribute AND item access (hrnet.py:571 vs :590)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    @staticmethod
    def wrap(d):
        if isinstance(d, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in d.items()})
        return d


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


def _load_by_path(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


_loaded = None


def load():
    """Returns a namespace with the reference classes/functions on the hot path."""
    global _loaded
    if _loaded is not None:
        return _loaded
    assert available(), "reference tree not present (expected only in the build container)"
    import torch.nn as nn
    from . import ops as oops

    if REF not in sys.path:
        sys.path.insert(0, REF)
    reg = importlib.import_module('utils.utils_registry')

    eng = _pkg('engine')
    dfl = _pkg('engine.defaults')
    dfl.TRAIN_PHASE, dfl.VAL_PHASE, dfl.TEST_PHASE = 'train', 'validate', 'test'
    cst = types.ModuleType('engine.defaults.constant')
    cst.MODEL_REGISTRY = reg.Registry('MODEL')
    cst.CORE_FUNCTION_REGISTRY = reg.Registry('CORE_FUNCTION')
    cst.DATASET_REGISTRY = reg.Registry('DATASET')
    sys.modules['engine.defaults.constant'] = cst
    eng.defaults, dfl.constant = dfl, cst

    _pkg('posetimation', os.path.join(REF, 'posetimation'))
    _pkg('posetimation.zoo', os.path.join(REF, 'posetimation', 'zoo'))
    # shadows the unrelated HuggingFace `datasets` in site-packages
    _pkg('datasets', os.path.join(REF, 'datasets'))
    _pkg('datasets.process', os.path.join(REF, 'datasets', 'process'))
    cv2 = sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    # the one cv2 function the post-processing calls (affine_transform.py:38-41); cv2 itself is absent here
    cv2.getAffineTransform = oops.cv2_get_affine_transform

    class DeformConv2d(nn.Module):
        def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, groups=1, bias=True):
            super().__init__()
            import torch
            self.stride, self.padding, self.dilation = stride, padding, dilation
            self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k))
            self.bias = nn.Parameter(torch.zeros(cout))
            nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)

        def forward(self, x, offset, mask=None):
            return oops.deform_conv2d(x, offset, mask, self.weight, self.bias,
                                      self.stride, self.padding, self.dilation)

    tv = _pkg('torchvision')
    tvo = _pkg('torchvision.ops')
    tvd = types.ModuleType('torchvision.ops.deform_conv')
    tvd.DeformConv2d = DeformConv2d
    tvo.DeformConv2d = DeformConv2d
    tvo.deform_conv = tvd
    tv.ops = tvo
    sys.modules['torchvision.ops.deform_conv'] = tvd
    ko = _pkg('kornia')
    kg = types.ModuleType('kornia.geometry')
    kg.warp_affine = oops.warp_affine_like
    ko.geometry = kg
    sys.modules['kornia.geometry'] = kg

    ns = types.SimpleNamespace()
    hr = importlib.import_module('posetimation.backbones.hrnet')
    layers = importlib.import_module('posetimation.layers')
    hp = importlib.import_module('datasets.process.heatmaps_process')
    ns.HRNet, ns.HRNetPlus, ns.HighResolutionModule = hr.HRNet, hr.HRNetPlus, hr.HighResolutionModule
    ns.BasicBlock, ns.Bottleneck = layers.BasicBlock, layers.Bottleneck
    ns.ChainOfBasicBlocks, ns.conv_bn_relu = layers.ChainOfBasicBlocks, layers.conv_bn_relu
    ns.generate_heatmaps, ns.get_max_preds = hp.generate_heatmaps, hp.get_max_preds
    ns.get_final_preds = hp.get_final_preds
    ns.JointMSELoss = _load_by_path('ref_mse_loss', 'posetimation/loss/mse_loss.py').JointMSELoss
    ns.accuracy = _load_by_path('ref_evaluate', 'engine/core/utils/evaluate.py').accuracy
    ns.Alignment_V15 = _load_by_path('ref_alignment_v15', 'posetimation/zoo/Alignment/Alignment_V15.py').Alignment_V15
    ns.AttrDict = AttrDict
    _loaded = ns
    return ns


def ref_cfg(width=48, freeze=False):
    from .model import make_cfg
    c = make_cfg(width=width, freeze=freeze)
    c.update({'DATASET': {'BBOX_ENLARGE_FACTOR': 1.25}, 'TRAIN': {'ROT_FACTOR': 45, 'SCALE_FACTOR': 0.35},
              'LOSS': {'HEATMAP_MSE': {'USE': True, 'WEIGHT': 1.0}}})
    return AttrDict.wrap(c)

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


def load_posetrack_evaluate():
    """-> the reference's PoseTrack_Alignment class (datasets/zoo/posetrack/PoseTrack_Alignment.py), imported only so
    that its `evaluate` method (:883-1037, the PoseTrack JSON writer) can be run on synthetic predictions.  Everything
    the module imports but the image lacks is a stand-in that `evaluate` never touches, with three exceptions that it
    does reach: the two joint-name lists (`datasets.zoo.coco.COCO_joint`, `datasets.zoo.posetrack.pose_topology.
    POSETRACK_joint` -- modules MISSING from the released reference, keypoints_ord.py:10-11; the standard COCO-17 /
    PoseTrack-15 orders are supplied) and `evaluate_simple.evaluate` (vendored poseval: needs shapely + ground truth;
    replaced by a function returning zeros AFTER the JSON files have been written)."""
    ns = load()
    import numpy as np

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    coco = ['nose', 'left_eye', 'right_eye', 'left_ear', 'right_ear', 'left_shoulder', 'right_shoulder', 'left_elbow',
            'right_elbow', 'left_wrist', 'right_wrist', 'left_hip', 'right_hip', 'left_knee', 'right_knee',
            'left_ankle', 'right_ankle']
    ptk = ['right_ankle', 'right_knee', 'right_hip', 'left_hip', 'left_knee', 'left_ankle', 'right_wrist',
           'right_elbow', 'right_shoulder', 'left_shoulder', 'left_elbow', 'left_wrist', 'neck', 'nose', 'head_top']
    _pkg('datasets.zoo', os.path.join(REF, 'datasets', 'zoo'))
    mod('datasets.zoo.coco', COCO_joint=coco, COCO_joint_paris=[])
    _pkg('datasets.zoo.posetrack', os.path.join(REF, 'datasets', 'zoo', 'posetrack'))
    mod('datasets.zoo.posetrack.pose_topology', POSETRACK_joint=ptk)
    _pkg('datasets.zoo.jhmdb')
    mod('datasets.zoo.jhmdb.pose_topology', JHMDB_Keypoint_Ordering=[])
    mod('pycocotools')
    mod('pycocotools.coco', COCO=object)
    mod('termcolor', colored=lambda s, *a, **k: s)
    # `from datasets.process import ...`: the package __init__ is broken as shipped; expose the names
    # PoseTrack_Alignment imports from the sub-modules that do import
    proc = sys.modules['datasets.process']
    aff = importlib.import_module('datasets.process.affine_transform')
    pp = importlib.import_module('datasets.process.pose_process')
    hp = importlib.import_module('datasets.process.heatmaps_process')
    st = importlib.import_module('datasets.process.structure.data_format')
    for m, names in ((aff, ('get_affine_transform', 'exec_affine_transform', 'dark_get_affine_transform')),
                     (pp, ('fliplr_joints', 'half_body_transform')), (hp, ('generate_heatmaps',)),
                     (st, ('convert_data_to_annorect_struct',))):
        for n in names:
            setattr(proc, n, getattr(m, n))
    mod('datasets.transforms', build_transforms=lambda *a, **k: None)
    mod('datasets.zoo.base', VideoDataset=object)
    sys.modules['engine.defaults.constant'].DATASET_REGISTRY = sys.modules['engine.defaults.constant'].DATASET_REGISTRY
    cv2 = sys.modules['cv2']
    for n in ('IMREAD_COLOR', 'IMREAD_IGNORE_ORIENTATION', 'COLOR_BGR2RGB', 'INTER_LINEAR'):
        setattr(cv2, n, 0)
    pu = _load_by_path('ref_posetrack_utils', 'datasets/zoo/posetrack/posetrack_utils/posetrack_utils.py')
    calls = []
    es = types.SimpleNamespace(evaluate=lambda annot_dir, out_dir, eval_track=False: (calls.append(out_dir), [np.zeros(8)])[1])
    mod('datasets.zoo.posetrack.posetrack_utils', video2filenames=pu.video2filenames, evaluate_simple=es)
    _pkg('thirdparty')
    mod('thirdparty.clustering', k_means=None)
    m = _load_by_path('datasets.zoo.posetrack.PoseTrack_Alignment', 'datasets/zoo/posetrack/PoseTrack_Alignment.py')
    ns.PoseTrack_Alignment = m.PoseTrack_Alignment
    ns.convert_data_to_annorect_struct = st.convert_data_to_annorect_struct
    ns.video2filenames = pu.video2filenames
    return ns


def ref_cfg(width=48, freeze=False):
    from .model import make_cfg
    c = make_cfg(width=width, freeze=freeze)
    c.update({'DATASET': {'BBOX_ENLARGE_FACTOR': 1.25}, 'TRAIN': {'ROT_FACTOR': 45, 'SCALE_FACTOR': 0.35},
              'LOSS': {'HEATMAP_MSE': {'USE': True, 'WEIGHT': 1.0}}})
    return AttrDict.wrap(c)

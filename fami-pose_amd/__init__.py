"""fami-pose_amd: MI355X-native (gfx950) implementation of the FAMI-Pose
temporal-alignment training hot path, behind the reference's model-registry API.

    from fami_pose_amd import build_model, default_cfg
    model = build_model(default_cfg(), 'train').cuda()
    final_hm, kf_bb_hm, mi = model(kf_x, sup_x)

Compute runs exclusively in libfami_hip.so (hand-written HIP); importing the
package does not need a GPU, running a model does.
"""
import os as _os

# ROCm runtime: kernel arguments in device memory (lower launch latency; -4..5 % step time with ~4000 launches per step).
# Only effective when this package is imported before the HIP runtime initialises; harmless otherwise.
_os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

from .config import CfgNode, default_cfg
from .zoo import (MODEL_REGISTRY, CORE_FUNCTION_REGISTRY, DATASET_REGISTRY, TRAIN_PHASE, VAL_PHASE, TEST_PHASE,
                  build_model, get_model_hyperparameter, Alignment_V15, HRNet, HRNetPlus)

__all__ = ['CfgNode', 'default_cfg', 'MODEL_REGISTRY', 'CORE_FUNCTION_REGISTRY', 'DATASET_REGISTRY', 'TRAIN_PHASE',
           'VAL_PHASE', 'TEST_PHASE', 'build_model', 'get_model_hyperparameter', 'Alignment_V15', 'HRNet',
           'HRNetPlus']

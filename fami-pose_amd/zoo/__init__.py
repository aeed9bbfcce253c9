"""posetimation/zoo front door: MODEL_REGISTRY + build_model (posetimation/zoo/__init__.py:9-12)."""
from .registry import (CORE_FUNCTION_REGISTRY, DATASET_REGISTRY, MODEL_REGISTRY, TEST_PHASE, TRAIN_PHASE,
                       VAL_PHASE, Registry)
from .build import build_model, get_model_hyperparameter
from .hrnet import HRNet, HRNetPlus
from .alignment_v15 import Alignment_V15, DeformConv2d

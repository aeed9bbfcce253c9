"""build_model / get_model_hyperparameter: posetimation/zoo/build.py:12-88.

Same call contract: `MODEL_REGISTRY.get(cfg.MODEL.NAME)(cfg, phase, **kwargs)`,
`.train()` for the train phase when MODEL.INIT_WEIGHTS, `.eval()` otherwise."""
from .registry import MODEL_REGISTRY, TRAIN_PHASE


def build_model(cfg, phase, **kwargs):
    model = MODEL_REGISTRY.get(cfg.MODEL.NAME)(cfg, phase, **kwargs)
    if phase == TRAIN_PHASE and cfg.MODEL.INIT_WEIGHTS:
        model.train()
    if phase != TRAIN_PHASE:
        model.eval()
    return model


def get_model_hyperparameter(cfg, **kwargs):
    return MODEL_REGISTRY.get(cfg.MODEL.NAME).get_model_hyper_parameters(cfg)

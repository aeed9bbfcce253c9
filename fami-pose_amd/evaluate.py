"""Inference step of the hot path (the model-side half): what the reference's eval loop does per
batch in engine/core/functions/alignment_mi_function_term6_1.py:256-276 -- `model(kf, sup) -> (pred, kf_bb)` under
`torch.no_grad()`, then `get_final_preds(pred, center, scale)` -- with the decode on device (loss.get_final_preds), so
only [B,J,2] coordinates and [B,J,1] confidences leave the GPU instead of two full heatmap stacks.
The dataset-level driver (PoseTrack JSON writer, vendored poseval AP) stays out of scope."""
import torch

from .loss import get_final_preds


@torch.no_grad()
def predict(model, kf_x, sup_x, center, scale):
    """-> (preds [B,J,2] image coordinates, maxvals [B,J,1], final_hm [B,J,H/4,W/4]) as device tensors.
    `model` is a val/test-phase Alignment_V15 (2-tuple forward) or a train-phase one (3-tuple; MI terms ignored)."""
    out = model(kf_x, sup_x)
    final_hm = out[0]
    preds, maxvals = get_final_preds(final_hm, center, scale)
    return preds, maxvals, final_hm

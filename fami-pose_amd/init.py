"""Realistic-scale weight initialisation for benchmarks and demos.

The reference's `init_weights` (Alignment_V15.py:185-214) draws every conv weight from N(0, 0.001^2), which makes
untrained outputs ~1e-10 and BatchNorm / argmax degenerate.  Benchmarks therefore re-initialise at
the scale of a trained network.  The oracle (test infrastructure) carries its own copy of this function for the
parity tests; tests/test_host.py checks the two stay identical.
"""
import torch
import torch.nn as nn


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

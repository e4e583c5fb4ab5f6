"""Fresh-weight initialisation with the reference's rule.

reference unet6.py:123-130 (`DEFAULT_INITIALIZER` = xavier_uniform with gain sqrt(scale or 1e-10)),
:165-168 / :212-215 (zero biases), nn.GroupNorm defaults (weight 1, bias 0); `init_scale=0.` on
ResidualBlock.conv2 (:352), AttentionBlock.project_out (:310) and the last conv (:414) -- which is
gain 1e-5, not exactly zero.  Same distribution as the reference, not the same stream.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

_ZERO_SCALE = (".conv2.weight", ".project_out.weight", "out_conv.2.weight")


def xavier_like_params(shapes, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for k, shp in shapes.items():
        if len(shp) == 1:
            is_norm_w = k.endswith(".weight")
            out[k] = torch.ones(shp) if is_norm_w else torch.zeros(shp)
            continue
        rf = 1
        for d in shp[2:]:
            rf *= d
        fan_in, fan_out = shp[1] * rf, shp[0] * rf
        scale = 0.0 if k.endswith(_ZERO_SCALE) else 1.0
        gain = math.sqrt(scale or 1e-10)
        a = gain * math.sqrt(6.0 / (fan_in + fan_out))
        out[k] = (torch.rand(shp, generator=g) * 2 - 1) * a
    return out

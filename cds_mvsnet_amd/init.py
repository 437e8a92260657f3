"""Deterministic initialiser used where no checkpoint is available (benchmarks, fixtures).

There is no network access on the build / GPU boxes and the reference ships no licence for its
checkpoints, so synthetic weights are generated here.  The values are chosen so that every
activation stays O(1) and BatchNorm running statistics are non-trivial (a fold bug would show).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


@torch.no_grad()
def seeded_init_(module: nn.Module, seed: int = 0) -> nn.Module:
    """Fill every parameter / buffer of ``module`` from a CPU generator seeded with ``seed``.

    Works on any module tree that uses the reference's state-dict names (ours or the reference's
    own classes), because it only looks at tensor names and shapes."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    for name in sorted(sd.keys()):
        t = sd[name]
        if name.endswith("num_batches_tracked"):
            t.fill_(100)
            continue
        shape = tuple(t.shape)
        if name.endswith("running_mean"):
            v = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("running_var"):
            v = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith(".bn.weight") or (len(shape) == 1 and name.endswith(".weight")):
            v = 0.75 + 0.5 * torch.rand(shape, generator=g)
        elif name.endswith(".bias"):
            v = 0.1 * torch.randn(shape, generator=g)
        else:  # convolution kernels
            if "att_convs" in name:
                std = 0.1
            elif "att_weights.3" in name:
                # blend logits of the size the trained checkpoints have.  Measured on the three shipped checkpoints
                # (profiles/r02_trained_blend_regime.md): per-layer logit gaps 0.002-0.09 at T = 0.01 (largest softmax
                # weight 0.5-0.9 on average, <= 3 % of pixels saturated), against 0.008-0.05 (0.5-0.95; up to 55 %
                # saturated in conv10) with this std: the seeded regime is at least as hard on fp32 round-off as the
                # trained one (the reference's own fp32 forward misses its float64 evaluation by 1.5e-4 .. 3.8e-4 with
                # these weights and by 1e-5 .. 7e-5 with the trained ones).  O(1) logits would make the blend a hard
                # switch everywhere, which no trained checkpoint does.
                std = 0.01
            else:
                is_transposed = "conv7.conv" in name or "conv9.conv" in name or "conv11.conv" in name or ".deconv." in name
                fan_in = (shape[0] if is_transposed else shape[1]) * math.prod(shape[2:])
                if is_transposed:
                    fan_in = fan_in / 8.0 if len(shape) == 5 else fan_in / 4.0  # stride-2: 1/2^d of taps hit
                std = math.sqrt(2.0 / fan_in)
            v = std * torch.randn(shape, generator=g)
        t.copy_(v.to(t.dtype))
    return module

"""Training loss of CDS-MVSNet (reference: models/losses.py:6-48): per stage smooth-L1 on depth / interval, balanced
binary cross-entropy on the feature-distance volume, curvature regulariser; smooth-L1 on the refined depth."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _masked_mean(values, mask, count):
    """mean(values[mask]) without the boolean gather (its backward sorts the indices: ~300 small launches per step)."""
    return torch.where(mask, values, 0.0).sum() / count


def final_loss(inputs, depth_gt_ms, mask_ms, **kwargs):
    weights = kwargs.get("dlossw", None)
    interval = kwargs.get("depth_interval", 1.0)
    interval = interval.unsqueeze(-1).unsqueeze(-1)
    total = torch.zeros((), dtype=torch.float32, device=mask_ms["stage1"].device)
    depth_loss = 0.0
    for key in ("stage1", "stage2", "stage3"):
        st = inputs[key]
        mask = mask_ms[key] > 0.5
        count = mask.sum()
        depth_loss = _masked_mean(F.smooth_l1_loss(st["depth"] / interval, depth_gt_ms[key] / interval, reduction="none"), mask, count)
        curv_reg = _masked_mean(st["norm_curv"].squeeze(1), mask, count)
        feat_loss = 0.0
        if "feat_distance" in st:
            dist, target = st["feat_distance"], st["feat_target"]
            m = mask.unsqueeze(1)
            n = count * target.size(1)
            pos = _masked_mean(target, m, 1.0)
            neg = n - pos
            bce = F.binary_cross_entropy_with_logits(dist, target, reduction="none", pos_weight=neg / pos)
            feat_loss = _masked_mean(bce, m, n)
        term = depth_loss + 5 * feat_loss + 0.1 * curv_reg
        total = total + (weights[int(key[-1]) - 1] * term if weights is not None else term)
    if "refined_depth" in inputs:
        mask = mask_ms["stage4"] > 0.5
        depth_loss = _masked_mean(F.smooth_l1_loss(inputs["refined_depth"] / interval, depth_gt_ms["stage4"] / interval, reduction="none"),
                                  mask, mask.sum())
        total = total + 2 * depth_loss
    return total, depth_loss

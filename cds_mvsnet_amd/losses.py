"""Training loss of CDS-MVSNet (reference: models/losses.py:6-48): per stage smooth-L1 on depth / interval, balanced
binary cross-entropy on the feature-distance volume, curvature regulariser; smooth-L1 on the refined depth."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def final_loss(inputs, depth_gt_ms, mask_ms, **kwargs):
    weights = kwargs.get("dlossw", None)
    interval = kwargs.get("depth_interval", 1.0)
    interval = interval.unsqueeze(-1).unsqueeze(-1)
    total = torch.zeros((), dtype=torch.float32, device=mask_ms["stage1"].device)
    depth_loss = 0.0
    for key in ("stage1", "stage2", "stage3"):
        st = inputs[key]
        mask = mask_ms[key] > 0.5
        depth_loss = F.smooth_l1_loss((st["depth"] / interval)[mask], (depth_gt_ms[key] / interval)[mask], reduction="mean")
        curv_reg = torch.mean(st["norm_curv"].squeeze(1)[mask])
        feat_loss = 0.0
        if "feat_distance" in st:
            dist, target = st["feat_distance"], st["feat_target"]
            m = mask.unsqueeze(1).repeat(1, target.size(1), 1, 1)
            pos = target[m].sum()
            neg = torch.numel(target[m]) - pos
            feat_loss = F.binary_cross_entropy_with_logits(dist[m], target[m], reduction="mean", pos_weight=neg / pos)
        term = depth_loss + 5 * feat_loss + 0.1 * curv_reg
        total = total + (weights[int(key[-1]) - 1] * term if weights is not None else term)
    if "refined_depth" in inputs:
        mask = mask_ms["stage4"] > 0.5
        depth_loss = F.smooth_l1_loss((inputs["refined_depth"] / interval)[mask], (depth_gt_ms["stage4"] / interval)[mask],
                                      reduction="mean")
        total = total + 2 * depth_loss
    return total, depth_loss

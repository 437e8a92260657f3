"""Training loss of CDS-MVSNet (reference: models/losses.py:6-48): per stage smooth-L1 on depth / interval, balanced
binary cross-entropy on the feature-distance volume, curvature regulariser; smooth-L1 on the refined depth.

Device tensors run the HIP kernels of csrc/loss.hip behind ONE autograd node (8 launches forward, 4 backward at three stages + the
refined depth; the ATen formulation below is ~90 + ~150 launches of a launch-bound step); CPU tensors - the golden-vector check of the
formula against the reference's values (golden set G11 under tests/) - run the same formula in ATen."""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn.functional as F

FUSED_LOSS = os.environ.get("CDS_FUSED_LOSS", "1") == "1"     # 0: the ATen formulation on device tensors as well (A/B, tests)


def _masked_mean(values, mask, count):
    """mean(values[mask]) without the boolean gather (its backward sorts the indices: ~300 small launches per step)."""
    return torch.where(mask, values, 0.0).sum() / count


class _FusedLoss(torch.autograd.Function):
    """(total, last depth loss) of final_loss from the flat tensor list of the stages: per stage depth, gt, mask, then norm_curv if
    `layout[s][0]`, then feat_distance, feat_target if `layout[s][1]`.  `weights[s]` multiplies the stage's term."""

    @staticmethod
    def forward(ctx, interval, weights, layout, *tensors):
        from . import _lib, ops
        from ._lib import check
        lib = _lib.load()
        dev = tensors[0].device
        st = ops._stream(tensors[0])
        interval = interval.detach().float().contiguous().reshape(-1)
        B0 = tensors[0].shape[0]
        if interval.numel() == 1 and B0 > 1:
            interval = interval.expand(B0).contiguous()      # a scalar interval broadcasts, as in final_loss_aten
        if interval.numel() != B0 or interval.device != tensors[0].device:
            # the kernels read interval[b] for b < B on the depth's device: anything else is an out-of-bounds / wrong-device read
            raise ValueError(f"final_loss: depth_interval must hold one value per batch item on the depth's device "
                             f"(got {interval.numel()} values on {interval.device} for a batch of {B0} on {tensors[0].device})")
        stages, k = [], 0
        for has_nc, has_feat in layout:
            depth, gt, mask = (t.detach().float().contiguous() for t in tensors[k:k + 3])
            k += 3
            nc = dist = target = None
            if has_nc:
                nc = tensors[k].detach().float().contiguous()
                k += 1
            if has_feat:
                dist, target = (t.detach().float().contiguous() for t in tensors[k:k + 2])
                k += 2
            B, hw = depth.shape[0], depth[0].numel()
            if gt.shape != depth.shape or mask.shape != depth.shape or (nc is not None and nc.numel() != depth.numel()) or \
                    (dist is not None and (dist.shape != target.shape or dist.shape[0] != B or dist[0, 0].numel() != hw)):
                raise ValueError("final_loss: stage tensors disagree in shape")
            Dp = dist.shape[1] if dist is not None else 1
            recA = torch.empty((lib.cds_loss_records(B * hw), 4), dtype=torch.float64, device=dev)
            recB = torch.empty((lib.cds_loss_records(B * hw * Dp),), dtype=torch.float64, device=dev) if dist is not None else None
            p = lambda t: t.data_ptr() if t is not None else None                                   # noqa: E731
            check(lib.cds_loss_stage_f32(p(depth), p(gt), p(mask), p(nc), p(dist), p(target), p(interval), B, hw, Dp, p(recA), p(recB), st),
                  "cds_loss_stage_f32")
            stages.append((depth, gt, mask, nc, dist, target, recA, recB, B, hw, Dp))
        n = len(stages)
        total = torch.empty((), dtype=torch.float32, device=dev)
        dloss = torch.empty((), dtype=torch.float32, device=dev)
        scalars = torch.empty((n, 4), dtype=torch.float64, device=dev)
        arr = lambda ctype, vals: (ctype * n)(*vals)                                                # noqa: E731
        check(lib.cds_loss_final_f32(arr(ctypes.c_void_p, [s[6].data_ptr() for s in stages]),
                                     arr(ctypes.c_void_p, [s[7].data_ptr() if s[7] is not None else None for s in stages]),
                                     arr(ctypes.c_longlong, [s[8] * s[9] for s in stages]), arr(ctypes.c_int, [s[10] for s in stages]),
                                     arr(ctypes.c_float, [float(w) for w in weights]),
                                     arr(ctypes.c_int, [1 if s[3] is not None else 0 for s in stages]), n, total.data_ptr(),
                                     dloss.data_ptr(), scalars.data_ptr(), st), "cds_loss_final_f32")
        ctx.stages = [(s[8], s[9], s[10], s[3] is not None, s[4] is not None) for s in stages]
        ctx.weights = [float(w) for w in weights]
        ctx.layout = layout
        saved = [interval, scalars]
        for s in stages:
            saved += [s[0], s[1], s[2]] + ([s[4], s[5]] if s[4] is not None else [])
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(dloss)
        return total, dloss

    @staticmethod
    def backward(ctx, gtotal, _gdl):
        from . import _lib, ops
        from ._lib import check
        lib = _lib.load()
        saved = ctx.saved_tensors
        interval, scalars = saved[0], saved[1]
        st = ops._stream(interval)
        gtotal = gtotal.detach().float().contiguous()
        grads, k = [], 2
        for s, (B, hw, Dp, has_nc, has_feat) in enumerate(ctx.stages):
            depth, gt, mask = saved[k:k + 3]
            k += 3
            dist = target = None
            if has_feat:
                dist, target = saved[k:k + 2]
                k += 2
            gdepth = torch.empty_like(depth)
            gnc = torch.empty((B, 1) + tuple(depth.shape[1:]), dtype=torch.float32, device=depth.device) if has_nc else None
            gdist = torch.empty_like(dist) if has_feat else None
            p = lambda t: t.data_ptr() if t is not None else None                                   # noqa: E731
            check(lib.cds_loss_stage_bwd_f32(p(depth), p(gt), p(mask), p(dist), p(target), p(interval), p(gtotal), scalars[s].data_ptr(),
                                             ctx.weights[s], B, hw, Dp, p(gdepth), p(gnc), p(gdist), st), "cds_loss_stage_bwd_f32")
            grads += [gdepth, None, None] + ([gnc] if has_nc else []) + ([gdist, None] if has_feat else [])
        return (None, None, None, *grads)


def _final_loss_fused(inputs, depth_gt_ms, mask_ms, weights, interval):
    layout, tensors, w = [], [], []
    for i, key in enumerate(("stage1", "stage2", "stage3")):
        st = inputs[key]
        has_feat = "feat_distance" in st
        layout.append((True, has_feat))
        tensors += [st["depth"], depth_gt_ms[key], mask_ms[key], st["norm_curv"]]
        if has_feat:
            tensors += [st["feat_distance"], st["feat_target"]]
        w.append(float(weights[i]) if weights is not None else 1.0)
    if "refined_depth" in inputs:
        layout.append((False, False))
        tensors += [inputs["refined_depth"], depth_gt_ms["stage4"], mask_ms["stage4"]]
        w.append(2.0)
    return _FusedLoss.apply(interval, tuple(w), tuple(layout), *tensors)


def final_loss(inputs, depth_gt_ms, mask_ms, **kwargs):
    weights = kwargs.get("dlossw", None)
    if FUSED_LOSS and mask_ms["stage1"].is_cuda and torch.is_tensor(kwargs.get("depth_interval", None)):
        return _final_loss_fused(inputs, depth_gt_ms, mask_ms, weights, kwargs["depth_interval"])
    return final_loss_aten(inputs, depth_gt_ms, mask_ms, **kwargs)


def final_loss_aten(inputs, depth_gt_ms, mask_ms, **kwargs):
    """The loss in ATen operations (the formulation the golden vectors G11 pin against the reference)."""
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

"""Autograd ops of the CostRegNet training path on the hand-written HIP kernels (reference:
models/module.py:80-160 Conv3d / Deconv3d + BatchNorm3d(train) + ReLU, :305-315 the U-Net wiring).

* :class:`Conv3dK3` — ``nn.Conv3d(k=3, p=1, stride 1|2, bias=False)`` / ``nn.ConvTranspose3d(k=3, s=2, p=1, op=1, bias=False)``:
  forward and data gradient on the inference kernels (``cds_conv3d_k3_f32`` / ``cds_deconv3d_k3s2_f32``: the data gradient
  of a stride-2 convolution is the transposed convolution with the same weights and vice versa, a stride-1 convolution's is
  the convolution with flipped, transposed weights), weight gradient on ``cds_conv3d_wgrad_f32``.
* :class:`BnRelu3d` — BatchNorm3d with batch statistics + ReLU + optional residual, fused forward (``cds_bn3d_stats_f32`` ->
  ``cds_bn3d_norm_f32``) and closed-form backward (``cds_bn3d_bwd_reduce_f32`` -> ``cds_bn3d_bwd_norm_f32``); running
  statistics are updated like ``nn.BatchNorm3d`` (momentum 0.1, unbiased variance).

All kernels are fp32; under bf16 autocast the Functions cast their inputs up (the reference trains in fp32)."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib, _scratch, ops
from ._lib import check

Tensor = torch.Tensor


def _dev(t: Tensor) -> int:
    return ops._dev(t, "tensor")


def conv3d_wgrad(g: Tensor, xin: Tensor, stride: int) -> Tensor:
    """dw[a][b][kz][ky][kx] = sum_{batch, o} g[:, a][o] * xin[:, b][stride * o - 1 + k]  ->  [Ca,Cb,3,3,3]."""
    B, Ca, Do, Ho, Wo = g.shape
    Bx, Cb, Di, Hi, Wi = xin.shape
    if Bx != B:
        raise ValueError("conv3d_wgrad: batch mismatch")
    dw = _scratch.zeros((Ca, Cb, 3, 3, 3), torch.float32, g.device)
    _scratch.audit_note(dw)
    side = _scratch.side_stream(g.device)
    if side is None:
        check(_lib.load().cds_conv3d_wgrad_f32(_dev(g), _dev(xin), dw.data_ptr(), B, Ca, Cb, Do, Ho, Wo, Di, Hi, Wi, stride,
                                               ops._stream(g)), "cds_conv3d_wgrad_f32")
        return dw
    with torch.cuda.stream(side):                                # a leaf of the backward pass: overlaps with the data-gradient chain
        check(_lib.load().cds_conv3d_wgrad_f32(_dev(g), _dev(xin), dw.data_ptr(), B, Ca, Cb, Do, Ho, Wo, Di, Hi, Wi, stride,
                                               side.cuda_stream), "cds_conv3d_wgrad_f32")
    g.record_stream(side)
    xin.record_stream(side)
    return dw


def _per_item(fn, x: Tensor) -> Tensor:
    """fn on every batch item (the inference kernels take one item per call); a batch of one is not copied again."""
    if x.shape[0] == 1:
        return fn(x[0]).unsqueeze(0)
    return torch.stack([fn(x[b]) for b in range(x.shape[0])])


def pack_conv3d(w: Tensor, mode: int, dgrad: bool):
    """(forward layout, data-gradient layout | None) of a 3x3x3 weight in one launch (cds_pack_conv3d_f32; mode 0 / 1: Conv3d stride
    1 / 2, mode 2: ConvTranspose3d)."""
    a, b = w.shape[:2]
    w = w.detach().float().contiguous()
    cin, cout = (a, b) if mode == 2 else (b, a)
    f = torch.empty((cin, 27, cout), dtype=torch.float32, device=w.device)
    d = torch.empty((cout, 27, cin), dtype=torch.float32, device=w.device) if dgrad else None
    check(_lib.load().cds_pack_conv3d_f32(_dev(w), f.data_ptr(), d.data_ptr() if dgrad else None, a, b, mode, ops._stream(w)),
          "cds_pack_conv3d_f32")
    return f, d


class Conv3dK3(torch.autograd.Function):
    """x [B,Cin,D,H,W], weight (Conv3d: [Cout,Cin,3,3,3]; ConvTranspose3d: [Cin,Cout,3,3,3]) -> y."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight, stride: int, transposed: bool):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.stride, ctx.transposed = stride, transposed
        wpk, ctx.dgrad_pack = pack_conv3d(weight, 2 if transposed else (0 if stride == 1 else 1), ctx.needs_input_grad[0])
        if transposed:
            return _per_item(lambda xb: ops.deconv3d_k3s2(xb, wpk, None, relu=False), x)
        return _per_item(lambda xb: ops.conv3d_k3(xb, wpk, None, stride=stride, relu=False), x)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous().float()
        w = weight.detach().float()
        B = x.shape[0]
        dx = dw = None
        if ctx.transposed:                                   # y = convT(x, w[Cin,Cout]):  dx = conv_s2(dy, w as [Cout'=Cin][Cin'=Cout])
            cin, cout = w.shape[:2]
            if ctx.needs_input_grad[0]:
                dx = _per_item(lambda gb: ops.conv3d_k3(gb, ctx.dgrad_pack, None, stride=2, relu=False), dy)
            if ctx.needs_input_grad[1]:
                dw = conv3d_wgrad(x, dy, 2)                  # [Cin,Cout,3,3,3]
        else:
            cout, cin = w.shape[:2]
            if ctx.needs_input_grad[0]:
                if ctx.stride == 1:                          # dx = conv(dy, flipped taps, channels swapped)
                    dx = _per_item(lambda gb: ops.conv3d_k3(gb, ctx.dgrad_pack, None, relu=False), dy)
                else:                                        # dx = convT(dy, w)
                    dx = _per_item(lambda gb: ops.deconv3d_k3s2(gb, ctx.dgrad_pack, None, relu=False), dy)
            if ctx.needs_input_grad[1]:
                dw = conv3d_wgrad(dy, x, ctx.stride)         # [Cout,Cin,3,3,3]
        return dx, dw, None, None


class BnRelu3d(torch.autograd.Function):
    """out = [skip +] relu?(batchnorm_train(y; gamma, beta)).  running_mean / running_var are updated in place."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, y, gamma, beta, skip, running_mean, running_var, momentum: float, eps: float, relu: bool):
        y = y.contiguous()
        B, C = y.shape[:2]
        V = y[0, 0].numel()
        n = B * V
        lib = _lib.load()
        sums = _scratch.zeros((C, 2), torch.float64, y.device)
        check(lib.cds_bn3d_stats_f32(_dev(y), sums.data_ptr(), B, C, V, ops._stream(y)), "cds_bn3d_stats_f32")
        # statistics pass, then ONE launch for the per-channel step (fp64: scale / shift, the saved mean / invstd, the
        # running-statistics update) + normalisation + ReLU + residual
        ss = torch.empty((2, C), dtype=torch.float32, device=y.device)
        mi = torch.empty((2, C), dtype=torch.float64, device=y.device)
        scale, shift, mean, invstd = ss[0], ss[1], mi[0], mi[1]
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        track = running_mean is not None
        out = torch.empty_like(y)
        skip_c = skip.contiguous() if skip is not None else None
        check(lib.cds_bn3d_norm_f32(_dev(y), sums.data_ptr(), _dev(g32), _dev(b32), float(n), float(eps), float(momentum),
                                    _dev(running_mean) if track else None, _dev(running_var) if track else None,
                                    _dev(skip_c) if skip_c is not None else None, out.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                    mean.data_ptr(), invstd.data_ptr(), B, C, V, 1 if relu else 0, ops._stream(y)),
              "cds_bn3d_norm_f32")
        ctx.save_for_backward(y, scale, shift, mean, invstd, gamma)
        ctx.relu, ctx.has_skip, ctx.n = relu, skip is not None, n
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dout):
        y, scale, shift, mean, invstd, gamma = ctx.saved_tensors
        dout = dout.contiguous().float()
        B, C = y.shape[:2]
        V = y[0, 0].numel()
        n = ctx.n
        lib = _lib.load()
        sums = _scratch.zeros((C, 2), torch.float64, y.device)
        check(lib.cds_bn3d_bwd_reduce_f32(_dev(dout), _dev(y), scale.data_ptr(), shift.data_ptr(), sums.data_ptr(), B, C, V,
                                          1 if ctx.relu else 0, ops._stream(y)), "cds_bn3d_bwd_reduce_f32")
        gb = torch.empty((2, C), dtype=torch.float32, device=y.device)
        dgamma, dbeta = gb[0], gb[1]
        dy = torch.empty_like(y)
        check(lib.cds_bn3d_bwd_norm_f32(_dev(dout), _dev(y), scale.data_ptr(), shift.data_ptr(), sums.data_ptr(), mean.data_ptr(),
                                        invstd.data_ptr(), float(n), dy.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), B, C, V,
                                        1 if ctx.relu else 0, ops._stream(y)), "cds_bn3d_bwd_norm_f32")
        return (dy, dgamma.to(gamma.dtype), dbeta.to(gamma.dtype), dout if ctx.has_skip else None,
                None, None, None, None, None)


def bn_momentum_and_count(bn) -> float:
    """The momentum of this training-mode call of a BatchNorm module, counting the call in num_batches_tracked (nn.BatchNorm semantics:
    momentum None = cumulative moving average)."""
    momentum = bn.momentum
    if bn.num_batches_tracked is not None:
        if momentum is None:
            bn.num_batches_tracked += 1
            momentum = 1.0 / float(bn.num_batches_tracked)
        else:
            _scratch.bump(bn.num_batches_tracked, 1)
    return 0.0 if momentum is None else float(momentum)


def conv_bn_relu3d(unit, x: Tensor, skip: Optional[Tensor] = None) -> Tensor:
    """One ConvBn3d holder (model.py) in its module mode: training -> batch statistics (and running-stat update),
    eval -> running statistics; Conv3d / ConvTranspose3d + BatchNorm3d + ReLU (+ skip), all on the HIP kernels."""
    y = Conv3dK3.apply(x, unit.conv.weight, unit.stride, unit.transposed)
    bn = unit.bn
    if bn.training:
        momentum = bn_momentum_and_count(bn)
        return BnRelu3d.apply(y, bn.weight, bn.bias, skip, bn.running_mean, bn.running_var, float(momentum), bn.eps, True)
    out = torch.relu(torch.nn.functional.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps))
    return out if skip is None else skip + out


def cost_regularization(cr, x: Tensor) -> Tensor:
    """models/module.py:305-315 on the HIP training ops.  x [B,C,D,h,w] -> [B,1,D,h,w]."""
    if x.shape[2] % 8 or x.shape[3] % 8 or x.shape[4] % 8:
        raise ValueError(f"CostRegNet needs D,h,w divisible by 8, got {tuple(x.shape[2:])}")
    c0 = conv_bn_relu3d(cr.conv0, x)
    c2 = conv_bn_relu3d(cr.conv2, conv_bn_relu3d(cr.conv1, c0))
    c4 = conv_bn_relu3d(cr.conv4, conv_bn_relu3d(cr.conv3, c2))
    y = conv_bn_relu3d(cr.conv6, conv_bn_relu3d(cr.conv5, c4))
    y = conv_bn_relu3d(cr.conv7, y, skip=c4)
    y = conv_bn_relu3d(cr.conv9, y, skip=c2)
    y = conv_bn_relu3d(cr.conv11, y, skip=c0)
    return Conv3dK3.apply(y, cr.prob.weight, 1, False)

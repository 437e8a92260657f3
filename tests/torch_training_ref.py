"""TEST INFRASTRUCTURE: the training-mode layers of the plane-sweep path restated with plain torch autograd ops.

These are the float64 / PyTorch-ROCm references the gradient tests hold the HIP training kernels to (tests/test_train2d_gpu.py,
tests/test_train_harness.py).  Nothing under ``cds_mvsnet_amd/`` imports this module: the product's training step runs on the HIP
kernels only.  ``torch_layers()`` swaps the layer functions of ``cds_mvsnet_amd.training`` for these restatements for the duration
of a ``with`` block, so a test can run the SAME step (``training.forward_train`` / ``model.train()(...)``) through torch ops.

Reference lines: models/dynamic_conv.py:81-122 (DynamicConv), models/module.py:28-71 (Conv2d + InstanceNorm + LeakyReLU),
:80-160,305-315 (CostRegNet), :318-370 (Refinement), :373-379 (depth regression), models/model.py:14 (visibility CNN).
"""
from __future__ import annotations

import contextlib
from typing import Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# Emulation of the product's bf16-storage policy (train2d_ops.activation_storage("bf16")) with torch ops: the tensors the policy stores
# as bfloat16 are rounded where they are produced, with a straight-through gradient (the kernels differentiate the widened values).
BF16_STORAGE = False


def _rb(x: Tensor) -> Tensor:
    return x + (x.float().bfloat16().to(x.dtype) - x).detach() if BF16_STORAGE else x



def att_weights_grouped(seq, curvs: Tensor, groups: int) -> Tensor:
    """``DynamicConv.att_weights`` (1x1 conv -> BatchNorm2d -> ReLU -> 1x1 conv, dynamic_conv.py:88-91) on a batch that
    stacks ``groups`` separate calls of the reference: the BatchNorm statistics are taken per group of N / groups samples
    (what each of those calls would have seen) and the running statistics receive the groups' updates in call order."""
    conv_a, bn, _, conv_b = seq[0], seq[1], seq[2], seq[3]
    h = conv_a(curvs)
    act_dtype = h.dtype
    if h.dtype in (torch.bfloat16, torch.float16):
        h = h.float()
    N, C, H, W = h.shape
    hg = h.view(groups, N // groups, C, H, W)
    mean = hg.mean(dim=(1, 3, 4))                                            # [G,C]
    var = hg.var(dim=(1, 3, 4), unbiased=False)
    y = (hg - mean.view(groups, 1, C, 1, 1)) * torch.rsqrt(var.view(groups, 1, C, 1, 1) + bn.eps)
    y = y * bn.weight.view(1, 1, C, 1, 1).to(y.dtype) + bn.bias.view(1, 1, C, 1, 1).to(y.dtype)
    if bn.training and bn.track_running_stats:
        with torch.no_grad():
            m = bn.momentum if bn.momentum is not None else 0.1
            n = (N // groups) * H * W
            rdt = bn.running_mean.dtype
            wts = m * (1.0 - m) ** torch.arange(groups - 1, -1, -1, device=h.device, dtype=rdt)   # call g, then g+1, ...
            bn.running_mean.mul_((1.0 - m) ** groups).add_((wts.view(-1, 1) * mean.detach().to(rdt)).sum(dim=0))
            bn.running_var.mul_((1.0 - m) ** groups).add_((wts.view(-1, 1) * (var.detach().to(rdt) * (n / max(n - 1, 1)))).sum(dim=0))
            bn.num_batches_tracked += groups
    return conv_b(F.relu(y.view(N, C, H, W)).to(act_dtype))


def dynamic_conv(dc, x: Tensor, epi: Tensor, T: float, groups: int = 1) -> Tuple[Tensor, Tensor]:
    """models/dynamic_conv.py:97-122 with torch ops.  x [N,Cin,H,W]; epi [N,2] on x's device.  groups > 1: the batch stacks
    that many separate calls of the reference (see ``att_weights_grouped``)."""
    N, _, H, W = x.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=x.device),
                            torch.arange(W, dtype=torch.float32, device=x.device), indexing="ij")
    u = xs.view(1, 1, H, W) - epi[:, 0].view(N, 1, 1, 1)
    v = ys.view(1, 1, H, W) - epi[:, 1].view(N, 1, 1, 1)
    nrm = torch.sqrt(u ** 2 + v ** 2)
    u, v = u / (nrm + 1e-6), v / (nrm + 1e-6)
    basis = torch.cat((u ** 2, 2 * u * v, v ** 2), dim=1)
    curvs, res = [], []
    for att, conv in zip(dc.att_convs, dc.convs):
        curvs.append((_rb(att(x)) * basis).sum(dim=1, keepdim=True))
        res.append(_rb(conv(x)).unsqueeze(1))
    curvs = torch.cat(curvs, dim=1)
    aw = dc.att_weights(curvs) if (groups == 1 or not dc.att_weights[1].training) else att_weights_grouped(dc.att_weights, curvs, groups)
    wts = F.softmax(aw / T, dim=1)
    return (torch.cat(res, dim=1) * wts.unsqueeze(2)).sum(dim=1), (curvs * wts).sum(dim=1, keepdim=True)


def in_act(y: Tensor, tanh: bool = False) -> Tensor:
    """InstanceNorm2d + LeakyReLU(0.1) (module.py:66-69) or + tanh (module.py:223)."""
    y = _rb(y)
    return torch.tanh(F.instance_norm(y)) if tanh else _rb(F.leaky_relu(F.instance_norm(y), 0.1))


def conv(conv_mod, x: Tensor) -> Tensor:
    return conv_mod(x)


def cbr2(unit, x: Tensor, groups: int = 1) -> Tensor:
    """ConvBn2d holder: Conv2d 3x3 -> BatchNorm2d (module mode) -> ReLU, one reference call."""
    if groups > 1:
        raise ValueError("the torch reference runs the visibility CNN one view at a time (training._stacked_operands() is False)")
    return F.relu(unit.bn(unit.conv(x)))


def visibility(seq, x: Tensor) -> Tensor:
    """models/model.py:14."""
    for i in range(3):
        x = cbr2(seq[i], x)
    return torch.sigmoid(seq[3](x))


def _cbr3(unit, x: Tensor) -> Tensor:
    return F.relu(unit.bn(unit.conv(x)))


def cost_regularization(cr, x: Tensor) -> Tensor:
    """models/module.py:305-315 (BatchNorm in the module's current mode).  x [B,C,D,h,w] -> [B,1,D,h,w]."""
    c0 = _cbr3(cr.conv0, x)
    c2 = _cbr3(cr.conv2, _cbr3(cr.conv1, c0))
    c4 = _cbr3(cr.conv4, _cbr3(cr.conv3, c2))
    y = _cbr3(cr.conv6, _cbr3(cr.conv5, c4))
    y = c4 + _cbr3(cr.conv7, y)
    y = c2 + _cbr3(cr.conv9, y)
    y = c0 + _cbr3(cr.conv11, y)
    return cr.prob(y)


def softargmin(prob_pre: Tensor, hyp: Tensor) -> Tensor:
    return torch.sum(F.softmax(prob_pre, dim=1) * hyp, dim=1)


def refinement(net, img: Tensor, depth0: Tensor, dmin: Tensor, dmax: Tensor) -> Tensor:
    """models/module.py:318-370."""
    B = dmin.shape[0]
    lo, hi = dmin.view(B, 1, 1, 1), dmax.view(B, 1, 1, 1)
    d = (depth0 - lo) / (hi - lo) * 10
    f_img = cbr2(net.conv0, img)
    f_d = F.relu(net.bn(net.deconv(cbr2(net.conv2, cbr2(net.conv1, d)))))
    res = net.res(cbr2(net.conv3, torch.cat((f_d, f_img), dim=1)))
    d = (F.interpolate(d, scale_factor=2, mode="bilinear", align_corners=True) + res) / 10
    return d * (hi - lo) + lo


@contextlib.contextmanager
def torch_layers(two_d: bool = True, three_d: bool = True):
    """Run ``cds_mvsnet_amd.training`` with its layer functions replaced by the torch restatements above (two_d: FeatureNet /
    DynamicConv, visibility CNN, soft-argmin, Refinement; three_d: CostRegNet).  Test-only: restores the HIP functions on exit."""
    from cds_mvsnet_amd import training
    saved = {}

    def swap(name, fn):
        saved[name] = getattr(training, name)
        setattr(training, name, fn)

    try:
        if two_d:
            swap("_dyn", dynamic_conv)
            swap("_in_act", in_act)
            swap("_conv", conv)
            swap("_curv", lambda a, b, c: ((a ** 2 + b ** 2 + c ** 2) / 3, c.abs()))      # module.py:250-251
            swap("_cbr2", cbr2)
            swap("_softargmin", softargmin)
            swap("_refinement", refinement)
            swap("_stacked_operands", lambda: False)
        if three_d:
            swap("cost_regularization", cost_regularization)
        yield
    finally:
        for name, fn in saved.items():
            setattr(training, name, fn)

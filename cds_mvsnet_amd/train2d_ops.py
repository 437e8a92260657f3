"""Training ops of the 2D stacks on the hand-written HIP kernels (csrc/train2d.hip + the forward kernels of conv2d.hip):
FeatureNet / DynamicConv, the visibility CNN, Refinement and the soft-argmin, each a ``torch.autograd.Function`` whose forward AND
backward are HIP kernels.  They replace the PyTorch-ROCm autograd ops (MIOpen convolutions, ATen elementwise
chains) ``training.py`` used for these stacks; parameters, ``state_dict`` and optimisers are unchanged (the functions take the
parameter tensors of the holder modules in ``model.py``).

Reference semantics (forward; the backward is what torch.autograd derives from it):
  models/dynamic_conv.py:97-122 (DynamicConv.forward), models/module.py:28-71 (conv -> InstanceNorm2d -> LeakyReLU(0.1)),
  models/module.py:234-267 (FeatureNet.forward), models/model.py:14,51 (visibility CNN), models/module.py:318-370 (Refinement),
  models/module.py:373-379 (depth_regression).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import _lib, _scratch, ops
from ._lib import ACT_ACCUM, ACT_LEAKY01, ACT_NONE, ACT_TANH, check

Tensor = torch.Tensor

# ---- activation storage policy ---------------------------------------------------------------------------------------------------
# "f32" (default: what the reference trains in, trainer/trainer.py:69-82) or "bf16" = bf16 STORAGE / fp32 ACCUMULATE for the FeatureNet
# activations (BASELINE config 5's label): every tensor of the 2D stack that lives from the forward to the backward pass - the layer
# inputs (InstanceNorm + LeakyReLU outputs), the DynamicConv branch responses [K,N,Cout+3,H,W] and the pre-normalisation maps - is a
# bfloat16 tensor, rounded once where it is produced; the backward kernels widen on load and accumulate in fp32 / fp64 as on the fp32
# path (csrc/train2d.hip: cds_instnorm_act_b16_f32, cds_f32_to_bf16, the *_b16 / *_xb16 / *_yb16 entries).  Between a producer and its
# consumer in the FORWARD pass the activation travels as a transient fp32 tensor (freed as soon as its consumers ran):
#   "bf16"          the transient is the unrounded value: the forward pass - loss, depth maps - is that of the fp32 step, the gradients
#                   differ by one bf16 rounding of each stored operand (the shipped policy; acceptance tests/test_train_bf16_gpu.py);
#   "bf16-forward"  the transient is the widened STORED value, so both passes see one and the same activation.  Kept as the stricter
#                   form: through nine DynamicConvs (InstanceNorm, LeakyReLU, softmax(. / T)) and a cascade whose hypothesis ranges
#                   switch discretely, 2^-9 of rounding per activation moves single gradient tensors of the randomly initialised G7
#                   step to cosines of 0.76-0.98 against fp32 (loss within 2e-3) - measured, and the reason it is not the default.
# Gradients, weights, statistics, the stage outputs (tanh features -> cost volume), the visibility CNN and Refinement (a few maps of
# 1-16 channels) stay fp32.
_STORAGE = {"dtype": torch.float32, "strict": False}


def set_activation_storage(kind) -> None:
    if kind in ("bf16", torch.bfloat16):
        _STORAGE["dtype"], _STORAGE["strict"] = torch.bfloat16, False
    elif kind == "bf16-forward":
        _STORAGE["dtype"], _STORAGE["strict"] = torch.bfloat16, True
    elif kind in ("f32", "fp32", torch.float32):
        _STORAGE["dtype"], _STORAGE["strict"] = torch.float32, False
    else:
        raise ValueError(f"activation storage {kind!r}: expected 'f32', 'bf16' or 'bf16-forward'")


set_activation_storage(__import__("os").environ.get("CDS_TRAIN_ACT_STORAGE", "f32").lower())


def activation_storage_dtype() -> torch.dtype:
    return _STORAGE["dtype"]


class activation_storage:
    """``with activation_storage("bf16"): loss = model(...); loss.backward()`` - the policy applies to the forward passes run inside."""

    def __init__(self, kind):
        self.kind = kind

    def __enter__(self):
        self.prev = dict(_STORAGE)
        set_activation_storage(self.kind)
        return self

    def __exit__(self, *exc):
        _STORAGE.update(self.prev)
        return False


def stored(x: Tensor) -> Optional[Tensor]:
    """The bf16-stored twin of an activation produced under the bf16 policy (None on the fp32 path)."""
    return getattr(x, "_cds_b16", None)


def with_store(x: Tensor, x16: Optional[Tensor]) -> Tensor:
    if x16 is not None:
        x._cds_b16 = x16
    return x


def cat_stored(parts: Sequence[Tensor]) -> Tensor:
    """torch.cat(parts, dim=1) that keeps the stored twins together (the FPN laterals concatenate two stored activations)."""
    out = torch.cat(tuple(parts), dim=1)
    tw = [stored(t) for t in parts]
    if any(t is not None for t in tw):       # a part without a twin (a tanh stage output, kept fp32 for the cost volume) is rounded here
        out._cds_b16 = torch.cat([t if t is not None else to_bf16(p.detach()) for t, p in zip(tw, parts)], dim=1)
    return out


def upsample2_stored(x: Tensor) -> Tensor:
    out = F.interpolate(x, scale_factor=2, mode="nearest")
    if stored(x) is not None:
        out._cds_b16 = F.interpolate(stored(x), scale_factor=2, mode="nearest")
    return out


def _p16(t: Tensor) -> int:
    if not (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
        raise ValueError("expected a contiguous bfloat16 device tensor")
    return t.data_ptr()


def to_bf16(x: Tensor) -> Tensor:
    """bfloat16(x) (round to nearest even) on the HIP conversion kernel."""
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(_lib.load().cds_f32_to_bf16(_p(x), out.data_ptr(), x.numel(), ops._stream(x)), "cds_f32_to_bf16")
    return out


_KS_DIRECT = (1, 3, 5, 7, 11)
SBF_MIN_PIXELS = 4 << 20      # DynamicConv forward: images x pixels from which the split-bf16 matrix-core branch kernel pays for its packing


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else ops._dev(t, "tensor")


def _p64(t: Tensor) -> int:
    if t.dtype != torch.float64 or not t.is_contiguous():
        raise ValueError("expected a contiguous float64 tensor")
    return t.data_ptr()


def pack_conv(w: Tensor) -> Tensor:
    """[Cout,Cin,k,k] -> [Cin,k*k,CoutP] (cout fastest, zero padded to a multiple of 8): the layout of cds_conv2d_f32."""
    cout, cin, k, _ = w.shape
    p = w.permute(1, 2, 3, 0).reshape(cin, k * k, cout)
    pad = (-cout) % 8
    if pad:
        p = F.pad(p, (0, pad))
    return p.contiguous()


def pack_conv2d(wa: Tensor, wb: Optional[Tensor], fwd: bool = True, dgrad: bool = False) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """pack_conv / pack_dgrad of cat(wa, wb) in ONE launch (cds_pack_conv2d_f32): (fwd [Cin,k*k,CoP] | None, dgrad [Co,k*k,CiP] | None)."""
    ca, cin, k, _ = wa.shape
    cb = wb.shape[0] if wb is not None else 0
    wa = wa.detach().float().contiguous()
    wb = wb.detach().float().contiguous() if wb is not None else None
    f = torch.empty((cin, k * k, (ca + cb + 7) // 8 * 8), dtype=torch.float32, device=wa.device) if fwd else None
    d = torch.empty((ca + cb, k * k, (cin + 7) // 8 * 8), dtype=torch.float32, device=wa.device) if dgrad else None
    check(_lib.load().cds_pack_conv2d_f32(_p(wa), _p(wb), f.data_ptr() if fwd else None, d.data_ptr() if dgrad else None, ca, cb, cin, k,
                                          ops._stream(wa)), "cds_pack_conv2d_f32")
    return f, d


def pack_dgrad(w: Tensor) -> Tensor:
    """Weights of the stride-1 data gradient as a forward convolution: dx = conv(dy, w'), w'[ci][co][ky][kx] = w[co][ci][k-1-ky][k-1-kx]."""
    return pack_conv(w.flip(2, 3).transpose(0, 1))


def conv2d_wgrad(g: Tensor, x: Tensor, k: int, stride: int, pad: int) -> Tensor:
    """dw[co][ci][ky][kx] = sum_{n, o} g[n][co][o] x[n][ci][stride o - pad + k]  ->  [Co,Cin,k,k]."""
    N, Co, Ho, Wo = g.shape
    Nx, Cin, H, W = x.shape
    if Nx != N:
        raise ValueError("conv2d_wgrad: batch mismatch")
    dw = _scratch.zeros((Co, Cin, k, k), torch.float32, g.device)
    _scratch.audit_note(dw)
    side = _scratch.side_stream(g.device)
    if x.dtype == torch.bfloat16:                                # the layer input in its stored form (bf16 policy)
        fn, px, what = _lib.load().cds_conv2d_wgrad_xb16_f32, _p16(x), "cds_conv2d_wgrad_xb16_f32"
    else:
        fn, px, what = _lib.load().cds_conv2d_wgrad_f32, _p(x), "cds_conv2d_wgrad_f32"
    if side is None:
        check(fn(_p(g), px, dw.data_ptr(), N, Co, Cin, Ho, Wo, H, W, k, stride, pad, ops._stream(g)), what)
        return dw
    with torch.cuda.stream(side):                                # a leaf of the backward pass: overlaps with the data-gradient chain
        check(fn(_p(g), px, dw.data_ptr(), N, Co, Cin, Ho, Wo, H, W, k, stride, pad, side.cuda_stream), what)
    g.record_stream(side)
    x.record_stream(side)
    return dw


def conv2d_dgrad_s2(g: Tensor, w: Tensor, H: int, W: int) -> Tensor:
    """Data gradient of Conv2d(k 3, stride 2, pad 1): g [N,Co,Ho,Wo], w [Co,Cin,3,3] -> [N,Cin,H,W]."""
    N, Co, Ho, Wo = g.shape
    Cin = w.shape[1]
    gx = torch.empty((N, Cin, H, W), dtype=torch.float32, device=g.device)
    check(_lib.load().cds_conv2d_dgrad_s2_f32(_p(g), _p(w.contiguous()), gx.data_ptr(), N, Co, Cin, Ho, Wo, H, W, ops._stream(g)),
          "cds_conv2d_dgrad_s2_f32")
    return gx


def _c16(w: Tensor, stride: int, pad: int, width: int) -> bool:
    return tuple(w.shape) == (16, 16, 3, 3) and stride == 1 and pad == 1 and width % 4 == 0 and ops.USE_CONV2D_MFMA


_C16_DGRAD_INDEX: dict = {}


def _c16_dgrad_index(device) -> Tensor:
    """Gather index of the data-gradient operand of the 16 -> 16 3x3 matrix-core kernel: [tap][ci][co] = w[co][ci][2-ky][2-kx] in ONE
    launch (flip + permute + contiguous were two)."""
    key = str(device)
    if key not in _C16_DGRAD_INDEX:
        t, ci, co = torch.meshgrid(torch.arange(9), torch.arange(16), torch.arange(16), indexing="ij")
        _C16_DGRAD_INDEX[key] = (co * 144 + ci * 9 + (8 - t)).reshape(-1).to(device)
    return _C16_DGRAD_INDEX[key]


class Conv2d(torch.autograd.Function):
    """y = conv2d(x, weight [Cout,Cin,k,k], bias | None; stride 1 with k in (1,3,5,7,11), or k 3 stride 2 pad 1)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias, stride: int, pad: int, x16=None):
        x = x.contiguous()
        k = weight.shape[-1]
        if k not in _KS_DIRECT or (stride == 2 and (k != 3 or pad != 1)) or stride not in (1, 2):
            raise ValueError(f"train2d_ops.Conv2d: unsupported kernel {k} / stride {stride} / pad {pad}")
        ctx.save_for_backward(x16 if x16 is not None else x, weight)    # bf16 policy: the stored twin is what lives until the backward
        ctx.stride, ctx.pad, ctx.has_bias = stride, pad, bias is not None
        if _c16(weight, stride, pad, x.shape[-1]):               # 16 -> 16, 3x3: the fp32 matrix-core kernel of the visibility CNN
            return ops.conv2d_k3_c16(x, weight.detach().permute(2, 3, 0, 1).reshape(9, 16, 16).contiguous(),
                                     bias.detach().contiguous() if bias is not None else None, ACT_NONE)
        return ops.conv2d(x, pack_conv2d(weight, None)[0], bias.detach().contiguous() if bias is not None else None, weight.shape[0], k,
                          stride, pad)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous().float()
        w = weight.detach().float()
        k = w.shape[-1]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if ctx.stride == 1:
                if 2 * ctx.pad != k - 1:
                    raise ValueError("train2d_ops.Conv2d: the stride-1 data gradient needs 'same' padding")
                if _c16(w, 1, ctx.pad, dy.shape[-1]):
                    dx = ops.conv2d_k3_c16(dy, w.reshape(-1).index_select(0, _c16_dgrad_index(w.device)).view(9, 16, 16), None, ACT_NONE)
                else:
                    dx = ops.conv2d(dy, pack_conv2d(w, None, fwd=False, dgrad=True)[1], None, w.shape[1], k, 1, ctx.pad)
            else:
                dx = conv2d_dgrad_s2(dy, w, x.shape[2], x.shape[3])
        if ctx.needs_input_grad[1]:
            dw = conv2d_wgrad(dy, x, k, ctx.stride, ctx.pad)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=(0, 2, 3))
        return dx, dw, db, None, None, None


class InstNormAct(torch.autograd.Function):
    """z = act(InstanceNorm2d(y)), act in (ACT_LEAKY01, ACT_TANH, ACT_NONE)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, y, act: int):
        y = y.contiguous()
        N, C, H, W = y.shape
        out = torch.empty_like(y)
        stats = torch.empty((N, C, 2), dtype=torch.float64, device=y.device)
        check(_lib.load().cds_instnorm_act_f32(_p(y), out.data_ptr(), stats.data_ptr(), N, C, H, W, act, 0, ops._stream(y)),
              "cds_instnorm_act_f32")
        ctx.save_for_backward(y, stats)
        ctx.act = act
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gz):
        y, stats = ctx.saved_tensors
        gz = gz.contiguous().float()
        N, C, H, W = y.shape
        gy = torch.empty_like(y)
        sums = _scratch.zeros((N, C, 2), torch.float64, y.device)
        check(_lib.load().cds_instnorm_bwd_f32(_p(gz), _p(y), _p64(stats), _p64(sums), gy.data_ptr(), N, C, H, W, ctx.act, 1,
                                               ops._stream(y)), "cds_instnorm_bwd_f32")
        return gy, None


class InstNormActB16(torch.autograd.Function):
    """InstNormAct under the bf16 storage policy: y (the transient fp32 convolution output) -> (z32, z16).  y16 = bfloat16(y) is what is
    kept for the backward; z16 = bfloat16(z) is the stored activation and z32 the transient for the next forward kernel (z itself, or
    under strict the widened z16 with the statistics taken from y16).  act = tanh (the stage outputs): z16 is empty, z32 unrounded."""

    @staticmethod
    def forward(ctx, y, act: int, strict: bool):
        y = y.contiguous()
        N, C, H, W = y.shape
        y16 = torch.empty(y.shape, dtype=torch.bfloat16, device=y.device)
        z32 = torch.empty_like(y)
        keep = act != ACT_TANH
        z16 = torch.empty(y.shape if keep else (0,), dtype=torch.bfloat16, device=y.device)
        stats = torch.empty((N, C, 2), dtype=torch.float64, device=y.device)
        check(_lib.load().cds_instnorm_act_b16_f32(_p(y), y16.data_ptr(), z32.data_ptr(), z16.data_ptr() if keep else None,
                                                   stats.data_ptr(), N, C, H, W, act, 1 if strict else 0, ops._stream(y)),
              "cds_instnorm_act_b16_f32")
        ctx.save_for_backward(y16, stats, z16)                    # z16: the tensor the consumer keeps anyway; here for its sign bits
        ctx.act = act
        ctx.mark_non_differentiable(z16)
        return z32, z16

    @staticmethod
    def backward(ctx, gz, _g16):
        y16, stats, z16 = ctx.saved_tensors
        gz = gz.contiguous().float()
        N, C, H, W = y16.shape
        gy = torch.empty(y16.shape, dtype=torch.float32, device=y16.device)
        sums = _scratch.zeros((N, C, 2), torch.float64, y16.device)
        check(_lib.load().cds_instnorm_bwd_yb16_f32(_p(gz), _p16(y16), _p16(z16) if z16.numel() else None, _p64(stats), _p64(sums), gy.data_ptr(), N, C, H, W, ctx.act, 1,
                                                    ops._stream(gz)), "cds_instnorm_bwd_yb16_f32")
        return gy, None, None


def instnorm_act(y: Tensor, act: int) -> Tensor:
    """act(InstanceNorm2d(y)) in the current storage policy; under "bf16" the result carries its stored twin (``stored(out)``)."""
    if _STORAGE["dtype"] is torch.bfloat16:
        z32, z16 = InstNormActB16.apply(y, act, _STORAGE["strict"])
        return with_store(z32, z16 if z16.numel() else None)
    return InstNormAct.apply(y, act)


class _DynConvFn(torch.autograd.Function):
    """One DynamicConv (dynamic_conv.py:97-122): (x, epipoles) -> (y [N,Cout,H,W], norm_curv [N,1,H,W]).
    Tensor arguments after the fixed ones: K convolution weights, K attention-convolution weights, K biases (or none), then
    att_weights[0].weight, BatchNorm weight, bias, att_weights[3].weight."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, epi, T: float, groups: int, bn, ksizes: Tuple[int, ...], has_bias: bool, x16, *params):
        K = len(ksizes)
        x = x.contiguous()
        N, Cin, H, W = x.shape
        convs, atts = params[:K], params[K:2 * K]
        biases = params[2 * K:3 * K] if has_bias else ()
        w1, gamma, beta, w2 = params[-4:]
        cout = convs[0].shape[0]
        dev = x.device
        epi = epi.detach().float().contiguous()
        lib = _lib.load()
        st = ops._stream(x)
        bcat = None
        if has_bias:                                              # [K, Cout + 3]: the attention convolutions have no bias
            bcat = _scratch.zeros((K, cout + 3), torch.float32, dev)
            bcat[:, :cout] = torch.stack([b.detach() for b in biases])
        branches = torch.empty((K, N, cout + 3, H, W), dtype=torch.float32, device=dev)
        need_dx = ctx.needs_input_grad[0]
        packs = [pack_conv2d(convs[i], atts[i], fwd=True, dgrad=need_dx) for i in range(K)]
        if N * H * W >= SBF_MIN_PIXELS and ops.dynconv_sbf_supported(Cin, cout + 3, ksizes, W):
            # all kernel sizes from one staged tile on the matrix cores (split-bf16 arithmetic, fp32-level accuracy); its weight layout
            # costs a dozen small launches, so small batches stay on the direct kernels
            wcat = [torch.cat((convs[i].detach(), atts[i].detach()), dim=0) for i in range(K)]        # [Cout+3,Cin,k,k]
            ops.dynconv_branches_sbf(x, ops.split_pack_dynconv(wcat), bcat, cout + 3, ksizes, out=branches)
        else:
            for i, k in enumerate(ksizes):
                ops.conv2d(x, packs[i][0], bcat[i] if has_bias else None, cout + 3, k, 1, (k - 1) // 2, out=branches[i])
        use_batch = bool(bn.training or not bn.track_running_stats)
        G = groups if use_batch else 1
        if N % G:
            raise ValueError(f"DynamicConv: {N} images do not split into {G} groups")
        w1m = w1.detach().reshape(4, K).contiguous()
        w2m = w2.detach().reshape(K, 4).contiguous()
        mean = torch.empty((G, 4), dtype=torch.float32, device=dev)
        rstd = torch.empty((G, 4), dtype=torch.float32, device=dev)
        mom = _scratch.zeros((G, K + K * (K + 1) // 2), torch.float64, dev)
        track = use_batch and bn.training and bn.track_running_stats
        momentum = bn.momentum if bn.momentum is not None else 0.1
        keep16 = _STORAGE["dtype"] is torch.bfloat16             # the branch responses are STORED as bf16 ...
        b16 = keep16 and _STORAGE["strict"]                       # ... and (strict) the epilogue reads the stored form
        if b16:
            br32, branches = branches, to_bf16(branches)
            del br32
        f_stats = lib.cds_dynconv_bn_stats_b16_f32 if b16 else lib.cds_dynconv_bn_stats_f32
        f_blend = lib.cds_dynconv_blend_train_b16_f32 if b16 else lib.cds_dynconv_blend_train_f32
        pbr = _p16(branches) if b16 else _p(branches)
        check(f_stats(pbr, _p(epi), _p(w1m), mom.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                           bn.running_mean.data_ptr() if (track or not use_batch) else None,
                                           bn.running_var.data_ptr() if (track or not use_batch) else None,
                                           N, G, K, cout, H, W, float(bn.eps), float(momentum), 1 if use_batch else 0, 1, st),
              "cds_dynconv_bn_stats_f32")
        if track:
            _scratch.bump(bn.num_batches_tracked, G)
        y = torch.empty((N, cout, H, W), dtype=torch.float32, device=dev)
        nc = torch.empty((N, 1, H, W), dtype=torch.float32, device=dev)
        gm, bt = gamma.detach().contiguous(), beta.detach().contiguous()
        check(f_blend(pbr, _p(epi), _p(w1m), _p(w2m), _p(gm), _p(bt), _p(mean), _p(rstd), float(T),
                      y.data_ptr(), nc.data_ptr(), N, G, K, cout, H, W, st), "cds_dynconv_blend_train_f32")
        if keep16 and not b16:
            branches = to_bf16(branches)
        ctx.save_for_backward(x16 if x16 is not None else x, epi, branches, mean, rstd, *params)
        ctx.dgrad_packs = [pk[1] for pk in packs]
        ctx.cfg = (float(T), G, tuple(ksizes), has_bias, use_batch)
        return y, nc

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy, gnc):
        T, G, ksizes, has_bias, use_batch = ctx.cfg
        K = len(ksizes)
        x, epi, branches, mean, rstd = ctx.saved_tensors[:5]
        params = ctx.saved_tensors[5:]
        convs, atts = params[:K], params[K:2 * K]
        w1, gamma, beta, w2 = params[-4:]
        N, Cin, H, W = x.shape
        cout = convs[0].shape[0]
        dev = x.device
        lib = _lib.load()
        st = ops._stream(x)
        gy = gy.contiguous().float() if gy is not None else torch.zeros((N, cout, H, W), dtype=torch.float32, device=dev)
        gnc = gnc.contiguous().float() if gnc is not None else None
        w1m = w1.detach().reshape(4, K).contiguous()
        w2m = w2.detach().reshape(K, 4).contiguous()
        gbr = torch.empty(branches.shape, dtype=torch.float32, device=dev)
        sums = _scratch.zeros((G * 8 + K * 4,), torch.float64, dev)
        dw1 = _scratch.zeros((4, K), torch.float64, dev)
        b16 = branches.dtype == torch.bfloat16
        f_bwd = lib.cds_dynconv_blend_bwd_b16_f32 if b16 else lib.cds_dynconv_blend_bwd_f32
        check(f_bwd(_p16(branches) if b16 else _p(branches), _p(epi), _p(w1m), _p(w2m), _p(gamma.detach().contiguous()),
                    _p(beta.detach().contiguous()), _p(mean), _p(rstd), T, _p(gy), _p(gnc), gbr.data_ptr(),
                    sums.data_ptr(), dw1.data_ptr(), N, G, K, cout, H, W, 1 if use_batch else 0, 1, st),
              "cds_dynconv_blend_bwd_f32")
        small = torch.empty((8 + 8 * K,), dtype=torch.float32, device=dev)
        check(lib.cds_dynconv_bwd_finish_f32(_p64(sums), _p64(dw1), G, K, small.data_ptr(), st), "cds_dynconv_bwd_finish_f32")
        g_beta, g_gamma = small[:4], small[4:8]
        g_w2 = small[8:8 + 4 * K].view_as(w2)
        g_w1 = small[8 + 4 * K:].view_as(w1)
        dx = None
        g_convs: List[Optional[Tensor]] = []
        g_atts: List[Optional[Tensor]] = []
        g_bias: List[Optional[Tensor]] = []
        for i, k in enumerate(ksizes):
            if ctx.needs_input_grad[0]:
                if dx is None:
                    dx = torch.empty(x.shape, dtype=torch.float32, device=dev)
                ops.conv2d(gbr[i], ctx.dgrad_packs[i], None, Cin, k, 1, (k - 1) // 2, act=ACT_NONE if i == 0 else ACT_ACCUM, out=dx)
            dw = conv2d_wgrad(gbr[i], x, k, 1, (k - 1) // 2)
            g_convs.append(dw[:cout])
            g_atts.append(dw[cout:])
            if has_bias:
                g_bias.append(gbr[i][:, :cout].sum(dim=(0, 2, 3)))
        return (dx, None, None, None, None, None, None, None, *g_convs, *g_atts, *g_bias, g_w1, g_gamma, g_beta, g_w2)


def dynamic_conv(dc, x: Tensor, epi: Tensor, T: float, groups: int = 1) -> Tuple[Tensor, Tensor]:
    """``DynamicConv.forward`` of the holder module `dc` on the HIP kernels.  x [N,Cin,H,W], epi [N,2] on x's device.
    groups > 1: the batch stacks that many separate calls of the reference (BatchNorm statistics per group of N / groups images)."""
    K = len(dc.size_kernels)
    has_bias = dc.convs[0].bias is not None
    params = [c.weight for c in dc.convs] + [a.weight for a in dc.att_convs]
    if has_bias:
        params += [c.bias for c in dc.convs]
    params += [dc.att_weights[0].weight, dc.att_weights[1].weight, dc.att_weights[1].bias, dc.att_weights[3].weight]
    return _DynConvFn.apply(x, epi, float(T), int(groups), dc.att_weights[1], tuple(dc.size_kernels), has_bias, stored(x), *params)


def conv_in_act(conv, x: Tensor, act: int = ACT_LEAKY01) -> Tensor:
    """Plain ConvUnit (module.py:28-71): Conv2d (no bias) -> InstanceNorm2d -> LeakyReLU(0.1)."""
    y = Conv2d.apply(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], stored(x))
    return instnorm_act(y, act)


class CurvatureStats(torch.autograd.Function):
    """((a^2 + b^2) + c^2) / 3 and |c| of the three norm-curvature maps of a FeatureNet level (module.py:250-251,257-258,264-265):
    one launch forward, one backward (14 ATen launches as `(a ** 2 + b ** 2 + c ** 2) / 3, c.abs()`)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, a, b, c):
        a, b, c = a.contiguous(), b.contiguous(), c.contiguous()
        ctx.save_for_backward(a, b, c)
        return ops.curvature_stats(a, b, c)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_sum, g_abs):
        a, b, c = ctx.saved_tensors
        ga, gb, gc = torch.empty_like(a), torch.empty_like(b), torch.empty_like(c)
        gs = g_sum.contiguous().float() if g_sum is not None else None
        gm = g_abs.contiguous().float() if g_abs is not None else None
        check(_lib.load().cds_curvature_stats_bwd_f32(_p(a), _p(b), _p(c), _p(gs), _p(gm), ga.data_ptr(), gb.data_ptr(), gc.data_ptr(),
                                                      a.numel(), ops._stream(a)), "cds_curvature_stats_bwd_f32")
        return ga, gb, gc


class SoftArgmin(torch.autograd.Function):
    """depth [B,h,w] = sum_d softmax(prob_pre, dim=1) * hyp  (module.py:373-379); prob_pre, hyp [B,D,h,w]."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, prob_pre, hyp):
        prob_pre, hyp = prob_pre.contiguous(), hyp.contiguous()
        ctx.save_for_backward(prob_pre, hyp)
        return torch.stack([ops.softargmin_conf(prob_pre[b], hyp[b])[0] for b in range(prob_pre.shape[0])])

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gd):
        prob_pre, hyp = ctx.saved_tensors
        gd = gd.contiguous().float()
        B, D, h, w = prob_pre.shape
        gp = torch.empty_like(prob_pre)
        lib = _lib.load()
        for b in range(B):
            check(lib.cds_softargmin_bwd_f32(_p(prob_pre[b]), _p(hyp[b]), _p(gd[b]), gp[b].data_ptr(), D, h, w, 1, ops._stream(gd)),
                  "cds_softargmin_bwd_f32")
        return gp, None


_GROUP_WEIGHTS = {}


def _group_weights(m: float, groups: int, device) -> Tensor:
    """m (1 - m)^(G-1-g), g = 0..G-1, on the device (cached: built once, no host-to-device copy per step)."""
    key = (m, groups, str(device))
    if key not in _GROUP_WEIGHTS:
        _GROUP_WEIGHTS[key] = torch.tensor([m * (1.0 - m) ** (groups - 1 - g) for g in range(groups)], dtype=torch.float32).to(device)
    return _GROUP_WEIGHTS[key]


def bn_relu2d(bn, y: Tensor, relu: bool = True, groups: int = 1) -> Tensor:
    """BatchNorm2d in the module's mode (+ ReLU) on the BatchNorm kernels of train3d.hip (a [B,C,H,W] map is a one-slice volume).
    groups > 1: y stacks that many separate calls of the module along the batch axis (group-major); the batch statistics are taken
    per group and the running statistics receive the groups' updates in call order - the same values as `groups` calls."""
    from . import train_ops
    if bn.training and groups > 1:
        GB, C, H, W = y.shape
        B = GB // groups
        if B * groups != GB:
            raise ValueError(f"bn_relu2d: {GB} maps do not split into {groups} groups")
        # group g, channel c -> channel g C + c of ONE BatchNorm call over B samples
        yv = y.reshape(1, GB * C, 1, H, W) if B == 1 else y.view(groups, B, C, H, W).transpose(0, 1).reshape(B, groups * C, 1, H, W)
        track = bn.track_running_stats and bn.running_mean is not None
        tm = _scratch.zeros((groups * C,), torch.float32, y.device) if track else None
        tv = _scratch.zeros((groups * C,), torch.float32, y.device) if track else None
        out = train_ops.BnRelu3d.apply(yv, bn.weight.repeat(groups), bn.bias.repeat(groups), None, tm, tv, 1.0, bn.eps, relu)
        if track:                                                        # momentum 1 left the batch mean / unbiased variance in tm / tv
            with torch.no_grad():
                tm, tv = tm.view(groups, C), tv.view(groups, C)
                if bn.momentum is not None:                          # one fused update: r (1-m)^G + sum_g m (1-m)^(G-1-g) stat_g
                    m = float(bn.momentum)
                    wts = _group_weights(m, groups, y.device)
                    keep = (1.0 - m) ** groups
                    if bn.running_mean.dtype == torch.float32 and bn.running_mean.is_contiguous() and bn.running_var.is_contiguous():
                        check(_lib.load().cds_bn_running_update_f32(tm.data_ptr(), tv.data_ptr(), wts.data_ptr(), float(keep), groups, C,
                                                                    bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                                                    ops._stream(y)), "cds_bn_running_update_f32")   # both statistics: one launch
                    else:
                        torch.addmv(bn.running_mean, tm.t(), wts, beta=keep, out=bn.running_mean)
                        torch.addmv(bn.running_var, tv.t(), wts, beta=keep, out=bn.running_var)
                    _scratch.bump(bn.num_batches_tracked, groups)
                else:
                    for g in range(groups):
                        bn.num_batches_tracked += 1
                        m = 1.0 / float(bn.num_batches_tracked)
                        bn.running_mean.mul_(1.0 - m).add_(tm[g], alpha=m)
                        bn.running_var.mul_(1.0 - m).add_(tv[g], alpha=m)
        out = out.view(GB, C, H, W) if B == 1 else out.view(B, groups, C, H, W).transpose(0, 1).reshape(GB, C, H, W)
        return out
    if bn.training:
        momentum = train_ops.bn_momentum_and_count(bn)
        return train_ops.BnRelu3d.apply(y.unsqueeze(2), bn.weight, bn.bias, None, bn.running_mean, bn.running_var, float(momentum),
                                        bn.eps, relu).squeeze(2)
    out = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    return torch.relu(out) if relu else out


class Deconv2dK3S2(torch.autograd.Function):
    """ConvTranspose2d(k 3, stride 2, pad 1, output_padding 1, no bias) with 8 output channels: x [B,Cin,H,W], weight [Cin,8,3,3]."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight):
        x = x.contiguous()
        cin, cout = weight.shape[:2]
        ctx.save_for_backward(x, weight)
        wpk = weight.detach().permute(0, 2, 3, 1).reshape(cin, 9, cout).contiguous()
        return torch.stack([ops.deconv2d_k3s2(x[b], wpk, None) for b in range(x.shape[0])])

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous().float()
        w = weight.detach().float()
        dx = dw = None
        if ctx.needs_input_grad[0]:                              # the stride-2 convolution this layer is the transpose of
            dx = ops.conv2d(dy, pack_conv(w), None, w.shape[0], 3, 2, 1)
        if ctx.needs_input_grad[1]:
            dw = conv2d_wgrad(x, dy, 3, 2, 1)                    # roles swapped: [Cin,Cout,3,3]
        return dx, dw


def _cbr(unit, x: Tensor) -> Tensor:
    return bn_relu2d(unit.bn, Conv2d.apply(x, unit.conv.weight, unit.conv.bias, 1, 1))


def refinement(net, img: Tensor, depth0: Tensor, dmin: Tensor, dmax: Tensor) -> Tensor:
    """Refinement.forward in training mode (module.py:351-370) on the HIP training ops.  img [B,3,H,W], depth0 [B,1,H/2,W/2]."""
    B = dmin.shape[0]
    lo, hi = dmin.view(B, 1, 1, 1).float(), dmax.view(B, 1, 1, 1).float()
    d = ((depth0.float() - lo) / (hi - lo) * 10).contiguous()
    f_img = _cbr(net.conv0, img.float())
    f_d = bn_relu2d(net.bn, Deconv2dK3S2.apply(_cbr(net.conv2, _cbr(net.conv1, d)), net.deconv.weight))
    res = Conv2d.apply(_cbr(net.conv3, torch.cat((f_d, f_img), dim=1)), net.res.weight, net.res.bias, 1, 1)
    d = (F.interpolate(d, scale_factor=2, mode="bilinear", align_corners=True) + res) / 10
    return d * (hi - lo) + lo

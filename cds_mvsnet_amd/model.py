"""Drop-in ``CDSMVSNet`` for MI355X: same constructor, ``forward`` signature, output dict and
state-dict keys as the reference's ``models.model.CDSMVSNet`` (models/model.py:97-223), with the
plane-sweep hot path executed by hand-written gfx950 kernels (libcdsmvs_hip.so).

The ``nn.Module`` tree below only *holds parameters* under the reference's names so that reference
checkpoints load unchanged (387 entries, e.g. ``feature.conv00.conv.att_convs.0.weight``,
``stage_net.vis.0.0.conv.weight``, ``cost_regularization.2.conv7.conv.weight``).  None of the holder
``nn.Conv*`` modules is ever called on the hot path: weights are BN-folded and packed once
(:class:`_Packed`) and handed to the C ABI through :mod:`cds_mvsnet_amd.ops`.

Scope: inference (``model.eval()``) runs entirely on the HIP kernels.  ``model.train()`` dispatches to
``training.forward_train``: every stack has hand-written HIP forward AND backward kernels behind
``torch.autograd.Function``s (train_ops.py, train2d_ops.py); CUDA (ROCm) tensors only, CPU tensors raise.
"""
from __future__ import annotations

import os
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import geometry, ops
from .ops import ACT_LEAKY01, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH

Tensor = torch.Tensor
BN_EPS = 1e-5
# CDS_CONV_EXACT=1: CostRegNet on the exact-fp32 kernels (sequential fmaf chains, planar volumes) instead of the
# split-bf16 matrix-core kernels (fp32-class error, channels-last volumes)
USE_SPLIT_BF16 = os.environ.get("CDS_CONV_EXACT", "0") != "1"
USE_FUSED_BLEND = os.environ.get("CDS_FUSED_BLEND", "1") != "0"   # A/B knob: 0 = DynamicConv branches and blend as two kernels
# stage 1 on a side stream next to FeatureNet's finer levels (CDS_OVERLAP_STAGE1=0: one stream): 1600x1184 19.04 -> 18.72 ms, 640x512 4.67 -> 4.48
OVERLAP_STAGE1 = os.environ.get("CDS_OVERLAP_STAGE1", "1") != "0"
OVERLAP_STAGE2 = os.environ.get("CDS_OVERLAP_STAGE2", "0") == "1"   # A/B knob: stage 2 as well (next to the full-resolution FPN level)
# FeatureNet on channels-last activations (csrc/feat_cl.hip, round 5); CDS_FEAT_CL=0: the planar kernels of rounds 1-4
USE_FEAT_CL = os.environ.get("CDS_FEAT_CL", "1") != "0"
USE_CONV00_MFMA = os.environ.get("CDS_CONV00_MFMA", "1") != "0"   # A/B knob: 0 = conv00's branches on the VALU kernels + blend kernel
# the stage-1 side stream inside a stream capture (fork / join become graph edges); CDS_OVERLAP_IN_CAPTURE=0: captured forwards are one chain
OVERLAP_IN_CAPTURE = os.environ.get("CDS_OVERLAP_IN_CAPTURE", "1") != "0"
_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}
USE_GRAPHS_DEFAULT = os.environ.get("CDS_GRAPH", "0") == "1"       # eval forwards of every model through graphed.CapturedForward
_GRAPH_RUNNERS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()   # model -> its CapturedForward (CDSMVSNet.use_graphs)


_UNIT_BOUNDS: Dict[int, Tensor] = {}


def _unit_bound(device) -> Tensor:
    """A device scalar 1.0 (the bound of |tanh features| products), one per device."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _UNIT_BOUNDS:
        _UNIT_BOUNDS[idx] = torch.ones((1,), dtype=torch.float32, device=device)
    return _UNIT_BOUNDS[idx]


def _side_stream(device) -> "torch.cuda.Stream":
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _SIDE_STREAMS:
        _SIDE_STREAMS[idx] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[idx]   # A/B knob: the prob layer as a z-marching matrix-core kernel


# ------------------------------------------------------------------------------------------------
# weight folding / packing
# ------------------------------------------------------------------------------------------------
def _bn_fold(bn: nn.Module) -> Tuple[Tensor, Tensor]:
    """Eval-mode BatchNorm as y = x*scale + shift."""
    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias.detach() - bn.running_mean * scale
    return scale, shift


def _pack2d(w: Tensor) -> Tensor:
    """[Cout,Cin,k,k] -> [Cin,k*k,CoutP] (cout fastest, zero padded to a multiple of 8)."""
    cout, cin, k, _ = w.shape
    p = w.permute(1, 2, 3, 0).reshape(cin, k * k, cout)
    pad = (-cout) % 8
    if pad:
        p = F.pad(p, (0, pad))
    return p.contiguous()


class _Packed:
    """Lazily built cache of the folded / packed weights of one holder module, one entry per device.

    An entry is rebuilt when the holder's parameters / buffers were replaced or modified through autograd-visible
    in-place ops (``data_ptr`` / ``_version`` signature), and the holders drop the whole cache from ``_apply`` (``.to()``,
    ``.cuda()``, ``.float()``), ``train()`` / ``eval()`` and ``load_state_dict``.  Edits through ``.data``
    (``p.data.copy_()``, EMA swaps) keep both the pointer and the version: call :meth:`CDSMVSNet.repack` after them.
    ``nn.DataParallel`` replicas (``_is_replica``: shallow copies that share this object and receive freshly broadcast
    weight tensors on every forward, base/base_trainer.py:17-18, test.py:185-186) are never cached: they pack from
    their own tensors each call."""

    def __init__(self):
        self._entries: Dict[object, Tuple[tuple, Dict[str, Tensor]]] = {}
        self.generation = 0       # bumped whenever packed tensors are dropped or rebuilt: a hipGraph holds their ADDRESSES (graphed.py)

    def invalidate(self) -> None:
        self._entries = {}
        self.generation += 1

    @staticmethod
    def _tensors(owner: nn.Module) -> List[Tensor]:
        return list(owner.parameters()) + list(owner.buffers())

    def get(self, owner: nn.Module, builder) -> Dict[str, Tensor]:
        if getattr(owner, "_is_replica", False):
            with torch.no_grad():
                return builder()
        ts = self._tensors(owner)
        key = ts[0].device if ts else None
        sig = tuple((t.data_ptr(), t._version) for t in ts)
        hit = self._entries.get(key)
        if hit is None or hit[0] != sig:
            with torch.no_grad():
                hit = (sig, builder())
            self._entries[key] = hit
            self.generation += 1
        return hit[1]


class _PackedHolder(nn.Module):
    """Mixin of the modules that own a :class:`_Packed` cache: every way PyTorch itself rewrites parameters drops it."""

    def _init_packed(self) -> None:
        self._packed = _Packed()
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._packed.invalidate())

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._packed.invalidate()
        return out

    def train(self, mode: bool = True):
        self._packed.invalidate()
        return super().train(mode)

    def repack(self) -> None:
        """Forget the packed weights (needed after edits that bypass autograd's version counter, e.g. ``p.data``)."""
        self._packed.invalidate()


# ------------------------------------------------------------------------------------------------
# parameter holders (names fix the state-dict keys)
# ------------------------------------------------------------------------------------------------
class DynamicConv(nn.Module):
    """Parameters of one dynamic-scale convolution (models/dynamic_conv.py:81-95)."""

    def __init__(self, in_c: int, out_c: int, size_kernels: Sequence[int], bias: bool = True, hidden_dim: int = 4):
        super().__init__()
        self.size_kernels = tuple(size_kernels)
        self.in_c, self.out_c = in_c, out_c
        self.att_convs = nn.ModuleList([nn.Conv2d(in_c, 3, k, padding=(k - 1) // 2, bias=False) for k in size_kernels])
        self.convs = nn.ModuleList([nn.Conv2d(in_c, out_c, k, padding=(k - 1) // 2, bias=bias) for k in size_kernels])
        nk = len(size_kernels)
        self.att_weights = nn.Sequential(nn.Conv2d(nk, hidden_dim, 1, bias=False), nn.BatchNorm2d(hidden_dim),
                                         nn.ReLU(inplace=True), nn.Conv2d(hidden_dim, nk, 1, bias=False))
        for p in self.att_convs.parameters():
            nn.init.normal_(p, std=0.1)


class ConvUnit(nn.Module):
    """``conv`` = DynamicConv or plain Conv2d; followed by InstanceNorm + LeakyReLU(0.1) (module.py:28-71)."""

    def __init__(self, in_c: int, out_c: int, kernel, stride: int = 1, dynamic: bool = False, padding: int = 0):
        super().__init__()
        self.dynamic, self.stride, self.padding = dynamic, stride, padding
        if dynamic:
            self.conv = DynamicConv(in_c, out_c, kernel, bias=False)
        else:
            self.conv = nn.Conv2d(in_c, out_c, kernel, stride=stride, padding=padding, bias=False)


class FeatureNet(_PackedHolder):
    """Parameter layout of the 3-level dynamic-conv pyramid (module.py:201-232)."""

    def __init__(self, base_channels: int = 8):
        super().__init__()
        b = base_channels
        self.conv00 = ConvUnit(3, b, (3, 7, 11), dynamic=True)
        self.conv01 = ConvUnit(b, b, (3, 5, 7), dynamic=True)
        self.downsample1 = ConvUnit(b, 2 * b, 3, stride=2, padding=1)
        self.conv10 = ConvUnit(2 * b, 2 * b, (3, 5), dynamic=True)
        self.conv11 = ConvUnit(2 * b, 2 * b, (3, 5), dynamic=True)
        self.downsample2 = ConvUnit(2 * b, 4 * b, 3, stride=2, padding=1)
        self.conv20 = ConvUnit(4 * b, 4 * b, (1, 3), dynamic=True)
        self.conv21 = ConvUnit(4 * b, 4 * b, (1, 3), dynamic=True)
        self.out1 = DynamicConv(4 * b, 4 * b, (1, 3))
        self.inner1 = ConvUnit(6 * b, 2 * b, 1)
        self.inner2 = ConvUnit(3 * b, b, 1)
        self.out2 = DynamicConv(2 * b, 2 * b, (1, 3))
        self.out3 = DynamicConv(b, b, (1, 3))
        self.out_channels = [4 * b, 2 * b, b]
        self._init_packed()


class ConvBn2d(nn.Module):
    def __init__(self, in_c: int, out_c: int):
        super().__init__()
        self.conv = nn.Conv2d(in_c, out_c, 3, padding=1, bias=False)
        self.bn = nn.BatchNorm2d(out_c)


class ConvBn3d(nn.Module):
    def __init__(self, in_c: int, out_c: int, stride: int = 1, transposed: bool = False):
        super().__init__()
        self.stride, self.transposed = stride, transposed
        if transposed:
            self.conv = nn.ConvTranspose3d(in_c, out_c, 3, stride=2, padding=1, output_padding=1, bias=False)
        else:
            self.conv = nn.Conv3d(in_c, out_c, 3, stride=stride, padding=1, bias=False)
        self.bn = nn.BatchNorm3d(out_c)


class CostRegNet(_PackedHolder):
    """3D U-Net regulariser (module.py:270-315); forward runs on the HIP conv kernels."""

    def __init__(self, in_channels: int, base_channels: int):
        super().__init__()
        b = base_channels
        self.conv0 = ConvBn3d(in_channels, b)
        self.conv1 = ConvBn3d(b, 2 * b, stride=2)
        self.conv2 = ConvBn3d(2 * b, 2 * b)
        self.conv3 = ConvBn3d(2 * b, 4 * b, stride=2)
        self.conv4 = ConvBn3d(4 * b, 4 * b)
        self.conv5 = ConvBn3d(4 * b, 8 * b, stride=2)
        self.conv6 = ConvBn3d(8 * b, 8 * b)
        self.conv7 = ConvBn3d(8 * b, 4 * b, transposed=True)
        self.conv9 = ConvBn3d(4 * b, 2 * b, transposed=True)
        self.conv11 = ConvBn3d(2 * b, b, transposed=True)
        self.prob = nn.Conv3d(b, 1, 3, stride=1, padding=1, bias=False)
        self._slab_operands = False     # slab.py: also pack the tiled operands of conv9 / conv11 (want_slab_operands())
        self._init_packed()

    def want_slab_operands(self) -> None:
        """The slab-parallel form (slab.py) exchanges halo rows between conv11 and prob and keeps the tiled transposed kernels
        for conv9 / conv11: pack their operands too (every other model skips them)."""
        if not self._slab_operands:
            self._slab_operands = True
            self._packed.invalidate()

    def _pack(self) -> Dict[str, Tensor]:
        out: Dict[str, Tensor] = {}
        for name in ("conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11"):
            unit: ConvBn3d = getattr(self, name)
            scale, shift = _bn_fold(unit.bn)
            w = unit.conv.weight.detach()
            if unit.transposed:   # [Cin,Cout,3,3,3]
                w = (w * scale.view(1, -1, 1, 1, 1)).permute(0, 2, 3, 4, 1)
            else:                 # [Cout,Cin,3,3,3]
                w = (w * scale.view(-1, 1, 1, 1, 1)).permute(1, 2, 3, 4, 0)
            out[name + ".w"] = w.reshape(w.shape[0], 27, w.shape[-1]).contiguous()
            out[name + ".b"] = shift.contiguous()
            if not unit.transposed and unit.stride == 1 and w.shape[0] % 16 == 0 and w.shape[-1] % 16 == 0:
                # ci-fastest copy [27,Cout,Cin] for the channels-last MFMA kernel (conv2, conv4, conv6)
                out[name + ".wcl"] = w.permute(1, 2, 3, 4, 0).reshape(27, w.shape[-1], w.shape[0]).contiguous()
        w = self.prob.weight.detach().permute(1, 2, 3, 4, 0)
        out["prob.w"] = w.reshape(w.shape[0], 27, 1).contiguous()
        if self.split_bf16_supported():
            # operands of the split-bf16 matrix-core kernels (csrc/conv3d_sbf.hip): BN-folded weights split exactly into
            # three bf16 terms, laid out per MFMA lane
            for name in ("conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11"):
                unit = getattr(self, name)
                scale, _ = _bn_fold(unit.bn)
                if name == "conv11":    # fused with the residual and prob (csrc/deconv_prob_zm.hip); the slab-parallel form
                    # (slab.py) exchanges halo rows between the two layers and keeps the separate kernels (".ws" below)
                    out[name + ".wz"] = ops.split_pack_deconv_prob(unit.conv.weight.detach() * scale.view(1, -1, 1, 1, 1))
                    out["prob.tab"] = ops.pack_prob_table(self.prob.weight)
                    if ops.USE_SPLIT_F16:
                        out[name + ".wh"], out[name + ".whs"] = ops.split_pack_deconv_prob(unit.conv.weight.detach() * scale.view(1, -1, 1, 1, 1), f16=True)
                if name == "conv9":     # 32 -> 16: z-marching class-per-wave kernel (csrc/deconv3d_zm.hip); ".ws" stays for slab.py
                    out[name + ".wc"] = ops.split_pack_deconv_cls(unit.conv.weight.detach() * scale.view(1, -1, 1, 1, 1))
                    if ops.USE_SPLIT_F16:
                        out[name + ".wh"], out[name + ".whs"] = ops.split_pack_deconv_cls(unit.conv.weight.detach() * scale.view(1, -1, 1, 1, 1), f16=True)
                if name == "conv7" and ops.USE_SPLIT_F16:
                    out[name + ".wh"], out[name + ".whs"] = ops.split_pack_deconv3d(unit.conv.weight.detach() * scale.view(1, -1, 1, 1, 1), f16=True)
                if unit.transposed:
                    if name != "conv7" and not self._slab_operands:
                        continue        # conv9 / conv11 run their z-marching forms; the tiled operands are slab.py's
                    out[name + ".ws"] = ops.split_pack_deconv3d(unit.conv.weight.detach() * scale.view(1, -1, 1, 1, 1))
                elif name == "conv0":   # Cout = 8, stride 1: voxel-pair columns (no matrix row multiplies padding)
                    out[name + ".ws"] = ops.split_pack_conv3d_pair(unit.conv.weight.detach() * scale.view(-1, 1, 1, 1, 1))
                else:
                    out[name + ".ws"] = ops.split_pack_conv3d(unit.conv.weight.detach() * scale.view(-1, 1, 1, 1, 1))
                # split-f16 operands (two fp16 terms of w x a power-of-two scale, + 1 / scale) of the layers that have the kernel
                if not unit.transposed:
                    wf = unit.conv.weight.detach() * scale.view(-1, 1, 1, 1, 1)
                    code = ops.SBF_PAIR if name == "conv0" else unit.stride
                    if ops.conv3d_sf16_supported(wf.shape[1], wf.shape[0], code):
                        out[name + ".wh"], out[name + ".whs"] = (ops.split_pack_conv3d_pair if name == "conv0" else ops.split_pack_conv3d)(wf, f16=True)
        return out

    def split_bf16_supported(self) -> bool:
        """The split-bf16 kernels cover base_channels = 8 with 8 / 16 / 32 input channels (the three cascade stages)."""
        return (USE_SPLIT_BF16 and self.conv0.conv.out_channels == 8 and self.conv0.conv.in_channels % 8 == 0
                and self.conv0.conv.weight.is_cuda)

    def forward(self, volume: Tensor, channels_last: bool = False, bound: Optional[Tensor] = None) -> Tensor:
        """volume [C,D,h,w] (one batch item; [D,h,w,C] with channels_last) -> [D,h,w].  D, h, w must be multiples of 8.
        The channels-last form runs the split-bf16 matrix-core kernels (fp32-class arithmetic, csrc/conv3d_sbf.hip), the
        planar form the exact-fp32 kernels (one fmaf chain per output).
        bound: a 1-element device tensor >= max |volume| (channels-last form): conv0 - conv3 then run in split-f16 arithmetic (half
        the matrix-pipe work at fp32-class error, csrc/sbf_common.hpp), each layer scaling its input by the bound its producer
        measured; None: split-bf16 throughout."""
        if self.training:
            # training / autograd path: HIP forward + backward kernels behind autograd Functions
            from . import training                                      # validates the dims
            v = volume.permute(3, 0, 1, 2) if channels_last else volume
            return training.cost_regularization(self, v.unsqueeze(0).contiguous())[0, 0]
        D, h, w = volume.shape[:3] if channels_last else volume.shape[1:]
        if D % 8 or h % 8 or w % 8:
            raise ValueError(f"CostRegNet needs D,h,w divisible by 8, got {(D, h, w)}")
        p = self._packed.get(self, self._pack)
        with ops.prof("costreg"):
            if channels_last:
                if "conv0.ws" not in p:
                    raise RuntimeError("CostRegNet: channels-last input needs the split-bf16 kernels (CDS_CONV_EXACT=1 disables them)")
                return self._run_cl(volume, p, bound)
            return self._run(volume, p)

    def regress(self, volume_cl: Tensor, hyp: Tensor, bound: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """volume [D,h,w,C] channels-last + hypotheses [D,h,w] -> (depth [h,w], confidence [h,w]): CostRegNet followed by the
        soft-argmin (models/model.py:83-92).  bound: see forward()."""
        if self.training:
            raise RuntimeError("CostRegNet.regress is the inference path")
        return ops.softargmin_conf(self.forward(volume_cl, channels_last=True, bound=bound), hyp)

    @staticmethod
    def _run_cl(v: Tensor, p: Dict[str, Tensor], bound: Optional[Tensor] = None) -> Tensor:
        f16 = bound is not None and all(f"conv{i}.wh" in p for i in (0, 1, 2, 3, 4, 5, 6, 7, 9, 11)) and ops.USE_SPLIT_F16
        if f16:
            # the whole network in split-f16: every layer leaves max |output| in its slot of `bnd` (zeroed once), the next one scales by it
            bnd = torch.zeros((16,), dtype=torch.float32, device=v.device)
            c0 = ops.conv3d_sbf(v, p["conv0.wh"], p["conv0.b"], 8, stride=ops.SBF_PAIR, in_bound=bound, w_inv_scale=p["conv0.whs"],
                                out_bound=bnd[0:1])
            c1 = ops.conv3d_sbf(c0, p["conv1.wh"], p["conv1.b"], 16, stride=2, in_bound=bnd[0:1], w_inv_scale=p["conv1.whs"],
                                out_bound=bnd[1:2])
            c2 = ops.conv3d_sbf(c1, p["conv2.wh"], p["conv2.b"], 16, in_bound=bnd[1:2], w_inv_scale=p["conv2.whs"], out_bound=bnd[2:3])
            del c1
            c3 = ops.conv3d_sbf(c2, p["conv3.wh"], p["conv3.b"], 32, stride=2, in_bound=bnd[2:3], w_inv_scale=p["conv3.whs"],
                                out_bound=bnd[3:4])
            c4 = ops.conv3d_sbf(c3, p["conv4.wh"], p["conv4.b"], 32, in_bound=bnd[3:4], w_inv_scale=p["conv4.whs"], out_bound=bnd[4:5])
            del c3
            c5 = ops.conv3d_sbf(c4, p["conv5.wh"], p["conv5.b"], 64, stride=2, in_bound=bnd[4:5], w_inv_scale=p["conv5.whs"],
                                out_bound=bnd[5:6])
            x = ops.conv3d_sbf(c5, p["conv6.wh"], p["conv6.b"], 64, in_bound=bnd[5:6], w_inv_scale=p["conv6.whs"], out_bound=bnd[6:7])
            del c5
            x = ops.deconv3d_sbf(x, p["conv7.wh"], p["conv7.b"], 32, skip=c4, in_bound=bnd[6:7], w_inv_scale=p["conv7.whs"],
                                 out_bound=bnd[7:8])
            del c4
            x = ops.deconv3d_zm(x, p["conv9.wh"], p["conv9.b"], skip=c2, in_bound=bnd[7:8], w_inv_scale=p["conv9.whs"], out_bound=bnd[8:9])
            del c2
            return ops.deconv_prob_zm(x, p["conv11.wh"], p["conv11.b"], c0, p["prob.tab"], in_bound=bnd[8:9], w_inv_scale=p["conv11.whs"])
        else:
            c0 = ops.conv3d_sbf(v, p["conv0.ws"], p["conv0.b"], 8, stride=ops.SBF_PAIR)
            c1 = ops.conv3d_sbf(c0, p["conv1.ws"], p["conv1.b"], 16, stride=2)
            c2 = ops.conv3d_sbf(c1, p["conv2.ws"], p["conv2.b"], 16)
            del c1
            c3 = ops.conv3d_sbf(c2, p["conv3.ws"], p["conv3.b"], 32, stride=2)
            c4 = ops.conv3d_sbf(c3, p["conv4.ws"], p["conv4.b"], 32)
            del c3
            c5 = ops.conv3d_sbf(c4, p["conv5.ws"], p["conv5.b"], 64, stride=2)
            x = ops.conv3d_sbf(c5, p["conv6.ws"], p["conv6.b"], 64)
            del c5
        x = ops.deconv3d_sbf(x, p["conv7.ws"], p["conv7.b"], 32, skip=c4)
        del c4
        x = ops.deconv3d_zm(x, p["conv9.wc"], p["conv9.b"], skip=c2)
        del c2
        # conv11 + the conv0 residual + prob: one z-marching kernel, the 8-channel volume between them never reaches HBM
        return ops.deconv_prob_zm(x, p["conv11.wz"], p["conv11.b"], c0, p["prob.tab"])

    @staticmethod
    def _run(volume: Tensor, p: Dict[str, Tensor]) -> Tensor:
        c0 = ops.conv3d_k3(volume, p["conv0.w"], p["conv0.b"])
        c1 = ops.conv3d_k3(c0, p["conv1.w"], p["conv1.b"], stride=2)
        c2 = ops.conv3d_k3(c1, p["conv2.w"], p["conv2.b"], wcl=p.get("conv2.wcl"))
        del c1
        c3 = ops.conv3d_k3(c2, p["conv3.w"], p["conv3.b"], stride=2)
        c4 = ops.conv3d_k3(c3, p["conv4.w"], p["conv4.b"], wcl=p.get("conv4.wcl"))
        del c3
        c5 = ops.conv3d_k3(c4, p["conv5.w"], p["conv5.b"], stride=2)
        x = ops.conv3d_k3(c5, p["conv6.w"], p["conv6.b"], wcl=p.get("conv6.wcl"))
        del c5
        x = ops.deconv3d_k3s2(x, p["conv7.w"], p["conv7.b"], skip=c4)
        del c4
        x = ops.deconv3d_k3s2(x, p["conv9.w"], p["conv9.b"], skip=c2)
        del c2
        x = ops.deconv3d_k3s2(x, p["conv11.w"], p["conv11.b"], skip=c0)
        del c0
        return ops.conv3d_k3(x, p["prob.w"], None, relu=False)[0]


class Refinement(_PackedHolder):
    """2x depth up-sampling with image guidance (module.py:318-370).  Eval mode runs on the HIP
    kernels (3x3 Conv+BN+ReLU units on cds_conv2d_f32 with the BatchNorm folded in, the transposed conv, the depth
    pre-scale and the bilinear-upsample + residual epilogue in refine.hip); training mode runs the HIP forward /
    backward training ops (train2d_ops.refinement; device tensors only)."""

    def __init__(self):
        super().__init__()
        self.conv0 = ConvBn2d(3, 8)
        self.conv1 = ConvBn2d(1, 8)
        self.conv2 = ConvBn2d(8, 8)
        self.deconv = nn.ConvTranspose2d(8, 8, 3, padding=1, output_padding=1, stride=2, bias=False)
        self.bn = nn.BatchNorm2d(8)
        self.conv3 = ConvBn2d(16, 8)
        self.res = nn.Conv2d(8, 1, 3, padding=1, bias=False)
        self._init_packed()

    def _pack(self) -> Dict[str, Tensor]:
        out: Dict[str, Tensor] = {}
        for name in ("conv0", "conv1", "conv2", "conv3"):
            unit: ConvBn2d = getattr(self, name)
            scale, shift = _bn_fold(unit.bn)
            out[name + ".w"] = _pack2d(unit.conv.weight.detach() * scale.view(-1, 1, 1, 1))
            out[name + ".b"] = shift.contiguous()
        scale, shift = _bn_fold(self.bn)
        w = self.deconv.weight.detach() * scale.view(1, -1, 1, 1)          # [Cin,Cout,3,3]
        out["deconv.w"] = w.permute(0, 2, 3, 1).reshape(8, 9, 8).contiguous()
        out["deconv.b"] = shift.contiguous()
        out["res.w"] = _pack2d(self.res.weight.detach())
        return out

    def forward(self, img: Tensor, depth0: Tensor, dmin, dmax=None) -> Tensor:
        """img [B,3,H,W], depth0 [B,1,H/2,W/2], dmin / dmax [B] -> refined depth [B,1,H,W]  (the reference's signature, module.py:344).
        Internal form (CDSMVSNet.forward_device): dmin = a list of B device triples (depth_min, depth_max, interval) of the geometry
        block and dmax None - depth0 is then the depth in depth units and the interval scaling of models/model.py:213-218 happens
        inside the first and last kernel."""
        if self.training:
            from . import training
            return training._refinement(self, img, depth0, dmin, dmax)         # HIP forward / backward kernels (train2d_ops.py)
        p = self._packed.get(self, self._pack)
        B, _, H, W = img.shape
        h, w = depth0.shape[-2:]
        if (2 * h, 2 * w) != (H, W):
            raise ValueError(f"Refinement: image {(H, W)} must be twice the depth map {(h, w)}")
        if dmax is None:
            ranges = list(dmin)
        else:
            lo_hi = _to_host([torch.stack((dmin.float(), dmax.float()), 1)])[0]   # reference signature: host scalars, uploaded
            ranges = [ops.geo([float(lo_hi[b, 0]), float(lo_hi[b, 1]), 1.0], img.device, "depth_range") for b in range(B)]
        outs = []
        with ops.prof("refinement"):
            for b in range(B):
                d = ops.depth_affine(depth0[b, 0].contiguous(), ranges[b])                           # [h,w]
                cat = torch.empty((1, 16, H, W), dtype=torch.float32, device=img.device)            # deconv | conv0
                ops.conv2d(img[b:b + 1].contiguous(), p["conv0.w"], p["conv0.b"], 8, 3, 1, 1, ACT_RELU, out=cat[0, 8:])
                x = ops.conv2d(d.view(1, 1, h, w), p["conv1.w"], p["conv1.b"], 8, 3, 1, 1, ACT_RELU)
                x = ops.conv2d(x, p["conv2.w"], p["conv2.b"], 8, 3, 1, 1, ACT_RELU)
                ops.deconv2d_k3s2(x[0], p["deconv.w"], p["deconv.b"], ACT_RELU, out=cat[0, :8])
                x = ops.conv2d(cat, p["conv3.w"], p["conv3.b"], 8, 3, 1, 1, ACT_RELU)
                res = ops.conv2d(x, p["res.w"], None, 1, 3, 1, 1, ACT_NONE)
                outs.append(ops.refine_finish(d, res[0, 0], ranges[b]))
        return torch.stack(outs).unsqueeze(1)


# ------------------------------------------------------------------------------------------------
# host readback of the (tiny) camera / depth-range tensors
# ------------------------------------------------------------------------------------------------
_READBACK_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def _to_host(tensors: List[Tensor]) -> List[Tensor]:
    """fp32 CPU copies of the small camera / depth-range tensors.  CPU inputs are used as they are: callers that keep
    these tensors on the host (they come from the data loader on the host anyway; `infer.py` and `bench.py` do) pay no
    readback, and the host side of forward k+1 overlaps the GPU side of forward k.  Device inputs are copied on a side
    stream that first waits for the work queued on the current stream (their producers may still be running there), which
    is what `.cpu()` costs as well."""
    out: List[Optional[Tensor]] = [None] * len(tensors)
    dev_idx = [i for i, t in enumerate(tensors) if t.is_cuda]
    for i, t in enumerate(tensors):
        if not t.is_cuda:
            out[i] = t.detach().float()
    if dev_idx:
        dev = tensors[dev_idx[0]].device
        side = _READBACK_STREAMS.get(dev.index)
        if side is None:
            side = _READBACK_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for i in dev_idx:
                out[i] = tensors[i].detach().float().to("cpu", non_blocking=True)
        side.synchronize()
    return out  # type: ignore[return-value]


# ------------------------------------------------------------------------------------------------
# FeatureNet on the HIP kernels (module.py:234-267, dynamic_conv.py:97-122)
# ------------------------------------------------------------------------------------------------
class _FeatureRunner:
    """FeatureNet forward over the HIP kernels.  Holds no state besides the net it is bound to for one call
    (``CDSMVSNet`` binds a fresh runner per forward, so ``nn.DataParallel`` replicas use their own parameter tensors)."""

    def __init__(self, net: FeatureNet):
        self.net = net

    def _pack(self) -> Dict[str, Tensor]:
        out: Dict[str, Tensor] = {}
        net = self.net

        def dyn(name: str, dc: DynamicConv):
            dev = dc.att_convs[0].weight.device
            for i, k in enumerate(dc.size_kernels):
                w = torch.cat((dc.convs[i].weight.detach(), dc.att_convs[i].weight.detach()), dim=0)
                out[f"{name}.w{i}"] = _pack2d(w)
                if dc.convs[i].bias is not None:
                    out[f"{name}.b{i}"] = torch.cat((dc.convs[i].bias.detach(), torch.zeros(3, device=dev))).contiguous()
            if ops.dynconv_sbf_supported(dc.in_c, dc.out_c + 3, dc.size_kernels, 4, fused=USE_FUSED_BLEND) and dc.att_convs[0].weight.is_cuda:
                out[f"{name}.ws"] = ops.split_pack_dynconv([torch.cat((dc.convs[i].weight.detach(), dc.att_convs[i].weight.detach()), dim=0)
                                                            for i in range(len(dc.size_kernels))])
                if ops.USE_SPLIT_F16 and (dc.in_c, tuple(dc.size_kernels)) in ops.DYNCONV_CL_SHAPES:
                    # split-f16 operands of the channels-last kernel (two fp16 terms of w x a power-of-two scale, + 1 / scale)
                    out[f"{name}.wh"], out[f"{name}.whs"] = ops.split_pack_dynconv(
                        [torch.cat((dc.convs[i].weight.detach(), dc.att_convs[i].weight.detach()), dim=0) for i in range(len(dc.size_kernels))],
                        f16=True)
                if dc.convs[0].bias is not None:
                    out[f"{name}.bs"] = torch.stack([out[f"{name}.b{i}"] for i in range(len(dc.size_kernels))]).contiguous()
            if name == "conv00" and USE_SPLIT_BF16 and ops.USE_CONV2D_SBF and dc.size_kernels == (3, 7, 11) and dc.att_convs[0].weight.is_cuda:
                # conv00 on the matrix cores (csrc/feat_cl.hip: tap-pair K-steps for the 3-channel input)
                out[f"{name}.ws00"] = ops.split_pack_conv00([torch.cat((dc.convs[i].weight.detach(), dc.att_convs[i].weight.detach()), dim=0)
                                                             for i in range(3)])
                if ops.USE_SPLIT_F16:
                    out[f"{name}.wh00"], out[f"{name}.whs00"] = ops.split_pack_conv00(
                        [torch.cat((dc.convs[i].weight.detach(), dc.att_convs[i].weight.detach()), dim=0) for i in range(3)], f16=True)
                if dc.convs[0].bias is not None:
                    out[f"{name}.bs"] = torch.stack([out[f"{name}.b{i}"] for i in range(3)]).contiguous()
            scale, shift = _bn_fold(dc.att_weights[1])
            nk = len(dc.size_kernels)
            out[f"{name}.m1"] = (dc.att_weights[0].weight.detach().reshape(4, nk) * scale.view(4, 1)).contiguous()
            out[f"{name}.mb"] = shift.contiguous()
            out[f"{name}.m2"] = dc.att_weights[3].weight.detach().reshape(nk, 4).contiguous()

        for name in ("conv00", "conv01", "conv10", "conv11", "conv20", "conv21"):
            dyn(name, getattr(net, name).conv)
        for name in ("out1", "out2", "out3"):
            dyn(name, getattr(net, name))
        for name in ("downsample1", "downsample2", "inner1", "inner2"):
            out[f"{name}.w"] = _pack2d(getattr(net, name).conv.weight.detach())
        # channels-last kernels (feat_cl.hip): [tap][cin][cout] for the stride-2 units, [cin][cout] for the FPN laterals
        for name in ("downsample1", "downsample2"):
            w = getattr(net, name).conv.weight.detach()
            out[f"{name}.w9"] = w.permute(2, 3, 1, 0).reshape(9, w.shape[1], w.shape[0]).contiguous()
            if ops.USE_SPLIT_F16 and w.is_cuda and (w.shape[1], w.shape[0]) in ((8, 16), (16, 32)):
                out[f"{name}.wh"], out[f"{name}.whs"] = ops.split_pack_dynconv([w], f16=True)      # the matrix-core form (split-f16)
        for name in ("inner1", "inner2"):
            w = getattr(net, name).conv.weight.detach()
            out[f"{name}.wt"] = w.reshape(w.shape[0], w.shape[1]).t().contiguous()
        return out

    # A layer output travels as (raw, affine): the un-normalised convolution result plus the [N,C,3] table
    # (1/std, -mean/std, leaky slope) of its InstanceNorm + LeakyReLU, which the consuming convolutions apply on load
    # (`cds_conv2d_affine_f32`) - the normalised tensor is never written.  affine = None marks a materialised tensor.
    def _dynamic(self, p, name: str, dc: DynamicConv, x: Tensor, epi: Tensor, T: float, n_shared: int = 1,
                 aff: Optional[Tensor] = None, stats_slope: Optional[float] = 0.1):
        """x [N,Cin,H,W] (with its pending affine), epi CPU [N,2] (pixels at this resolution) ->
        (out [N,Cout,H,W], norm_curv [N,H,W]).  n_shared > 1: the first n_shared images are copies of one image (SURVEY
        §8(f)-4): their epipole-independent branch responses are convolved once (image n_shared - 1 stands for all)."""
        N, Cin, H, W = x.shape
        nk = len(dc.size_kernels)
        xs = x[n_shared - 1:] if n_shared > 1 else x
        affs = aff[n_shared - 1:].contiguous() if (aff is not None and n_shared > 1) else aff
        fusable = USE_FUSED_BLEND and n_shared == 1 and stats_slope is not None and nk >= 2 and N <= ops.MAX_IMAGES
        sbf = f"{name}.ws" in p and ops.dynconv_sbf_supported(Cin, dc.out_c + 3, dc.size_kernels, W, fused=fusable)
        if sbf and fusable:
            # branch convolutions + blend epilogue in one kernel: the [K, N, Cout + 3] branch tensor never exists
            return ops.dynconv_fused_sbf(x.contiguous(), p[f"{name}.ws"], p.get(f"{name}.bs"), dc.out_c, dc.size_kernels,
                                         p[f"{name}.m1"], p[f"{name}.mb"], p[f"{name}.m2"], epi, T, stats_slope, in_affine=aff)
        branches = torch.empty((nk, xs.shape[0], dc.out_c + 3, H, W), dtype=torch.float32, device=x.device)
        if sbf:
            # all kernel sizes from one staged tile on the matrix cores (split-bf16 arithmetic)
            ops.dynconv_branches_sbf(xs.contiguous(), p[f"{name}.ws"], p.get(f"{name}.bs"), dc.out_c + 3, dc.size_kernels,
                                     out=branches, in_affine=affs)
        else:
            for i, k in enumerate(dc.size_kernels):
                ops.conv2d(xs, p[f"{name}.w{i}"], p.get(f"{name}.b{i}"), dc.out_c + 3, k, 1, (k - 1) // 2, ACT_NONE,
                           out=branches[i], in_affine=affs)
        return ops.dynconv_blend(branches, p[f"{name}.m1"], p[f"{name}.mb"], p[f"{name}.m2"], epi, T, n_shared,
                                 stats_slope=stats_slope)

    def _dyn_unit(self, p, name, x, epi, T, n_shared: int = 1, aff: Optional[Tensor] = None):
        # the blend kernel leaves the InstanceNorm statistics of its output: (raw, affine) without another pass
        y, nc, _, a = self._dynamic(p, name, getattr(self.net, name).conv, x, epi, T, n_shared, aff)
        return y, a, nc

    def _plain_unit(self, p, name, x, aff: Optional[Tensor] = None):
        unit: ConvUnit = getattr(self.net, name)
        k = unit.conv.kernel_size[0]
        y = ops.conv2d(x, p[f"{name}.w"], None, unit.conv.out_channels, k, unit.stride, unit.padding, in_affine=aff)
        return y, ops.instnorm_affine(y, 0.1)

    def _lateral_unit(self, p, name, coarse, a_coarse, skip, a_skip):
        """FPN lateral: ConvUnit 1x1 over cat(nearest2x(coarse), skip), neither of which is materialised."""
        unit: ConvUnit = getattr(self.net, name)
        return ops.conv2d_fpn(coarse, skip, p[f"{name}.w"], unit.conv.out_channels, a_coarse, a_skip, stats_slope=0.1)

    @staticmethod
    def _final(o: Tensor, st: Tensor, n_chw: int) -> Tuple[Tensor, Optional[Tensor]]:
        """InstanceNorm + tanh of the stage output (st = its statistics from the blend kernel); the first n_chw images
        stay [C,h,w] (reference features), the rest are emitted channels-last [h,w,C] (source features, gathered by K1/K3)."""
        N = o.shape[0]
        chw = ops.instnorm_apply(o[:n_chw], st[:n_chw], ACT_TANH) if n_chw > 0 else None
        hwc = ops.instnorm_apply(o[n_chw:], st[n_chw:], ACT_TANH, out_hwc=True) if n_chw < N else None
        return chw, hwc

    def __call__(self, imgs: Tensor, epipoles: Tensor, T: float, n_chw: Optional[int] = None, n_shared: int = 1, on_stage1=None):
        """imgs [N,3,H,W]; epipoles: one per image in pixels - a CPU tensor [N,2] at full resolution, or the three device tensors [N,2]
        of the geometry block at full / half / quarter resolution.
        Returns {'stageK': (fea_chw [n_chw,C,h,w] | None, fea_hwc [N-n_chw,h,w,C] | None, nc_sum [N,h,w], |nc| [N,h,w])}."""
        net = self.net
        if net.training:
            raise NotImplementedError("_FeatureRunner is the inference runner (InstanceNorm applied on load, packed weights); "
                                      "in training mode CDSMVSNet.forward uses training.feature_net (autograd ops)")
        N = imgs.shape[0]
        if n_chw is None:
            n_chw = N
        if N > ops.MAX_IMAGES:
            raise ValueError(f"at most {ops.MAX_IMAGES} images per FeatureNet batch")
        p = net._packed.get(net, self._pack)
        if isinstance(epipoles, (tuple, list)):
            e0, e1, e2 = epipoles                 # device slices of the call's geometry block (CDSMVSNet.geometry_block)
        else:                                     # direct callers: CPU [N,2] at full resolution, uploaded here
            e0 = epipoles.detach().float().cpu().contiguous()
            e0, e1, e2 = (ops.geo(e, imgs.device, "epipoles") for e in (e0, e0 / 2, e0 / 4))
        if USE_FEAT_CL and net.conv00.conv.out_c == 8 and all(f"{nm}.ws" in p for nm in self._CL_LAYERS):
            return self._call_cl(p, imgs, e0, e1, e2, T, n_chw, n_shared, on_stage1)
        # conv00 sees the raw images: with n_shared copies of the reference image its branch convolutions (3x3, 7x7,
        # 11x11) run once for all of them; from conv01 on the inputs differ (the blend depends on the epipole)
        c00, a00, n00 = self._dyn_unit(p, "conv00", imgs, e0, T, n_shared)
        c01, a01, n01 = self._dyn_unit(p, "conv01", c00, e0, T, aff=a00)
        d0, ad0 = self._plain_unit(p, "downsample1", c01, a01)
        c10, a10, n10 = self._dyn_unit(p, "conv10", d0, e1, T, aff=ad0)
        c11, a11, n11 = self._dyn_unit(p, "conv11", c10, e1, T, aff=a10)
        d1, ad1 = self._plain_unit(p, "downsample2", c11, a11)
        c20, a20, n20 = self._dyn_unit(p, "conv20", d1, e2, T, aff=ad1)
        c21, a21, n21 = self._dyn_unit(p, "conv21", c20, e2, T, aff=a20)

        out = {}
        o1, n22, s1, _ = self._dynamic(p, "out1", net.out1, c21, e2, T, aff=a21)
        out["stage1"] = self._final(o1, s1, n_chw) + ops.curvature_stats(n20, n21, n22)
        if on_stage1 is not None:          # the coarse stage can start (on another stream) while the finer FPN levels are computed
            on_stage1("stage1", out["stage1"])

        # FPN: nearest-neighbour up-sampling and concatenation move raw values; the affine tables concatenate alike
        x, ax = self._lateral_unit(p, "inner1", c21, a21, c11, a11)
        o2, n12, s2, _ = self._dynamic(p, "out2", net.out2, x, e1, T, aff=ax)
        o2n = ops.instnorm_apply(o2, s2, ACT_TANH)
        hwc2 = torch.stack([ops.chw_to_hwc(o2n[i]) for i in range(n_chw, N)]) if n_chw < N else None
        out["stage2"] = (o2n[:n_chw] if n_chw > 0 else None, hwc2) + ops.curvature_stats(n10, n11, n12)
        if on_stage1 is not None:
            on_stage1("stage2", out["stage2"])

        x, ax = self._lateral_unit(p, "inner2", o2n, None, c01, a01)      # o2n is materialised (tanh features)
        o3, n02, s3, _ = self._dynamic(p, "out3", net.out3, x, e0, T, aff=ax)
        out["stage3"] = self._final(o3, s3, n_chw) + ops.curvature_stats(n00, n01, n02)
        return out


    _CL_LAYERS = ("conv01", "conv10", "conv11", "conv20", "conv21", "out1", "out2", "out3")

    def _call_cl(self, p, imgs: Tensor, e0: Tensor, e1: Tensor, e2: Tensor, T: float, n_chw: int, n_shared: int, on_stage1):
        """The same network on CHANNELS-LAST activations [N,h,w,C] (csrc/feat_cl.hip): staged tile rows are contiguous runs, the
        DynamicConv kernel stores 16-byte channel quads, the stage outputs leave channels-last for the source views (what K1 / K3
        gather) and planar for the reference views in the same pass - no transposition kernels."""
        net = self.net
        N, _, H, W = imgs.shape
        # conv00 (3 input channels, kernel sizes 3 / 7 / 11) sees the raw planar images; the n_shared reference copies are convolved
        # once and blended per copy (its own epipole).  Default: one matrix-core kernel (tap-pair K-steps); CDS_CONV00_MFMA=0: the
        # VALU branch kernels + a blend kernel that writes channels-last
        dc = net.conv00.conv
        xs = imgs[n_shared - 1:] if n_shared > 1 else imgs
        if "conv00.ws00" in p and USE_CONV00_MFMA:
            if ops.USE_SPLIT_F16 and "conv00.wh00" in p:      # split-f16: the images' scale from their largest magnitude (one small reduction)
                xs = xs.contiguous()
                c00, n00, _, a00 = ops.conv00_cl(xs, p["conv00.wh00"], p.get("conv00.bs"), p["conv00.m1"], p["conv00.mb"], p["conv00.m2"],
                                                 e0, T, n_shared, 0.1, in_bound=xs.abs().amax().reshape(1), w_inv_scale=p["conv00.whs00"])
                return self._after_conv00(p, c00, n00, a00, e0, e1, e2, T, n_chw, on_stage1)
            c00, n00, _, a00 = ops.conv00_cl(xs.contiguous(), p["conv00.ws00"], p.get("conv00.bs"), p["conv00.m1"], p["conv00.mb"],
                                             p["conv00.m2"], e0, T, n_shared, 0.1)
            return self._after_conv00(p, c00, n00, a00, e0, e1, e2, T, n_chw, on_stage1)
        branches = torch.empty((len(dc.size_kernels), xs.shape[0], dc.out_c + 3, H, W), dtype=torch.float32, device=imgs.device)
        for i, k in enumerate(dc.size_kernels):
            ops.conv2d(xs, p[f"conv00.w{i}"], p.get(f"conv00.b{i}"), dc.out_c + 3, k, 1, (k - 1) // 2, ACT_NONE, out=branches[i])
        c00, n00, _, a00 = ops.dynconv_blend_cl(branches, p["conv00.m1"], p["conv00.mb"], p["conv00.m2"], e0, T, n_shared, 0.1)
        del branches
        return self._after_conv00(p, c00, n00, a00, e0, e1, e2, T, n_chw, on_stage1)

    def _after_conv00(self, p, c00: Tensor, n00: Tensor, a00: Tensor, e0: Tensor, e1: Tensor, e2: Tensor, T: float, n_chw: int, on_stage1):
        net = self.net
        N = c00.shape[0]

        def dyn(name: str, dc: DynamicConv, x: Tensor, epi: Tensor, aff: Optional[Tensor]):
            if ops.USE_SPLIT_F16 and f"{name}.wh" in p and aff is not None:
                # split-f16: the input is InstanceNorm-ed on load, so sqrt(h w) bounds it (Samuelson's inequality; LeakyReLU shrinks)
                return ops.dynconv_cl(x, p[f"{name}.wh"], p.get(f"{name}.bs"), dc.size_kernels, p[f"{name}.m1"], p[f"{name}.mb"],
                                      p[f"{name}.m2"], epi, T, 0.1, in_affine=aff, x_bound=float(x.shape[1] * x.shape[2]) ** 0.5,
                                      w_inv_scale=p[f"{name}.whs"])
            return ops.dynconv_cl(x, p[f"{name}.ws"], p.get(f"{name}.bs"), dc.size_kernels, p[f"{name}.m1"], p[f"{name}.mb"],
                                  p[f"{name}.m2"], epi, T, 0.1, in_affine=aff)

        def down(name: str, x: Tensor, aff: Tensor):
            if ops.USE_SPLIT_F16 and f"{name}.wh" in p and aff is not None:
                y = ops.conv2d_k3s2_cl(x, None, getattr(net, name).conv.out_channels, aff, wsplit=p[f"{name}.wh"],
                                       w_inv_scale=p[f"{name}.whs"], x_bound=float(x.shape[1] * x.shape[2]) ** 0.5)
                return y, ops.instnorm_stats_cl(y, 0.1)[1]
            y = ops.conv2d_k3s2_cl(x, p[f"{name}.w9"], getattr(net, name).conv.out_channels, aff)
            return y, ops.instnorm_stats_cl(y, 0.1)[1]

        c01, n01, _, a01 = dyn("conv01", net.conv01.conv, c00, e0, a00)
        del c00
        d0, ad0 = down("downsample1", c01, a01)
        c10, n10, _, a10 = dyn("conv10", net.conv10.conv, d0, e1, ad0)
        c11, n11, _, a11 = dyn("conv11", net.conv11.conv, c10, e1, a10)
        del d0, c10
        d1, ad1 = down("downsample2", c11, a11)
        c20, n20, _, a20 = dyn("conv20", net.conv20.conv, d1, e2, ad1)
        c21, n21, _, a21 = dyn("conv21", net.conv21.conv, c20, e2, a20)
        del d1, c20

        out = {}
        o1, n22, s1, _ = dyn("out1", net.out1, c21, e2, a21)
        hwc, chw = ops.instnorm_apply_cl(o1, s1, ACT_TANH, n_chw, cl_from=n_chw)
        out["stage1"] = (chw, hwc) + ops.curvature_stats(n20, n21, n22)
        if on_stage1 is not None:
            on_stage1("stage1", out["stage1"])

        x, ax = ops.conv2d_fpn_cl(c21, c11, p["inner1.wt"], net.inner1.conv.out_channels, a21, a11, 0.1)
        o2, n12, s2, _ = dyn("out2", net.out2, x, e1, ax)
        o2n, chw = ops.instnorm_apply_cl(o2, s2, ACT_TANH, n_chw, cl_from=0)       # all images channels-last: inner2 reads them
        out["stage2"] = (chw, o2n[n_chw:] if n_chw < N else None) + ops.curvature_stats(n10, n11, n12)
        if on_stage1 is not None:
            on_stage1("stage2", out["stage2"])

        x, ax = ops.conv2d_fpn_cl(o2n, c01, p["inner2.wt"], net.inner2.conv.out_channels, None, a01, 0.1)
        o3, n02, s3, _ = dyn("out3", net.out3, x, e0, ax)
        hwc, chw = ops.instnorm_apply_cl(o3, s3, ACT_TANH, n_chw, cl_from=n_chw)
        out["stage3"] = (chw, hwc) + ops.curvature_stats(n00, n01, n02)
        return out


# ------------------------------------------------------------------------------------------------
# StageNet: one cost-volume stage (models/model.py:11-94)
# ------------------------------------------------------------------------------------------------
class StageNet(_PackedHolder):
    def __init__(self, num_mvs_stages: int = 3):
        super().__init__()
        self.vis = nn.ModuleList([nn.Sequential(ConvBn2d(2, 16), ConvBn2d(16, 16), ConvBn2d(16, 16),
                                                nn.Conv2d(16, 1, 1), nn.Sigmoid()) for _ in range(num_mvs_stages)])
        self._init_packed()

    def _pack(self) -> Dict[str, Tensor]:
        out: Dict[str, Tensor] = {}
        for s, seq in enumerate(self.vis):
            for i in range(3):
                scale, shift = _bn_fold(seq[i].bn)
                wf = seq[i].conv.weight.detach() * scale.view(-1, 1, 1, 1)
                out[f"{s}.w{i}"] = _pack2d(wf)
                out[f"{s}.b{i}"] = shift.contiguous()
                if i > 0:    # 16 -> 16: matrix-core layouts: fp32 MFMA [tap][cout][cin]; split-bf16 (the DynamicConv kernel's)
                    out[f"{s}.wcl{i}"] = wf.permute(2, 3, 0, 1).reshape(9, 16, 16).contiguous()
                    if USE_SPLIT_BF16 and wf.is_cuda:
                        out[f"{s}.ws{i}"] = ops.split_pack_dynconv([wf])
            out[f"{s}.w3"] = _pack2d(seq[3].weight.detach())
            out[f"{s}.b3"] = seq[3].bias.detach().contiguous()
            out[f"{s}.hw"] = seq[3].weight.detach().reshape(16).contiguous()
        return out

    def visibility(self, entropy: Tensor, ref_nc: Tensor, stage_idx: int) -> Tensor:
        """entropy, ref_nc [V,h,w] -> visibility weight [V,h,w]   (model.py:14,51)."""
        p = self._packed.get(self, self._pack)
        s = stage_idx
        with ops.prof("visibility_cnn"):
            if USE_FEAT_CL and f"{s}.ws1" in p and ops.USE_CONV2D_SBF:
                # channels-last activations (csrc/feat_cl.hip): layer 1 straight from the two maps, layers 2 / 3 on the matrix cores
                x = ops.vis_layer1_cl(entropy.contiguous(), ref_nc.contiguous(), p[f"{s}.w0"], p[f"{s}.b0"])
                x = ops.conv2d_k3_relu_cl(x, p[f"{s}.ws1"], p[f"{s}.b1"])
                return ops.conv2d_k3_relu_cl(x, p[f"{s}.ws2"], p[f"{s}.b2"], head_w=p[f"{s}.hw"], head_b=p[f"{s}.b3"])
            x = torch.stack((entropy, ref_nc), dim=1)
            x = ops.conv2d(x, p[f"{s}.w0"], p[f"{s}.b0"], 16, 3, 1, 1, ACT_RELU)
            if f"{s}.ws1" in p and ops.USE_CONV2D_SBF and x.shape[-1] % 4 == 0:
                # layers 2 and 3 on the bf16 matrix cores in split-bf16 arithmetic; the last one applies the 1x1 head + sigmoid
                x = ops.conv2d_k3_relu_sbf(x, p[f"{s}.ws1"], p[f"{s}.b1"])
                return ops.conv2d_k3_relu_sbf(x, p[f"{s}.ws2"], p[f"{s}.b2"], head_w=p[f"{s}.hw"], head_b=p[f"{s}.b3"])
            if ops.conv2d_c16_supported(x):
                # matrix-core layers; the last one also applies the 1x1 head + sigmoid
                x = ops.conv2d_k3_c16(x, p[f"{s}.wcl1"], p[f"{s}.b1"], ACT_RELU)
                return ops.conv2d_k3_c16(x, p[f"{s}.wcl2"], p[f"{s}.b2"], ACT_RELU, head_w=p[f"{s}.hw"], head_b=p[f"{s}.b3"])
            for i in (1, 2):
                x = ops.conv2d(x, p[f"{s}.w{i}"], p[f"{s}.b{i}"], 16, 3, 1, 1, ACT_RELU)
            return ops.conv2d(x, p[f"{s}.w3"], p[f"{s}.b3"], 1, 1, 1, 0, ACT_SIGMOID)[:, 0]

    def aggregate(self, ref_chw: Tensor, src_hwc: Tensor, ref_nc: Tensor, mats: Tensor, hyp: Tensor, stage_idx: int,
                  normalize: bool = True, channels_last: bool = False):
        """K1 -> vis CNN -> K3 for the given source views.  Returns (volume, vis_sum, entropy, vis_w); the volume is
        [C,D,h,w], or [D,h,w,C] with channels_last."""
        ent = ops.warp_entropy(ref_chw, src_hwc, mats, hyp)
        vis = self.visibility(ent, ref_nc, stage_idx).contiguous()
        volume, vis_sum = ops.warp_aggregate(ref_chw, src_hwc, vis, mats, hyp, normalize=normalize, channels_last=channels_last)
        return volume, vis_sum, ent, vis

    def run_single(self, ref_chw, src_hwc, ref_nc, nc_sums, mats, hyp, cost_regularization, stage_idx, vol_bound=None, nc_mean=None):
        """One batch item.  ref_chw [V,C,h,w], src_hwc [V,h,w,C], ref_nc [V,h,w], nc_sums [V,h,w] (already
        (ref+src)/2 per view), hyp [D,h,w].  vol_bound: a 1-element device tensor >= max |volume| for CostRegNet's split-f16 layers
        (the model passes 1: its features are tanh outputs); None = max |ref| x max |src|, which bounds the normalised volume - a
        visibility-weighted average of ref x (bilinear, zero-padded samples of src) - for any features."""
        cl = isinstance(cost_regularization, CostRegNet) and cost_regularization.split_bf16_supported()
        volume, _, _, _ = self.aggregate(ref_chw, src_hwc, ref_nc, mats, hyp, stage_idx, channels_last=cl)
        if cl:
            if vol_bound is None and ops.USE_SPLIT_F16:
                vol_bound = (ref_chw.abs().amax() * src_hwc.abs().amax()).reshape(1)
            depth, conf = cost_regularization.regress(volume, hyp, bound=vol_bound)
        else:
            depth, conf = ops.softargmin_conf(cost_regularization(volume), hyp)
        del volume
        if nc_mean is None:        # (ops.stage_inputs has averaged the per-pair curvature sums already)
            nc_mean = ops.view_mean(nc_sums.contiguous())
        return depth, conf, nc_mean

    def forward(self, features, proj_matrices, depth_values, num_depth, cost_regularization, prob_volume_init=None,
                stage_idx=0, gt_depth=None):
        """Reference signature (model.py:16).  features: list over source views of
        {'ref': (fea [B,C,h,w], nc_sum [B,1,h,w], nc [B,1,h,w]), 'src': (fea, nc_sum, _)}; proj_matrices
        [B,N,2,4,4]; depth_values [B,D,h,w]."""
        if prob_volume_init is not None:
            raise NotImplementedError("prob_volume_init is dead code in the reference (never passed)")
        assert len(features) == proj_matrices.shape[1] - 1, "Different number of images and projection matrices"
        assert depth_values.shape[1] == num_depth, f"depth_values.shape[1]:{depth_values.shape[1]}  num_depth:{num_depth}"
        if self.training or gt_depth is not None:
            # training branch (model.py:52-56,63-69): autograd path on the HIP forward / backward kernels
            from .training import stage_forward_train
            dv = depth_values
            if dv.dim() == 2:
                h, w = features[0]["ref"][0].shape[-2:]
                dv = dv.view(*dv.shape, 1, 1).expand(-1, -1, h, w)
            with torch.cuda.device(dv.device):
                return stage_forward_train(self, features, proj_matrices, dv, cost_regularization, stage_idx, gt_depth)
        with torch.cuda.device(depth_values.device):   # launches go to the current device's stream
            return self._forward(features, proj_matrices, depth_values, cost_regularization, stage_idx)

    def _forward(self, features, proj_matrices, depth_values, cost_regularization, stage_idx):
        B = depth_values.shape[0]
        cams = _to_host([proj_matrices])[0]
        V = len(features)
        depths, confs, ncs = [], [], []
        geo = geometry.GeoBlock()                  # the homographies of the call: one asynchronous upload, read by K1 and K3
        for b in range(B):
            geo.add(f"b{b}.mats", geometry.warp_matrices(cams[b]))
        geo.upload(depth_values.device)
        for b in range(B):
            rf = [f["ref"][0][b] for f in features]
            sf = [f["src"][0][b] for f in features]
            maps = ([f["ref"][2][b, 0] for f in features], [f["ref"][1][b, 0] for f in features],
                    [f["src"][1][b, 0] for f in features])
            nc_mean = vol_bound = None
            if V <= ops.MAX_VIEWS and ops.stage_inputs_supported(rf, sf, maps):
                # the whole re-layout of the reference's feature dicts, the curvature mean and the volume bound in one launch
                ref, src, ref_nc, nc_mean, vol_bound = ops.stage_inputs(rf, sf, *maps, want_bound=ops.USE_SPLIT_F16)
                nc_sums = None
            else:
                ref = torch.stack(rf).contiguous()
                C, h, w = ref.shape[1:]
                src = torch.empty((V, h, w, C), dtype=torch.float32, device=ref.device)
                for v in range(V):   # transposed straight into the stacked buffer (no second copy)
                    ops.chw_to_hwc(sf[v].contiguous(), out=src[v])
                ref_nc = torch.stack(maps[0]).contiguous()
                nc_sums = ops.pair_mean(torch.stack(maps[1] + maps[2]), V)
            hyp = depth_values[b]
            if hyp.dim() == 1:
                h, w = ref.shape[-2:]
                hyp = hyp.view(-1, 1, 1).expand(-1, h, w)
            d, c, n = self.run_single(ref, src, ref_nc, nc_sums, geo[f"b{b}.mats"], hyp.contiguous(), cost_regularization, stage_idx,
                                      vol_bound=vol_bound, nc_mean=nc_mean)
            depths.append(d)
            confs.append(c)
            ncs.append(n.unsqueeze(0))
        if B == 1:      # a batch of one: views of the fresh outputs, no copy
            return {"depth": depths[0].unsqueeze(0), "photometric_confidence": confs[0].unsqueeze(0), "norm_curv": ncs[0].unsqueeze(0)}
        return {"depth": torch.stack(depths), "photometric_confidence": torch.stack(confs),
                "norm_curv": torch.stack(ncs)}


# ------------------------------------------------------------------------------------------------
# the model (models/model.py:97-223)
# ------------------------------------------------------------------------------------------------
class CDSMVSNet(nn.Module):
    def __init__(self, refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4, 2, 1), share_cr=False,
                 grad_method="detach", arch_mode="fpn", cr_base_chs=(8, 8, 8)):
        super().__init__()
        assert len(ndepths) == len(depth_interals_ratio)
        if share_cr:
            raise NotImplementedError("share_cr=True crashes in the reference too (model.py:129-130 passes a list)")
        self.refine = refine
        self.share_cr = share_cr
        self.ndepths = tuple(ndepths)
        self.depth_interals_ratio = tuple(depth_interals_ratio)
        self.grad_method = grad_method
        self.arch_mode = arch_mode
        self.cr_base_chs = tuple(cr_base_chs)
        self.num_stage = len(ndepths)
        self.stage_infos = {"stage1": {"scale": 4.0}, "stage2": {"scale": 2.0}, "stage3": {"scale": 1.0}}

        self.feature = FeatureNet(base_channels=8)
        self.stage_net = StageNet(num_mvs_stages=self.num_stage)
        self.cost_regularization = nn.ModuleList([CostRegNet(self.feature.out_channels[i], self.cr_base_chs[i])
                                                  for i in range(self.num_stage)])
        if self.refine:
            self.refine_network = Refinement()
        # view-shard hook: set by cds_mvsnet_amd.distributed.shard_views(); None = all views on this GPU
        self._view_shard = None

    # -- helpers -----------------------------------------------------------------------------
    def extract_features(self, ref_img: Tensor, src_imgs: List[Tensor], epi_groups, T: float, on_stage1=None):
        """FeatureNet for every (reference, source) pair in ONE batched pass: V copies of the reference image (each
        conditioned on its pair's epipole — DynamicConv makes reference features pair specific, model.py:154-161)
        followed by the V source images.  epi_groups: per group of at most MAX_IMAGES / 2 pairs the epipoles of its images
        ([reference copies ..., sources ...]) at the three FeatureNet resolutions as device tensors [2 g, 2] (slices of the call's
        geometry block, :meth:`geometry_block`).  Returns the runner's dict; rows [0,V) are the reference features (CHW), the source
        features come back channels-last."""
        V = len(src_imgs)
        G = ops.MAX_IMAGES // 2   # pairs per batched pass (2 images per pair); more views run in groups
        parts = []
        for gi, v0 in enumerate(range(0, V, G)):
            n = min(G, V - v0)
            batch = torch.stack([ref_img] * n + list(src_imgs[v0:v0 + G]))
            parts.append((n, _FeatureRunner(self.feature)(batch, epi_groups[gi], T, n_chw=n, n_shared=n,
                                                          on_stage1=on_stage1 if V <= G else None)))
        if len(parts) == 1:
            return parts[0][1]
        out = {}
        for name in parts[0][1]:
            chw = torch.cat([p[name][0] for _, p in parts])
            hwc = torch.cat([p[name][1] for _, p in parts])
            # per-image maps are ordered [reference copies ..., sources ...] inside a group: regroup to that order overall
            maps = []
            for k in (2, 3):
                maps.append(torch.cat([p[name][k][:n] for n, p in parts] + [p[name][k][n:] for n, p in parts]))
            out[name] = (chw, hwc, maps[0], maps[1])
        return out

    def repack(self) -> None:
        """Drop every cached folded / packed weight table.  PyTorch's own parameter rewrites (``load_state_dict``,
        ``.to()`` / ``.cuda()``, ``train()`` / ``eval()``, optimizer steps) are tracked automatically; call this after
        edits that bypass them (``p.data.copy_(...)``, EMA / SWA swaps through ``.data``)."""
        for m in self.modules():
            if isinstance(m, _PackedHolder):
                m.repack()

    def use_graphs(self, on: bool = True, check_weights: bool = True) -> "CDSMVSNet":
        """Route eval-mode ``forward`` calls through a hipGraph replay (graphed.CapturedForward: one capture per image shape /
        view count / temperature, replayed with this call's cameras through the geometry block; outputs are copies, as from the eager
        path).  Off by default; ``CDS_GRAPH=1`` in the environment turns it on for every model.  Training mode, view-sharded
        models and nn.DataParallel replicas always run eagerly."""
        if on:
            from .graphed import CapturedForward
            _GRAPH_RUNNERS[self] = CapturedForward(self, check_weights=check_weights, weak=True)
        else:
            _GRAPH_RUNNERS.pop(self, None)
        return self

    def forward(self, imgs, proj_matrices, depth_values, gt_depths=None, temperature=0.001):
        if not imgs.is_cuda:
            raise RuntimeError("cds_mvsnet_amd.CDSMVSNet runs on a ROCm device only (no CPU fallback); "
                               "move the model and inputs with .cuda()")
        if not self.training and gt_depths is None and self._view_shard is None and not getattr(self, "_is_replica", False):
            runner = _GRAPH_RUNNERS.get(self)
            if runner is None and USE_GRAPHS_DEFAULT:
                runner = self.use_graphs(True) and _GRAPH_RUNNERS[self]
            if runner is not None:
                return runner(imgs, proj_matrices, depth_values, temperature=temperature, clone=True)
        # the kernels are launched on the CURRENT device's stream: make the inputs' device current for the whole call
        # (nn.DataParallel replicas, models on cuda:k in a process whose current device is another one)
        with torch.cuda.device(imgs.device):
            return self._forward(imgs, proj_matrices, depth_values, gt_depths, temperature)

    def _forward(self, imgs, proj_matrices, depth_values, gt_depths, temperature):
        B, N, _, Him, Wim = imgs.shape
        H, W = (Him // 2, Wim // 2) if self.refine else (Him, Wim)
        if H % 32 or W % 32:
            raise ValueError("internal resolution must be a multiple of 32 (three stride-2 levels at 1/4 scale)")
        if self.training:
            from .training import forward_train   # autograd path: HIP forward / backward kernels for every stack
            return forward_train(self, imgs.float(), proj_matrices, depth_values, gt_depths, temperature)
        # host part: camera algebra -> the geometry block, one asynchronous copy; device part: launches only (capturable)
        geo = self.geometry_block(proj_matrices, depth_values, N).upload(imgs.device)
        return self.forward_device(imgs, geo, float(temperature))

    def geometry_block(self, proj_matrices, depth_values, N: int) -> "geometry.GeoBlock":
        """The HOST side of an inference forward: everything the reference derives from the cameras and the depth range inside its
        forward (projection composition + relative homographies per stage, models/model.py:40-43, warping.py:80-82; epipoles of every
        pair at the three FeatureNet resolutions, dynamic_conv.py:19-47, model.py:156-158, module.py:239,242; depth range and the
        stages' hypothesis spacings, model.py:165-175), packed into a :class:`geometry.GeoBlock`.  CPU inputs are used as they are;
        device inputs are read back (`_to_host`)."""
        keys = list(proj_matrices.keys())
        host = _to_host([depth_values] + [proj_matrices[k] for k in keys])
        dv, cams = host[0], dict(zip(keys, host[1:]))
        geo = geometry.GeoBlock()
        views = self._my_views(N - 1)
        sh = self._view_shard
        mat_views = list(range(N - 1)) if (sh is not None and sh.exchange == "slab") else views   # slab: every rank warps every view
        G = ops.MAX_IMAGES // 2
        for b in range(dv.shape[0]):
            dint = dv[b, 1] - dv[b, 0]
            geo.add(f"b{b}.range", [float(dv[b, 0]), float(dv[b, -1]), float(dint)])          # depth_min, depth_max, interval
            for s in range(self.num_stage):
                geo.add(f"b{b}.interval{s}", [float(self.depth_interals_ratio[s] * dint)])    # model.py:174
            if views:
                e_ref, e_src = geometry.pairs_epipoles(cams["stage3"][b])       # [N-1,2] each, every pair in one batched pass
            for gi, v0 in enumerate(range(0, len(views), G)):
                g = views[v0:v0 + G]
                e0 = torch.cat((e_ref[g], e_src[g]))                              # [reference copies ..., sources ...]
                geo.add(f"b{b}.epi{gi}.0", e0)               # full resolution; halved / quartered for the coarser levels
                geo.add(f"b{b}.epi{gi}.1", e0 / 2)
                geo.add(f"b{b}.epi{gi}.2", e0 / 4)
            if mat_views:
                for s in range(self.num_stage):
                    name = f"stage{s + 1}"
                    geo.add(f"b{b}.mats.{name}", geometry.warp_matrices(cams[name][b])[mat_views].contiguous())
        return geo

    def forward_device(self, imgs: Tensor, geo: "geometry.GeoBlock", T: float):
        """The DEVICE side of an inference forward: kernel launches only - no host readback, no host-side camera algebra, every
        per-call number read from `geo`'s device block.  This is the function a hipGraph capture records (graphed.py)."""
        B, N, _, Him, Wim = imgs.shape
        H, W = (Him // 2, Wim // 2) if self.refine else (Him, Wim)
        imgs = imgs.float()
        capturing = torch.cuda.is_current_stream_capturing()

        per_b: List[Dict[str, object]] = []
        for b in range(B):
            rng = geo[f"b{b}.range"]
            # ---- features, one FeatureNet pass per image of every (ref, src) pair ----
            ref_img = _resize_nearest(imgs[b, 0], H, W)
            views = self._my_views(N - 1)
            V = len(views)
            feats = None
            early: Dict[str, object] = {}
            sh = self._view_shard
            if V:
                on_stage1 = None
                if OVERLAP_STAGE1 and sh is None and V <= ops.MAX_IMAGES // 2 and (not capturing or OVERLAP_IN_CAPTURE):
                    # stage 1 (quarter resolution: kernels that do not fill 256 CUs) runs on a side stream from the moment its
                    # features exist, next to the finer FPN levels of FeatureNet on the main stream; joined before stage 2
                    main = torch.cuda.current_stream(imgs.device)
                    side = _side_stream(imgs.device)

                    def on_stage1(name, f, b=b, V=V, rng=rng):
                        if name == "stage2" and not OVERLAP_STAGE2:
                            return
                        s_ = int(name[-1]) - 1
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            scale = int(self.stage_infos[name]["scale"])
                            if s_ == 0:
                                hyp_ = ops.depth_planes(self.ndepths[0], H // scale, W // scale, rng, None, imgs.device)
                            else:
                                hyp_ = ops.depth_hypotheses(early["stage1"][0], self.ndepths[s_], H, W, scale, geo[f"b{b}.interval{s_}"], rng)
                            ref, src, nc_sum, nc_abs = f
                            early[name] = self._run_stage(ref, src, nc_abs[:V].contiguous(), ops.pair_mean(nc_sum, V),
                                                          geo[f"b{b}.mats.{name}"], hyp_, s_, N - 1)
                G = ops.MAX_IMAGES // 2
                epi_groups = [tuple(geo[f"b{b}.epi{gi}.{k}"] for k in range(3)) for gi in range((V + G - 1) // G)]
                feats = self.extract_features(ref_img, [_resize_nearest(imgs[b, v + 1], H, W) for v in views], epi_groups, T,
                                              on_stage1=on_stage1)
                if early:
                    torch.cuda.current_stream(imgs.device).wait_stream(side)
                    for st_out in early.values():
                        for t in st_out:
                            t.record_stream(torch.cuda.current_stream(imgs.device))
            if sh is not None and sh.exchange == "slab":
                # pixel-slab sharding: FeatureNet stays sharded by view; every rank then needs every view's maps for its rows
                C_s = self.feature.out_channels
                sh.set_feature_shapes({f"stage{s + 1}": ((C_s[s], H // sc, W // sc), (H // sc, W // sc, C_s[s]), (H // sc, W // sc),
                                                         (H // sc, W // sc))
                                       for s, sc in enumerate(int(self.stage_infos[f"stage{k + 1}"]["scale"]) for k in range(self.num_stage))},
                                      imgs.device)
                feats = sh.gather_features(feats, N - 1)
                views = list(range(N - 1))
                V = len(views)
            out_b: Dict[str, object] = {}
            depth = None
            for s in range(self.num_stage):
                name = f"stage{s + 1}"
                scale = int(self.stage_infos[name]["scale"])
                h, w = H // scale, W // scale
                D = self.ndepths[s]
                if name in early:
                    depth, conf, nc = early[name]
                    out_b[name] = {"depth": depth, "photometric_confidence": conf, "norm_curv": nc.unsqueeze(0)}
                    continue
                if depth is None:
                    hyp = ops.depth_planes(D, h, w, rng, None, imgs.device)
                else:
                    hyp = ops.depth_hypotheses(depth, D, H, W, scale, geo[f"b{b}.interval{s}"], rng)
                if V:
                    ref, src, nc_sum, nc_abs = feats[name]
                    ref_nc = nc_abs[:V].contiguous()
                    nc_sums = ops.pair_mean(nc_sum, V)
                    mats = geo[f"b{b}.mats.{name}"]
                else:  # a view-shard rank without a source view of its own (more GPUs than views)
                    ref = src = ref_nc = nc_sums = mats = None
                depth, conf, nc = self._run_stage(ref, src, ref_nc, nc_sums, mats, hyp, s, N - 1)
                out_b[name] = {"depth": depth, "photometric_confidence": conf, "norm_curv": nc.unsqueeze(0)}
            per_b.append(out_b)

        outputs: Dict[str, object] = {}
        for s in range(self.num_stage):
            name = f"stage{s + 1}"
            st = {k: (per_b[0][name][k].unsqueeze(0) if B == 1 else torch.stack([pb[name][k] for pb in per_b]))
                  for k in ("depth", "photometric_confidence", "norm_curv")}   # B = 1: a view, no copy launch
            outputs[name] = st
            outputs.update(st)
        depth = outputs["depth"]
        if self.refine:
            # models/model.py:213-218: depth and limits in interval units in, refined depth times the interval out - the divisions and the
            # product are inside the first / last Refinement kernel (device scalars of the geometry block)
            refined = self.refine_network(imgs[:, 0], depth.unsqueeze(1), [geo[f"b{b}.range"] for b in range(B)], None)
            outputs["refined_depth"] = refined[:, 0]
        else:
            outputs["refined_depth"] = depth
        return outputs

    # -- view sharding --------------------------------------------------------
    def _my_views(self, V: int) -> List[int]:
        sh = self._view_shard
        return list(range(V)) if sh is None else sh.local_views(V)

    def _run_stage(self, ref, src, ref_nc, nc_sums, mats, hyp, s, V_total):
        sh = self._view_shard
        if sh is None:
            # FeatureNet's stage outputs are tanh outputs: |feature| < 1, so 1 bounds the normalised volume (no reduction launches)
            return self.stage_net.run_single(ref, src, ref_nc, nc_sums, mats, hyp, self.cost_regularization[s], s,
                                             vol_bound=_unit_bound(ref.device))
        return sh.run_stage(self, ref, src, ref_nc, nc_sums, mats, hyp, s, V_total, C=self.feature.out_channels[s],
                            vol_bound=_unit_bound(hyp.device))


def _resize_nearest(img: Tensor, H: int, W: int) -> Tensor:
    """F.interpolate(img, (H, W)) with the default nearest mode (model.py:159-160): identity when the size
    already matches, plain subsampling for the refine=True half-resolution case."""
    _, h, w = img.shape
    if (h, w) == (H, W):
        return img.contiguous()
    if h == 2 * H and w == 2 * W:
        return img[:, ::2, ::2].contiguous()
    return F.interpolate(img.unsqueeze(0), (H, W))[0].contiguous()

"""Training-mode forward of CDSMVSNet (reference: models/model.py:52-56,63-69,140-223 with ``self.training``).

SURVEY §8(f)-2.  Every stack runs on the hand-written HIP kernels in BOTH directions: the fused homography warp + visibility-weighted
aggregation (``ops.WarpAggregate``: K3 forward, scatter-add backward), CostRegNet with batch-statistics BatchNorm (``train_ops.py``),
and the 2D stacks - FeatureNet / DynamicConv, visibility CNN, Refinement, soft-argmin (``train2d_ops.py``); the no-gradient pieces
(K1 entropy, hypothesis generation, confidence) use the inference kernels.  The parameter-holder modules are the inference ones, so
``state_dict`` / optimisers / checkpoints are unchanged.  There is no other backend: CPU tensors raise (the torch-op restatement that
serves as the float64 reference of the gradient tests is test infrastructure, tests/torch_training_ref.py).  Gradient topology follows the reference: the
sampling grid has no gradient, the entropy input of the visibility CNN is detached, depth is detached between stages.
"""
from __future__ import annotations

import contextlib
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import _scratch, geometry, ops

Tensor = torch.Tensor


def _need_hip(x: Tensor) -> None:
    if not x.is_cuda:
        raise RuntimeError("the training step runs on the HIP kernels only: CPU tensors are not supported (the torch-op restatement "
                           "used as the float64 reference of the gradient tests lives in tests/torch_training_ref.py)")


def _stacked_operands() -> bool:
    """The K1 / K3 operands and the visibility CNN's input are read straight from the stacked FeatureNet outputs, and the V
    visibility calls of a stage run as ONE grouped call.  (Patched to False by tests/torch_training_ref.py, whose per-call torch
    reference follows the reference's own call structure.)"""
    return True


def _dyn(dc, x: Tensor, epi: Tensor, T: float, groups: int = 1) -> Tuple[Tensor, Tensor]:
    """One DynamicConv (models/dynamic_conv.py:97-122) on the HIP forward / backward kernels.  groups > 1: the batch stacks that many
    separate calls of the reference; the BatchNorm2d of the attention MLP takes its statistics per group."""
    _need_hip(x)
    from . import train2d_ops
    return train2d_ops.dynamic_conv(dc, x, epi, T, groups)


def _in_act(y: Tensor, tanh: bool = False) -> Tensor:
    """InstanceNorm2d + LeakyReLU(0.1) (module.py:66-69) or + tanh (module.py:223)."""
    _need_hip(y)
    from . import train2d_ops
    return train2d_ops.instnorm_act(y, ops.ACT_TANH if tanh else ops.ACT_LEAKY01)


def _conv(conv, x: Tensor) -> Tensor:
    """A plain nn.Conv2d holder (3x3 stride 1 | 2, 1x1) on the HIP training kernels."""
    _need_hip(x)
    from . import train2d_ops
    return train2d_ops.Conv2d.apply(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], train2d_ops.stored(x))


def _cat_up(coarse: Tensor, skip: Tensor) -> Tensor:
    """cat(nearest2x(coarse), skip) of an FPN lateral (module.py:253-254,260-261); under the bf16 storage policy the stored twins
    (train2d_ops.stored) are concatenated alongside, so the lateral's weight gradient reads bf16."""
    from . import train2d_ops
    return train2d_ops.cat_stored((train2d_ops.upsample2_stored(coarse), skip))


def _unit(unit, x: Tensor, epi: Optional[Tensor], T: float, groups: int = 1):
    if unit.dynamic:
        y, nc = _dyn(unit.conv, x, epi, T, groups)
        return _in_act(y), nc
    return _in_act(_conv(unit.conv, x))


def _curv(a: Tensor, b: Tensor, c: Tensor) -> Tuple[Tensor, Tensor]:
    """((a^2 + b^2) + c^2) / 3 and |c| (module.py:250-251), HIP forward / backward."""
    _need_hip(a)
    from . import train2d_ops
    return train2d_ops.CurvatureStats.apply(a, b, c)


def feature_net(net, x: Tensor, epi: Tensor, T: float, groups: int = 1) -> Dict[str, Tuple[Tensor, Tensor, Tensor]]:
    """models/module.py:234-267.  x [N,3,H,W], epi [N,2] -> {'stageK': (fea, nc_sum, |nc|)} batched over N.  groups > 1: x stacks
    that many separate calls of the reference (the only cross-sample operation in FeatureNet is the BatchNorm2d inside each
    DynamicConv's attention MLP: its statistics are then taken per group)."""
    e0, e1, e2 = epi, epi / 2, epi / 4
    c00, n00 = _unit(net.conv00, x, e0, T, groups)
    c01, n01 = _unit(net.conv01, c00, e0, T, groups)
    d0 = _unit(net.downsample1, c01, None, T)
    c10, n10 = _unit(net.conv10, d0, e1, T, groups)
    c11, n11 = _unit(net.conv11, c10, e1, T, groups)
    d1 = _unit(net.downsample2, c11, None, T)
    c20, n20 = _unit(net.conv20, d1, e2, T, groups)
    c21, n21 = _unit(net.conv21, c20, e2, T, groups)
    out = {}
    o1, n22 = _dyn(net.out1, c21, e2, T, groups)
    out["stage1"] = (_in_act(o1, tanh=True), *_curv(n20, n21, n22))
    t = _unit(net.inner1, _cat_up(c21, c11), None, T)
    o2, n12 = _dyn(net.out2, t, e1, T, groups)
    o2 = _in_act(o2, tanh=True)
    out["stage2"] = (o2, *_curv(n10, n11, n12))
    t = _unit(net.inner2, _cat_up(o2, c01), None, T)
    o3, n02 = _dyn(net.out3, t, e0, T, groups)
    out["stage3"] = (_in_act(o3, tanh=True), *_curv(n00, n01, n02))
    return out


BATCH_FEATURES = os.environ.get("CDS_TRAIN_BATCH_FEATURES", "1") != "0"   # 0 = one FeatureNet call per image of every pair, as the reference


def cost_regularization(cr, x: Tensor) -> Tensor:
    """models/module.py:305-315 with BatchNorm in batch-statistics mode.  x [B,C,D,h,w] -> [B,1,D,h,w] on the hand-written HIP
    training ops (train_ops.py: convolution forward / data gradient / weight gradient and fused BatchNorm(train) + ReLU + skip)."""
    _need_hip(x)
    from . import train_ops
    return train_ops.cost_regularization(cr, x)


def _softargmin(prob_pre: Tensor, hyp: Tensor) -> Tensor:
    """softmax over D + depth regression (models/module.py:373-379), HIP forward / backward."""
    _need_hip(prob_pre)
    from . import train2d_ops
    return train2d_ops.SoftArgmin.apply(prob_pre, hyp)


def _refinement(net, img: Tensor, depth0: Tensor, dmin: Tensor, dmax: Tensor) -> Tensor:
    """Refinement (models/module.py:318-370) in training mode, HIP forward / backward."""
    _need_hip(img)
    from . import train2d_ops
    return train2d_ops.refinement(net, img, depth0, dmin, dmax)


def _cbr2(unit, x: Tensor, groups: int = 1) -> Tensor:
    """ConvBn2d holder: Conv2d 3x3 -> BatchNorm2d (module mode) -> ReLU.  groups > 1: the batch stacks that many calls."""
    from . import train2d_ops
    return train2d_ops.bn_relu2d(unit.bn, _conv(unit.conv, x), True, groups)


def _visibility(seq, x: Tensor, groups: int = 1) -> Tensor:
    """models/model.py:14.  groups > 1: x stacks that many calls (one per source view) along the batch axis, group-major."""
    for i in range(3):
        x = _cbr2(seq[i], x, groups)
    return torch.sigmoid(_conv(seq[3], x))


class StackedFeatures:
    """The FeatureNet outputs of one stage for all 2 V B images of a stacked call, in the reference's call order (pair v: its reference
    image for every batch item, then its source image): fea [2 V B,C,h,w], nc_sum / nc [2 V B,1,h,w].  `stage_forward_train` reads the
    K1 / K3 operands, the visibility CNN's curvature input and the curvature regulariser straight from these tensors: slicing them per
    (pair, image) and stacking the slices again costs a full-size zero fill + copy + add per slice in the backward (~250 launches and
    2 ms of the config-5 step)."""

    def __init__(self, fea: Tensor, nc_sum: Tensor, nc: Tensor, V: int, B: int):
        self.fea, self.nc_sum, self.nc, self.V, self.B = fea, nc_sum, nc, V, B

    def __len__(self) -> int:
        return self.V

    def as_list(self):
        """The reference's argument layout (list over source views of {'ref': ..., 'src': ...})."""
        B = self.B
        ts = (self.fea, self.nc_sum, self.nc)
        return [{"ref": tuple(t[2 * v * B:(2 * v + 1) * B] for t in ts), "src": tuple(t[(2 * v + 1) * B:(2 * v + 2) * B] for t in ts)}
                for v in range(self.V)]


class _PairSplit(torch.autograd.Function):
    """fea [2 V B,C,h,w] (pair-major: ref images of pair v, then its src images) -> for every batch item b the K1 / K3 operands
    ref_b [V,C,h,w] and src_b [V,h,w,C] (channels-last), 2 B copies; the backward writes every gradient into ONE dense tensor."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, fea, V: int, B: int):
        f = fea.view(V, 2, B, *fea.shape[1:])
        ctx.cfg = (V, B, tuple(fea.shape))
        ctx.set_materialize_grads(False)
        out = []
        for b in range(B):
            out.append(f[:, 0, b].contiguous())
            out.append(f[:, 1, b].permute(0, 2, 3, 1).contiguous())
        return tuple(out)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, *grads):
        V, B, shape = ctx.cfg
        ref = next(g for g in grads if g is not None)
        gf = torch.empty(shape, dtype=torch.float32, device=ref.device).view(V, 2, B, *shape[1:])
        for b in range(B):
            g_ref, g_src = grads[2 * b], grads[2 * b + 1]
            if g_ref is None:
                gf[:, 0, b].zero_()
            else:
                gf[:, 0, b].copy_(g_ref)
            if g_src is None:
                gf[:, 1, b].zero_()
            else:
                gf[:, 1, b].copy_(g_src.permute(0, 3, 1, 2))
        return gf.view(shape), None, None


def _stack(ts: List[Tensor]) -> Tensor:
    """torch.stack along a new batch axis; a batch of one is a view, not a copy."""
    return ts[0].unsqueeze(0) if len(ts) == 1 else torch.stack(ts)


def stage_forward_train(stage_net, features, cams: Optional[Tensor], depth_values: Tensor, cost_reg, stage_idx: int,
                        gt_depth: Optional[Tensor] = None, mats: Optional[List[Tensor]] = None) -> Dict[str, Tensor]:
    """``StageNet.forward`` in training mode (models/model.py:16-94 with ``self.training``), reference argument layout:
    features = list over source views of {'ref': (fea [B,C,h,w], nc_sum [B,1,h,w], nc [B,1,h,w]), 'src': (fea, nc_sum, _)};
    cams [B,N,2,4,4] (host) - or `mats`, the per-item homographies [V,12] as device slices of the step's geometry block (cams then
    unused); depth_values [B,D,h,w].  Returns depth / photometric_confidence / feat_distance / norm_curv.
    K1 (detached, model.py:49), hypotheses and confidence run on the HIP kernels without gradient; K3 forward / backward and
    CostRegNet, the visibility CNN and the soft-argmin forward / backward on the HIP kernels with gradient."""
    stacked = features if isinstance(features, StackedFeatures) else None
    if stacked is not None and not _stacked_operands():
        features, stacked = stacked.as_list(), None
    V = len(features)
    B = depth_values.shape[0]
    if gt_depth is not None and gt_depth.dim() == 4:                    # the reference passes gt_depths[stage].unsqueeze(1): [B,1,h,w]
        if gt_depth.shape[1] != 1:
            raise ValueError(f"gt_depth must be [B,h,w] or [B,1,h,w], got {tuple(gt_depth.shape)}")
        gt_depth = gt_depth[:, 0]
    if mats is None:                                                     # reference signature: cameras in, one upload per batch item
        cams = cams.detach().float().cpu()
        mats = [ops.geo(geometry.warp_matrices(cams[b]), depth_values.device, "mats") for b in range(B)]
    hyps = [depth_values[b].detach().float().contiguous() for b in range(B)]
    # .float(): under bf16 autocast the convolution stacks hand over bf16 activations; the HIP kernels are fp32
    if stacked is not None:
        parts = _PairSplit.apply(stacked.fea, V, B)
        ref, src = [parts[2 * b] for b in range(B)], [parts[2 * b + 1] for b in range(B)]
        nc6 = stacked.nc.float().view(V, 2, B, *stacked.nc.shape[2:])                                     # [V,2,B,h,w]
    else:
        ref = [torch.stack([features[v]["ref"][0][b] for v in range(V)]).float() for b in range(B)]         # [V,C,h,w]
        src = [torch.stack([features[v]["src"][0][b] for v in range(V)]).float().permute(0, 2, 3, 1).contiguous() for b in range(B)]
    with torch.no_grad():                                               # K1, detached input (model.py:49)
        ent = torch.stack([ops.warp_entropy(ref[b].detach().contiguous(), src[b].detach(), mats[b], hyps[b])
                           for b in range(B)])                           # [B,V,h,w]
    if _stacked_operands() and V > 1:
        # the V calls of model.py:51 as ONE call on the views stacked along the batch axis (BatchNorm statistics per view)
        nc_ref = nc6[:, 0] if stacked is not None else torch.stack([features[v]["ref"][2].float()[:, 0] for v in range(V)])   # [V,B,h,w]
        x = torch.stack((ent.transpose(0, 1), nc_ref), dim=2).reshape(V * B, 2, ent.shape[-2], ent.shape[-1])
        vis_all = _visibility(stage_net.vis[stage_idx], x, groups=V).view(V, B, ent.shape[-2], ent.shape[-1])
        vis = [vis_all[v] for v in range(V)]
    else:
        flist = stacked.as_list() if stacked is not None else features
        vis = [_visibility(stage_net.vis[stage_idx], torch.cat((ent[:, v:v + 1], flist[v]["ref"][2].float()), dim=1))[:, 0]
               for v in range(V)]                                        # V x [B,h,w]  (model.py:51)
    vols, fds = [], []
    for b in range(B):
        vis_b = torch.stack([vis[v][b] for v in range(V)]).float()       # [V,h,w]
        vol_sum = ops.WarpAggregate.apply(ref[b], src[b], vis_b, mats[b], hyps[b])       # K3 fwd / bwd kernels
        gt_sum = None
        if gt_depth is not None:                                         # model.py:63-69,76-78
            gt_sum = ops.WarpAggregate.apply(ref[b], src[b], vis_b, mats[b], gt_depth[b:b + 1].float().contiguous())
        # volume / (sum_v vis + 1e-6) (model.py:74) and the feature distances sum_v sim_v vis_v / vis_sum (model.py:56,75-78): one launch
        vol, fd = ops.VolumeFinish.apply(vol_sum, gt_sum, vis_b)
        vols.append(vol)
        fds.append(fd)
    if stacked is not None:                                                                         # sum_v ((ref + src) / 2) / V
        nc_mean = stacked.nc_sum.view(V, 2, B, *stacked.nc_sum.shape[1:]).mean(dim=(0, 1))         # [B,1,h,w]
    else:
        nc_mean = sum((features[v]["ref"][1] + features[v]["src"][1]) / 2 for v in range(V)) / V   # [B,1,h,w]
    hyp_b = _stack(hyps)
    prob_pre = cost_regularization(cost_reg, _stack(vols)).squeeze(1).float()
    depth = _softargmin(prob_pre, hyp_b)
    with torch.no_grad():
        conf = torch.stack([ops.softargmin_conf(prob_pre[b].detach().contiguous(), hyp_b[b])[1] for b in range(B)])
    return {"depth": depth, "photometric_confidence": conf, "feat_distance": _stack(fds), "norm_curv": nc_mean}


STAGE_STREAMS = os.environ.get("CDS_TRAIN_STAGE_STREAMS", "0") == "1"   # experiment: one stream per cascade stage (see forward_train)
_STAGE_STREAMS: Dict[Tuple[int, int], "torch.cuda.Stream"] = {}


def _stage_stream(dev, s: int) -> "torch.cuda.Stream":
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), s)
    if key not in _STAGE_STREAMS:
        _STAGE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _STAGE_STREAMS[key]


_PAIR_ORDER: Dict[Tuple[int, str], Tensor] = {}


def _pair_order(V: int, dev) -> Tensor:
    """View indices of the stacked FeatureNet call (ref, src_1, ref, src_2, ...), cached on the device."""
    key = (V, str(dev))
    if key not in _PAIR_ORDER:
        _PAIR_ORDER[key] = torch.tensor([n for v in range(V) for n in (0, v + 1)], device=dev)
    return _PAIR_ORDER[key]


def train_geometry(model, proj_matrices: Dict[str, Tensor], depth_values: Tensor, N: int) -> "geometry.GeoBlock":
    """The HOST side of a training forward (the counterpart of CDSMVSNet.geometry_block): epipoles of every pair and batch item in
    the orders FeatureNet's calls want them, homographies per stage and item, depth range / interval / per-stage spacing - one
    geometry block, uploaded with one asynchronous copy; `forward_train(..., geo=)` then only launches kernels, which is what the
    captured training step (train.CapturedTrainStep) records.  Device inputs are read back here (a synchronisation: keep the cameras
    and depth values on the host, as the data loader delivers them, to let the host run ahead of the GPU)."""
    from .model import _to_host
    keys = list(proj_matrices.keys())
    host = _to_host([depth_values] + [proj_matrices[k] for k in keys])
    dv, cams = host[0], dict(zip(keys, host[1:]))
    B, V = dv.shape[0], N - 1
    geo = geometry.GeoBlock()
    epi = [geometry.pairs_epipoles(cams["stage3"][b]) for b in range(B)]            # per item: ([V,2] in the reference, [V,2] in the sources)
    # one stacked FeatureNet call: images ordered (pair v: reference, source) x batch (forward_train, BATCH_FEATURES)
    geo.add("epi.all", torch.stack([epi[b][k][v] for v in range(V) for k in (0, 1) for b in range(B)]))
    for v in range(V):                                                                # the reference's 2 V separate calls
        geo.add(f"epi.ref{v}", torch.stack([epi[b][0][v] for b in range(B)]))
        geo.add(f"epi.src{v}", torch.stack([epi[b][1][v] for b in range(B)]))
    dint = dv[:, 1] - dv[:, 0]
    geo.add("dmin", dv[:, 0])
    geo.add("dmax", dv[:, -1])
    geo.add("dint", dint)
    for b in range(B):
        geo.add(f"b{b}.range", [float(dv[b, 0]), float(dv[b, -1]), float(dint[b])])
        for s in range(model.num_stage):
            geo.add(f"b{b}.interval{s}", [float(model.depth_interals_ratio[s] * dint[b])])
            geo.add(f"b{b}.mats.stage{s + 1}", geometry.warp_matrices(cams[f"stage{s + 1}"][b]))
    return geo


def forward_train(model, imgs: Tensor, proj_matrices: Optional[Dict[str, Tensor]], depth_values: Optional[Tensor],
                  gt_depths: Optional[Dict[str, Tensor]], temperature: float, geo: Optional["geometry.GeoBlock"] = None):
    """CDSMVSNet.forward in training mode.  Same inputs / outputs as the reference (adds 'feat_distance' and
    'feat_target' per stage).  Module calls are grouped exactly like the reference's (FeatureNet once per image of
    every pair over the batch, the visibility CNN once per source view over the batch, CostRegNet once per stage over the
    batch) so the batch statistics of every BatchNorm layer are taken over the same sets.
    geo: the step's geometry block (:func:`train_geometry`, uploaded); None = built from proj_matrices / depth_values here."""
    B, N, _, Him, Wim = imgs.shape
    H, W = (Him // 2, Wim // 2) if model.refine else (Him, Wim)
    T = float(temperature)
    dev = imgs.device
    if geo is None:
        geo = train_geometry(model, proj_matrices, depth_values, N).upload(dev)
    _scratch.begin_step(dev)                                             # zero arena + deferred counters of this step's HIP training ops
    V = N - 1
    # model.py:154-161 calls FeatureNet once per image of every pair.  Its InstanceNorms are per sample and the BatchNorm2d of
    # each DynamicConv's attention MLP is evaluated per group of B samples (_att_weights_grouped), so the 2 V calls are ONE call
    # on the 2 V B images stacked along the batch axis: same values, an eighth of the kernel launches and of the per-parameter
    # gradient accumulations (the step is launch-bound).
    feats = []
    if BATCH_FEATURES:
        # one call on the 2 V B images, stacked in the reference's call order (ref of pair 0, src of pair 0, ref of pair 1, ...)
        e_all = geo["epi.all"]
        # all N views resized in one call, then gathered in the reference's call order (ref, src_1, ref, src_2, ...)
        small = imgs if (Him, Wim) == (H, W) else F.interpolate(imgs.reshape(B * N, 3, Him, Wim), (H, W)).view(B, N, 3, H, W)
        order = _pair_order(V, dev)
        x_all = small.index_select(1, order).transpose(0, 1).reshape(2 * V * B, 3, H, W)
        f_all = feature_net(model.feature, x_all, e_all, T, groups=2 * V)
        stacked_feats = {k: StackedFeatures(*f_all[k], V, B) for k in f_all}
    else:
        ref_img = F.interpolate(imgs[:, 0], (H, W))
        for v in range(V):
            e_ref, e_src = geo[f"epi.ref{v}"], geo[f"epi.src{v}"]
            feats.append((feature_net(model.feature, ref_img, e_ref, T),
                          feature_net(model.feature, F.interpolate(imgs[:, v + 1], (H, W)), e_src, T)))
    outputs: Dict[str, object] = {}
    depth = None
    dint_dev = geo["dint"]
    # One stream per stage (CDS_TRAIN_STAGE_STREAMS=1, experiment): the forward stays serial (a stage's hypotheses need the previous
    # stage's depth) but depth is DETACHED between stages, so the three backward chains are independent and autograd runs each on the
    # stream of its forward - next to each other.  Every tensor that crosses streams is recorded on the other stream (caching allocator).
    use_streams = STAGE_STREAMS and BATCH_FEATURES and _stacked_operands()
    main = torch.cuda.current_stream(dev) if use_streams else None
    prev_stream = None
    for s in range(model.num_stage):
        name = f"stage{s + 1}"
        scale = int(model.stage_infos[name]["scale"])
        h, w = H // scale, W // scale
        D = model.ndepths[s]
        st_s = None
        if use_streams:
            st_s = _stage_stream(dev, s)
            st_s.wait_stream(main)                                           # FeatureNet outputs, the zero arena
            if prev_stream is not None:
                st_s.wait_stream(prev_stream)                                # the previous stage's depth
        with (torch.cuda.stream(st_s) if use_streams else contextlib.nullcontext()):
            hyps = []
            for b in range(B):
                if depth is None:
                    hyps.append(ops.depth_planes(D, h, w, geo[f"b{b}.range"], None, dev))
                else:
                    hyps.append(ops.depth_hypotheses(depth[b].detach().contiguous(), D, H, W, scale, geo[f"b{b}.interval{s}"],
                                                     geo[f"b{b}.range"]))
            features = stacked_feats[name] if BATCH_FEATURES else [{"ref": feats[v][0][name], "src": feats[v][1][name]} for v in range(V)]
            hyp_b = torch.stack(hyps)
            if use_streams:
                for t in (features.fea, features.nc_sum, features.nc):
                    t.record_stream(st_s)
                if gt_depths is not None:
                    gt_depths[name].record_stream(st_s)
                if depth is not None:
                    depth.record_stream(st_s)
            st = stage_forward_train(model.stage_net, features, None, hyp_b, model.cost_regularization[s], s,
                                     gt_depth=gt_depths[name] if gt_depths is not None else None,
                                     mats=[geo[f"b{b}.mats.{name}"] for b in range(B)])
            depth = st["depth"]
            if gt_depths is not None:                                            # model.py:202-207
                st["feat_target"] = ops.feat_target(hyp_b, gt_depths[name], dint_dev, float(scale), 0.5 / float(scale))
            if use_streams:
                for t in st.values():
                    if isinstance(t, torch.Tensor):
                        t.record_stream(main)
        prev_stream = st_s
        outputs[name] = st
        outputs.update(st)
    if use_streams:
        for s in range(model.num_stage):
            main.wait_stream(_stage_stream(dev, s))
    if model.refine:
        dint = dint_dev.view(B, 1, 1)
        refined = model.refine_network(imgs[:, 0], (depth.detach() / dint).unsqueeze(1), geo["dmin"] / dint_dev, geo["dmax"] / dint_dev)
        outputs["refined_depth"] = refined.squeeze(1) * dint
    else:
        outputs["refined_depth"] = depth
    _scratch.flush_counters(dev)
    return outputs

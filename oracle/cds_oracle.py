"""CPU oracle for the CDS-MVSNet plane-sweep hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cds_mvsnet_amd/`` may import this
module; it is used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` as the *checker*, never as the thing that
is shipped or measured.

What it is: a functional, state-dict driven restatement (plain PyTorch fp32 on
CPU) of the algorithm the reference implements in
``/root/reference/models/{model,module,dynamic_conv}.py`` and
``models/utils/warping.py``.  Every function cites the reference lines it
follows.  Parity status: PINNED — ``tests/golden/*.npz`` were produced by
importing the reference itself in the build container
(``tests/golden/make_golden.py``) and ``tests/test_oracle_golden.py`` checks
this file against them.

The reference is floating point throughout, so the oracle keeps torch fp32
ops for convolutions / norms (the same ATen kernels the reference calls) and
restates everything else explicitly.  The homography warp has two modes:

* ``exact=True``  – explicit bilinear gather with the fp32 operation order of
  ATen's CPU ``grid_sample`` (measured bit-identical on 2e5 random samples:
  ``o = v_nw*nw; o = fma(v_ne,ne,o); o = fma(v_sw,sw,o); o = fma(v_se,se,o)``);
  fp32 FMA is emulated through float64.  This is the op order the HIP kernels
  follow.
* ``exact=False`` – ``F.grid_sample`` (what the reference calls); used for the
  timed CPU baseline.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5
IN_EPS = 1e-5


# ----------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------
def _fma32(a: Tensor, b: Tensor, c: Tensor) -> Tensor:
    """fp32 fused multiply-add emulated through float64 (24+24 < 53 bits)."""
    return (a.double() * b.double() + c.double()).float()


def _bn_eval(x: Tensor, sd: Dict[str, Tensor], key: str) -> Tensor:
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"],
                        sd[key + ".weight"], sd[key + ".bias"], False, 0.0, BN_EPS)


# ----------------------------------------------------------------------------
# a2  projection composition                      models/model.py:40-43
# ----------------------------------------------------------------------------
def compose_projection(cam: Tensor) -> Tensor:
    """cam [B,2,4,4] (0: extrinsic 4x4, 1: intrinsic in [:3,:3]) -> P [B,4,4].

    P = extr with the top 3x4 block replaced by K @ extr[:3,:4]; the bottom row
    of the extrinsic is kept (models/model.py:40-43)."""
    P = cam[:, 0].clone()
    P[:, :3, :4] = torch.matmul(cam[:, 1, :3, :3], cam[:, 0, :3, :4])
    return P


def relative_projection(P_src: Tensor, P_ref: Tensor) -> Tensor:
    """M = P_src @ inverse(P_ref)  [B,4,4]     (models/utils/warping.py:80)."""
    return torch.matmul(P_src, torch.inverse(P_ref))


# ----------------------------------------------------------------------------
# a1  homography warp                             models/utils/warping.py:69-104
# ----------------------------------------------------------------------------
def sample_positions(M: Tensor, hyp: Tensor, h: int, w: int) -> Tuple[Tensor, Tensor]:
    """Source-image sample positions (pixels) for every reference voxel.

    M [B,4,4]; hyp [B,D,h,w] (or [B,D]).  Returns ix, iy  [B,D,h*w] in the
    exact fp32 operation order of warping.py:81-97 followed by ATen's
    align_corners=True un-normalisation ``(g+1)*((size-1)/2)``:
      r   = fma-chain(R[:,0]*x, R[:,1]*y, R[:,2]*1)        (sgemm, K=3, k-ordered)
      p   = r*d  (rounded)  + t (rounded)                  (warping.py:90-92)
      u,v = p.x/(p.z+1e-6), p.y/(p.z+1e-6)                 (warping.py:94, true division)
      g   = u/((w-1)/2) - 1                                (warping.py:95-96, true division)
      ix  = (g+1)*((w-1)/2)
    """
    B = M.shape[0]
    D = hyp.shape[1]
    R = M[:, :3, :3]
    t = M[:, :3, 3]
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32),
                            torch.arange(w, dtype=torch.float32), indexing="ij")
    xs = xs.reshape(1, 1, -1)
    ys = ys.reshape(1, 1, -1)
    r = R[:, :, 0:1] * xs                                   # [B,3,hw]
    r = _fma32(R[:, :, 1:2].expand(-1, -1, h * w), ys.expand(B, 3, -1), r)
    r = _fma32(R[:, :, 2:3].expand(-1, -1, h * w), torch.ones_like(r), r)
    if hyp.dim() == 2:
        d = hyp.view(B, 1, D, 1).expand(B, 1, D, h * w)
    else:
        d = hyp.reshape(B, 1, D, h * w)
    p = r.unsqueeze(2) * d                                  # [B,3,D,hw] rounded product
    p = p + t.view(B, 3, 1, 1)                              # rounded sum (no fma)
    z = p[:, 2] + 1e-6
    u = p[:, 0] / z
    v = p[:, 1] / z
    half_w = (w - 1) / 2
    half_h = (h - 1) / 2
    gx = u / half_w - 1
    gy = v / half_h - 1
    ix = (gx + 1) * half_w
    iy = (gy + 1) * half_h
    return ix, iy


def bilinear_gather(src: Tensor, ix: Tensor, iy: Tensor) -> Tensor:
    """Zero-padded bilinear sampling at pixel positions (ATen grid_sample
    semantics, align_corners=True; warping.py:100-101).

    src [B,C,h,w]; ix,iy [B,D,hw] -> [B,C,D,hw].  Each tap whose integer
    coordinate falls outside the image contributes 0 individually."""
    B, C, h, w = src.shape
    D = ix.shape[1]
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    wx = ix - x0
    ex = 1 - wx
    ny = iy - y0
    sy = 1 - ny
    w_nw, w_ne, w_sw, w_se = sy * ex, sy * wx, ny * ex, ny * wx
    flat = src.reshape(B, C, h * w)

    def tap(xi: Tensor, yi: Tensor) -> Tensor:
        ok = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1)
        lin = (yi.clamp(0, h - 1).long() * w + xi.clamp(0, w - 1).long())
        lin = torch.where(ok, lin, torch.zeros_like(lin))
        val = torch.gather(flat, 2, lin.reshape(B, 1, -1).expand(B, C, -1)).reshape(B, C, D, -1)
        return val * ok.unsqueeze(1)

    o = tap(x0, y0) * w_nw.unsqueeze(1)
    o = _fma32(tap(x0 + 1, y0), w_ne.unsqueeze(1).expand_as(o), o)
    o = _fma32(tap(x0, y0 + 1), w_sw.unsqueeze(1).expand_as(o), o)
    o = _fma32(tap(x0 + 1, y0 + 1), w_se.unsqueeze(1).expand_as(o), o)
    return o


def warp_volume(src: Tensor, P_src: Tensor, P_ref: Tensor, hyp: Tensor, exact: bool = True) -> Tensor:
    """Warped source feature volume [B,C,D,h,w]  (warping.py:69-104)."""
    B, C, h, w = src.shape
    D = hyp.shape[1]
    M = relative_projection(P_src, P_ref)
    if exact:
        ix, iy = sample_positions(M, hyp, h, w)
        return bilinear_gather(src, ix, iy).reshape(B, C, D, h, w)
    # fast path: the op the reference calls.
    R, t = M[:, :3, :3], M[:, :3, 3:4]
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32),
                            torch.arange(w, dtype=torch.float32), indexing="ij")
    pix = torch.stack((xs.reshape(-1), ys.reshape(-1), torch.ones(h * w))).unsqueeze(0).expand(B, -1, -1)
    r = torch.matmul(R, pix)
    d = hyp.view(B, 1, D, 1) if hyp.dim() == 2 else hyp.reshape(B, 1, D, h * w)
    p = r.unsqueeze(2) * d + t.view(B, 3, 1, 1)
    uv = p[:, :2] / (p[:, 2:3] + 1e-6)
    gx = uv[:, 0] / ((w - 1) / 2) - 1
    gy = uv[:, 1] / ((h - 1) / 2) - 1
    grid = torch.stack((gx, gy), dim=3).view(B, D * h, w, 2)
    out = F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    return out.view(B, C, D, h, w)


# ----------------------------------------------------------------------------
# a3  correlation + entropy                       models/model.py:46-50
# ----------------------------------------------------------------------------
def correlation_entropy(ref_fea: Tensor, warped: Tensor) -> Tuple[Tensor, Tensor]:
    """in_prod = ref (x) warped [B,C,D,h,w]; entropy over D of softmax(sum_C in_prod)
    -> (in_prod, entropy [B,1,h,w])."""
    in_prod = ref_fea.unsqueeze(2) * warped
    sim = in_prod.sum(dim=1)
    p = F.softmax(sim, dim=1)
    entropy = (-p * torch.log(p)).sum(dim=1, keepdim=True)
    return in_prod, entropy


# ----------------------------------------------------------------------------
# a4  visibility CNN                              models/model.py:14,51 ; module.py:169-198
# ----------------------------------------------------------------------------
def vis_cnn(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """x [B,2,h,w] = cat(entropy, |curvature|) -> visibility weight [B,1,h,w] in (0,1).
    prefix e.g. 'stage_net.vis.0'."""
    for i in range(3):
        x = F.conv2d(x, sd[f"{prefix}.{i}.conv.weight"], None, padding=1)
        x = F.relu(_bn_eval(x, sd, f"{prefix}.{i}.bn"))
    x = F.conv2d(x, sd[f"{prefix}.3.weight"], sd[f"{prefix}.3.bias"])
    return torch.sigmoid(x)


# ----------------------------------------------------------------------------
# a6  CostRegNet                                  models/module.py:270-315
# ----------------------------------------------------------------------------
def _c3(x, sd, key, stride=1):
    x = F.conv3d(x, sd[key + ".conv.weight"], None, stride=stride, padding=1)
    return F.relu(_bn_eval(x, sd, key + ".bn"))


def _d3(x, sd, key):
    x = F.conv_transpose3d(x, sd[key + ".conv.weight"], None, stride=2, padding=1, output_padding=1)
    return F.relu(_bn_eval(x, sd, key + ".bn"))


def cost_regularization(vol: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """3D U-Net [B,C,D,h,w] -> [B,1,D,h,w]  (module.py:305-315); eval-mode BN."""
    c0 = _c3(vol, sd, prefix + ".conv0")
    c2 = _c3(_c3(c0, sd, prefix + ".conv1", 2), sd, prefix + ".conv2")
    c4 = _c3(_c3(c2, sd, prefix + ".conv3", 2), sd, prefix + ".conv4")
    x = _c3(_c3(c4, sd, prefix + ".conv5", 2), sd, prefix + ".conv6")
    x = c4 + _d3(x, sd, prefix + ".conv7")
    x = c2 + _d3(x, sd, prefix + ".conv9")
    x = c0 + _d3(x, sd, prefix + ".conv11")
    return F.conv3d(x, sd[prefix + ".prob.weight"], None, padding=1)


# ----------------------------------------------------------------------------
# a7 / a8  soft-argmin depth + confidence         models/module.py:373-391
# ----------------------------------------------------------------------------
def softargmin(prob_pre: Tensor, hyp: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """prob_pre [B,D,h,w] -> (prob, depth [B,h,w], confidence [B,h,w]).

    confidence = sum of prob over the window [i-1, i+2] (zero outside) at
    i = clamp(trunc(sum_d prob*d_index), 0, D-1)   (module.py:382-391)."""
    B, D = prob_pre.shape[:2]
    prob = F.softmax(prob_pre, dim=1)
    hv = hyp.view(B, D, 1, 1) if hyp.dim() == 2 else hyp
    depth = torch.sum(prob * hv, dim=1)
    padded = F.pad(prob, (0, 0, 0, 0, 1, 2))
    win = padded[:, 0:D] + padded[:, 1:D + 1] + padded[:, 2:D + 2] + padded[:, 3:D + 3]
    idx = torch.sum(prob * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1), dim=1)
    idx = idx.long().clamp(0, D - 1)
    conf = torch.gather(win, 1, idx.unsqueeze(1)).squeeze(1)
    return prob, depth, conf


def confidence_window_reference(prob: Tensor) -> Tensor:
    """Same window sum the way the reference spells it (4*avg_pool3d), used only
    to double check ``softargmin`` in the oracle tests (module.py:386-387)."""
    return 4 * F.avg_pool3d(F.pad(prob.unsqueeze(1), (0, 0, 0, 0, 1, 2)), (4, 1, 1), stride=1).squeeze(1)


# ----------------------------------------------------------------------------
# a9  depth hypotheses                            models/module.py:394-439 ; model.py:176-193
# ----------------------------------------------------------------------------
def stage_hypotheses(cur_depth: Tensor, ndepth: int, interval: Tensor, depth_min: Tensor,
                     depth_max: Tensor, H: int, W: int, scale: int) -> Tensor:
    """Hypotheses handed to StageNet for one stage: [B,ndepth,H/scale,W/scale].

    cur_depth is either the global plane list [B,Dg] (first stage) or the
    previous stage's depth map [B,h',w'].  interval/depth_min/depth_max are
    [B,1,1] tensors (interval already multiplied by the stage ratio).
    Follows model.py:176-193 and module.py:394-439: bilinear (align_corners=
    False) upsample to HxW, centred sampling, two-sided clamp, then trilinear
    (align_corners=False) resize to the stage grid."""
    B = cur_depth.shape[0]
    if cur_depth.dim() == 2:
        lo, hi = cur_depth[:, 0], cur_depth[:, -1]
        step = (hi - lo) / (ndepth - 1)
        planes = lo.unsqueeze(1) + torch.arange(ndepth, dtype=torch.float32).view(1, -1) * step.unsqueeze(1)
        full = planes.view(B, ndepth, 1, 1).repeat(1, 1, H, W)
    else:
        up = F.interpolate(cur_depth.unsqueeze(1), [H, W], mode="bilinear", align_corners=False).squeeze(1)
        nl = (ndepth - 1) // 2
        first = up - nl * interval
        step = torch.ones_like(up) * interval
        full = first.unsqueeze(1) + torch.arange(ndepth, dtype=torch.float32).view(1, -1, 1, 1) * step.unsqueeze(1)
        dmin = depth_min.view(B, 1, 1, 1)
        dmax = depth_max.view(B, 1, 1, 1)
        full = dmin + (full - dmin).clamp(min=0)
        full = dmax + (full - dmax).clamp(max=0)
    return F.interpolate(full.unsqueeze(1), [ndepth, H // scale, W // scale], mode="trilinear",
                         align_corners=False).squeeze(1)


# ----------------------------------------------------------------------------
# a12  epipolar geometry                          models/dynamic_conv.py:7-47
# ----------------------------------------------------------------------------
def _cross_matrix(v: Tensor) -> Tensor:
    B = v.shape[0]
    S = torch.zeros(B, 3, 3, dtype=v.dtype)
    S[:, 0, 1], S[:, 0, 2] = -v[:, 2], v[:, 1]
    S[:, 1, 0], S[:, 1, 2] = v[:, 2], -v[:, 0]
    S[:, 2, 0], S[:, 2, 1] = -v[:, 1], v[:, 0]
    return S


def fundamental_matrix(cam1: Tensor, cam2: Tensor) -> Tensor:
    """F = [P2 (c1-c2)]_x P2 P1^-1 with P_i = K_i R_i, c_i = -R_i^-1 t_i
    (dynamic_conv.py:19-38).  cam* [B,2,4,4]."""
    K1, R1, t1 = cam1[:, 1, :3, :3], cam1[:, 0, :3, :3], cam1[:, 0, :3, 3:4]
    K2, R2, t2 = cam2[:, 1, :3, :3], cam2[:, 0, :3, :3], cam2[:, 0, :3, 3:4]
    c1 = -torch.inverse(R1) @ t1
    c2 = -torch.inverse(R2) @ t2
    P1 = torch.matmul(K1, R1)
    P2 = torch.matmul(K2, R2)
    e = torch.matmul(P2, c1 - c2)
    return _cross_matrix(e.squeeze(2)) @ P2 @ torch.inverse(P1)


def epipole_from_F(Fm: Tensor) -> Tensor:
    """Solve two linear combinations (c=1e3) of F's rows for the epipole [B,2]
    (dynamic_conv.py:41-47)."""
    c = 1e3
    r1 = c * Fm[:, 0] + Fm[:, 1] + Fm[:, 2]
    r2 = c * Fm[:, 0] - Fm[:, 1] - Fm[:, 2]
    A = torch.stack((r1, r2), dim=1)
    return (-torch.inverse(A[:, :, :2]) @ A[:, :, 2:3]).squeeze(2)


# ----------------------------------------------------------------------------
# a10  dynamic-scale convolution                  models/dynamic_conv.py:81-122
# ----------------------------------------------------------------------------
def dynamic_conv(x: Tensor, epipole: Tensor, temperature: float, sd: Dict[str, Tensor], prefix: str,
                 sizes: Sequence[int], stride: int = 1) -> Tuple[Tensor, Tensor]:
    """x [B,Cin,H,W], epipole [B,2] (pixels at this resolution) -> (out, norm_curv)."""
    B, _, H, W = x.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                            indexing="ij")
    u = xs.view(1, 1, H, W) - epipole[:, 0].view(B, 1, 1, 1)
    v = ys.view(1, 1, H, W) - epipole[:, 1].view(B, 1, 1, 1)
    nrm = torch.sqrt(u ** 2 + v ** 2)
    u, v = u / (nrm + 1e-6), v / (nrm + 1e-6)
    basis = torch.cat((u ** 2, 2 * u * v, v ** 2), dim=1)
    curvs, branches = [], []
    for i, k in enumerate(sizes):
        a = F.conv2d(x, sd[f"{prefix}.att_convs.{i}.weight"], None, padding=(k - 1) // 2)
        curvs.append((a * basis).sum(dim=1, keepdim=True))
        branches.append(F.conv2d(x, sd[f"{prefix}.convs.{i}.weight"], sd.get(f"{prefix}.convs.{i}.bias"),
                                 stride=stride, padding=(k - 1) // 2))
    curvs = torch.cat(curvs, dim=1)
    a = F.conv2d(curvs, sd[f"{prefix}.att_weights.0.weight"])
    a = F.relu(_bn_eval(a, sd, f"{prefix}.att_weights.1"))
    a = F.conv2d(a, sd[f"{prefix}.att_weights.3.weight"])
    wts = F.softmax(a / temperature, dim=1)
    out = (torch.stack(branches, dim=1) * wts.unsqueeze(2)).sum(dim=1)
    nc = (curvs * wts).sum(dim=1, keepdim=True)
    return out, nc


# ----------------------------------------------------------------------------
# a11  feature pyramid                            models/module.py:201-267, 28-71
# ----------------------------------------------------------------------------
def _dyn_block(x, epi, T, sd, key, sizes):
    y, nc = dynamic_conv(x, epi, T, sd, key + ".conv", sizes)
    return F.leaky_relu(F.instance_norm(y, eps=IN_EPS), 0.1), nc


def _plain_block(x, sd, key, stride, padding):
    y = F.conv2d(x, sd[key + ".conv.weight"], None, stride=stride, padding=padding)
    return F.leaky_relu(F.instance_norm(y, eps=IN_EPS), 0.1)


def feature_net(img: Tensor, epipole: Tensor, temperature: float, sd: Dict[str, Tensor],
                prefix: str = "feature") -> Dict[str, Tuple[Tensor, Tensor, Tensor]]:
    """img [B,3,H,W] -> {'stageK': (feat, nc_sum, |nc_out|)}  (module.py:234-267)."""
    p = prefix
    c00, n00 = _dyn_block(img, epipole, temperature, sd, p + ".conv00", (3, 7, 11))
    c01, n01 = _dyn_block(c00, epipole, temperature, sd, p + ".conv01", (3, 5, 7))
    e1 = epipole / 2
    d0 = _plain_block(c01, sd, p + ".downsample1", 2, 1)
    c10, n10 = _dyn_block(d0, e1, temperature, sd, p + ".conv10", (3, 5))
    c11, n11 = _dyn_block(c10, e1, temperature, sd, p + ".conv11", (3, 5))
    e2 = epipole / 4
    d1 = _plain_block(c11, sd, p + ".downsample2", 2, 1)
    c20, n20 = _dyn_block(d1, e2, temperature, sd, p + ".conv20", (1, 3))
    c21, n21 = _dyn_block(c20, e2, temperature, sd, p + ".conv21", (1, 3))

    out = {}
    o1, n22 = dynamic_conv(c21, e2, temperature, sd, p + ".out1", (1, 3))
    o1 = torch.tanh(F.instance_norm(o1, eps=IN_EPS))
    out["stage1"] = (o1, (n20 ** 2 + n21 ** 2 + n22 ** 2) / 3, n22.abs())

    x = torch.cat((F.interpolate(c21, scale_factor=2, mode="nearest"), c11), dim=1)
    x = _plain_block(x, sd, p + ".inner1", 1, 0)
    o2, n12 = dynamic_conv(x, e1, temperature, sd, p + ".out2", (1, 3))
    o2 = torch.tanh(F.instance_norm(o2, eps=IN_EPS))
    out["stage2"] = (o2, (n10 ** 2 + n11 ** 2 + n12 ** 2) / 3, n12.abs())

    x = torch.cat((F.interpolate(o2, scale_factor=2, mode="nearest"), c01), dim=1)
    x = _plain_block(x, sd, p + ".inner2", 1, 0)
    o3, n02 = dynamic_conv(x, epipole, temperature, sd, p + ".out3", (1, 3))
    o3 = torch.tanh(F.instance_norm(o3, eps=IN_EPS))
    out["stage3"] = (o3, (n00 ** 2 + n01 ** 2 + n02 ** 2) / 3, n02.abs())
    return out


# ----------------------------------------------------------------------------
# a13  one cost-volume stage                      models/model.py:16-94 (eval branch)
# ----------------------------------------------------------------------------
def aggregate_views(pairs: List[Dict[str, Tensor]], cams: Tensor, hyp: Tensor, sd: Dict[str, Tensor],
                    stage_idx: int, exact: bool = True) -> Dict[str, Tensor]:
    """Visibility-weighted mean of ref (x) warp over the source views (a1-a5).

    pairs[v] = {'ref': (fea, nc_sum, nc_abs), 'src': (fea, nc_sum, _)}; cams
    [B,N,2,4,4] (view 0 = reference); hyp [B,D,h,w].
    Returns volume_mean, nc_mean, plus per-view entropy / vis_w lists."""
    P_ref = compose_projection(cams[:, 0])
    vol_sum, vis_sum, nc_sum = 0.0, 0.0, 0.0
    ent_list, vis_list = [], []
    for v, pair in enumerate(pairs):
        ref_fea, ref_ncs, ref_nc = pair["ref"]
        src_fea, src_ncs, _ = pair["src"]
        P_src = compose_projection(cams[:, v + 1])
        warped = warp_volume(src_fea, P_src, P_ref, hyp, exact=exact)
        in_prod, ent = correlation_entropy(ref_fea, warped)
        vis = vis_cnn(torch.cat((ent, ref_nc), dim=1), sd, f"stage_net.vis.{stage_idx}")
        vol_sum = vol_sum + in_prod * vis.unsqueeze(1)
        vis_sum = vis_sum + vis
        nc_sum = nc_sum + (ref_ncs + src_ncs) / 2
        ent_list.append(ent)
        vis_list.append(vis)
    return {"volume_mean": vol_sum / (vis_sum.unsqueeze(1) + 1e-6),
            "vis_sum": vis_sum, "nc_mean": nc_sum / len(pairs),
            "entropy": ent_list, "vis_w": vis_list}


def stage_forward(pairs, cams, hyp, sd, stage_idx, exact=True) -> Dict[str, Tensor]:
    agg = aggregate_views(pairs, cams, hyp, sd, stage_idx, exact=exact)
    reg = cost_regularization(agg["volume_mean"], sd, f"cost_regularization.{stage_idx}").squeeze(1)
    _, depth, conf = softargmin(reg, hyp)
    return {"depth": depth, "photometric_confidence": conf, "norm_curv": agg["nc_mean"],
            "_volume_mean": agg["volume_mean"], "_prob_pre": reg}


# ----------------------------------------------------------------------------
# a15  refinement                                  models/module.py:318-370
# ----------------------------------------------------------------------------
def refinement(img: Tensor, depth0: Tensor, dmin: Tensor, dmax: Tensor, sd: Dict[str, Tensor],
               prefix: str = "refine_network") -> Tensor:
    B = dmin.shape[0]
    lo, hi = dmin.view(B, 1, 1, 1), dmax.view(B, 1, 1, 1)
    d = (depth0 - lo) / (hi - lo) * 10

    def cbr(x, key):
        return F.relu(_bn_eval(F.conv2d(x, sd[f"{prefix}.{key}.conv.weight"], None, padding=1), sd,
                               f"{prefix}.{key}.bn"))

    f_img = cbr(img, "conv0")
    f_d = cbr(cbr(d, "conv1"), "conv2")
    f_d = F.conv_transpose2d(f_d, sd[f"{prefix}.deconv.weight"], None, stride=2, padding=1, output_padding=1)
    f_d = F.relu(_bn_eval(f_d, sd, f"{prefix}.bn"))
    res = F.conv2d(cbr(torch.cat((f_d, f_img), dim=1), "conv3"), sd[f"{prefix}.res.weight"], None, padding=1)
    d = (F.interpolate(d, scale_factor=2, mode="bilinear", align_corners=True) + res) / 10
    return d * (hi - lo) + lo


# ----------------------------------------------------------------------------
# a14  full forward                                models/model.py:140-223 (eval)
# ----------------------------------------------------------------------------
def forward(imgs: Tensor, proj_matrices: Dict[str, Tensor], depth_values: Tensor, sd: Dict[str, Tensor],
            ndepths=(48, 32, 8), ratios=(4.0, 1.5, 0.75), refine: bool = False,
            temperature: float = 0.001, exact: bool = True) -> Dict[str, object]:
    B, N, _, H, W = imgs.shape
    dmin = depth_values[:, 0].view(B, 1, 1)
    dmax = depth_values[:, -1].view(B, 1, 1)
    dint = (depth_values[:, 1] - depth_values[:, 0]).view(B, 1, 1)
    if refine:
        H, W = H // 2, W // 2
    cams3 = proj_matrices["stage3"]
    pairs_all = []
    for v in range(1, N):
        Fm = fundamental_matrix(cams3[:, 0], cams3[:, v])
        e_ref = epipole_from_F(Fm)
        e_src = epipole_from_F(Fm.transpose(1, 2))
        f_ref = feature_net(F.interpolate(imgs[:, 0], (H, W)), e_ref, temperature, sd)
        f_src = feature_net(F.interpolate(imgs[:, v], (H, W)), e_src, temperature, sd)
        pairs_all.append({"ref": f_ref, "src": f_src})

    out: Dict[str, object] = {}
    depth = None
    for s in range(len(ndepths)):
        name = f"stage{s + 1}"
        scale = (4, 2, 1)[s]
        cur = depth_values if depth is None else depth
        hyp = stage_hypotheses(cur, ndepths[s], ratios[s] * dint, dmin, dmax, H, W, scale)
        pairs = [{"ref": p["ref"][name], "src": p["src"][name]} for p in pairs_all]
        st = stage_forward(pairs, proj_matrices[name], hyp, sd, s, exact=exact)
        depth = st["depth"]
        st["_hyp"] = hyp
        out[name] = st
        out.update({k: v for k, v in st.items() if not k.startswith("_")})
    if refine:
        lo = depth_values[:, 0] / dint[:, 0, 0]
        hi = depth_values[:, -1] / dint[:, 0, 0]
        r = refinement(imgs[:, 0], (depth / dint).unsqueeze(1), lo, hi, sd)
        out["refined_depth"] = r.squeeze(1) * dint
    else:
        out["refined_depth"] = depth
    return out


# ----------------------------------------------------------------------------
# a16  depth-map filtering + average fusion  (fusion.py:7-114, test.py:334-351)
# ----------------------------------------------------------------------------
def _pixel_centres(h: int, w: int) -> Tensor:
    """[h,w,3,1] homogeneous pixel centres (x+0.5, y+0.5, 1) (fusion.py:7-12)."""
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5,
                            indexing="ij")
    return torch.stack([xs, ys, torch.ones_like(xs)], -1).unsqueeze(-1)


def _lift_and_project(pix: Tensor, depth: Tensor, cam_from: Tensor, cam_to: Tensor) -> Tuple[Tensor, Tensor]:
    """pix [h,w,3,1], depth [h,w], cams [2,4,4] -> (image coords [h,w,2] in cam_to, depth [h,w] in cam_to), with the
    reference's +1e-9 homogeneous normalisation after every step (fusion.py:24-46)."""
    ray = torch.inverse(cam_from[1, :3, :3]) @ pix
    pc = ray / (ray[..., -1:, :] + 1e-9) * depth[..., None, None]
    pc = torch.cat([pc, torch.ones_like(pc[..., -1:, :])], -2)
    pw = torch.inverse(cam_from[0]) @ pc
    pw = pw / (pw[..., -1:, :] + 1e-9)
    po = cam_to[0] @ pw
    po = po / (po[..., -1:, :] + 1e-9)
    q = po[..., :3, :] / (po[..., 3:4, :] + 1e-9)
    im = cam_to[1, :3, :3] @ q
    im = im / (im[..., -1:, :] + 1e-9)
    return im[..., :2, 0], po[..., 2, 0]


def confidence_mask(conf: Tensor, thresh) -> Tensor:
    """conf [3,h,w] -> bool [h,w]: every stage confidence above its threshold (fusion.py:64-72)."""
    m = torch.ones_like(conf[0], dtype=torch.bool)
    for i, p in enumerate(thresh):
        m = m & (conf[i] > p)
    return m


def fuse_view(ref_depth: Tensor, ref_conf: Tensor, ref_cam: Tensor, src_depths: Tensor, src_confs: Tensor,
              src_cams: Tensor, conf=(0.0, 0.0, 0.0), thres_disp: float = 1.0, thres_view: int = 3,
              depth_thresh: float = 0.01) -> Dict[str, Tensor]:
    """One reference view of test.py:334-351.  ref_depth [h,w], ref_conf [3,h,w], src_depths [V,h,w],
    src_confs [V,3,h,w], cams [2,4,4] / [V,2,4,4] -> fused depth, final mask, world points, per-view masks."""
    h, w = ref_depth.shape
    pix = _pixel_centres(h, w)
    view_masks, reproj_d = [], []
    for v in range(src_depths.shape[0]):
        sd = src_depths[v] * confidence_mask(src_confs[v], conf).float()
        xy_sr, d_sr = _lift_and_project(pix, sd, src_cams[v], ref_cam)            # source pixel -> reference view
        xyd = torch.cat([xy_sr, d_sr.unsqueeze(-1)], -1).permute(2, 0, 1).unsqueeze(0)   # [1,3,h,w]
        xy_rs, _ = _lift_and_project(pix, ref_depth, ref_cam, src_cams[v])        # reference pixel -> source view
        grid = torch.stack([xy_rs[..., 0] / w, xy_rs[..., 1] / h], -1)
        grid = (grid * 2 - 1).clamp(-1.1, 1.1)
        inside = ((grid[..., 0] >= -1) & (grid[..., 0] <= 1) & (grid[..., 1] >= -1) & (grid[..., 1] <= 1))
        rep = F.grid_sample(xyd, grid.unsqueeze(0), mode="bilinear", padding_mode="zeros", align_corners=True)[0]
        dist_ok = (rep[:2] - pix[..., :2, 0].permute(2, 0, 1)).norm(dim=0) < thres_disp
        depth_ok = (ref_depth - rep[2]).abs() < torch.max(ref_depth, rep[2]) * depth_thresh
        view_masks.append((inside & dist_ok & depth_ok).float())
        reproj_d.append(rep[2])
    vm, rz = torch.stack(view_masks), torch.stack(reproj_d)
    geo = vm.sum(0) >= (thres_view - 1.1)
    fused = ((rz * vm).sum(0) + ref_depth) / (vm.sum(0) + 1)
    mask = geo & confidence_mask(ref_conf, conf)
    ray = torch.inverse(ref_cam[1, :3, :3]) @ pix
    pc = ray / (ray[..., -1:, :] + 1e-9) * fused[..., None, None]
    pc = torch.cat([pc, torch.ones_like(pc[..., -1:, :])], -2)
    pw = torch.inverse(ref_cam[0]) @ pc
    pw = pw / (pw[..., -1:, :] + 1e-9)
    return {"depth": fused, "mask": mask.float(), "points": pw[..., :3, 0].permute(2, 0, 1), "view_masks": vm,
            "reproj_depth": rz}

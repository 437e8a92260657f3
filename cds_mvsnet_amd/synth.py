"""Seeded synthetic inputs for tests, fixtures and bench.py.

No dataset exists on the build / GPU boxes, so every workload is generated here: DTU-like cameras
(finite epipoles, ~90 % in-image samples), images in [0,1), per-pair feature maps in (-1,1) and
jittered per-pixel depth hypotheses.  Everything is created on CPU from explicit seeds so the same
numbers are obtained in the build container (where the fixtures are captured from the reference)
and on the GPU box.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _rot_xy(a: float, b: float) -> np.ndarray:
    rx = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
    ry = np.array([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]])
    return rx @ ry


def make_cameras(n_views: int, H: int, W: int, refine: bool = False, seed: int = 0,
                 baseline: Tuple[float, float, float] = (40.0, 15.0, 10.0)) -> Dict[str, Tensor]:
    """Multi-scale projection dict as the reference's datasets build it (general_eval.py:167-200):
    ``stageK`` -> [1,N,2,4,4] with [:, :, 0] = world-to-camera extrinsic and [:, :, 1, :3, :3] = intrinsic
    scaled per stage (refine=False: K/4, K/2, K; refine=True adds stage4 = K and halves the others).
    View 0 is the reference camera (identity extrinsic)."""
    rs = np.random.RandomState(seed)
    K = np.array([[0.9 * W, 0, W / 2.0], [0, 0.9 * W, H / 2.0], [0, 0, 1.0]])
    ext = []
    for i in range(n_views):
        E = np.eye(4)
        if i > 0:
            a, b = rs.uniform(-0.06, 0.06, 2)
            t = rs.uniform(-1, 1, 3) * np.array(baseline)
            t[2] += 5.0 if i % 2 else -5.0
            E[:3, :3] = _rot_xy(a, b)
            E[:3, 3] = t
        ext.append(E)
    names = ["stage1", "stage2", "stage3"] + (["stage4"] if refine else [])
    divs = [8.0, 4.0, 2.0, 1.0] if refine else [4.0, 2.0, 1.0]
    out = {}
    for name, dv in zip(names, divs):
        mats = np.zeros((1, n_views, 2, 4, 4), dtype=np.float32)
        for i in range(n_views):
            mats[0, i, 0] = ext[i]
            Ks = K.copy()
            Ks[:2] /= dv
            mats[0, i, 1, :3, :3] = Ks
        out[name] = torch.from_numpy(mats)
    return out


def stage_cameras(n_views: int, h: int, w: int, seed: int = 0) -> Tensor:
    """[1,N,2,4,4] cameras whose intrinsics address an h x w grid directly (single-stage workloads)."""
    return make_cameras(n_views, h, w, refine=False, seed=seed)["stage3"]


def make_depth_values(n: int = 192, start: float = 425.0, step: float = 2.5) -> Tensor:
    return (start + step * torch.arange(n, dtype=torch.float32)).unsqueeze(0)


def make_images(n_views: int, H: int, W: int, seed: int = 0, smooth: bool = True) -> Tensor:
    """[1,N,3,H,W] in [0,1): bicubic-upsampled low-resolution noise plus fine noise (image-like spectrum)."""
    g = torch.Generator().manual_seed(seed)
    if not smooth:
        return torch.rand(1, n_views, 3, H, W, generator=g)
    low = torch.rand(n_views, 3, max(2, H // 16), max(2, W // 16), generator=g)
    img = F.interpolate(low, (H, W), mode="bicubic", align_corners=False)
    img = 0.8 * img + 0.2 * torch.rand(n_views, 3, H, W, generator=g)
    return img.clamp(0.0, 0.999).unsqueeze(0).contiguous()


def _boxblur5(x: Tensor) -> Tensor:
    c = x.shape[1]
    k = torch.full((c, 1, 5, 5), 1.0 / 25.0)
    return F.conv2d(F.pad(x, (2, 2, 2, 2), mode="replicate"), k, groups=c)


def make_pair_features(n_src: int, C: int, h: int, w: int, seed: int = 1,
                       sharp: bool = False) -> List[Dict[str, Tuple[Tensor, Tensor, Tensor]]]:
    """Per-pair stage features in the reference's layout: list over source views of
    {'ref': (fea [1,C,h,w], nc_sum [1,1,h,w], |nc| [1,1,h,w]), 'src': (...)} with fea in (-1,1).
    sharp=False: 5x5 box-blurred (smooth, image-feature like); sharp=True: white tanh noise (large
    correlation dynamic range and maximal sensitivity to the sampling position)."""
    g = torch.Generator().manual_seed(seed)
    feats = []
    for _ in range(n_src):
        d = {}
        for key in ("ref", "src"):
            fea = torch.tanh(torch.randn(1, C, h, w, generator=g) * (1.5 if sharp else 1.0))
            if not sharp:
                fea = _boxblur5(fea)
            nc_sum = torch.rand(1, 1, h, w, generator=g)
            nc = torch.rand(1, 1, h, w, generator=g)
            d[key] = (fea.contiguous(), nc_sum, nc)
        feats.append(d)
    return feats


def make_hypotheses(D: int, h: int, w: int, lo: float = 425.0, hi: float = 902.5, jitter: float = 3.0,
                    seed: int = 1) -> Tensor:
    """[1,D,h,w] planes linspace(lo,hi,D) plus a per-pixel uniform jitter."""
    g = torch.Generator().manual_seed(seed + 1000)
    planes = torch.linspace(lo, hi, D).view(1, D, 1, 1)
    return (planes + jitter * torch.rand(1, D, h, w, generator=g)).contiguous()


def make_fusion_scene(n_views: int, h: int, w: int, seed: int = 0, outlier_frac: float = 0.15) -> Dict[str, Tensor]:
    """Geometrically consistent depth maps for the filtering / fusion step: every camera of ``make_cameras`` looks at
    the height field Z = 650 + 40 sin(X/60) cos(Y/50) (world = reference-camera frame); per-view depth = camera-frame z of
    the ray/surface intersection (fixed-point iteration in float64).  A fraction of the pixels gets a ±(2..6)% depth
    error so that the geometric masks are mixed; confidences are uniform in [0,1).
    -> depths [N,h,w], confs [N,3,h,w], cams [N,2,4,4] (intrinsic [3,3] = 1), imgs [N,h,w,3]."""
    cams = make_cameras(n_views, h, w, refine=False, seed=seed)["stage3"][0].clone()
    cams[:, 1, 3, 3] = 1.0
    rs = np.random.RandomState(seed + 77)
    ys, xs = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing="ij")
    pix = np.stack([xs, ys, np.ones_like(xs)], 0).reshape(3, -1)
    depths = []
    for i in range(n_views):
        E = cams[i, 0].double().numpy()
        K = cams[i, 1, :3, :3].double().numpy()
        R, t = E[:3, :3], E[:3, 3:4]
        d = np.linalg.inv(K) @ pix                         # camera-frame rays with z = 1
        rd, rt = R.T @ d, R.T @ t
        lam = np.full(pix.shape[1], 650.0)
        for _ in range(20):
            P = rd * lam - rt
            surf = 650.0 + 40.0 * np.sin(P[0] / 60.0) * np.cos(P[1] / 50.0)
            lam = (surf + rt[2]) / rd[2]
        dep = lam.reshape(h, w)
        bad = rs.rand(h, w) < outlier_frac
        dep = np.where(bad, dep * (1.0 + rs.choice([-1.0, 1.0], (h, w)) * rs.uniform(0.02, 0.06, (h, w))), dep)
        depths.append(dep.astype(np.float32))
    g = torch.Generator().manual_seed(seed + 5)
    return {"depths": torch.from_numpy(np.stack(depths)), "confs": torch.rand(n_views, 3, h, w, generator=g),
            "cams": cams.contiguous(), "imgs": torch.rand(n_views, h, w, 3, generator=g)}

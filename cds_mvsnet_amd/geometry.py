"""Host-side camera algebra of the plane-sweep path (tiny 3x3 / 4x4 fp32 work).

Mirrors what the reference computes with torch ops inside its forward:
  * projection composition                   models/model.py:40-43
  * relative homography  P_src * P_ref^-1     models/utils/warping.py:80-82
  * fundamental matrix and epipoles           models/dynamic_conv.py:19-47, models/model.py:156-158

Everything here runs on CPU tensors in fp32 with the same torch primitives (matmul / inverse) so the
twelve numbers handed to the HIP kernels are the ones the reference's own CPU forward would use.
"""
from __future__ import annotations

from typing import Tuple

import torch

Tensor = torch.Tensor


def full_projection(cam: Tensor) -> Tensor:
    """cam [2,4,4] = (extrinsic, intrinsic) -> 4x4 projection with K @ [R|t] in the top three rows."""
    cam = cam.unsqueeze(0)
    P = cam[:, 0].clone()
    P[:, :3, :4] = torch.matmul(cam[:, 1, :3, :3], cam[:, 0, :3, :4])
    return P  # [1,4,4]


def warp_matrices(cams: Tensor) -> Tensor:
    """cams [N,2,4,4] (view 0 = reference) -> [N-1,12]: rows of M[:3,:3] then M[:3,3], M = P_v P_0^-1."""
    P_ref = full_projection(cams[0])
    P_ref_inv = torch.inverse(P_ref)
    rows = []
    for v in range(1, cams.shape[0]):
        M = torch.matmul(full_projection(cams[v]), P_ref_inv)[0]
        rows.append(torch.cat((M[:3, :3].reshape(9), M[:3, 3].reshape(3))))
    return torch.stack(rows).contiguous()


def _cross(v: Tensor) -> Tensor:
    S = torch.zeros(1, 3, 3, dtype=v.dtype)
    S[:, 0, 1], S[:, 0, 2] = -v[:, 2], v[:, 1]
    S[:, 1, 0], S[:, 1, 2] = v[:, 2], -v[:, 0]
    S[:, 2, 0], S[:, 2, 1] = -v[:, 1], v[:, 0]
    return S


def fundamental(cam_a: Tensor, cam_b: Tensor) -> Tensor:
    """F such that the epipolar line of a pixel of view a lies in view b.  cam_* [2,4,4] -> [1,3,3]."""
    a, b = cam_a.unsqueeze(0), cam_b.unsqueeze(0)
    Ka, Ra, ta = a[:, 1, :3, :3], a[:, 0, :3, :3], a[:, 0, :3, 3:4]
    Kb, Rb, tb = b[:, 1, :3, :3], b[:, 0, :3, :3], b[:, 0, :3, 3:4]
    centre_a = -torch.inverse(Ra) @ ta
    centre_b = -torch.inverse(Rb) @ tb
    Pa = torch.matmul(Ka, Ra)
    Pb = torch.matmul(Kb, Rb)
    e = torch.matmul(Pb, centre_a - centre_b)
    return _cross(e.squeeze(2)) @ Pb @ torch.inverse(Pa)


def epipole(Fm: Tensor) -> Tuple[float, float]:
    """Epipole (x, y) as the solution of two row combinations of F (weight 1e3 on row 0)."""
    c = 1e3
    r1 = c * Fm[:, 0] + Fm[:, 1] + Fm[:, 2]
    r2 = c * Fm[:, 0] - Fm[:, 1] - Fm[:, 2]
    A = torch.stack((r1, r2), dim=1)
    e = (-torch.inverse(A[:, :, :2]) @ A[:, :, 2:3]).squeeze(2)[0]
    return float(e[0]), float(e[1])


def pair_epipoles(cam_ref: Tensor, cam_src: Tensor) -> Tuple[Tuple[float, float], Tuple[float, float]]:
    Fm = fundamental(cam_ref, cam_src)
    return epipole(Fm), epipole(Fm.transpose(1, 2))

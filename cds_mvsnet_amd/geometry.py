"""Host-side camera algebra of the plane-sweep path (tiny 3x3 / 4x4 fp32 work).

Mirrors what the reference computes with torch ops inside its forward:
  * projection composition                   models/model.py:40-43
  * relative homography  P_src * P_ref^-1     models/utils/warping.py:80-82
  * fundamental matrix and epipoles           models/dynamic_conv.py:19-47, models/model.py:156-158

Everything here runs on CPU tensors in fp32 with the same torch primitives (matmul / inverse) so the
twelve numbers handed to the HIP kernels are the ones the reference's own CPU forward would use.

:class:`GeoBlock` is how those numbers reach the kernels since round 6: every homography, epipole and depth-range scalar of one
forward is written into ONE float32 buffer - pinned host memory mirrored by a device tensor, one asynchronous copy per forward - and
the kernels read device slices of it (they were by-value kernel arguments before).  That is what makes a captured hipGraph of the
forward or of the training step replayable for new cameras: rewrite the block, replay (``graphed.py``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor


class GeoBlock:
    """The per-call geometry of one forward as one float32 buffer.  ``add(name, values)`` registers CPU values (a tensor or a
    sequence of floats); ``upload(device)`` packs them into pinned host memory and issues ONE asynchronous host -> device copy on the
    current stream (into a fresh device tensor, or ``into=`` the static block of a captured graph); ``block[name]`` is the device
    view the kernels read.  Entries start on 64-byte boundaries.  The pinned buffer comes from PyTorch's caching host allocator, which
    does not hand it out again before the copy has run, so the host may run several forwards ahead of the GPU."""

    ALIGN = 16   # floats

    def __init__(self) -> None:
        self._values: List[Tuple[str, Tensor]] = []
        self._index: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self._size = 0
        self.dev: Optional[Tensor] = None

    def add(self, name: str, values) -> None:
        if name in self._index:
            raise KeyError(f"GeoBlock: duplicate entry {name}")
        if not isinstance(values, torch.Tensor):
            values = torch.tensor([float(v) for v in values], dtype=torch.float32)
        values = values.detach().to(dtype=torch.float32, device="cpu")
        self._index[name] = (self._size, tuple(values.shape))
        self._values.append((name, values))
        self._size += (values.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN

    def __contains__(self, name: str) -> bool:
        return name in self._index

    def layout(self) -> Tuple[Tuple[str, int, Tuple[int, ...]], ...]:
        """(name, offset, shape) of every entry: two forwards with equal layouts can share a captured graph."""
        return tuple((n,) + self._index[n] for n, _ in self._values)

    def numel(self) -> int:
        return max(self._size, self.ALIGN)

    def pack(self, pin: bool = True) -> Tensor:
        host = torch.zeros((self.numel(),), dtype=torch.float32, pin_memory=pin)
        for name, v in self._values:
            off = self._index[name][0]
            host[off:off + v.numel()] = v.reshape(-1)
        return host

    def upload(self, device, into: Optional[Tensor] = None) -> "GeoBlock":
        host = self.pack(pin=torch.cuda.is_available())
        if into is None:
            into = torch.empty((self.numel(),), dtype=torch.float32, device=device)
        elif into.numel() != self.numel() or into.dtype != torch.float32:
            raise ValueError("GeoBlock.upload: the static block does not match this layout")
        into.copy_(host, non_blocking=True)
        self.dev = into
        return self

    def bind(self, dev: Tensor) -> "GeoBlock":
        """Use `dev` (a device tensor that already holds / will hold this layout's values) as the block: the capture pass of a graph."""
        if dev.numel() != self.numel():
            raise ValueError("GeoBlock.bind: size mismatch")
        self.dev = dev
        return self

    def __getitem__(self, name: str) -> Tensor:
        if self.dev is None:
            raise RuntimeError("GeoBlock: upload() first")
        off, shape = self._index[name]
        n = 1
        for d in shape:
            n *= d
        return self.dev[off:off + n].view(shape)

    def host(self, name: str) -> Tensor:
        for n, v in self._values:
            if n == name:
                return v
        raise KeyError(name)


def full_projection(cam: Tensor) -> Tensor:
    """cam [2,4,4] = (extrinsic, intrinsic) -> 4x4 projection with K @ [R|t] in the top three rows."""
    cam = cam.unsqueeze(0)
    P = cam[:, 0].clone()
    P[:, :3, :4] = torch.matmul(cam[:, 1, :3, :3], cam[:, 0, :3, :4])
    return P  # [1,4,4]


def warp_matrices(cams: Tensor) -> Tensor:
    """cams [N,2,4,4] (view 0 = reference) -> [N-1,12]: rows of M[:3,:3] then M[:3,3], M = P_v P_0^-1.
    All views in one batched matmul / inverse (round 6: the host side of a forward is on the latency path of a replayed graph);
    bit-identical to the per-view form (each batch item runs the same 3x3 / 4x4 kernel; checked on 400 camera sets)."""
    ext = cams[:, 0]
    P = ext.clone()
    P[:, :3, :4] = torch.matmul(cams[:, 1, :3, :3], ext[:, :3, :4])
    M = torch.matmul(P[1:], torch.inverse(P[0:1]))
    return torch.cat((M[:, :3, :3].reshape(-1, 9), M[:, :3, 3]), dim=1).contiguous()


def _cross(v: Tensor) -> Tensor:
    S = torch.zeros(v.shape[0], 3, 3, dtype=v.dtype)
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


def pairs_epipoles(cams: Tensor) -> Tuple[Tensor, Tensor]:
    """cams [N,2,4,4] (view 0 = reference) -> (epipole in the reference image of every pair [N-1,2], epipole in each source image
    [N-1,2]): :func:`pair_epipoles` for all source views in one batched pass (same values, a quarter of the host time at N = 5)."""
    a, b = cams[0:1], cams[1:]
    Ka, Ra, ta = a[:, 1, :3, :3], a[:, 0, :3, :3], a[:, 0, :3, 3:4]
    Kb, Rb, tb = b[:, 1, :3, :3], b[:, 0, :3, :3], b[:, 0, :3, 3:4]
    centre_a = -torch.inverse(Ra) @ ta
    centre_b = -torch.inverse(Rb) @ tb
    Pa = torch.matmul(Ka, Ra)
    Pb = torch.matmul(Kb, Rb)
    e = torch.matmul(Pb, centre_a - centre_b)
    Fm = _cross(e.squeeze(2)) @ Pb @ torch.inverse(Pa)

    def solve(Fx: Tensor) -> Tensor:
        c = 1e3
        r1 = c * Fx[:, 0] + Fx[:, 1] + Fx[:, 2]
        r2 = c * Fx[:, 0] - Fx[:, 1] - Fx[:, 2]
        A = torch.stack((r1, r2), dim=1)
        return (-torch.inverse(A[:, :, :2]) @ A[:, :, 2:3]).squeeze(2)

    return solve(Fm), solve(Fm.transpose(1, 2))
